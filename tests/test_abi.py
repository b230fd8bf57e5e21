"""C-ABI checks that need no GPU: the library loads, exports every symbol include/lm_engine.h declares, the
ctypes mirror of lm_params has the C layout, and argument validation returns error codes (no exceptions /
crashes across the ABI)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from conftest import ROOT
from locomanipulationrl_amd import lib as lmlib

HEADERS = [os.path.join(ROOT, "include", h) for h in ("lm_engine.h", "lm_policy.h")]


@pytest.fixture(scope="module")
def so():
    lmlib.build_library()
    return lmlib.load_library()


def declared_symbols():
    text = "".join(open(h).read() for h in HEADERS)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lm_[a-z_0-9]+)\s*\(", text)))


def test_exports_match_header(so):
    names = declared_symbols()
    assert set(names) == set(lmlib.EXPORTS), (names, lmlib.EXPORTS)
    for n in names:
        assert hasattr(so, n), f"{n} declared in lm_engine.h but not exported"
    assert b"gfx950" in so.lm_version() and so.lm_abi_version() == lmlib.ABI_VERSION
    hdr = open(HEADERS[0]).read()
    assert int(re.search(r"#define LM_ABI_VERSION (\d+)", hdr).group(1)) == lmlib.ABI_VERSION


def test_params_struct_layout_matches_c():
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "lm_engine.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(lm_params), ' \
          'offsetof(lm_params, substeps), offsetof(lm_params, init_q), offsetof(lm_params, corner), offsetof(lm_params, plate_phi), ' \
          'offsetof(lm_params, variant), offsetof(lm_params, dr), offsetof(lm_params, acc_dt_inv), sizeof(lm_dr_channel));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c"); open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        size, o_sub, o_q, o_corner, o_phi, o_var, o_dr, o_acc, s_ch = map(int, subprocess.check_output([exe]).split())
    P = lmlib.LmParams
    assert C.sizeof(P) == size and C.sizeof(lmlib.LmDrChannel) == s_ch
    assert (P.substeps.offset, P.init_q.offset, P.corner.offset, P.plate_phi.offset) == (o_sub, o_q, o_corner, o_phi)
    assert (P.variant.offset, P.dr.offset, P.acc_dt_inv.offset) == (o_var, o_dr, o_acc)


def test_argument_validation_without_gpu(so):
    h = C.c_void_p()
    assert so.lm_create(C.byref(h), 0, None, None, 1, 0, 0) == -1            # LM_EINVAL: null table/params
    assert b"null" in so.lm_last_error()
    assert so.lm_step(None, None, None, None, None, None, None, None, None) == -1
    assert so.lm_reset_all(None, None) == -1
    assert so.lm_num_envs(None) == 0 and so.lm_ptr(None, 0) is None
    assert so.lm_destroy(None) == 0
    import numpy as np
    from locomanipulationrl_amd.engine_config import loco_params
    tab = np.zeros(lmlib.TABLE_FLOATS, np.float32)
    arr = (lmlib.LmParams * 2)(lmlib.make_params(loco_params()), lmlib.make_params(loco_params()))
    assert so.lm_create(C.byref(h), -5, tab.ctypes.data_as(C.c_void_p), arr, 1, 0, 0) == -1
    # a caller built against another header: wrong version stamp, a shorter struct, a shorter table -> LM_EINVAL before anything else is read
    for field, value in (("abi_version", lmlib.ABI_VERSION - 1), ("params_size", C.sizeof(lmlib.LmParams) - 4), ("table_floats", 486)):
        stale = (lmlib.LmParams * 1)(lmlib.make_params(loco_params())); setattr(stale[0], field, value)
        assert so.lm_create(C.byref(h), 64, tab.ctypes.data_as(C.c_void_p), stale, 1, 0, 0) == -1 and b"ABI mismatch" in so.lm_last_error(), field
    stale = (lmlib.LmParams * 2)(lmlib.make_params(loco_params()), lmlib.make_params(loco_params())); stale[1].abi_version = 2
    assert so.lm_create(C.byref(h), 64, tab.ctypes.data_as(C.c_void_p), stale, 2, 32, 0) == -1 and b"second parameter block" in so.lm_last_error()
    assert so.lm_create(C.byref(h), 64, tab.ctypes.data_as(C.c_void_p), arr, 3, 0, 0) == -1
    assert so.lm_create(C.byref(h), 64, tab.ctypes.data_as(C.c_void_p), arr, 2, 24, 0) == -1     # split not a multiple of 16
    bad = (lmlib.LmParams * 1)(lmlib.make_params(loco_params(dt=0.0)))
    assert so.lm_create(C.byref(h), 64, tab.ctypes.data_as(C.c_void_p), bad, 1, 0, 0) == -1
    bad = (lmlib.LmParams * 1)(lmlib.make_params(loco_params(variant=1)))                    # custom controller needs the 88-wide observation
    assert so.lm_create(C.byref(h), 64, tab.ctypes.data_as(C.c_void_p), bad, 1, 0, 0) == -1 and b"variant" in so.lm_last_error()
    from locomanipulationrl_amd.engine_config import DRChannel
    dr = [DRChannel() for _ in range(8)]; dr[1] = DRChannel(enabled=1, operation=0, distribution=0, interval=0)      # on_interval noise without an interval
    bad = (lmlib.LmParams * 1)(lmlib.make_params(loco_params(dr_enabled=1, dr=dr)))
    assert so.lm_create(C.byref(h), 64, tab.ctypes.data_as(C.c_void_p), bad, 1, 0, 0) == -1 and b"randomisation" in so.lm_last_error()
    # policy / rollout entry points
    assert so.lm_mlp_param_count_obs(64) == so.lm_mlp_param_count() and so.lm_mlp_param_count_obs(88) > so.lm_mlp_param_count() and so.lm_mlp_param_count_obs(70) == -1
    assert so.lm_rollout_create(None, None, 0, None, None, 0, 0, None, None, None, None, None, None, None) == -1
    assert so.lm_rollout_run(None, 1, None) == -1 and b"lm_rollout_run" in so.lm_last_error() and so.lm_rollout_destroy(None) == 0
    assert so.lm_sample_actions(None, None, None, 0, 0, None, None, None) == -1
