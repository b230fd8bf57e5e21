"""Row a12 (RobotLearning/omniisaacgymenvs/utils/math.py:33-193) on the host, and the pin of what pins the task-layer oracle:

  * the seven third-party quaternion helpers that tools/gen_golden.py supplies in place of omni.isaac.core.utils.torch.rotations
    (source absent from the reference) are checked against scipy.spatial.transform.Rotation -- the library the reference itself routes
    every rotation through (utils/math.py:15-18,29-31);
  * locomanipulationrl_amd/utils/math.py (the scipy-free mirror user scripts import) is checked against scipy AND against
    tests/golden/math.npz, the outputs of the reference's own utils/math.py on the same inputs: all six functions
    (rotate_orientations, inverse_rotate_orientations, transform_vectors, inverse_transform_vectors, rand_quaternions and the
    `__main__` self-check input of :212-216).
"""
import importlib.util
import os

import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

from conftest import GOLDEN, ROOT
from locomanipulationrl_amd.utils import math as M


def _gen_golden_module():
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(ROOT, "tools", "gen_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)          # defines the helpers; touches /root/reference only in main()
    return mod


def _rand_quats(n, seed):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(n, 4, generator=g, dtype=torch.float64)
    return q / q.norm(dim=1, keepdim=True)


def _rot(q):          # scalar-first -> scipy (scalar-last), utils/math.py:16
    q = np.asarray(q)
    return Rotation.from_quat(q[:, [1, 2, 3, 0]])


@pytest.mark.parametrize("source", ["standin", "host"])
def test_quaternion_helpers_against_scipy(source):
    H = _gen_golden_module() if source == "standin" else M
    q, r = _rand_quats(200, 1), _rand_quats(200, 2)
    v = torch.randn(200, 3, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    Rq, Rr = _rot(q), _rot(r)
    assert np.abs(H.quat_rotate(q, v).numpy() - Rq.apply(v.numpy())).max() < 1e-12                      # R(q) v
    assert np.abs(H.quat_rotate_inverse(q, v).numpy() - Rq.inv().apply(v.numpy())).max() < 1e-12        # R(q)^T v
    qm = H.quat_mul(q, r).numpy()                                                                        # Hamilton: R(q r) = R(q) R(r)
    assert np.abs(_rot(qm).as_matrix() - Rq.as_matrix() @ Rr.as_matrix()).max() < 1e-12
    assert np.abs(_rot(H.quat_conjugate(q).numpy()).as_matrix() - Rq.inv().as_matrix()).max() < 1e-12
    if source == "standin":
        for k in range(3):
            assert np.abs(H.quat_axis(q, k).numpy() - Rq.as_matrix()[:, :, k]).max() < 1e-12             # R(q) e_k
        assert np.abs(H.quat_apply(q, v).numpy() - Rq.apply(v.numpy())).max() < 1e-12
        lo, up = torch.tensor([-3.0, -1.5]), torch.tensor([3.0, 2.5])
        x = torch.tensor([[-1.0, 1.0], [0.0, 0.0], [0.5, -0.5]])
        assert torch.allclose(H.unscale_transform(x, lo, up), torch.tensor([[-3.0, 2.5], [0.0, 0.5], [1.5, -0.5]]))
    e = torch.rand(200, 3, generator=torch.Generator().manual_seed(4), dtype=torch.float64) * 2 - 1
    e = e * torch.tensor([0.4, 0.4, 1.57])
    qe = H.quat_from_euler_xyz(e[:, 0], e[:, 1], e[:, 2]).numpy()                                        # extrinsic xyz = Rz Ry Rx
    assert np.abs(_rot(qe).as_matrix() - Rotation.from_euler("xyz", e.numpy()).as_matrix()).max() < 1e-12


def test_host_math_against_the_reference_outputs():
    """locomanipulationrl_amd/utils/math.py vs the reference's own utils/math.py (tests/golden/math.npz, tools/gen_golden.py:gen_math)."""
    g = np.load(os.path.join(GOLDEN, "math.npz"))
    q, r, t, V = (torch.from_numpy(g[k]) for k in ("q", "r", "t", "V"))
    assert np.abs(M.transform_vectors(q, t, V).numpy() - g["transform"]).max() < 2e-6                   # :102-113
    assert np.abs(M.inverse_transform_vectors(q, t, V).numpy() - g["inverse_transform"]).max() < 2e-6   # :115-127
    assert np.abs(M.rotate_orientations(r, q).numpy() - g["rotate"]).max() < 2e-6                       # :33-57
    assert np.abs(M.inverse_rotate_orientations(r, q).numpy() - g["inverse_rotate"]).max() < 2e-6       # :59-83
    assert (g["rotate"][:, 0] >= 0).all() and (g["inverse_rotate"][:, 0] >= 0).all()                    # the w >= 0 rule
    u = torch.from_numpy(g["rand_u"])                                                                    # :176-193 on the same uniform draws
    rq = M.quat_from_euler_xyz(-0.4 + 0.8 * u[:, 0], -0.4 + 0.8 * u[:, 1], -1.57 + 3.14 * u[:, 2])
    assert np.abs(rq.numpy() - g["rand_quat"]).max() < 2e-6
    sc = M.inverse_rotate_orientations(torch.tensor([-0.5, -0.5, 0.5, 0.5]).repeat(2, 1), torch.tensor([0.7071, 0, 0, 0.7071]).repeat(2, 1))
    # the file's only self-check (:212-216) is the one ambiguous input of the w >= 0 rule: the result has w = 0 exactly, where scipy's
    # matrix -> quaternion branch decides the sign; equal up to that sign
    assert abs(float(g["selfcheck"][0, 0])) < 1e-6
    assert min(np.abs(sc.numpy() - g["selfcheck"]).max(), np.abs(sc.numpy() + g["selfcheck"]).max()) < 1e-4


def test_host_math_against_scipy():
    q, r = _rand_quats(100, 7), _rand_quats(100, 8)
    g = torch.Generator().manual_seed(9)
    t = torch.randn(100, 3, generator=g, dtype=torch.float64); V = torch.randn(100, 5, 3, generator=g, dtype=torch.float64)
    Rq, Rr = _rot(q).as_matrix(), _rot(r).as_matrix()
    assert np.abs(M.transform_vectors(q, t, V).numpy() - (np.einsum("nij,nmj->nmi", Rq, V.numpy()) + t.numpy()[:, None])).max() < 1e-12
    assert np.abs(M.inverse_transform_vectors(q, t, V).numpy() - np.einsum("nji,nmj->nmi", Rq, V.numpy() - t.numpy()[:, None])).max() < 1e-12
    ro = M.rotate_orientations(r, q).numpy(); iro = M.inverse_rotate_orientations(r, q).numpy()
    assert np.abs(_rot(ro).as_matrix() - Rr @ Rq).max() < 1e-12 and (ro[:, 0] >= 0).all()
    assert np.abs(_rot(iro).as_matrix() - np.transpose(Rr, (0, 2, 1)) @ Rq).max() < 1e-12 and (iro[:, 0] >= 0).all()
    rq = M.rand_quaternions(64, -0.4, 0.4, -0.4, 0.4, -1.57, 1.57, "cpu", generator=torch.Generator().manual_seed(0))
    e = _rot(rq.double().numpy()).as_euler("xyz")
    assert (np.abs(e[:, :2]) <= 0.4 + 1e-6).all() and (np.abs(e[:, 2]) <= 1.57 + 1e-6).all() and np.abs(rq.norm(dim=1) - 1).max() < 1e-6
