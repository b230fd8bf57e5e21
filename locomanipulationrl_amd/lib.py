"""ctypes binding of the C-ABI engine library (include/lm_engine.h) + zero-copy torch views.

The HIP library is the product path; there is no CPU fallback.  Importing this module never
touches the GPU; constructing an :class:`Engine` does, and fails loudly when the shared object or
a HIP device is missing.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
_SO = os.environ.get("LM_ENGINE_SO", os.path.join(_CSRC, "liblm_engine.so"))     # override: kernel experiments only

STATE_ROWS, CNT_ROWS, NUM_OBS, NUM_STATES, NUM_ACTIONS, NUM_EXTRAS, TABLE_FLOATS, TERM_ROWS, READBACK = 115, 6, 64, 93, 12, 13, 502, 11, 99
PTR_STATE, PTR_CNT, PTR_OBS_BUF, PTR_STATES_BUF, PTR_REW_BUF, PTR_EXTRAS, PTR_STATS, PTR_TERMS, PTR_DR_CNT, PTR_DR_PHYS = range(10)

# names of the exported C symbols (checked by tests/test_abi.py against include/lm_engine.h)
EXPORTS = ["lm_create", "lm_destroy", "lm_step", "lm_post_physics", "lm_reset_all", "lm_task_eval", "lm_apply_resets", "lm_substeps",
           "lm_forward_kinematics", "lm_debug_dynamics", "lm_ptr", "lm_num_envs", "lm_num_obs", "lm_set_seed", "lm_last_error", "lm_version", "lm_abi_version",
           "lm_gnn_param_count", "lm_gnn_forward", "lm_mlp_param_count", "lm_mlp_forward", "lm_mlp_param_count_obs", "lm_mlp_forward_obs",
           "lm_sample_actions", "lm_rollout_create", "lm_rollout_run", "lm_rollout_destroy"]

# rows of the SoA float state
ROW = dict(base_pos=0, base_quat=3, base_lin=7, base_ang=10, q=13, qd=25, plate_pos=37, plate_quat=40, plate_lin=44,
           plate_ang=47, last_actions=50, last_qd=62, last_tip=74, goal=86)
CNT = dict(successes=0, consecutive_successes=1, goal_reset_buf=2, reset_buf=3, progress_buf=4, episode_count=5)


class LmDrChannel(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("operation", C.c_int32), ("distribution", C.c_int32), ("interval", C.c_int32),
                ("p0", C.c_float * 3), ("p1", C.c_float * 3)]


ABI_VERSION = 4          # LM_ABI_VERSION of include/lm_engine.h this mirror was written against


class LmParams(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("params_size", C.c_int32), ("table_floats", C.c_int32), ("reserved0", C.c_int32),
        ("dt", C.c_float), ("kd", C.c_float), ("tau_max", C.c_float), ("act_scale", C.c_float), ("mu", C.c_float),
        ("tip_radius", C.c_float), ("baumgarte", C.c_float), ("max_depen_vel", C.c_float), ("max_joint_vel", C.c_float), ("gravity", C.c_float),
        ("substeps", C.c_int32), ("pgs_iters", C.c_int32), ("mode", C.c_int32),
        ("fixed_base_pos", C.c_float * 3), ("fixed_base_quat", C.c_float * 4),
        ("plate_mass", C.c_float), ("plate_com", C.c_float * 3), ("plate_inertia", C.c_float * 3),
        ("plate_half", C.c_float * 3), ("plate_center", C.c_float * 3),
        ("init_q", C.c_float * 12), ("init_base_pos", C.c_float * 3), ("init_base_quat", C.c_float * 4),
        ("init_plate_pos", C.c_float * 3), ("init_plate_quat", C.c_float * 4),
        ("default_tip", C.c_float * 12), ("goal_lo", C.c_float * 3), ("goal_hi", C.c_float * 3),
        ("s_pos", C.c_float), ("s_lin", C.c_float), ("s_ang", C.c_float), ("s_q", C.c_float), ("s_qd", C.c_float),
        ("quat_scale", C.c_float), ("rot_eps", C.c_float), ("trans_scale", C.c_float), ("acc_scale", C.c_float),
        ("rate_scale", C.c_float), ("bonus", C.c_float), ("limit_pen", C.c_float), ("fall_pen", C.c_float),
        ("succ_thresh", C.c_float),
        ("max_consec", C.c_int32), ("max_episode", C.c_int32),
        ("d23_pen", C.c_float * 2), ("d23_rst", C.c_float * 2), ("d1_pen", (C.c_float * 2) * 4), ("d1_rst", (C.c_float * 2) * 4),
        ("h_base", C.c_float), ("h_corner", C.c_float), ("h_knee", C.c_float), ("corner", (C.c_float * 3) * 4),
        ("clip_obs", C.c_float), ("clip_actions", C.c_float),
        ("max_reset_counts", C.c_int32),
        ("variant", C.c_int32), ("num_obs", C.c_int32), ("pd_kp", C.c_float), ("joint_damping", C.c_float), ("act_scale_se", C.c_float),
        ("se_lo", C.c_float * 12), ("se_hi", C.c_float * 12), ("init_se", C.c_float * 12),
        ("torque_div", C.c_float), ("power_scale", C.c_float), ("target_err_scale", C.c_float), ("rot_dec_scale", C.c_float),
        ("rot_dec_thresh", C.c_float), ("cc_update_last_tgt", C.c_int32), ("acc_substeps", C.c_int32),
        ("dr_enabled", C.c_int32), ("dr_min_frequency", C.c_int32), ("dr", LmDrChannel * 9), ("drive_mode", C.c_int32), ("pd_second_pass", C.c_int32),
        ("plate_si", C.c_float * 10), ("plate_phi", C.c_float * 36), ("ctrl_dt_inv", C.c_float), ("acc_dt_inv", C.c_float),
    ]


_DERIVED = {"plate_si", "plate_phi", "ctrl_dt_inv", "acc_dt_inv", "abi_version", "params_size", "table_floats", "reserved0"}


def make_params(ep, clip_obs: float = 5.0, clip_actions: float = 1.0) -> LmParams:
    """EngineParams -> C struct (clipObservations / clipActions: QuadrupedPoseControl.yaml:11-12)."""
    if getattr(ep, "solver", 0):
        raise ValueError("EngineParams.solver != 0 is an oracle-only experiment (DESIGN.md 2.2): the engine implements solver 0")
    if getattr(ep, "pyramid", 0):
        raise ValueError("EngineParams.pyramid is an oracle-only evidence switch: the engine implements the friction cone only")
    p = LmParams()
    p.abi_version, p.params_size, p.table_floats = ABI_VERSION, C.sizeof(LmParams), TABLE_FLOATS      # the stamp lm_create checks
    for name, _ in LmParams._fields_:
        if name in _DERIVED:
            continue
        if name == "clip_obs":
            p.clip_obs = float(clip_obs); continue
        if name == "clip_actions":
            p.clip_actions = float(clip_actions); continue
        if name == "dr":
            for i, ch in enumerate(ep.dr):
                p.dr[i].enabled, p.dr[i].operation, p.dr[i].distribution, p.dr[i].interval = int(ch.enabled), int(ch.operation), int(ch.distribution), int(ch.interval)
                for c in range(3):
                    p.dr[i].p0[c] = float(ch.p0[c]); p.dr[i].p1[c] = float(ch.p1[c])
            continue
        val = getattr(ep, name)
        if isinstance(val, (list, tuple, np.ndarray)):
            arr = np.asarray(val, dtype=np.float64)
            dst = getattr(p, name)
            if arr.ndim == 1:
                for i, x in enumerate(arr):
                    dst[i] = float(x)
            else:
                for i, row in enumerate(arr):
                    for j, x in enumerate(row):
                        dst[i][j] = float(x)
        else:
            setattr(p, name, val)
    return p


def hipcc_command(extra, out):
    """The ONE hipcc command line of the engine: the product build and every diagnostic build (tools/stamp_profile*.py, tools/ab_build.py: `extra` =
    their -D switches) compile with the same flags, so that a diagnostic library measures the product's code."""
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
    # -fno-slp-vectorize: the SLP vectoriser's own packed-fp32 pairs cost more v_mov / AGPR shuffles than they save (round 1: 56.0 -> 58.2 M
    #   env-steps/s without it); the packed fp32 the kernel relies on is written by hand on the f2 type (lm_math.h) and is not affected
    # -fno-hip-fp32-correctly-rounded-divide-sqrt: 1/x and sqrt as v_rcp / v_sqrt (1 ulp) instead of the ~10-instruction IEEE sequences
    # -amdgpu-mfma-vgpr-form: the MFMA accumulators of the policy tiles live in ordinary VGPRs, so the VALU work on them (ELU, max aggregation,
    # LDS stores) needs no v_accvgpr_read per element (504 of them in k_gnn_forward)
    return [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fno-hip-fp32-correctly-rounded-divide-sqrt",
            "-mllvm", "-amdgpu-mfma-vgpr-form", "-fPIC", "-shared", *extra, os.path.join(_CSRC, "lm_engine.hip"), os.path.join(_CSRC, "lm_engine_w2.hip"), os.path.join(_CSRC, "lm_policy.hip"), "-o", out]


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/lm_engine.hip for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_CSRC, "lm_engine.hip")
    src2 = os.path.join(_CSRC, "lm_policy.hip")
    inc = os.path.join(os.path.dirname(os.path.dirname(_CSRC)), "include")
    import glob
    deps = sorted(glob.glob(os.path.join(_CSRC, "*.hip")) + glob.glob(os.path.join(_CSRC, "*.h")) + glob.glob(os.path.join(inc, "*.h")))      # every source the .so is built from
    if not force and os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(d) for d in deps):
        return _SO
    cmd = hipcc_command(extra=[], out=_SO)
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return _SO


_lib = None


def load_library() -> C.CDLL:
    """dlopen the engine; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise RuntimeError(f"{_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the engine has no CPU fallback)")
    lib = C.CDLL(_SO)
    vp, fp, ip = C.c_void_p, C.c_void_p, C.c_int
    lib.lm_create.argtypes = [C.POINTER(vp), ip, C.c_void_p, C.POINTER(LmParams), ip, ip, C.c_uint32]
    lib.lm_destroy.argtypes = [vp]
    lib.lm_step.argtypes = [vp, fp, fp, fp, fp, fp, fp, fp, vp]
    lib.lm_post_physics.argtypes = [vp, fp, fp, fp, fp, fp, fp, vp]
    lib.lm_reset_all.argtypes = [vp, vp]
    lib.lm_task_eval.argtypes = [vp, fp, fp, fp, fp, fp, fp, fp, vp]
    lib.lm_apply_resets.argtypes = [vp, fp, vp]
    lib.lm_substeps.argtypes = [vp, fp, ip, vp]
    lib.lm_forward_kinematics.argtypes = [vp, fp, fp, vp]
    lib.lm_debug_dynamics.argtypes = [vp, fp, fp, vp]
    lib.lm_ptr.argtypes = [vp, ip]; lib.lm_ptr.restype = vp
    lib.lm_num_envs.argtypes = [vp]
    lib.lm_num_obs.argtypes = [vp]
    lib.lm_set_seed.argtypes = [vp, C.c_uint32]
    lib.lm_gnn_forward.argtypes = [fp, ip, fp, fp, fp, vp]
    lib.lm_mlp_forward.argtypes = [fp, ip, fp, fp, fp, vp]
    lib.lm_mlp_param_count_obs.argtypes = [ip]
    lib.lm_mlp_forward_obs.argtypes = [fp, ip, ip, fp, fp, fp, vp]
    lib.lm_sample_actions.argtypes = [fp, fp, vp, ip, C.c_uint32, fp, fp, vp]
    lib.lm_rollout_create.argtypes = [C.POINTER(vp), vp, ip, fp, fp, ip, C.c_uint32, fp, fp, fp, fp, fp, vp, fp]
    lib.lm_rollout_run.argtypes = [vp, ip, vp]
    lib.lm_rollout_destroy.argtypes = [vp]
    lib.lm_last_error.restype = C.c_char_p
    lib.lm_version.restype = C.c_char_p
    _lib = lib
    return lib


class EngineError(RuntimeError):
    pass


class _DevArray:
    """Minimal __cuda_array_interface__ carrier so torch can wrap handle-owned device memory zero-copy."""

    def __init__(self, ptr: int, shape, typestr: str, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}
        self._owner = owner


class Engine:
    """One engine instance = N lock-step environments on one GPU (one process per GPU)."""

    def __init__(self, robot_model, params: Sequence, num_envs: int, split_env: Optional[int] = None, seed: int = 42,
                 device: str = "cuda:0", clip_obs: float = 5.0, clip_actions: float = 1.0):
        import torch
        if not torch.cuda.is_available():
            raise EngineError("no HIP device visible: the engine runs on MI355X only (no CPU fallback)")
        self.torch = torch
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise EngineError(f"the engine runs on a HIP device, not {self.device}")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.lib = load_library()
        self.num_envs = int(num_envs)
        params = list(params)
        arr = (LmParams * len(params))(*[make_params(p, clip_obs, clip_actions) for p in params])
        table = np.ascontiguousarray(robot_model.packed_table(), dtype=np.float32)
        assert table.shape == (TABLE_FLOATS,)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.lm_create(C.byref(self._h), self.num_envs, table.ctypes.data_as(C.c_void_p), arr, len(params),
                                    int(split_env or 0), C.c_uint32(seed))
        self._check(rc)
        self.seed = int(seed) & 0xFFFFFFFF
        N = self.num_envs
        self.state = self._wrap(PTR_STATE, (STATE_ROWS, N), "<f4")
        self.cnt = self._wrap(PTR_CNT, (CNT_ROWS, N), "<i8")
        self.num_obs = int(self.lib.lm_num_obs(self._h))
        self._views = {}            # obs_buf / states_buf / terms: asked for on first access (lm_step writes them only from then on)
        self._steps_run = 0         # lm_step calls so far (a first view request after stepping is stale: warned about)
        self.rew_buf = self._wrap(PTR_REW_BUF, (N,), "<f4")
        self.extras_buf = self._wrap(PTR_EXTRAS, (NUM_EXTRAS,), "<f4")
        self.stats_i64 = self._wrap(PTR_STATS, (6,), "<i8")
        self._stats_i32 = self._wrap(PTR_STATS, (16,), "<i4")         # word 15: contained blow-ups
        self.dr_cnt = self._wrap(PTR_DR_CNT, (5, N), "<i8")          # domain-randomisation counters
        self.dr_phys = self._wrap(PTR_DR_PHYS, (42, N), "<f4")       # attributes sampled for the last step

    def _view(self, kind, shape):
        if kind not in self._views:
            if self._steps_run:
                # lm_step starts writing this buffer with the NEXT step: what it holds now is zeros or an older step's values
                import warnings
                warnings.warn("lm_engine: an unclipped view (obs_buf / states_buf / terms) was first requested after %d step(s): it is stale until the next "
                              "step (ask for it before stepping; include/lm_engine.h, lm_ptr_kind)" % self._steps_run, RuntimeWarning, stacklevel=3)
            self._views[kind] = self._wrap(kind, shape, "<f4")
        return self._views[kind]

    # The engine's own unclipped copies of what step() hands out clipped (task.obs_buf / states_buf of rl_task.py:104-113, the per-env reward
    # terms).  lm_step keeps one current only after its pointer has been asked for (lm_ptr), or when step() is called without the matching
    # output tensor: a caller that consumes out_obs / out_states alone does not pay for the second copy (672 B per env-step of stores).
    @property
    def obs_buf(self):
        return self._view(PTR_OBS_BUF, (self.num_envs, self.num_obs))

    @property
    def states_buf(self):
        return self._view(PTR_STATES_BUF, (self.num_envs, NUM_STATES))

    @property
    def terms(self):
        return self._view(PTR_TERMS, (TERM_ROWS, self.num_envs))

    @property
    def blowups(self) -> int:
        """Envs whose state became non-finite or exploded and was replaced by the reset pose (contained by the kernel's guard)."""
        return int(self._stats_i32[15].item()) & 0xFFFFFFFF

    # ------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != 0:
            raise EngineError(f"lm_engine error {rc}: {self.lib.lm_last_error().decode()}")

    def _wrap(self, kind: int, shape, typestr: str):
        ptr = self.lib.lm_ptr(self._h, kind)
        if not ptr:
            raise EngineError("lm_ptr returned NULL")
        return self.torch.as_tensor(_DevArray(ptr, shape, typestr, self), device=self.device)

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _p(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def _f32(self, t, shape):
        assert t.device == self.device and t.dtype == self.torch.float32 and t.is_contiguous() and tuple(t.shape) == tuple(shape), \
            (t.device, self.device, t.dtype, tuple(t.shape), shape)
        return t

    # ------------------------------------------------------------------ C-ABI calls
    def step(self, actions, goal_rand=None, out_obs=None, out_states=None, out_rew=None, out_resets=None, out_extras=None):
        N = self.num_envs
        self._f32(actions, (N, NUM_ACTIONS))
        if goal_rand is not None:
            self._f32(goal_rand, (N, 3))
        if out_obs is not None: self._f32(out_obs, (N, self.num_obs))
        if out_states is not None: self._f32(out_states, (N, NUM_STATES))
        if out_rew is not None: self._f32(out_rew, (N,))
        if out_extras is not None: self._f32(out_extras, (NUM_EXTRAS,))
        if out_resets is not None:
            assert out_resets.dtype == self.torch.int64 and tuple(out_resets.shape) == (N,) and out_resets.device == self.device and out_resets.is_contiguous()
        self._check(self.lib.lm_step(self._h, self._p(actions), self._p(goal_rand), self._p(out_obs), self._p(out_states),
                                     self._p(out_rew), self._p(out_resets), self._p(out_extras), self._stream()))
        self._steps_run += 1

    def post_physics(self, actions, out_obs=None, out_states=None, out_rew=None, out_resets=None, out_extras=None):
        self._f32(actions, (self.num_envs, NUM_ACTIONS))
        self._check(self.lib.lm_post_physics(self._h, self._p(actions), self._p(out_obs), self._p(out_states), self._p(out_rew),
                                             self._p(out_resets), self._p(out_extras), self._stream()))

    def task_eval(self, readback, actions, out_obs=None, out_states=None, out_rew=None, out_resets=None, out_extras=None):
        N = self.num_envs
        self._f32(readback, (N, READBACK)); self._f32(actions, (N, NUM_ACTIONS))
        self._check(self.lib.lm_task_eval(self._h, self._p(readback), self._p(actions), self._p(out_obs), self._p(out_states),
                                          self._p(out_rew), self._p(out_resets), self._p(out_extras), self._stream()))

    def reset_all(self):
        self._check(self.lib.lm_reset_all(self._h, self._stream()))

    def apply_resets(self, goal_rand=None):
        if goal_rand is not None:
            self._f32(goal_rand, (self.num_envs, 3))
        self._check(self.lib.lm_apply_resets(self._h, self._p(goal_rand), self._stream()))

    def substeps(self, targets, n: int = 1):
        self._f32(targets, (self.num_envs, NUM_ACTIONS))
        self._check(self.lib.lm_substeps(self._h, self._p(targets), int(n), self._stream()))

    def forward_kinematics(self):
        N = self.num_envs
        tips = self.torch.empty((N, 4, 3), dtype=self.torch.float32, device=self.device)
        knees = self.torch.empty((N, 8, 3), dtype=self.torch.float32, device=self.device)
        self._check(self.lib.lm_forward_kinematics(self._h, self._p(tips), self._p(knees), self._stream()))
        return tips, knees

    def debug_dynamics(self):
        N = self.num_envs
        M = self.torch.zeros((N, 18, 18), dtype=self.torch.float32, device=self.device)
        h = self.torch.zeros((N, 18), dtype=self.torch.float32, device=self.device)
        self._check(self.lib.lm_debug_dynamics(self._h, self._p(M), self._p(h), self._stream()))
        return M, h

    def set_seed(self, seed: int):
        self._check(self.lib.lm_set_seed(self._h, C.c_uint32(seed)))
        self.seed = int(seed) & 0xFFFFFFFF

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.torch.cuda.synchronize(self.device)
            self.lib.lm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ convenience (tests / tooling)
    def set_phys_env_major(self, phys):
        """phys: (N,50) env-major array in the oracle's LMO_PHYS layout -> rows 0..49 of the SoA state."""
        t = self.torch.as_tensor(np.ascontiguousarray(phys, dtype=np.float32).T.copy(), device=self.device)
        self.state[:50].copy_(t)

    def get_phys_env_major(self):
        return self.state[:50].T.contiguous().cpu().numpy()

    def set_task_env_major(self, task):
        t = self.torch.as_tensor(np.ascontiguousarray(task, dtype=np.float32).T.copy(), device=self.device)
        self.state[50:50 + t.shape[0]].copy_(t)

    def get_task_env_major(self):
        return self.state[50:115].T.contiguous().cpu().numpy()

    def set_cnt_env_major(self, cnt):
        self.cnt.copy_(self.torch.as_tensor(np.ascontiguousarray(cnt, dtype=np.int64).T.copy(), device=self.device))

    def get_cnt_env_major(self):
        return self.cnt.T.contiguous().cpu().numpy()

    # ------------------------------------------------------------------ simulator checkpoint (the reference never
    # checkpoints simulator state; with a deterministic engine a snapshot + the same actions reproduces a run bit for bit)
    def state_dict(self):
        """Everything lm_step reads besides its arguments: SoA state, counters, success windows, randomisation counters (host tensors)."""
        self.torch.cuda.synchronize(self.device)
        return {"state": self.state.cpu().clone(), "cnt": self.cnt.cpu().clone(), "stats": self._stats_i32.cpu().clone(),
                "dr_cnt": self.dr_cnt.cpu().clone(), "num_envs": self.num_envs, "seed": self.seed}

    def load_state_dict(self, sd):
        assert int(sd["num_envs"]) == self.num_envs and tuple(sd["state"].shape) == tuple(self.state.shape)
        self.state.copy_(sd["state"].to(self.device)); self.cnt.copy_(sd["cnt"].to(self.device))
        self._stats_i32.copy_(sd["stats"].to(self.device)); self.dr_cnt.copy_(sd["dr_cnt"].to(self.device))
        if "seed" in sd:          # goal sampling / domain randomisation are keyed by the seed: a resumed engine must carry the checkpoint's
            self.set_seed(int(sd["seed"]))


POLICY_MLP, POLICY_GNN = 0, 1


class Rollout:
    """T steps of  policy forward (MFMA) -> gaussian sampling -> lm_step  recorded into rollout buffers and replayed as one hipGraph
    (include/lm_policy.h).  `packed_params` / `log_std` are device tensors read at run time: refresh them in place
    between runs.  Buffers: obs (T+1,N,64) with obs[0] = current observations, actions (T,N,12), logp (T,N), values (T+1,N),
    rewards (T,N), dones int64 (T,N), extras (T,13)."""

    def __init__(self, engine: Engine, policy: int, packed_params, log_std, T: int, noise_seed: int = 0):
        torch = engine.torch
        self.engine, self.T, self.N = engine, int(T), engine.num_envs
        assert engine.num_obs == 64 or (engine.num_obs == 88 and policy == POLICY_MLP), "the GNN reads the 64-wide layout; the MLP 64 or 88"
        dev = engine.device
        assert packed_params.is_cuda and packed_params.dtype == torch.float32 and packed_params.is_contiguous()
        assert log_std.is_cuda and log_std.dtype == torch.float32 and log_std.numel() == 12 and log_std.is_contiguous()
        n_expected = engine.lib.lm_mlp_param_count_obs(engine.num_obs) if policy == POLICY_MLP else engine.lib.lm_gnn_param_count()
        assert packed_params.numel() == n_expected, (packed_params.numel(), n_expected)
        self.params, self.log_std = packed_params, log_std
        z = lambda *s, dt=torch.float32: torch.zeros(*s, device=dev, dtype=dt)
        T, N = self.T, self.N
        self.obs, self.actions, self.logp, self.values = z(T + 1, N, engine.num_obs), z(T, N, 12), z(T, N), z(T + 1, N)
        self.rewards, self.dones, self.extras = z(T, N), z(T, N, dt=torch.int64), z(T, NUM_EXTRAS)
        self._h = C.c_void_p()
        p = Engine._p
        rc = engine.lib.lm_rollout_create(C.byref(self._h), engine._h, int(policy), p(self.params), p(self.log_std), T, C.c_uint32(noise_seed),
                                          p(self.obs), p(self.actions), p(self.logp), p(self.values), p(self.rewards), p(self.dones), p(self.extras))
        if rc != 0:
            raise EngineError(f"lm_rollout_create failed ({rc})")

    MODES = {"enqueue": 0, "graph": 1, "persistent": 2, "auto": 3}

    def run(self, use_graph=True):
        """use_graph: False / "enqueue" (2T+1 launches), True / "graph" (one hipGraph replay) or "persistent" (the whole rollout in one kernel:
        un-randomised engines; pays up to 16 envs x compute units = 4096 envs on MI355X) or "auto" (persistent where it pays, else the graph);
        identical buffers in all modes."""
        mode = self.MODES[use_graph] if isinstance(use_graph, str) else (1 if use_graph else 0)
        rc = self.engine.lib.lm_rollout_run(self._h, mode, self.engine._stream())
        if rc != 0:
            raise EngineError(f"lm_rollout_run failed ({rc}): {self.engine.lib.lm_last_error().decode()}")

    def close(self):
        if self._h:
            self.engine.lib.lm_rollout_destroy(self._h); self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sample_actions(engine: Engine, mean, log_std, seed: int):
    """lm_sample_actions: (actions, logp) for the engine's current counters."""
    torch = engine.torch
    N = engine.num_envs
    act = torch.empty((N, 12), device=engine.device); logp = torch.empty((N,), device=engine.device)
    p = Engine._p
    rc = engine.lib.lm_sample_actions(p(mean.contiguous()), p(log_std.contiguous()), p(engine.cnt), N, C.c_uint32(seed), p(act), p(logp), engine._stream())
    if rc != 0:
        raise EngineError(f"lm_sample_actions failed ({rc})")
    return act, logp
