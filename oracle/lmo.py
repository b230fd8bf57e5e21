"""ctypes binding of the CPU oracle (oracle/lm_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
The model / params objects are duck-typed (attributes of RobotModel / EngineParams).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
MAXB, NTREE, NQ = 24, 20, 12
PHYS, TASK, CNT, READBACK, TERMS = 50, 65, 6, 99, 11


class LmoModel(C.Structure):
    _fields_ = [
        ("nb", C.c_int32), ("parent", C.c_int32 * MAXB), ("dof", C.c_int32 * MAXB),
        ("Rt", (C.c_double * 9) * MAXB), ("pt", (C.c_double * 3) * MAXB), ("axis", (C.c_double * 3) * MAXB),
        ("mass", C.c_double * MAXB), ("com", (C.c_double * 3) * MAXB), ("inertia", (C.c_double * 9) * MAXB),
        ("nclos", C.c_int32), ("clos_p", C.c_int32 * 8), ("clos_a", C.c_int32 * 8), ("clos_b", C.c_int32 * 8),
        ("clos_s", C.c_double * 8),
        ("tip_body", C.c_int32 * 4), ("tip_off", (C.c_double * 3) * 4), ("knee_body", C.c_int32 * 8),
        ("contact_body", C.c_int32 * 4), ("contact_off", (C.c_double * 3) * 4),
    ]


DR_CNT = 5


class LmoDrChannel(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("operation", C.c_int32), ("distribution", C.c_int32), ("interval", C.c_int32),
                ("p0", C.c_double * 3), ("p1", C.c_double * 3)]


class LmoParams(C.Structure):
    _fields_ = [
        ("dt", C.c_double), ("substeps", C.c_int32), ("pgs_iters", C.c_int32), ("gravity", C.c_double),
        ("kd", C.c_double), ("tau_max", C.c_double), ("act_scale", C.c_double), ("mu", C.c_double),
        ("tip_radius", C.c_double), ("baumgarte", C.c_double), ("max_depen_vel", C.c_double), ("max_joint_vel", C.c_double),
        ("mode", C.c_int32), ("pyramid", C.c_int32),
        ("fixed_base_pos", C.c_double * 3), ("fixed_base_quat", C.c_double * 4),
        ("plate_mass", C.c_double), ("plate_com", C.c_double * 3), ("plate_inertia", C.c_double * 3),
        ("plate_half", C.c_double * 3), ("plate_center", C.c_double * 3),
        ("init_q", C.c_double * 12), ("init_base_pos", C.c_double * 3), ("init_base_quat", C.c_double * 4),
        ("init_plate_pos", C.c_double * 3), ("init_plate_quat", C.c_double * 4), ("default_tip", C.c_double * 12),
        ("goal_lo", C.c_double * 3), ("goal_hi", C.c_double * 3),
        ("s_pos", C.c_double), ("s_lin", C.c_double), ("s_ang", C.c_double), ("s_q", C.c_double), ("s_qd", C.c_double),
        ("quat_scale", C.c_double), ("rot_eps", C.c_double), ("trans_scale", C.c_double), ("acc_scale", C.c_double),
        ("rate_scale", C.c_double), ("bonus", C.c_double), ("limit_pen", C.c_double), ("fall_pen", C.c_double),
        ("succ_thresh", C.c_double), ("max_consec", C.c_int32), ("max_episode", C.c_int32),
        ("d23_pen", C.c_double * 2), ("d23_rst", C.c_double * 2),
        ("d1_pen", (C.c_double * 2) * 4), ("d1_rst", (C.c_double * 2) * 4),
        ("h_base", C.c_double), ("h_corner", C.c_double), ("h_knee", C.c_double),
        ("corner", (C.c_double * 3) * 4), ("ctrl_dt", C.c_double),
        ("variant", C.c_int32), ("num_obs", C.c_int32), ("pd_kp", C.c_double), ("joint_damping", C.c_double), ("act_scale_se", C.c_double),
        ("se_lo", C.c_double * 12), ("se_hi", C.c_double * 12), ("init_se", C.c_double * 12), ("torque_div", C.c_double),
        ("power_scale", C.c_double), ("target_err_scale", C.c_double), ("rot_dec_scale", C.c_double), ("rot_dec_thresh", C.c_double), ("cc_update_last_tgt", C.c_int32), ("acc_substeps", C.c_int32),
        ("dr_enabled", C.c_int32), ("dr_min_frequency", C.c_int32), ("dr", LmoDrChannel * 9),
        ("drive_mode", C.c_int32), ("pd_second_pass", C.c_int32),
        ("solver", C.c_int32), ("vel_iters", C.c_int32), ("drive_iter_impulse", C.c_double), ("tgs_flags", C.c_int32),
    ]


def build(force: bool = False) -> None:
    """Compile the oracle shared objects next to the sources (gcc only)."""
    if force or not all(os.path.exists(os.path.join(_DIR, f"liblmoracle_{s}.so")) for s in ("f64", "f32")) or \
            os.path.getmtime(os.path.join(_DIR, "lm_oracle.c")) > os.path.getmtime(os.path.join(_DIR, "liblmoracle_f64.so")):
        subprocess.check_call(["make", "-C", _DIR, "-s", "-B"])


def _set(arr, values):
    v = np.asarray(values, dtype=np.float64)
    if v.ndim == 1:
        for i, x in enumerate(v):
            arr[i] = float(x)
    else:
        for i, row in enumerate(v):
            for j, x in enumerate(row):
                arr[i][j] = float(x)


def make_model(rm) -> LmoModel:
    m = LmoModel()
    m.nb = rm.nb
    for k in range(rm.nb):
        m.parent[k] = int(rm.parent[k]); m.dof[k] = int(rm.dof[k]); m.mass[k] = float(rm.mass[k])
    _set(m.Rt, rm.Rt); _set(m.pt, rm.pt); _set(m.axis, rm.axis); _set(m.com, rm.com); _set(m.inertia, rm.inertia)
    m.nclos = 8
    for c in range(8):
        m.clos_p[c] = int(rm.clos_p[c]); m.clos_a[c] = int(rm.clos_a[c]); m.clos_b[c] = int(rm.clos_b[c])
        m.clos_s[c] = float(rm.clos_s[c])
    for i in range(4):
        m.tip_body[i] = int(rm.tip_body[i])
    _set(m.tip_off, rm.tip_off)
    for i in range(8):
        m.knee_body[i] = int(rm.knee_body[i])
    for i in range(4):
        m.contact_body[i] = int(rm.contact_body[i])
    _set(m.contact_off, rm.contact_off)
    return m


def make_params(ep) -> LmoParams:
    p = LmoParams()
    for name, ctype in LmoParams._fields_:
        if name == "pyramid":
            p.pyramid = int(getattr(ep, "pyramid", 0)); continue
        if name in ("solver", "vel_iters", "tgs_flags", "drive_iter_impulse"):            # oracle-only experiment of round 4
            setattr(p, name, getattr(ep, name, 0)); continue
        if name == "dr":
            for i, ch in enumerate(ep.dr):
                p.dr[i].enabled, p.dr[i].operation, p.dr[i].distribution, p.dr[i].interval = int(ch.enabled), int(ch.operation), int(ch.distribution), int(ch.interval)
                for c in range(3):
                    p.dr[i].p0[c] = float(ch.p0[c]); p.dr[i].p1[c] = float(ch.p1[c])
            continue
        val = getattr(ep, name)
        if isinstance(val, (list, tuple, np.ndarray)):
            _set(getattr(p, name), val)
        else:
            setattr(p, name, val)
    return p


class Oracle:
    """CPU oracle instance (float64 by default; precision='f32' for the float build)."""

    def __init__(self, robot_model, engine_params, precision: str = "f64"):
        build()
        self.lib = C.CDLL(os.path.join(_DIR, f"liblmoracle_{precision}.so"))
        self.dtype = np.float64 if precision == "f64" else np.float32
        assert self.lib.lmo_sizeof_real() == np.dtype(self.dtype).itemsize
        self.model = make_model(robot_model)
        self.params = make_params(engine_params)
        self._rm, self._ep = robot_model, engine_params

    def set_params(self, engine_params):
        self.params = make_params(engine_params); self._ep = engine_params

    def _p(self, a):
        return a.ctypes.data_as(C.c_void_p)

    def _arr(self, a, shape=None):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if shape is not None:
            assert a.shape == shape, (a.shape, shape)
        return a

    # -- state helpers ---------------------------------------------------------
    def new_state(self, N):
        phys = np.zeros((N, PHYS), self.dtype); phys[:, 3] = 1; phys[:, 40] = 1
        task = np.zeros((N, TASK), self.dtype); task[:, 36] = 1
        cnt = np.zeros((N, CNT), np.int64); cnt[:, 3] = 1       # reset_buf starts as ones (rl_task.py:111)
        return phys, task, cnt

    def fk(self, phys):
        phys = self._arr(phys); N = phys.shape[0]
        tips = np.zeros((N, 4, 3), self.dtype); knees = np.zeros((N, 8, 3), self.dtype)
        for e in range(N):
            self.lib.lmo_fk(C.byref(self.model), C.byref(self.params), self._p(phys[e]), self._p(tips[e]), self._p(knees[e]))
        return tips, knees

    def substep(self, phys, targets):
        assert phys.dtype == self.dtype and phys.flags.c_contiguous
        targets = self._arr(targets, (phys.shape[0], 12))
        self.lib.lmo_substep(C.byref(self.model), C.byref(self.params), C.c_int(phys.shape[0]), self._p(phys), self._p(targets))

    def substep_tau(self, phys, targets):
        """One sub-step that also returns the drive torque applied over it (what the PD-actuator tasks log)."""
        assert phys.dtype == self.dtype and phys.flags.c_contiguous
        targets = self._arr(targets, (phys.shape[0], 12)); tau = np.zeros((phys.shape[0], 12), self.dtype)
        self.lib.lmo_substep_tau(C.byref(self.model), C.byref(self.params), C.c_int(phys.shape[0]), self._p(phys), self._p(targets), self._p(tau))
        return tau

    def dyn_terms(self, phys_row):
        phys_row = self._arr(phys_row, (PHYS,))
        M = np.zeros((18, 18), self.dtype); h = np.zeros(18, self.dtype)
        self.lib.lmo_dyn_terms(C.byref(self.model), C.byref(self.params), self._p(phys_row), self._p(M), self._p(h))
        return M, h

    def task_eval(self, readback, actions, task, cnt):
        N = readback.shape[0]
        readback = self._arr(readback, (N, READBACK)); actions = self._arr(actions, (N, 12))
        assert task.dtype == self.dtype and cnt.dtype == np.int64
        obs = np.zeros((N, self._ep.num_obs), self.dtype); states = np.zeros((N, 93), self.dtype)
        rew = np.zeros(N, self.dtype); terms = np.zeros((N, TERMS), self.dtype)
        self.lib.lmo_task_eval(C.byref(self.params), C.c_int(N), self._p(readback), self._p(actions), self._p(task),
                               self._p(cnt), self._p(obs), self._p(states), self._p(rew), self._p(terms))
        return obs, states, rew, terms

    def reset(self, phys, task, cnt, goal_rand=None, seed=0):
        N = phys.shape[0]
        gr = None if goal_rand is None else self._arr(goal_rand, (N, 3))
        self.lib.lmo_reset(C.byref(self.params), C.c_int(N), self._p(phys), self._p(task), self._p(cnt),
                           None if gr is None else self._p(gr), C.c_uint32(seed))

    def step(self, phys, task, cnt, actions, goal_rand=None, seed=0):
        N = phys.shape[0]
        actions = self._arr(actions, (N, 12))
        gr = None if goal_rand is None else self._arr(goal_rand, (N, 3))
        obs = np.zeros((N, self._ep.num_obs), self.dtype); states = np.zeros((N, 93), self.dtype)
        rew = np.zeros(N, self.dtype); terms = np.zeros((N, TERMS), self.dtype)
        self.lib.lmo_step(C.byref(self.model), C.byref(self.params), C.c_int(N), self._p(phys), self._p(task), self._p(cnt),
                          self._p(actions), None if gr is None else self._p(gr), C.c_uint32(seed),
                          self._p(obs), self._p(states), self._p(rew), self._p(terms))
        return obs, states, rew, terms

    def new_dr_counters(self, N):
        return np.zeros((N, DR_CNT), np.int64)

    def step_dr(self, phys, task, cnt, drc, actions_raw, clip_actions=1.0, goal_rand=None, seed=0, env_offset=0):
        """lmo_step with domain randomisation; returns obs (noisy, unclipped), states, rew, terms, the clamped noisy actions and the
        sampled physics attributes (N x 42: max efforts 12, max velocities 12, gravity 3, base force 3, joint damping 12)."""
        N = phys.shape[0]
        a = self._arr(actions_raw, (N, 12)); gr = None if goal_rand is None else self._arr(goal_rand, (N, 3))
        assert drc.dtype == np.int64 and drc.shape == (N, DR_CNT)
        obs = np.zeros((N, self._ep.num_obs), self.dtype); states = np.zeros((N, 93), self.dtype)
        rew = np.zeros(N, self.dtype); terms = np.zeros((N, TERMS), self.dtype); used = np.zeros((N, 12), self.dtype); phd = np.zeros((N, 42), self.dtype)
        cl = C.c_double(clip_actions) if self.dtype == np.float64 else C.c_float(clip_actions)
        self.lib.lmo_set_env_offset(C.c_uint32(env_offset))
        self.lib.lmo_step_dr(C.byref(self.model), C.byref(self.params), C.c_int(N), self._p(phys), self._p(task), self._p(cnt), self._p(drc),
                             self._p(a), cl, None if gr is None else self._p(gr), C.c_uint32(seed),
                             self._p(obs), self._p(states), self._p(rew), self._p(terms), self._p(used), self._p(phd))
        self.lib.lmo_set_env_offset(C.c_uint32(0))
        return obs, states, rew, terms, used, phd

    def dr_noise(self, on_reset, on_interval, seed, stream, buf, reset_flags, counter, corr_key, step_key):
        """randomize.py:212-306 on a float buffer (in place); on_reset / on_interval are DRChannel-like objects or None."""
        def ch(c):
            if c is None:
                return None
            x = LmoDrChannel(); x.enabled, x.operation, x.distribution, x.interval = int(c.enabled), int(c.operation), int(c.distribution), int(c.interval)
            for k in range(3):
                x.p0[k] = float(c.p0[k]); x.p1[k] = float(c.p1[k])
            return x
        r, i = ch(on_reset), ch(on_interval)
        N, D = buf.shape
        assert buf.dtype == self.dtype and buf.flags.c_contiguous
        for a in (reset_flags, counter, corr_key, step_key):
            assert a.dtype == np.int64 and a.shape == (N,)
        self.lib.lmo_dr_noise(None if r is None else C.byref(r), None if i is None else C.byref(i), C.c_uint32(seed), C.c_uint32(stream),
                              C.c_int(N), C.c_int(D), self._p(buf), self._p(reset_flags), self._p(counter), self._p(corr_key), self._p(step_key))

    def dr_sample(self, seed, stream, env, key, idx, dist, p0, p1):
        f = self.lib.lmo_dr_sample
        rt = C.c_double if self.dtype == np.float64 else C.c_float
        f.restype = rt
        return float(f(C.c_uint32(seed), C.c_uint32(stream), C.c_uint32(env), C.c_uint32(key), C.c_uint32(idx), C.c_int(dist), rt(p0), rt(p1)))

    def hash_uniform3(self, seed, env, episode):
        u = np.zeros(3, self.dtype)
        self.lib.lmo_hash_uniform3(C.c_uint32(seed), C.c_uint32(env), C.c_uint32(episode), self._p(u))
        return u
