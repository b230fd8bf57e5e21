#!/usr/bin/env python3
"""Generate golden vectors for the TASK LAYER from the reference's own Python.

Runs only in the build container (needs /root/reference); the .npz files it writes under
tests/golden/ are data (inputs + expected outputs) and are what travels to the GPU box.

Method: the reference task classes import Isaac Sim packages that do not exist
here (ordinary ModuleNotFoundError).  Inert placeholder modules are registered for those
names so that the reference's *own* torch code in
    tasks/quadruped_pose_control_tasks/quadruped_pose_control.py   (get_observations,
        calculate_metrics, is_done, reset_idx, pre_physics_step)
    tasks/quadruped_manipulate_plate/quadruped_manipulate_plate.py
    utils/math.py, robot/base/robot.py (take_action)
executes unmodified on CPU tensors.  The only behaviour supplied from outside the reference
is the 7 quaternion helpers + unscale_transform of omni.isaac.core.utils.torch (third-party,
source absent); their semantics are fixed by the call-site evidence listed in SURVEY 8(c)
and they are unit-tested against scipy in tests/test_math_host.py.

Run:  python -B tools/gen_golden.py
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF_RL = "/root/reference/RobotLearning/omniisaacgymenvs"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


# ----------------------------------------------------------------------------- placeholders
class _InertMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return lambda *a, **k: None


class _Inert(metaclass=_InertMeta):
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Inert()

    def __getattr__(self, name):
        return _Inert()


class _InertModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return type(name, (_Inert,), {})


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = ("omni", "pxr", "carb", "gym")

    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _InertModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


# ---- the 7 third-party quaternion helpers (omni.isaac.core.utils.torch.rotations; source absent) with the semantics the reference's
# call sites fix: scalar-first, Hamilton product, quat_rotate_inverse(q, v) = R(q)^T v, quat_axis(q, k) = R(q) e_k,
# euler 'xyz' = Rz(yaw) Ry(pitch) Rx(roll).  Module level so that tests/test_math_host.py can check them against scipy.
def quat_mul(a, b):
    w1, x1, y1, z1 = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    w2, x2, y2, z2 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return torch.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
                        w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], dim=-1)

def quat_conjugate(a):
    return torch.cat((a[..., :1], -a[..., 1:]), dim=-1)

def quat_rotate(q, v):
    qw, qv = q[:, 0:1], q[:, 1:]
    return v * (2.0 * qw ** 2 - 1.0) + torch.cross(qv, v, dim=-1) * qw * 2.0 + qv * (qv * v).sum(-1, keepdim=True) * 2.0

def quat_rotate_inverse(q, v):
    qw, qv = q[:, 0:1], q[:, 1:]
    return v * (2.0 * qw ** 2 - 1.0) - torch.cross(qv, v, dim=-1) * qw * 2.0 + qv * (qv * v).sum(-1, keepdim=True) * 2.0

def quat_axis(q, axis=0):
    b = torch.zeros(q.shape[0], 3, device=q.device, dtype=q.dtype)
    b[:, axis] = 1
    return quat_rotate(q, b)

def quat_apply(q, v):
    return quat_rotate(q, v)

def quat_from_euler_xyz(roll, pitch, yaw):
    cy, sy = torch.cos(yaw * 0.5), torch.sin(yaw * 0.5)
    cr, sr = torch.cos(roll * 0.5), torch.sin(roll * 0.5)
    cp, sp = torch.cos(pitch * 0.5), torch.sin(pitch * 0.5)
    return torch.stack([cy * cr * cp + sy * sr * sp, cy * sr * cp - sy * cr * sp,
                        cy * cr * sp + sy * sr * cp, sy * cr * cp - cy * sr * sp], dim=-1)


def unscale_transform(x, lower, upper):
    return x * (upper - lower) * 0.5 + (upper + lower) * 0.5


STANDIN_HELPERS = (quat_mul, quat_conjugate, quat_rotate, quat_rotate_inverse, quat_axis, quat_apply, quat_from_euler_xyz)


def _install_placeholders():
    sys.meta_path.insert(0, _Finder())
    rot = types.ModuleType("omni.isaac.core.utils.torch.rotations")

    for f in STANDIN_HELPERS:
        setattr(rot, f.__name__, f)
    maths = types.ModuleType("omni.isaac.core.utils.torch.maths")
    maths.unscale_transform = unscale_transform
    import omni.isaac.core.utils.torch  # noqa: F401  (placeholder package)
    sys.modules[rot.__name__] = rot
    sys.modules[maths.__name__] = maths
    # the reference's own omniisaacgymenvs.* imports inside rl_task.py need pxr-heavy modules: make them inert too
    for name in ("omniisaacgymenvs", "omniisaacgymenvs.tasks", "omniisaacgymenvs.tasks.utils",
                 "omniisaacgymenvs.tasks.utils.usd_utils", "omniisaacgymenvs.utils",
                 "omniisaacgymenvs.utils.domain_randomization", "omniisaacgymenvs.utils.domain_randomization.randomize"):
        m = _InertModule(name); m.__path__ = []
        sys.modules[name] = m
    return rot


# ----------------------------------------------------------------------------- fakes handed to the reference code
class FakeRobot:
    num_modules = 4

    def __init__(self, N):
        z = lambda *s: torch.zeros(*s, dtype=torch.float32)
        self.joint_positions, self.joint_velocities, self.joint_accelerations = z(N, 12), z(N, 12), z(N, 12)
        self.last_joint_velcoties = z(N, 12)
        self.base_positions, self.base_quaternions = z(N, 3), z(N, 4)
        self.base_linear_velocities, self.base_angular_velocities = z(N, 3), z(N, 3)
        self.tip_positions, self.knee_positions = z(N, 4, 3), z(N, 8, 3)

    def __getattr__(self, name):          # every setter / updater is a no-op
        if name == "_robot_articulation":
            return _Inert()
        return lambda *a, **k: None


class FakeObj:
    def __init__(self, N):
        self.pos = torch.zeros(N, 3); self.quat = torch.zeros(N, 4); self.lin = torch.zeros(N, 3); self.ang = torch.zeros(N, 3)

    def get_object_poses(self):
        return self.pos, self.quat

    def get_object_velocities(self):
        return self.lin, self.ang

    def __getattr__(self, name):
        return lambda *a, **k: None


class FakeDR:
    randomize = False


def _rand_unit_quat(g, n, small=None):
    q = torch.randn(n, 4, generator=g)
    if small is not None:
        q = torch.cat((torch.ones(n, 1), small * torch.randn(n, 3, generator=g)), dim=-1)
    return q / q.norm(dim=-1, keepdim=True)


def make_task(kind, N):
    cc = kind.endswith("_cc"); pc = kind.endswith("_pc")
    if kind == "loco_pc":
        from tasks.quadruped_pose_control_tasks.quadruped_pose_control_position_control import QuadrupedPoseControlPositionControl as T
        mangle = "_QuadrupedPoseControlPositionControl"
    elif kind == "mani_pc":
        from tasks.quadruped_manipulate_plate.quadruped_manipulate_plate_position_control import QuadrupedManipulatePlatePositionControl as T
        mangle = "_QuadrupedManipulatePlatePositionControl"
    elif kind == "loco_cc":
        from tasks.quadruped_pose_control_tasks.quadruped_pose_control_custom_controller import QuadrupedPoseControlCustomController as T
        mangle = "_QuadrupedPoseControlCustomController"
    elif kind == "mani_cc":
        from tasks.quadruped_manipulate_plate.quadruped_manipulate_plate_custom_controller import QuadrupedManipulatePlateCustomController as T
        mangle = "_QuadrupedManipulatePlateCustomController"
    elif kind == "loco":
        from tasks.quadruped_pose_control_tasks.quadruped_pose_control import QuadrupedPoseControl as T
        mangle = "_QuadrupedPoseControl"
    elif kind == "loco_v":
        from tasks.quadruped_pose_control_tasks.quadruped_pose_control_vertical import QuadrupedPoseControlVertical as T
        mangle = "_QuadrupedPoseControlVertical"
    elif kind == "mani_v":
        from tasks.quadruped_manipulate_plate.quadruped_manipulate_plate_vertical import QuadrupedManipulatePlateVertical as T
        mangle = "_QuadrupedManipulatePlateVertical"
    else:
        from tasks.quadruped_manipulate_plate.quadruped_manipulate_plate import QuadrupedManipulatePlate as T
        mangle = "_QuadrupedManipulatePlate"
    from utils.math import transform_vectors
    t = object.__new__(T)
    dev = "cpu"
    t._device, t._num_envs, t._num_actions, t._num_observations, t._num_states = dev, N, 12, (88 if cc else 64), (64 if pc else 93)
    t._max_episode_length = {"loco_pc": 500, "mani_pc": 450}.get(kind, 500 if cc else 300)
    t._env = types.SimpleNamespace(_world=None)
    t._dr_randomizer = FakeDR()
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt)
    t.obs_buf, t.states_buf, t.rew_buf = z(N, 88 if cc else 64), z(N, 64 if pc else 93), z(N)
    t.reset_buf = torch.ones(N, dtype=torch.long); t.progress_buf = z(N, dt=torch.long); t.extras = {}
    t.last_actions, t.current_actions = z(N, 12), z(N, 12)
    t.last_base_tip_positions = z(N, 4, 3)
    t.default_base_tip_positions = torch.tensor([[-0.0937, 0.1223, -0.1774], [0.0937, 0.1408, -0.1773],
                                                 [-0.0937, -0.1408, -0.1773], [0.0937, -0.1223, -0.1774]]).repeat((N, 1, 1))
    vert = kind.endswith("_v")
    cpts = [[0.0, -0.115, -0.1853], [0.0, 0.115, -0.1853], [-0.115, 0.0, -0.1853], [0.115, 0.0, -0.1853]] if vert else \
        [[0.075, 0.1835, -0.04], [-0.075, 0.1835, -0.04], [0.075, -0.1835, -0.04], [-0.075, -0.1835, -0.04]]      # (…_vertical.py:122-127 / :132-137)
    corner = torch.cat([torch.tensor(c).repeat(N, 1) for c in cpts], dim=-1).view(N, 4, 3).to(torch.float32)
    setattr(t, mangle + "__corner_pos_robot", corner)
    t.goal_quaternions = z(N, 4)
    t.successes, t.consecutive_successes, t.goal_reset_buf = z(N, dt=torch.long), z(N, dt=torch.long), z(N, dt=torch.long)
    t.max_reset_counts = torch.tensor(2048, dtype=torch.long)
    t.num_successes = torch.tensor(0, dtype=torch.long); t.num_resets = torch.tensor(0, dtype=torch.long)
    t.success_rate = torch.tensor(0.0)
    t.randomization_buf = z(N, dt=torch.long)
    init_q = torch.tensor([-1.57, 1.57, 1.57, -1.57, -1.04, -2.09, 2.09, 1.04, 2.09, 1.04, -1.04, -2.09] + [1.37, -1.37] * 4)
    if vert:   # class-default pose of the vertical robot (robot/quadruped_robot.py:81-87); the vertical task files do not override it
        init_q = torch.tensor([0.0] * 4 + [0.35, -0.35] * 4 + [0.95, -0.95] * 4)
    if cc:   # class-default pose (robot/quadruped_robot.py:45-52); custom-controller state (…_custom_controller.py:175-195)
        init_q = torch.tensor([-1.2, 1.2, 1.2, -1.2, -1.22, -1.92, 1.92, 1.22, 1.92, 1.22, -1.22, -1.92] + [0.953, -0.953] * 4)
        t.current_joint_position_targets = init_q[:12].repeat(N, 1).clone()
        t.last_joint_position_targets = t.current_joint_position_targets.clone()
        t.current_joint_position_targets_se = torch.tensor([-1.2, 1.2, 1.2, -1.2, -1.57, 0.7, 1.57, 0.7, 1.57, 0.7, -1.57, 0.7]).repeat(N, 1)
        t.joint_position_target_se_upper = torch.tensor(T.max_joint_pos_swing_ext, dtype=torch.float32)
        t.joint_position_target_se_lower = torch.tensor(T.min_joint_pos_swing_ext, dtype=torch.float32)
        t.last_rot_dist = torch.zeros(N)
        if kind == "loco_cc":
            t.joint_positions_loco = init_q[:12].repeat(N, 1).clone(); t.joint_velocities_loco = torch.zeros(N, 12)
        else:
            t.joint_positions_mani = init_q[:12].repeat(N, 1).clone(); t.joint_velocities_mani = torch.zeros(N, 12)
    if pc:   # position-control state (…_position_control.py:181-199)
        t.current_joint_position_targets = init_q[:12].repeat(N, 1).clone()
        t.last_joint_position_targets = t.current_joint_position_targets.clone()
        t.current_joint_position_targets_se = torch.tensor([-1.57, 1.57, 1.57, -1.57, -1.57, 1.05, 1.57, 1.05, 1.57, 1.05, -1.57, 1.05]).repeat(N, 1)
        t.joint_position_target_se_upper = torch.tensor(T.max_joint_pos_swing_ext, dtype=torch.float32)
        t.joint_position_target_se_lower = torch.tensor(T.min_joint_pos_swing_ext, dtype=torch.float32)
        if kind == "loco_pc":
            t.joint_positions_loco = init_q[:12].repeat(N, 1).clone(); t.joint_velocities_loco = torch.zeros(N, 12)
        else:
            t.joint_positions_mani = init_q[:12].repeat(N, 1).clone(); t.joint_velocities_mani = torch.zeros(N, 12)
    zfix = 0.3 if kind == "mani_pc" else 0.0
    if kind in ("loco", "loco_cc", "loco_pc", "loco_v"):
        t.robot_locomotion = FakeRobot(N)
        t.pose_indicator_loco = FakeObj(N)
        t.default_joint_positions_loco = init_q.repeat((N, 1))
        t.default_robot_positions_loco = torch.tensor([0.0, 0.0, 0.35 if vert else (0.18 if cc else 0.14)]).repeat((N, 1))
        t.default_robot_quaternions_loco = torch.tensor([1.0, 0, 0, 0]).repeat((N, 1))
        t.default_pose_indicator_loco_positions = torch.tensor([[0.0, 0.0, 0.4 if vert else 0.3]]).repeat((N, 1))
    else:
        t.robot_manipulation = FakeRobot(N)
        t.pose_indicator_mani = FakeObj(N); t.obj = FakeObj(N)
        t.default_joint_positions_mani = init_q.repeat((N, 1))
        t.default_robot_positions_mani = torch.tensor([0.0, 0.0, zfix]).repeat((N, 1))
        t.default_robot_quaternions_mani = torch.tensor([0.0, 1.0, 0, 0]).repeat((N, 1))
        t.default_obj_positions_mani = torch.tensor([0.0, 0.0, 0.35 if vert else (0.44 if pc else (0.18 if cc else 0.14))]).repeat((N, 1))
        t.default_obj_quaternions_mani = torch.tensor([0.0, 1.0, 0, 0]).repeat((N, 1))
        t.default_pose_indicator_mani_positions = torch.tensor([[0.0, 0.0, 0.4 if vert else 0.3]]).repeat((N, 1))
        setattr(t, mangle + "__corner_pos_world",
                transform_vectors(t.default_robot_quaternions_mani, t.default_robot_positions_mani, corner, dev))
    return t


def gen_task(kind, N=32, T=26, seed=7):
    """Drive the reference task code through T post-physics evaluations on synthetic read-back states."""
    from omni.isaac.core.utils.torch.rotations import quat_conjugate, quat_mul
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    t = make_task(kind, N)
    cc = kind.endswith("_cc"); pc = kind.endswith("_pc"); base_kind = kind[:4]
    zfix = 0.3 if kind == "mani_pc" else 0.0
    robot = t.robot_locomotion if base_kind == "loco" else t.robot_manipulation
    vert = kind.endswith("_v")
    init_q = torch.tensor([-1.2, 1.2, 1.2, -1.2, -1.22, -1.92, 1.92, 1.22, 1.92, 1.22, -1.22, -1.92]) if cc else \
        torch.tensor([-1.57, 1.57, 1.57, -1.57, -1.04, -2.09, 2.09, 1.04, 2.09, 1.04, -1.04, -2.09])
    if vert: init_q = torch.tensor([0.0] * 4 + [0.35, -0.35] * 4)
    zb = 0.30 if vert else 0.13            # nominal height of the free body in the synthetic read-back (vertical: corners hang 0.185 below the base)
    rec = {k: [] for k in ("readback", "actions", "goal_rand", "obs", "states", "rew", "reset_buf", "goal_reset_buf",
                           "successes", "consecutive_successes", "progress_buf", "last_actions", "last_base_tip",
                           "goal_quaternions", "extras", "success_rate", "num_successes", "num_resets", "joint_reset", "torque", "se", "last_targets", "last_rot_dist")}
    extras_keys = None
    for step in range(T):
        actions = (torch.rand(N, 12, generator=g) * 2 - 1).clamp(-1, 1)
        if vert: actions = torch.zeros(N, 12)      # the vertical files zero their argument in place (…_vertical.py:204): feed zeros
        # ---- pre_physics_step (resets flagged envs; consumes the global RNG exactly like the reference)
        ids = t.reset_buf.nonzero(as_tuple=False).squeeze(-1)
        goal_rand = torch.zeros(N, 3)
        if len(ids) > 0:
            st = torch.get_rng_state()
            goal_rand[ids] = torch.rand((len(ids), 3))
            torch.set_rng_state(st)
        t.pre_physics_step(actions.clone())      # (the position-control files zero their argument in place, :261)
        # ---- synthetic read-back state, spread across the thresholds of Appendix D
        q = init_q.repeat(N, 1) + 0.25 * torch.randn(N, 12, generator=g)
        q[0:4, 5] = q[0:4, 4] - torch.tensor([0.40, 0.39, 2.55, 2.62])        # |dof3-dof2| around penalty/reset windows
        q[4:8, 0] = torch.tensor([-2.40, -2.45, 0.80, 0.90])                  # a1 dof1 windows
        q[8:12, 1] = torch.tensor([2.40, 2.45, -0.80, -0.90])                 # a2 dof1 mirrored windows
        if vert:                                                              # symmetric windows +-2.09 / +-2.26 on all four dof1 (…_vertical.py:84-87,452-459)
            q[4:8, 0] = torch.tensor([-2.10, -2.27, 2.08, 2.25]); q[8:12, 1] = torch.tensor([2.10, 2.27, -2.08, -2.25])
        qd = 2.0 * torch.randn(N, 12, generator=g)
        acc = 20.0 * torch.randn(N, 12, generator=g)
        pos = torch.cat((0.05 * torch.randn(N, 2, generator=g), zb + 0.02 * torch.randn(N, 1, generator=g)), dim=-1)
        quat = _rand_unit_quat(g, N, small=0.25)
        pos[12, 2] = 0.049; pos[13, 2] = 0.051                                  # base height threshold
        quat[14] = torch.tensor([0.0, 1.0, 0.0, 0.0])                           # upside down -> ground above robot
        lin = 0.3 * torch.randn(N, 3, generator=g); ang = 1.0 * torch.randn(N, 3, generator=g)
        tips = 0.15 * torch.randn(N, 4, 3, generator=g)
        knees = torch.cat((0.15 * torch.randn(N, 8, 2, generator=g), 0.10 + 0.03 * torch.randn(N, 8, 1, generator=g)), dim=-1)
        knees[15, 3, 2] = 0.0399; knees[16, 3, 2] = 0.0401
        if base_kind == "mani":
            # plate near its rest pose above the inverted robot: world z ~ 0.13, flipped about x
            flip = torch.tensor([0.0, 1.0, 0.0, 0.0]).repeat(N, 1)
            quat = quat_mul(_rand_unit_quat(g, N, small=0.2), flip)
            pos[12, 2] = 0.051; pos[13, 2] = 0.049
            quat[14] = torch.tensor([1.0, 0.05, 0.02, 0.01])   # un-flipped plate (exact w=0 in the robot frame is sign-ambiguous in scipy)
            pos[17, 2] = -0.01                                                  # plate below the robot base
            knees = torch.cat((0.15 * torch.randn(N, 8, 2, generator=g), 0.03 + 0.03 * torch.randn(N, 8, 1, generator=g)), dim=-1)
        # envs 20..27 track their goal for the whole sequence -> consecutive-success path + 600 bonus + goal reset
        track = torch.arange(20, 28)
        goal = t.goal_quaternions[track]
        small = _rand_unit_quat(g, len(track), small=0.02)
        if base_kind == "loco":
            quat[track] = quat_conjugate(quat_mul(small, goal))         # conj(bq) (x) conj(goal) ~ identity
        else:
            flip = torch.tensor([0.0, 1.0, 0.0, 0.0]).repeat(len(track), 1)
            quat[track] = quat_mul(flip, quat_mul(small, goal))         # conj(q_r) (x) pq = small (x) goal
        quat = quat / quat.norm(dim=-1, keepdim=True)
        pos[track] = torch.tensor([0.0, 0.0, zb]); q[track] = init_q
        knees[track, :, 2] = 0.1 if base_kind == "loco" else 0.02
        robot.joint_positions, robot.joint_velocities, robot.joint_accelerations = q, qd, acc
        robot.tip_positions, robot.knee_positions = tips, knees
        if zfix:   # the fixed robot (and everything measured relative to it) sits at z = zfix in the world
            pos[:, 2] += zfix; tips[:, :, 2] += zfix; knees[:, :, 2] += zfix
        if base_kind == "loco":
            robot.base_positions, robot.base_quaternions = pos, quat
            robot.base_linear_velocities, robot.base_angular_velocities = lin, ang
        else:
            t.obj.pos, t.obj.quat, t.obj.lin, t.obj.ang = pos, quat, lin, ang
        # ---- post_physics_step (rl_task.py:240-260)
        t.progress_buf[:] += 1
        t.get_observations(); t.calculate_metrics(); t.is_done()
        rb = torch.cat((q, qd, acc, pos, quat, lin, ang, tips.reshape(N, 12), knees.reshape(N, 24), torch.zeros(N, 2)), dim=-1)
        if cc or pc:
            rb = torch.cat((rb, t.torque), dim=-1)
            rec["torque"].append(t.torque.clone()); rec["se"].append(t.current_joint_position_targets_se.clone())
            rec["last_targets"].append(t.last_joint_position_targets.clone())
            if cc:
                rec["last_rot_dist"].append(t.last_rot_dist.clone())
        if extras_keys is None:
            extras_keys = sorted(t.extras.keys())
        rec["readback"].append(rb); rec["actions"].append(actions); rec["goal_rand"].append(goal_rand)
        rec["obs"].append(t.obs_buf.clone()); rec["states"].append(t.states_buf.clone()); rec["rew"].append(t.rew_buf.clone())
        rec["reset_buf"].append(t.reset_buf.clone()); rec["goal_reset_buf"].append(t.goal_reset_buf.clone())
        rec["successes"].append(t.successes.clone()); rec["consecutive_successes"].append(t.consecutive_successes.clone())
        rec["progress_buf"].append(t.progress_buf.clone()); rec["last_actions"].append(t.last_actions.clone())
        rec["last_base_tip"].append(t.last_base_tip_positions.reshape(N, 12).clone())
        rec["goal_quaternions"].append(t.goal_quaternions.clone())
        rec["extras"].append(torch.stack([torch.as_tensor(t.extras[k], dtype=torch.float32) for k in extras_keys]))
        rec["success_rate"].append(torch.as_tensor(t.success_rate, dtype=torch.float32).clone())
        rec["num_successes"].append(t.num_successes.clone()); rec["num_resets"].append(t.num_resets.clone())
        rec["joint_reset"].append((t.joint1_pos_reset + t.joint23_pos_reset).clone())
    out = {k: torch.stack(v).numpy() for k, v in rec.items() if len(v) > 0}
    out["extras_keys"] = np.array(extras_keys)
    return out


# ----------------------------------------------------------------------------- co-training tasks (a14, f-1)
def make_cotrain_task(kind, N):
    """JointLocomanipulation / JointLocomanipulationPositionControl on fakes: envs [0, N/2) locomotion, [N/2, N) manipulation."""
    pc = kind == "cotrain_pc"; vert = kind == "cotrain_v"
    if pc:
        from tasks.joint_train_locomanipulation.joint_locomanipulation_position_control import JointLocomanipulationPositionControl as T
        mangle = "_JointLocomanipulationPositionControl"
    elif vert:
        from tasks.joint_train_locomanipulation.joint_locomanipulation_vertical import JointLocomanipulationVertical as T
        mangle = "_JointLocomanipulationVertical"
    else:
        from tasks.joint_train_locomanipulation.joint_locomanipulation import JointLocomanipulation as T
        mangle = "_JointLocomanipulation"
    from utils.math import transform_vectors
    h = N // 2
    t = object.__new__(T)
    t.RECORD_JOINT = False          # the class default dumps joint trajectories to an absolute path of the authors' machine and exits
    t._device, t._num_envs, t._num_actions, t._num_observations, t._num_states = "cpu", N, 12, 64, 64
    t._single_task_num_envs = h
    t._max_episode_length = 300
    t._env = types.SimpleNamespace(_world=None)
    t._dr_randomizer = FakeDR()
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt)
    t.obs_buf, t.states_buf, t.rew_buf = z(N, 64), z(N, 64), z(N)
    t.reset_buf = torch.ones(N, dtype=torch.long); t.progress_buf = z(N, dt=torch.long); t.extras = {}
    t.last_actions, t.current_actions = z(N, 12), z(N, 12)
    t.last_base_tip_positions = z(N, 4, 3)
    t.default_base_tip_positions = torch.tensor([[-0.0937, 0.1223, -0.1774], [0.0937, 0.1408, -0.1773],
                                                 [-0.0937, -0.1408, -0.1773], [0.0937, -0.1223, -0.1774]]).repeat((N, 1, 1))
    cpts = [[0.0, -0.115, -0.1853], [0.0, 0.115, -0.1853], [-0.115, 0.0, -0.1853], [0.115, 0.0, -0.1853]] if vert else \
        [[0.075, 0.1835, -0.04], [-0.075, 0.1835, -0.04], [0.075, -0.1835, -0.04], [-0.075, -0.1835, -0.04]]      # (joint_locomanipulation_vertical.py:147-152 / :178-183)
    corner = torch.cat([torch.tensor(c).repeat(h, 1) for c in cpts], dim=-1).view(h, 4, 3).to(torch.float32)
    setattr(t, mangle + "__corner_pos_robot", corner)
    t.goal_quaternions = z(N, 4)
    t.successes, t.consecutive_successes, t.goal_reset_buf = z(N, dt=torch.long), z(N, dt=torch.long), z(N, dt=torch.long)
    for sfx in ("", "_loco", "_mani"):
        setattr(t, "max_reset_counts" + sfx, torch.tensor(2048, dtype=torch.long))
        setattr(t, "num_successes" + sfx, torch.tensor(0, dtype=torch.long)); setattr(t, "num_resets" + sfx, torch.tensor(0, dtype=torch.long))
        setattr(t, "success_rate" + sfx, torch.tensor(0.0))
    t.randomization_buf = z(N, dt=torch.long)
    init_q = torch.tensor([-1.57, 1.57, 1.57, -1.57, -1.04, -2.09, 2.09, 1.04, 2.09, 1.04, -1.04, -2.09] + [1.37, -1.37] * 4)
    if vert: init_q = torch.tensor([0.0] * 4 + [0.35, -0.35] * 4 + [0.95, -0.95] * 4)          # robot/quadruped_robot.py:81-87
    t.robot_locomotion, t.robot_manipulation = FakeRobot(h), FakeRobot(h)
    t.pose_indicator_loco, t.pose_indicator_mani, t.obj = FakeObj(h), FakeObj(h), FakeObj(h)
    t.default_joint_positions_loco = init_q.repeat((h, 1)); t.default_joint_positions_mani = init_q.repeat((h, 1))
    t.default_robot_positions_loco = torch.tensor([0.0, 0.0, 0.35 if vert else 0.14]).repeat((h, 1))
    t.default_robot_quaternions_loco = torch.tensor([1.0, 0, 0, 0]).repeat((h, 1))
    t.default_robot_positions_mani = torch.tensor([0.0, 0.0, 0.6 if vert else 0.5]).repeat((h, 1))           # joint_locomanipulation.py:139, …_vertical.py:108
    t.default_robot_quaternions_mani = torch.tensor([0.0, 1.0, 0, 0]).repeat((h, 1))
    t.default_obj_positions_mani = torch.tensor([0.0, 0.0, 0.95 if vert else (0.64 if pc else 0.68)]).repeat((h, 1))
    t.default_obj_quaternions_mani = torch.tensor([0.0, 1.0, 0, 0]).repeat((h, 1))
    t.default_pose_indicator_loco_positions = torch.tensor([[0.0, 0.0, 0.5 if vert else 0.3]]).repeat((h, 1))
    t.default_pose_indicator_mani_positions = torch.tensor([[0.0, 0.0, 1.1 if vert else 0.8]]).repeat((h, 1))
    setattr(t, mangle + "__corner_pos_world", transform_vectors(t.default_robot_quaternions_mani, t.default_robot_positions_mani, corner, "cpu"))
    if pc:   # joint_locomanipulation_position_control.py:218-243
        t.joint_positions_combined = init_q[:12].repeat(N, 1).clone(); t.joint_velocities_combined = z(N, 12)
        t.default_joint_positions_combined = init_q[:12].repeat(N, 1).clone()
        t.current_joint_position_targets = init_q[:12].repeat(N, 1).clone()
        t.last_joint_position_targets = t.current_joint_position_targets.clone()
        t.current_joint_position_targets_se = torch.tensor([-1.57, 1.57, 1.57, -1.57, -1.57, 1.05, 1.57, 1.05, 1.57, 1.05, -1.57, 1.05]).repeat(N, 1)
        t.joint_position_target_se_upper = torch.tensor(T.max_joint_pos_swing_ext, dtype=torch.float32)
        t.joint_position_target_se_lower = torch.tensor(T.min_joint_pos_swing_ext, dtype=torch.float32)
    return t


def _synth_half(g, mani, n, init_q, goal, zfix, vert=False):
    """Synthetic read-back state of one half (n envs), straddling the thresholds of Appendix D; envs n-6.. track their goal."""
    from omni.isaac.core.utils.torch.rotations import quat_conjugate, quat_mul
    q = init_q.repeat(n, 1) + 0.25 * torch.randn(n, 12, generator=g)
    q[0:2, 5] = q[0:2, 4] - torch.tensor([0.40, 2.62]); q[2:4, 0] = torch.tensor([-2.40, 0.90]); q[4:6, 1] = torch.tensor([2.40, -0.90])
    if vert: q[2:4, 0] = torch.tensor([-2.27, 2.10]); q[4:6, 1] = torch.tensor([2.27, -2.10])      # symmetric dof1 windows of the vertical files
    zb = 0.30 if vert else 0.13
    qd = 2.0 * torch.randn(n, 12, generator=g); acc = 20.0 * torch.randn(n, 12, generator=g)
    pos = torch.cat((0.05 * torch.randn(n, 2, generator=g), zb + 0.02 * torch.randn(n, 1, generator=g)), dim=-1)
    quat = _rand_unit_quat(g, n, small=0.25)
    lin = 0.3 * torch.randn(n, 3, generator=g); ang = 1.0 * torch.randn(n, 3, generator=g)
    tips = 0.15 * torch.randn(n, 4, 3, generator=g)
    kz = 0.03 if mani else 0.10
    knees = torch.cat((0.15 * torch.randn(n, 8, 2, generator=g), kz + 0.03 * torch.randn(n, 8, 1, generator=g)), dim=-1)
    flip = torch.tensor([0.0, 1.0, 0.0, 0.0])
    if mani:
        quat = quat_mul(_rand_unit_quat(g, n, small=0.2), flip.repeat(n, 1))
        pos[6, 2] = 0.051; pos[7, 2] = 0.049; pos[8, 2] = -0.01
    else:
        pos[6, 2] = 0.049; pos[7, 2] = 0.051
        quat[8] = flip
        knees[9, 3, 2] = 0.0399; knees[10, 3, 2] = 0.0401
    track = torch.arange(n - 6, n)
    small = _rand_unit_quat(g, len(track), small=0.02)
    if mani:
        quat[track] = quat_mul(flip.repeat(len(track), 1), quat_mul(small, goal[track]))
    else:
        quat[track] = quat_conjugate(quat_mul(small, goal[track]))
    quat = quat / quat.norm(dim=-1, keepdim=True)
    pos[track] = torch.tensor([0.0, 0.0, zb]); q[track] = init_q
    knees[track, :, 2] = 0.02 if mani else 0.1
    knees[track, :, :2] *= 0.2          # keep the tracking envs' knees clear of the tilted plate so they reach the bonus
    pos[:, 2] += zfix; tips[:, :, 2] += zfix; knees[:, :, 2] += zfix
    return q, qd, acc, pos, quat, lin, ang, tips, knees


def gen_cotrain(kind, N=32, T=30, seed=13):
    """The co-training tasks through T post-physics evaluations (same protocol as gen_task; both halves in one call)."""
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    t = make_cotrain_task(kind, N); h = N // 2; pc = kind == "cotrain_pc"; vert = kind == "cotrain_v"
    init_q = torch.tensor([-1.57, 1.57, 1.57, -1.57, -1.04, -2.09, 2.09, 1.04, 2.09, 1.04, -1.04, -2.09])
    if vert: init_q = torch.tensor([0.0] * 4 + [0.35, -0.35] * 4)
    rec = {k: [] for k in ("readback", "actions", "goal_rand", "obs", "rew", "reset_buf", "goal_reset_buf", "successes", "consecutive_successes",
                           "progress_buf", "last_actions", "last_base_tip", "goal_quaternions", "extras", "joint_reset", "se", "last_targets", "torque")}
    extras_keys = None
    for step in range(T):
        actions = (torch.rand(N, 12, generator=g) * 2 - 1).clamp(-1, 1)
        if vert: actions = torch.zeros(N, 12)      # the vertical file zeroes both halves of its argument in place (:258-259)
        # reset_idx draws the locomotion goals first, then the manipulation goals (joint_locomanipulation.py:319-334)
        ids = t.reset_buf.nonzero(as_tuple=False).squeeze(-1)
        goal_rand = torch.zeros(N, 3)
        if len(ids) > 0:
            st = torch.get_rng_state()
            il, im = ids[ids < h], ids[ids >= h]
            goal_rand[il] = torch.rand((len(il), 3)); goal_rand[im] = torch.rand((len(im), 3))
            torch.set_rng_state(st)
        t.pre_physics_step(actions.clone())
        ql, qdl, accl, posl, quatl, linl, angl, tipsl, kneesl = _synth_half(g, False, h, init_q, t.goal_quaternions[:h], 0.0, vert)
        qm, qdm, accm, posm, quatm, linm, angm, tipsm, kneesm = _synth_half(g, True, h, init_q, t.goal_quaternions[h:], 0.6 if vert else 0.5, vert)
        rl, rm = t.robot_locomotion, t.robot_manipulation
        rl.joint_positions, rl.joint_velocities, rl.joint_accelerations, rl.tip_positions, rl.knee_positions = ql, qdl, accl, tipsl, kneesl
        rl.base_positions, rl.base_quaternions, rl.base_linear_velocities, rl.base_angular_velocities = posl, quatl, linl, angl
        rm.joint_positions, rm.joint_velocities, rm.joint_accelerations, rm.tip_positions, rm.knee_positions = qm, qdm, accm, tipsm, kneesm
        t.obj.pos, t.obj.quat, t.obj.lin, t.obj.ang = posm, quatm, linm, angm
        t.progress_buf[:] += 1
        t.get_observations(); t.calculate_metrics(); t.is_done()
        cat = lambda a, b: torch.cat((a, b), dim=0)
        rb = torch.cat((cat(ql, qm), cat(qdl, qdm), cat(accl, accm), cat(posl, posm), cat(quatl, quatm), cat(linl, linm), cat(angl, angm),
                        cat(tipsl, tipsm).reshape(N, 12), cat(kneesl, kneesm).reshape(N, 24), torch.zeros(N, 2),
                        t.torque if pc else torch.zeros(N, 12)), dim=-1)
        if pc:
            rec["se"].append(t.current_joint_position_targets_se.clone()); rec["last_targets"].append(t.last_joint_position_targets.clone())
            rec["torque"].append(t.torque.clone())
        if extras_keys is None:
            extras_keys = sorted(t.extras.keys())
        rec["readback"].append(rb); rec["actions"].append(actions); rec["goal_rand"].append(goal_rand)
        rec["obs"].append(t.obs_buf.clone()); rec["rew"].append(t.rew_buf.clone())
        rec["reset_buf"].append(t.reset_buf.clone()); rec["goal_reset_buf"].append(t.goal_reset_buf.clone())
        rec["successes"].append(t.successes.clone()); rec["consecutive_successes"].append(t.consecutive_successes.clone())
        rec["progress_buf"].append(t.progress_buf.clone()); rec["last_actions"].append(t.last_actions.clone())
        rec["last_base_tip"].append(t.last_base_tip_positions.reshape(N, 12).clone())
        rec["goal_quaternions"].append(t.goal_quaternions.clone())
        rec["extras"].append(torch.stack([torch.as_tensor(t.extras[k], dtype=torch.float32) for k in extras_keys]))
        rec["joint_reset"].append((t.joint1_pos_reset + t.joint23_pos_reset).clone())
    out = {k: torch.stack(v).numpy() for k, v in rec.items() if len(v) > 0}
    out["extras_keys"] = np.array(extras_keys)
    out["counters"] = np.array([int(getattr(t, n + s)) for s in ("", "_loco", "_mani") for n in ("num_successes", "num_resets")])
    return out


# ----------------------------------------------------------------------------- domain randomisation noise (f-3)
DR_NOISE_CASES = [   # (on_reset, on_interval) entries in the YAML schema of cfg/task/QuadrupedPoseControl.yaml:116-135; std 0 / lo == hi
    ({"operation": "additive", "distribution": "gaussian", "distribution_parameters": [0.25, 0.0]},
     {"frequency_interval": 1, "operation": "additive", "distribution": "gaussian", "distribution_parameters": [0.5, 0.0]}),
    ({"operation": "additive", "distribution": "gaussian", "distribution_parameters": [-0.125, 0.0]},
     {"frequency_interval": 3, "operation": "scaling", "distribution": "uniform", "distribution_parameters": [1.5, 1.5]}),
    ({"operation": "scaling", "distribution": "uniform", "distribution_parameters": [0.75, 0.75]},
     {"frequency_interval": 4, "operation": "additive", "distribution": "normal", "distribution_parameters": [2.0, 0.0]}),
    (None, {"frequency_interval": 2, "operation": "additive", "distribution": "gaussian", "distribution_parameters": [1.0, 0.0]}),
    ({"operation": "scaling", "distribution": "loguniform", "distribution_parameters": [2.0, 2.0]}, None),
]


def gen_dr_noise(seed=17, N=16, T=24):
    """The reference's own Randomizer.apply_observations_randomization / apply_actions_randomization (randomize.py:212-306) with
    degenerate distributions (std 0, low == high): what is left is the structure - per-env counters, reset handling, the order
    correlated -> uncorrelated, additive / scaling - which is what the engine must reproduce; its random stream is its own."""
    from utils.domain_randomization.randomize import Randomizer
    g = torch.Generator().manual_seed(seed)
    out = {}
    for ci, (on_reset, on_interval) in enumerate(DR_NOISE_CASES):
        for kind, D in (("observations", 9), ("actions", 12)):
            params = {}
            if on_reset is not None: params["on_reset"] = dict(on_reset)
            if on_interval is not None: params["on_interval"] = dict(on_interval)
            task_cfg = {"env": {"numEnvs": N}, "domain_randomization": {"randomize": True, "min_frequency": 1,
                        "randomization_params": {kind: params}}}
            r = Randomizer(types.SimpleNamespace(task_config=task_cfg, config={"rl_device": "cpu", "seed": 0}))
            task = types.SimpleNamespace(num_observations=D, num_actions=D, randomize_observations=False, randomize_actions=False)
            (r._set_up_observations_randomization if kind == "observations" else r._set_up_actions_randomization)(task)
            assert task.randomize_observations or task.randomize_actions
            ins, flags, outs, counters = [], [], [], []
            for t in range(T):
                buf = torch.randn(N, D, generator=g)
                rf = torch.ones(N, dtype=torch.long) if t == 0 else (torch.rand(N, generator=g) < 0.2).long()
                ins.append(buf.clone()); flags.append(rf.clone())
                fn = r.apply_observations_randomization if kind == "observations" else r.apply_actions_randomization
                o = fn(buf, rf) if kind == "observations" else fn(actions=buf, reset_buf=rf)
                outs.append(o.clone())
                counters.append((r._observations_counter_buffer if kind == "observations" else r._actions_counter_buffer).clone().long())
            out[f"c{ci}_{kind}_in"] = torch.stack(ins).numpy(); out[f"c{ci}_{kind}_reset"] = torch.stack(flags).numpy()
            out[f"c{ci}_{kind}_out"] = torch.stack(outs).numpy(); out[f"c{ci}_{kind}_counter"] = torch.stack(counters).numpy()
    return out


def gen_math(seed=3, N=64):
    from utils.math import (inverse_rotate_orientations, inverse_transform_vectors, rand_quaternions,
                            rotate_orientations, transform_vectors)
    g = torch.Generator().manual_seed(seed)
    q = _rand_unit_quat(g, N); r = _rand_unit_quat(g, N)
    t = torch.randn(N, 3, generator=g); V = torch.randn(N, 5, 3, generator=g)
    torch.manual_seed(seed)
    st = torch.get_rng_state(); u = torch.rand((N, 3)); torch.set_rng_state(st)
    rq = rand_quaternions(N, -0.4, 0.4, -0.4, 0.4, -1.57, 1.57, "cpu")
    out = dict(q=q, r=r, t=t, V=V, transform=transform_vectors(q, t, V, "cpu"),
               inverse_transform=inverse_transform_vectors(q, t, V, "cpu"),
               rotate=rotate_orientations(r, q, "cpu"), inverse_rotate=inverse_rotate_orientations(r, q, "cpu"),
               rand_u=u, rand_quat=rq)
    # the reference's only self-check input (utils/math.py:212-216)
    out["selfcheck"] = inverse_rotate_orientations(torch.tensor([-0.5, -0.5, 0.5, 0.5]).repeat(2, 1),
                                                   torch.tensor([0.7071, 0, 0, 0.7071]).repeat(2, 1))
    return {k: v.numpy() for k, v in out.items()}


def gen_take_action(seed=5, N=8):
    """RobotOmni.take_action scaling for the three control modes (robot/base/robot.py:444-461)."""
    from robot.base.robot import RobotOmni
    g = torch.Generator().manual_seed(seed)
    a = torch.rand(N, 12, generator=g) * 2 - 1
    out = {"actions": a.numpy()}

    class Arti:
        num_dof = 20

        def set_joint_position_targets(self, x, joint_indices=None):
            self.got = x

        set_joint_velocity_targets = set_joint_position_targets

        def set_joint_efforts(self, x):
            self.got = x

    for mode in ("position", "velocity", "effort"):
        r = object.__new__(RobotOmni)
        r.num_envs, r.device = N, "cpu"
        r.robot_description = types.SimpleNamespace(control_mode=mode)
        r._robot_articulation = Arti()
        r._omni_dof_indices = torch.arange(12)
        r._positions_upper = torch.tensor([np.pi], dtype=torch.float32).repeat(1, 12); r._positions_lower = -r._positions_upper
        r._velocity_upper = torch.full((12,), 3.0); r._velocity_lower = -r._velocity_upper
        r._torque_upper = torch.full((12,), 1.5); r._torque_lower = -r._torque_upper
        r.take_action(a.clone())
        out[mode] = r._robot_articulation.got.numpy()
    return out


def gen_gnn(seed=11, B=40):
    """The reference's GraphNet / Action_Layer / Value_Layer (scripts/graph_model_orebot_ov.py) on random inputs with seeded
    weights.  torch_scatter (third-party, absent) is replaced by a stand-in with its documented semantics:
    scatter(src, index, dim, dim_size, reduce='max') = per-index maximum, 0 where an index receives nothing."""
    ts = types.ModuleType("torch_scatter")

    def scatter(src, index, dim=-1, dim_size=None, reduce="sum"):
        assert reduce == "max" and dim == -2
        out = torch.zeros(src.shape[:-2] + (dim_size, src.shape[-1]), dtype=src.dtype)
        idx = index.view(1, -1, 1).expand(src.shape)
        return out.scatter_reduce(-2, idx, src, reduce="amax", include_self=False)

    ts.scatter = scatter
    sys.modules["torch_scatter"] = ts
    sys.path.insert(0, os.path.join(REF_RL, "scripts"))
    import graph_model_orebot_ov as G
    torch.manual_seed(seed)
    net = G.GraphNet(hidden_features=32, out_features=32); act = G.Action_Layer(32, 12); val = G.Value_Layer(32)
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(B, 64, generator=g) * 1.5
    with torch.no_grad():
        h = net(obs); mean = act(h); value = val(h)
    out = {"obs": obs, "h": h, "mean": mean, "value": value, "edge_index": net.edge_index}
    for k, v in net.state_dict().items():
        out["net." + k] = v
    for k, v in act.state_dict().items():
        out["mean_layer." + k] = v
    for k, v in val.state_dict().items():
        out["value_layer." + k] = v
    return {k: v.numpy() for k, v in out.items()}


def main():
    assert os.path.isdir(REF_RL), "reference tree not present: golden vectors can only be regenerated in the build container"
    _install_placeholders()
    sys.path[:0] = [REF_RL, os.path.dirname(REF_RL)]
    os.makedirs(OUT, exist_ok=True)
    for kind in ("loco", "mani", "loco_cc", "mani_cc", "loco_pc", "mani_pc", "loco_v", "mani_v"):
        d = gen_task(kind, T=26 if kind in ("loco", "mani", "loco_v", "mani_v") else 32)
        np.savez_compressed(os.path.join(OUT, f"task_{kind}.npz"), **d)
        print(kind, {k: v.shape for k, v in d.items() if k in ("obs", "states", "rew", "extras")},
              "resets/step", d["reset_buf"].sum(1)[:8], "max consec", d["consecutive_successes"].max(),
              "bonus steps", int((d["rew"] > 300).sum()))
    for kind in ("cotrain", "cotrain_pc", "cotrain_v"):
        d = gen_cotrain(kind)
        np.savez_compressed(os.path.join(OUT, f"task_{kind}.npz"), **d)
        print(kind, {k: v.shape for k, v in d.items() if k in ("obs", "rew", "extras")}, "resets/step", d["reset_buf"].sum(1)[:8],
              "max consec", d["consecutive_successes"].max(), "bonus steps", int((d["rew"] > 300).sum()), "counters", d["counters"])
    np.savez_compressed(os.path.join(OUT, "dr_noise.npz"), **gen_dr_noise())
    np.savez_compressed(os.path.join(OUT, "math.npz"), **gen_math())
    np.savez_compressed(os.path.join(OUT, "take_action.npz"), **gen_take_action())
    np.savez_compressed(os.path.join(OUT, "gnn.npz"), **gen_gnn())
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
