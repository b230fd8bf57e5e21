"""Numeric contract of the engine: every simulation / task constant in one place.

Values and where they come from in the reference:
  dt 0.0083, controlFrequencyInv 4, max_episode_length 300, gravity -9.81
      RobotLearning/omniisaacgymenvs/cfg/task/QuadrupedPoseControl.yaml:13,18,19,21
  drive: velocity mode, kp 0, kd 100, max effort 1.5, velocity limit 3.0 (action scale)
      robot/quadruped_robot.py:42,61-64 ; robot/base/robot.py:252-254,452-454
  reward / termination constants
      tasks/quadruped_pose_control_tasks/quadruped_pose_control.py:27-87,127-137
  vertical deltas
      tasks/quadruped_pose_control_tasks/quadruped_pose_control_vertical.py:84-87,123-126

Contact parameters (mu, baumgarte, max_depen_vel, pgs_iters) belong to THIS engine's
contact model; the reference delegates contact to PhysX (TGS, 16+2
iterations, contact_offset 0.005, max depenetration velocity 100 -- YAML :41-50) whose
algorithm is not reproducible here.  Values chosen on the evidence of the reference's recorded PhysX joint
trajectories (tests/golden/npy_traj.npz; tests/npy_replay_evidence.py prints the tables):
  tip_radius 0.005 (round 2): the foot is the 5 mm hemisphere that ends the long distal link's collision mesh
  tau_max = max_effort / dt (round 2; PARITY UNPINNED): the recorded joints follow their velocity commands through full-speed
      reversals within one control period, which a 1.5 N m torque clamp on this 2.26 kg robot cannot do in this engine (section 2.1's
      negative control); the drive limit is therefore read as PhysX's per-step IMPULSE limit (1.5 N m s per 0.0083 s step = 180.7 N m,
      never binding).  Whether the reference's Isaac build raised PxArticulationFlag::eDRIVE_LIMITS_ARE_FORCES cannot be checked here,
      and row 0 of the same recordings (the joints give way by 3e-3 ... 1.25e-2 rad while the scene settles) shows what the limit really is:
      an impulse limit of 1.5 N m x dt PER SOLVER ITERATION (16 per step), which binds only on loads that arrive inside one iteration - the
      plate scene's start-in-penetration - and amounts to 24 N m on sustained ones (reproduced in a sub-iterated oracle run; round 4
      tried the drive as rows of the contact iteration with that per-iteration bound - 17 variants, none reproduces the recordings' later rows,
      tests/drive_rows_experiment.py).  The torque reading (tau_max = 1.5 for the velocity drive) is the replay test's negative control; its YAML
      switch `drive_limits_are_impulses` and bench leg were removed in round 4 (an unpinned reading at +36 % step time is not a feature):
      `sim.engine.tau_max` sets an explicit limit for experiments.
  friction CONE, mu = 0.8 x the nominal coefficient, pgs_iters 8 (round 3): with the axis-aligned friction pyramid of rounds 1-2 the replay's
      orientation outcomes depended on the sweep count (1 of 4 locomotion files entered PhysX's success window at 8 sweeps, 3 of 4 from 12 on)
      and missed PhysX's rows by up to 6; with the isotropic cone |lam_t| <= mu lam_n they are THE SAME FROM 2 TO 128 SWEEPS and land on
      PhysX's rows: locomotion enters on rows 10, 12, 10, 12 (PhysX 9, 15, 10, 12), manipulation on 12, 13, 14 (PhysX 12, 12, 13), `test` falls
      on PhysX's row, 97 of PhysX's 119 window rows are shared (pyramid: 59).  The coefficient: all seven episodes enter for mu in
      [0.75, 0.85] x nominal, five of seven at the nominal 1.0 (tables in DESIGN.md 2.1) - a fitted effective coefficient, parity unpinned (PhysX's
      friction rows are part of an unconverged iterative solve; a link material at PhysX's default 0.5 averaged with the ground's 1.0 would
      give 0.75, the robot's USD is not in the reference).  Sweep counts: see PGS_ITERS_* below.
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Dict, List


# Contact sweeps per solve, chosen on convergence (DESIGN.md 2.1, table `convergence` of tests/npy_replay_evidence.py): the smallest count of 4, 8,
# 16 ... whose contact velocity change is within 1 % (median) / 20 % (90th percentile) of the 128-sweep solve's on states from random-action
# rollouts.  The rigid velocity drives couple the four feet through the base (ground: 8 sweeps: 0.6 % / 12 %; 4 sweeps: 6 % / 33 %); the plate
# (4: 0.0 % / 0.9 %) and the soft PD actuators of the f-1 families (4: 0.1 % / 0.8 %) converge faster.  The replayed PhysX episodes do not tell
# 2 sweeps from 128 (same section).
PGS_ITERS_GROUND, PGS_ITERS_PLATE, PGS_ITERS_PD = 8, 4, 4
FRICTION_SCALE = 0.8             # effective / nominal friction coefficient (see the module docstring)
MODE_LOCO = 0    # free base on a ground plane
MODE_MANI = 1    # fixed (inverted) base + free plate
DRIVE_VELOCITY, DRIVE_POSITION, DRIVE_EFFORT = 0, 1, 2


def _f(x):
    return field(default_factory=lambda: list(x))


# ---- domain randomisation (YAML schema of cfg/task/QuadrupedPoseControl.yaml:102-173)
DR_OBS_RESET, DR_OBS_INTERVAL, DR_ACT_RESET, DR_ACT_INTERVAL, DR_GRAVITY, DR_BASE_FORCE, DR_MAX_EFFORT, DR_MAX_VELOCITY, DR_JOINT_DAMPING = range(9)
DR_CHANNELS = 9
DR_OPERATIONS = {"additive": 0, "scaling": 1, "direct": 2}
DR_DISTRIBUTIONS = {"gaussian": 0, "normal": 0, "uniform": 1, "loguniform": 2, "log_uniform": 2}


@dataclass
class DRChannel:
    """One randomised quantity: operation / distribution / distribution_parameters of the YAML; `interval` = frequency_interval
    for on_interval entries, 0 for on_reset entries.  p0 / p1 = gaussian mean / std or uniform low / high (3 components for vectors)."""
    enabled: int = 0
    operation: int = 0
    distribution: int = 0
    interval: int = 0
    p0: List[float] = _f([0.0, 0.0, 0.0])
    p1: List[float] = _f([0.0, 0.0, 0.0])


def _no_dr():
    return [DRChannel() for _ in range(DR_CHANNELS)]


@dataclass
class EngineParams:
    # ---- physics
    dt: float = 0.0083
    substeps: int = 4
    pgs_iters: int = -1                 # contact sweeps per solve; -1 = by contact surface / actuator (PGS_ITERS_GROUND / _PLATE / _PD: 8 / 4 / 4, see above)
    gravity: float = 9.81
    kd: float = 100.0
    max_effort: float = 1.5             # ArticulationView.set_max_efforts (robot.py:347-355); host-side only, tau_max is what the engine reads
    tau_max: float = -1.0               # -1 = max_effort / dt: the velocity drive's limit read as an impulse limit per step (see above); the PD families set 1.5
    act_scale: float = 3.0
    mu: float = FRICTION_SCALE * 1.0     # effective friction coefficient of the foot contacts: FRICTION_SCALE x the nominal (combined) 1.0
    pyramid: int = 0                    # ORACLE-ONLY evidence switch: 1 = the axis-aligned friction pyramid of rounds 1-2 (the kernel implements the cone only)
    tip_radius: float = 0.005
    baumgarte: float = 0.2
    max_depen_vel: float = 1.0
    max_joint_vel: float = 7.853981633974483     # 450 deg/s: physxJoint:maxJointVelocity set by Design/Scripts/config_module_joints.py:11,61-69
    mode: int = MODE_LOCO
    fixed_base_pos: List[float] = _f([0.0, 0.0, 0.0])
    fixed_base_quat: List[float] = _f([0.0, 1.0, 0.0, 0.0])
    plate_mass: float = 2.4                              # Design/ObjectURDF/plate.urdf:5-23
    plate_com: List[float] = _f([0.0, 0.0, 0.004])
    plate_inertia: List[float] = _f([0.05, 0.05, 0.1])
    plate_half: List[float] = _f([0.25, 0.25, 0.004])
    plate_center: List[float] = _f([0.0, 0.0, 0.004])
    # ---- reset
    init_q: List[float] = _f([-1.57, 1.57, 1.57, -1.57, -1.04, -2.09, 2.09, 1.04, 2.09, 1.04, -1.04, -2.09])
    init_base_pos: List[float] = _f([0.0, 0.0, 0.14])
    init_base_quat: List[float] = _f([1.0, 0.0, 0.0, 0.0])
    init_plate_pos: List[float] = _f([0.0, 0.0, 0.14])
    init_plate_quat: List[float] = _f([0.0, 1.0, 0.0, 0.0])
    default_tip: List[float] = _f([-0.0937, 0.1223, -0.1774, 0.0937, 0.1408, -0.1773,
                                   -0.0937, -0.1408, -0.1773, 0.0937, -0.1223, -0.1774])
    goal_lo: List[float] = _f([-0.4, -0.4, -1.57])
    goal_hi: List[float] = _f([0.4, 0.4, 1.57])
    # ---- observation scales
    s_pos: float = 5.0
    s_lin: float = 2.0
    s_ang: float = 0.25
    s_q: float = 0.3
    s_qd: float = 0.3
    # ---- reward
    quat_scale: float = 0.5
    rot_eps: float = 0.1
    trans_scale: float = -2.5
    acc_scale: float = -0.0005
    rate_scale: float = -0.02
    bonus: float = 600.0
    limit_pen: float = -5.0
    fall_pen: float = 0.0
    succ_thresh: float = 0.15
    max_consec: int = 15
    max_episode: int = 300
    d23_pen: List[float] = _f([0.43, 2.53])
    d23_rst: List[float] = _f([0.384, 2.61])
    # dof1 windows per limb a1..a4: a1,a4 [min,max]; a2,a3 mirrored (quadruped_pose_control.py:482-501)
    d1_pen: List[List[float]] = field(default_factory=lambda: [[-2.35, 0.78], [-0.78, 2.35], [-0.78, 2.35], [-2.35, 0.78]])
    d1_rst: List[List[float]] = field(default_factory=lambda: [[-2.44, 0.87], [-0.87, 2.44], [-0.87, 2.44], [-2.44, 0.87]])
    h_base: float = 0.05
    h_corner: float = 0.01
    h_knee: float = 0.04
    corner: List[List[float]] = field(default_factory=lambda: [[0.075, 0.1835, -0.04], [-0.075, 0.1835, -0.04],
                                                               [0.075, -0.1835, -0.04], [-0.075, -0.1835, -0.04]])
    # ---- custom-controller task family (quadruped_pose_control_custom_controller.py:24-52,88-97)
    variant: int = 0                    # 0 velocity-drive tasks, 1 custom-controller (PD actuator, swing/extension actions, obs 88),
                                        # 2 position-control (same actuator / actions; obs 64 with the targets in place of the actions; base reward)
    num_obs: int = 64
    pd_kp: float = 4.5
    joint_damping: float = 0.0
    act_scale_se: float = 0.1
    se_lo: List[float] = _f([-2.35, -0.78, -0.78, -2.35, -2.09, 0.52, 1.05, 0.52, 1.05, 0.52, -2.09, 0.52])
    se_hi: List[float] = _f([0.78, 2.35, 2.35, 0.78, -1.05, 2.09, 2.09, 2.09, 2.09, 2.09, -1.05, 2.09])
    init_se: List[float] = _f([-1.2, 1.2, 1.2, -1.2, -1.57, 0.7, 1.57, 0.7, 1.57, 0.7, -1.57, 0.7])
    torque_div: float = 4.0
    power_scale: float = -0.02
    target_err_scale: float = -0.05
    rot_dec_scale: float = 0.0
    rot_dec_thresh: float = 0.3
    acc_substeps: int = 1               # variants 1/2: the joint acceleration spans the trailing controlFrequencyInv sub-steps (robot.py:289-291)
    cc_update_last_tgt: int = 1         # loco: last targets follow the targets (:723-725); the mani variant never updates them after reset
    # ---- domain randomisation (off in every YAML of the measured path)
    dr_enabled: int = 0
    dr_min_frequency: int = 1
    dr: List[DRChannel] = field(default_factory=_no_dr)
    # ---- RobotOmni.take_action control mode (robot/base/robot.py:444-461), variant 0 only: 0 velocity (every task of the path), 1 position
    # (target a * act_scale rad with act_scale = pi, PD gains pd_kp / kd), 2 effort (torque a * act_scale N m with act_scale = torque limit)
    drive_mode: int = 0
    # PD-actuator families (variants 1 / 2).  0: the joints on the +-tau_max limit are those whose PD torque on the pre-step state is outside it (the
    # reference's explicit clamp, ...custom_controller.py:289-293), one pass.  1: unsaturated joints whose implicit end-of-step torque left the limit
    # (0.02 % of the joint-sub-steps) are put on it too and the sub-step is solved again (+6 us per step at 4096 envs)
    pd_second_pass: int = 0
    # ---- ORACLE-ONLY experiment of round 4 (oracle/lm_oracle.h `solver`): the drive rows inside the contact iteration, every drive
    # row's impulse bounded per iteration by drive_iter_impulse (-1 = max_effort x dt).  The engine implements solver 0 only (lib.make_params refuses).
    solver: int = 0
    vel_iters: int = 0
    drive_iter_impulse: float = -1.0
    tgs_flags: int = 0
    # ---- bookkeeping
    max_reset_counts: int = 2048        # success-rate window (quadruped_pose_control.py:151)
    # values the -1 sentinels above were resolved to ({field: value}).  dataclasses.replace() hands every field back to __init__, the resolved ones
    # included: a field that still holds the value recorded here was not set by the caller and is derived again from the (possibly replaced) dt,
    # max_effort, mode and variant - replace(loco_params(), dt=0.005).tau_max is 300, not 180.7
    derived: Dict[str, float] = field(default_factory=dict, repr=False, compare=False)

    def __post_init__(self):
        auto = dict(self.derived)

        def resolve(name, value):
            if getattr(self, name) < 0 or (name in auto and getattr(self, name) == auto[name]):
                setattr(self, name, value); auto[name] = value
            else:
                auto.pop(name, None)
        resolve("tau_max", self.max_effort / self.dt if self.dt > 0 else self.max_effort)
        resolve("drive_iter_impulse", self.max_effort * self.dt)
        resolve("pgs_iters", PGS_ITERS_PD if self.variant != 0 else PGS_ITERS_GROUND if self.mode == MODE_LOCO else PGS_ITERS_PLATE)
        self.derived = auto

    @property
    def ctrl_dt(self) -> float:
        return self.dt * self.substeps


def loco_params(**kw) -> EngineParams:
    """Horizontal locomotion task (QuadrupedPoseControl)."""
    return EngineParams(**kw)


def mani_params(**kw) -> EngineParams:
    """Horizontal manipulation task (QuadrupedManipulatePlate): fixed inverted base at the origin,
    plate dropped from z = 0.14 (quadruped_manipulate_plate.py:91-94,150-151)."""
    return EngineParams(**{**dict(mode=MODE_MANI), **kw})


_CLASS_DEFAULT_Q = [-1.2, 1.2, 1.2, -1.2, -1.22, -1.92, 1.92, 1.22, 1.92, 1.22, -1.22, -1.92]      # robot/quadruped_robot.py:45-52


def _cc(**kw):
    """Constants shared by the custom-controller tasks (quadruped_pose_control_custom_controller.py:24-52,88-108;
    cfg/task/QuadrupedPoseControlCustomController.yaml:13,18-19: dt 0.005, controlFrequencyInv 1, 500-step episodes;
    control_decimal 4 in-task sub-steps + 1 wrapper step = 5 sub-steps per action)."""
    base = dict(variant=1, num_obs=88, dt=0.005, substeps=5, kd=0.2, pd_kp=4.5, joint_damping=0.008, tau_max=1.5, act_scale_se=0.1,
                torque_div=4.0, init_q=list(_CLASS_DEFAULT_Q), acc_scale=-0.00015, rate_scale=-0.01, max_consec=20, max_episode=500,
                power_scale=-0.02, target_err_scale=-0.05, rot_dec_scale=0.0, rot_dec_thresh=0.3)
    base.update(kw)
    return base


def loco_cc_params(**kw) -> EngineParams:
    """QuadrupedPoseControlCustomController: class-default pose, base at z 0.18, fixed goal yaw 1.57 (:67-78)."""
    return EngineParams(**_cc(**{**dict(init_base_pos=[0.0, 0.0, 0.18], goal_lo=[0.0, 0.0, 1.57], goal_hi=[0.0, 0.0, 1.57]), **kw}))


def mani_cc_params(**kw) -> EngineParams:
    """QuadrupedManipulatePlateCustomController: plate dropped from z 0.18 onto the inverted fixed robot."""
    return EngineParams(**_cc(**{**dict(mode=MODE_MANI, init_plate_pos=[0.0, 0.0, 0.18], cc_update_last_tgt=0), **kw}))


_PC_INIT_SE = [-1.57, 1.57, 1.57, -1.57, -1.57, 1.05, 1.57, 1.05, 1.57, 1.05, -1.57, 1.05]


def _pc(**kw):
    """Constants shared by the position-control tasks (quadruped_pose_control_position_control.py:24-118,148-149,183-199,261;
    cfg/task/QuadrupedPoseControlPositionControl.yaml:13,18-19).  Same PD actuator and swing/extension action space as the
    custom-controller tasks; 64-wide observation with the scaled joint position targets in place of the actions; the reward of
    the velocity-drive tasks; the last-target buffer is only written at reset.  The committed `actions[:] = 0.0` debug line
    (:261) that freezes the targets in the two single-task files is not reproduced."""
    base = dict(variant=2, num_obs=64, dt=0.005, substeps=5, kd=0.2, pd_kp=4.5, joint_damping=0.008, tau_max=1.5, act_scale_se=0.1,
                torque_div=4.0, init_se=list(_PC_INIT_SE), cc_update_last_tgt=0, acc_substeps=1, power_scale=0.0, target_err_scale=0.0,
                goal_lo=[-0.5, -0.5, -3.14], goal_hi=[0.5, 0.5, 3.14], succ_thresh=0.1, max_consec=15, max_episode=500)
    base.update(kw)
    return base


def loco_pc_params(**kw) -> EngineParams:
    """QuadrupedPoseControlPositionControl."""
    return EngineParams(**_pc(**kw))


def mani_pc_params(**kw) -> EngineParams:
    """QuadrupedManipulatePlatePositionControl: inverted fixed robot at z 0.3, plate dropped from z 0.44, 450-step episodes
    (quadruped_manipulate_plate_position_control.py:122-124,177; cfg/task/QuadrupedManipulatePlatePositionControl.yaml:18)."""
    return EngineParams(**_pc(**{**dict(mode=MODE_MANI, fixed_base_pos=[0.0, 0.0, 0.3], init_plate_pos=[0.0, 0.0, 0.44], max_episode=450), **kw}))
