"""Concrete tasks of the hot path (names, constructor signature and class constants of the reference):

  QuadrupedPoseControl            tasks/quadruped_pose_control_tasks/quadruped_pose_control.py
  QuadrupedPoseControlVertical    tasks/quadruped_pose_control_tasks/quadruped_pose_control_vertical.py
  QuadrupedManipulatePlate        tasks/quadruped_manipulate_plate/quadruped_manipulate_plate.py
  QuadrupedManipulatePlateVertical tasks/quadruped_manipulate_plate/quadruped_manipulate_plate_vertical.py
  JointLocomanipulation(/Vertical) tasks/joint_train_locomanipulation/joint_locomanipulation(_vertical).py
  …CustomController               tasks/*/quadruped_*_custom_controller.py                  
  …PositionControl                tasks/*/quadruped_*_position_control.py, joint_locomanipulation_position_control.py   

The class attributes are the reference's (quadruped_pose_control.py:27-87); they are compiled into the
engine's parameter block, so editing them on a subclass changes the kernels' behaviour exactly as it changes
the reference's torch code.
"""
from __future__ import annotations

import sys
from typing import List, Optional

from ..engine_config import FRICTION_SCALE, MODE_LOCO, MODE_MANI, EngineParams
from ..robot.quadruped_robot import (QuadrupedRobotOVFixedBaseOmni, QuadrupedRobotOVOmni, QuadrupedRobotVerticalOVFixedOmni,
                                     QuadrupedRobotVerticalOVOmni)
from .base.rl_task import CC_EXTRAS_KEYS, RLTask

_TASK_INIT_Q = [-1.57, 1.57, 1.57, -1.57, -1.04, -2.09, 2.09, 1.04, 2.09, 1.04, -1.04, -2.09] + [1.37, -1.37] * 4
_H_CORNERS = [[0.075, 0.1835, -0.04], [-0.075, 0.1835, -0.04], [0.075, -0.1835, -0.04], [-0.075, -0.1835, -0.04]]
_V_CORNERS = [[0.0, -0.115, -0.1853], [0.0, 0.115, -0.1853], [-0.115, 0.0, -0.1853], [0.115, 0.0, -0.1853]]


class _QuadrupedTask(RLTask):
    # observation scales (quadruped_pose_control.py:27-34)
    ground_position_scale = 5
    ground_quaternion_scale = 1
    ground_linear_vel_scale = 2
    ground_angular_vel_scale = 0.25
    base_tip_position_scale = 1
    joint_position_scale = 0.3
    joint_velocity_scale = 0.3
    clip_reward = False
    # goal ranges (:38-43)
    min_roll, max_roll, min_pitch, max_pitch, min_yaw, max_yaw = -0.4, 0.4, -0.4, 0.4, -1.57, 1.57
    # reward scales (:54-61)
    quaternion_scale = 0.5
    rot_eps = 0.1
    translation_scale = -2.5
    joint_acc_scale = -0.0005
    action_rate_scale = -0.02
    # reset thresholds (:64-67)
    baseline_knee_height = 0.04
    fall_penalty = 0.0
    baseline_height = 0.05
    baseline_corner_height = 0.01
    # goal thresholds (:70-72); the bonus actually paid is the literal 600 (:457)
    success_thresh = 0.15
    success_bonus = 3
    consecutive_success_bonus = 600.0
    max_consecutive_successes = 15
    # joint limits (:79-87)
    joint_limit_penalty = -5
    min_joint_23_diff, max_joint_23_diff = 0.43, 2.53
    reset_min_joint_23_diff, reset_max_joint_23_diff = 0.384, 2.61
    min_joint_1_pos, max_joint_1_pos = -2.35, 0.78
    reset_min_joint_1_pos, reset_max_joint_1_pos = -2.44, 0.87
    mirrored_dof1_limits = True          # a2/a3 use the mirrored window (:482-501); vertical tasks use one symmetric window
    corner_points = _H_CORNERS

    def __init__(self, sim_config, name, env=None, offset=None) -> None:
        super().__init__(name, env, offset, sim_config=sim_config)
        if self.clip_reward:
            raise NotImplementedError("clip_reward=True is never set on the reference's hot path")

    # ---- helpers ------------------------------------------------------------------
    def _common(self, robot, **kw) -> EngineParams:
        sim = self._task_cfg["sim"]; eng = sim.get("engine", {})
        if "drive_limits_are_impulses" in eng:
            raise ValueError("sim.engine.drive_limits_are_impulses was removed (round 4): the velocity drive's limit is max_effort / dt; "
                             "set sim.engine.tau_max (N m) for an explicit torque limit")
        mat = sim.get("default_physics_material", {}); gnd = sim.get("ground_material", None)
        mu_body = float(mat.get("dynamic_friction", 1.0))
        if gnd is not None and kw.get("mode", MODE_LOCO) == MODE_LOCO:
            comb = eng.get("friction_combine", "average")       # PhysX default combine mode
            mu_g = float(gnd.get("dynamic_friction", 0.0))
            mu = {"average": 0.5 * (mu_body + mu_g), "min": min(mu_body, mu_g), "max": max(mu_body, mu_g), "multiply": mu_body * mu_g}[comb]
        else:
            mu = mu_body
        scale = float(eng.get("friction_scale", FRICTION_SCALE))     # effective / nominal coefficient (engine_config.py, DESIGN.md 2.1: fitted, parity unpinned)
        if scale != 1.0 and not getattr(_QuadrupedTask, "_friction_scale_logged", False):
            # said once per process, next to the YAML's own number: the friction materials of the task YAMLs do NOT mean what they say in this engine
            _QuadrupedTask._friction_scale_logged = True
            print(f"[locomanipulationrl_amd] foot friction: nominal (YAML, combined) {mu:.3g} x sim.engine.friction_scale {scale:.3g} = {mu * scale:.3g} "
                  f"(fitted on the reference's PhysX recordings, parity unpinned; set sim.engine.friction_scale: 1.0 for the YAML value)", file=sys.stderr)
        mu *= scale
        rd = robot.robot_description
        if rd.control_mode not in ("velocity", "position", "effort"):          # robot.py:323-333
            raise AttributeError(f"Invalid control mode name {rd.control_mode!r}")
        lo1, hi1 = self.min_joint_1_pos, self.max_joint_1_pos
        rlo1, rhi1 = self.reset_min_joint_1_pos, self.reset_max_joint_1_pos
        if self.mirrored_dof1_limits:
            d1_pen = [[lo1, hi1], [-hi1, -lo1], [-hi1, -lo1], [lo1, hi1]]
            d1_rst = [[rlo1, rhi1], [-rhi1, -rlo1], [-rhi1, -rlo1], [rlo1, rhi1]]
        else:
            d1_pen = [[lo1, hi1]] * 4; d1_rst = [[rlo1, rhi1]] * 4
        g = sim.get("gravity", [0, 0, -9.81])
        # Drive effort limit: ArticulationView.set_max_efforts(1.5) (robot.py:347-355).  Read as PhysX's per-step impulse limit (max_effort / dt,
        # never binding) by default - PARITY UNPINNED, engine_config.py and DESIGN.md 2.1 / 2.2 give the evidence for and against;
        # `sim.engine.tau_max` sets an explicit limit in N m (experiments / the replay test's negative control; the `drive_limits_are_impulses`
        # switch of rounds 2-3 is gone).  The PD-actuator tasks clamp their torque in Python
        # (…custom_controller.py:289-307): a real 1.5 N m either way.
        mode = kw.get("mode", MODE_LOCO)
        sweeps = eng.get("pgs_iters", {})          # contact sweeps per solve: {ground: 8, plate: 4}, {ground: 4, plate: 4} for the PD-actuator tasks (engine_config.PGS_ITERS_*); a plain integer sets both
        if isinstance(sweeps, dict):
            sweeps = sweeps.get("ground" if mode == MODE_LOCO else "plate", -1)
        base = dict(
            dt=float(sim["dt"]), substeps=int(self.control_frequency_inv), pgs_iters=int(sweeps), gravity=float(-g[2]),
            kd=float(rd.joint_kds[0]), max_effort=float(rd.torque_limits[0]), tau_max=float(eng.get("tau_max", -1.0)), act_scale=float(rd.velocity_limits[0]), mu=mu, drive_mode=0,
            tip_radius=float(eng.get("tip_radius", 0.005)), baumgarte=float(eng.get("baumgarte", 0.2)),
            max_depen_vel=float(eng.get("max_depenetration_velocity", 1.0)), pd_second_pass=int(bool(eng.get("pd_second_pass", False))),
            max_joint_vel=float(eng.get("max_joint_velocity_deg", 450.0)) * 3.141592653589793 / 180.0,
            init_q=list(rd.init_joint_pos[:12]),
            goal_lo=[self.min_roll, self.min_pitch, self.min_yaw], goal_hi=[self.max_roll, self.max_pitch, self.max_yaw],
            s_pos=float(self.ground_position_scale), s_lin=float(self.ground_linear_vel_scale), s_ang=float(self.ground_angular_vel_scale),
            s_q=float(self.joint_position_scale), s_qd=float(self.joint_velocity_scale),
            quat_scale=float(self.quaternion_scale), rot_eps=float(self.rot_eps), trans_scale=float(self.translation_scale),
            acc_scale=float(self.joint_acc_scale), rate_scale=float(self.action_rate_scale), bonus=float(self.consecutive_success_bonus),
            limit_pen=float(self.joint_limit_penalty), fall_pen=float(self.fall_penalty), succ_thresh=float(self.success_thresh),
            max_consec=int(self.max_consecutive_successes), max_episode=int(self._max_episode_length),
            d23_pen=[self.min_joint_23_diff, self.max_joint_23_diff], d23_rst=[self.reset_min_joint_23_diff, self.reset_max_joint_23_diff],
            d1_pen=d1_pen, d1_rst=d1_rst, h_base=float(self.baseline_height), h_corner=float(self.baseline_corner_height),
            h_knee=float(self.baseline_knee_height), corner=[list(c) for c in self.corner_points],
        )
        if not getattr(self, "custom_controller", False) and rd.control_mode != "velocity":
            # RobotOmni.take_action's other two modes (robot.py:444-461): position targets a * pi against the PD gains of the robot
            # description (robot_description.py:37-41), or joint efforts a * torque limit with the gains off (switch_control_mode)
            if rd.control_mode == "position":
                if not float(rd.joint_kds[0]) > 0:
                    raise ValueError("position control needs joint_kds > 0 (the drive is solved as kd (kp/kd (q* - q) - qd))")
                base.update(drive_mode=1, act_scale=3.141592653589793, pd_kp=float(rd.joint_kps[0]))
            else:
                base.update(drive_mode=2, act_scale=float(rd.torque_limits[0]))
        if getattr(self, "custom_controller", False):      # quadruped_pose_control_custom_controller.py:24-52,88-97
            v = int(self.controller_variant)
            base.update(variant=v, num_obs=88 if v == 1 else 64, acc_substeps=int(self.control_frequency_inv), kd=float(self.control_kd), pd_kp=float(self.control_kp), joint_damping=float(self.joint_damping),
                        tau_max=float(self.max_effort), act_scale_se=float(self.action_scale), se_lo=list(self.min_joint_pos_swing_ext),
                        se_hi=list(self.max_joint_pos_swing_ext), init_se=list(self.init_joint_pos_swing_ext),
                        substeps=int(self.control_decimal) + int(self.control_frequency_inv), torque_div=float(self.control_decimal),
                        power_scale=float(self.mechanical_power_penalty_scale), target_err_scale=float(self.position_target_error_penalty_scale),
                        rot_dec_scale=float(self.rot_dist_decreasing_reward_scale), rot_dec_thresh=float(self.no_rot_dist_decreasing_reward_thresh),
                        cc_update_last_tgt=int(self.update_last_targets))
        base.update(self._dr_randomizer.engine_dr())
        base.update(kw)
        return EngineParams(**base)

    def _loco_params(self, robot) -> EngineParams:
        rd = robot.robot_description
        return self._common(robot, mode=MODE_LOCO, init_base_pos=list(rd.default_position), init_base_quat=list(rd.default_quaternion))

    def _mani_params(self, robot, plate_pos) -> EngineParams:
        rd = robot.robot_description
        return self._common(robot, mode=MODE_MANI, fixed_base_pos=list(rd.default_position), fixed_base_quat=list(rd.default_quaternion),
                            init_plate_pos=list(plate_pos), init_plate_quat=list(rd.default_quaternion))


class QuadrupedPoseControl(_QuadrupedTask):
    """Horizontal locomotion (configs 1, 2)."""

    def __init__(self, sim_config, name="QuadrupedPoseControl", env=None, offset=None) -> None:
        self.robot_locomotion = QuadrupedRobotOVOmni()
        self.robot_locomotion.robot_description.init_joint_pos = list(_TASK_INIT_Q)          # quadruped_pose_control.py:94-102
        self.robot_locomotion.robot_description.default_position = [0.0, 0.0, 0.14]          # :103
        super().__init__(sim_config, name, env, offset)

    def engine_params(self) -> List[EngineParams]:
        return [self._loco_params(self.robot_locomotion)]

    def create_engine(self, engine_factory=None):
        e = super().create_engine(engine_factory); self.robot_locomotion.bind(e); return e


class QuadrupedPoseControlVertical(QuadrupedPoseControl):
    """Vertical configuration (config 5): quadfinger model, symmetric dof1 limits, different corner points
    (quadruped_pose_control_vertical.py:84-87,123-126).  The committed reference zeroes the actions
    (:204, a debug leftover); that is deliberately not reproduced."""
    model_asset = "quadfinger"
    success_bonus = 5
    min_joint_1_pos, max_joint_1_pos = -2.09, 2.09
    reset_min_joint_1_pos, reset_max_joint_1_pos = -2.26, 2.26
    mirrored_dof1_limits = False
    corner_points = _V_CORNERS

    def __init__(self, sim_config, name="QuadrupedPoseControlVertical", env=None, offset=None) -> None:
        self.robot_locomotion = QuadrupedRobotVerticalOVOmni()
        _QuadrupedTask.__init__(self, sim_config, name, env, offset)


class QuadrupedManipulatePlate(_QuadrupedTask):
    """Horizontal manipulation (config 3): inverted fixed-base robot + 2.4 kg plate."""
    default_obj_position = [0.0, 0.0, 0.14]           # quadruped_manipulate_plate.py:150

    def __init__(self, sim_config, name="QuadrupedManipulatePlate", env=None, offset=None) -> None:
        self.robot_manipulation = QuadrupedRobotOVFixedBaseOmni()
        rd = self.robot_manipulation.robot_description
        rd.default_quaternion = [0.0, 1.0, 0.0, 0.0]; rd.default_position = [0.0, 0.0, 0.0]    # :92-93
        rd.init_joint_pos = list(_TASK_INIT_Q)                                                  # :94-102
        super().__init__(sim_config, name, env, offset)

    def engine_params(self) -> List[EngineParams]:
        return [self._mani_params(self.robot_manipulation, self.default_obj_position)]

    def create_engine(self, engine_factory=None):
        e = super().create_engine(engine_factory); self.robot_manipulation.bind(e); return e

    # plate read-back (objects/base/rigid_object.py:52-58)
    @property
    def plate_pos_ground(self): return self.engine.state[37:40].T
    @property
    def plate_quat_ground(self): return self.engine.state[40:44].T


class QuadrupedManipulatePlateVertical(QuadrupedManipulatePlate):
    model_asset = "quadfinger"
    success_bonus = 5
    min_joint_1_pos, max_joint_1_pos = -2.09, 2.09
    reset_min_joint_1_pos, reset_max_joint_1_pos = -2.26, 2.26
    mirrored_dof1_limits = False
    corner_points = _V_CORNERS
    default_obj_position = [0.0, 0.0, 0.35]           # quadruped_manipulate_plate_vertical.py:141

    def __init__(self, sim_config, name="QuadrupedManipulatePlateVertical", env=None, offset=None) -> None:
        self.robot_manipulation = QuadrupedRobotVerticalOVFixedOmni()
        rd = self.robot_manipulation.robot_description
        rd.default_quaternion = [0.0, 1.0, 0.0, 0.0]; rd.default_position = [0.0, 0.0, 0.0]
        _QuadrupedTask.__init__(self, sim_config, name, env, offset)


class JointLocomanipulation(_QuadrupedTask):
    """Co-training (config 4): envs [0, N/2) locomotion, [N/2, N) manipulation, one observation tensor,
    states_buf aliases obs_buf (joint_locomanipulation.py:25-34,139,198,544-548).  The committed reference pins
    the goal to a single orientation and exits after recording two trajectories (:61-66,861-874); the ranges of
    the single tasks are used instead."""
    _num_states = 64
    default_obj_position = [0.0, 0.0, 0.68]
    mani_base_position = [0.0, 0.0, 0.5]

    def _make_robots(self):
        self.robot_locomotion = QuadrupedRobotOVOmni()
        self.robot_manipulation = QuadrupedRobotOVFixedBaseOmni()
        rd = self.robot_manipulation.robot_description
        rd.default_quaternion = [0.0, 1.0, 0.0, 0.0]; rd.default_position = list(self.mani_base_position)

    def __init__(self, sim_config, name="JointLocomanipulation", env=None, offset=None) -> None:
        self._make_robots()
        super().__init__(sim_config, name, env, offset)
        self._single_task_num_envs = self._num_envs // 2
        assert self._single_task_num_envs * 2 == self._num_envs, "Number of envs must be a multiplier of 2. "
        assert self._single_task_num_envs % 16 == 0, "each half must be a multiple of 16 envs (one wavefront = 16 envs)"

    def engine_params(self) -> List[EngineParams]:
        return [self._loco_params(self.robot_locomotion), self._mani_params(self.robot_manipulation, self.default_obj_position)]

    def split_env(self) -> Optional[int]:
        return self._num_envs // 2

    def create_engine(self, engine_factory=None):
        e = super().create_engine(engine_factory)
        h = self._num_envs // 2
        self.robot_locomotion.bind(e, slice(0, h)); self.robot_manipulation.bind(e, slice(h, None))
        return e


class JointLocomanipulationVertical(JointLocomanipulation):
    model_asset = "quadfinger"
    success_bonus = 5
    min_joint_1_pos, max_joint_1_pos = -2.09, 2.09
    reset_min_joint_1_pos, reset_max_joint_1_pos = -2.26, 2.26
    mirrored_dof1_limits = False
    corner_points = _V_CORNERS
    default_obj_position = [0.0, 0.0, 0.95]       # joint_locomanipulation_vertical.py:166
    mani_base_position = [0.0, 0.0, 0.6]          # :108

    def __init__(self, sim_config, name="JointLocomanipulationVertical", env=None, offset=None) -> None:
        self.robot_locomotion = QuadrupedRobotVerticalOVOmni()
        self.robot_manipulation = QuadrupedRobotVerticalOVFixedOmni()
        rd = self.robot_manipulation.robot_description
        rd.default_quaternion = [0.0, 1.0, 0.0, 0.0]; rd.default_position = list(self.mani_base_position)
        _QuadrupedTask.__init__(self, sim_config, name, env, offset)
        self._single_task_num_envs = self._num_envs // 2
        assert self._single_task_num_envs * 2 == self._num_envs and self._single_task_num_envs % 16 == 0


class _CustomControllerMixin:
    """Custom-controller task family: explicit PD actuator re-evaluated every physics sub-step, swing / extension
    action space, 88 observations, mechanical-power / target-error reward terms.  Class constants are the reference's
    (quadruped_pose_control_custom_controller.py:24-52,67-108)."""
    custom_controller = True
    controller_variant = 1
    cc_extras_keys = CC_EXTRAS_KEYS
    _num_observations = 88
    max_effort = 1.5
    control_kp = 4.5
    control_kd = 0.2
    joint_damping = 0.008
    joint_friction = 0.007            # not modelled
    action_scale = 0.1
    control_decimal = 4
    min_joint_pos_swing_ext = [-2.35, -0.78, -0.78, -2.35, -2.09, 0.52, 1.05, 0.52, 1.05, 0.52, -2.09, 0.52]
    max_joint_pos_swing_ext = [0.78, 2.35, 2.35, 0.78, -1.05, 2.09, 2.09, 2.09, 2.09, 2.09, -1.05, 2.09]
    init_joint_pos_swing_ext = [-1.2, 1.2, 1.2, -1.2, -1.57, 0.7, 1.57, 0.7, 1.57, 0.7, -1.57, 0.7]
    joint_acc_scale = -0.00015
    action_rate_scale = -0.01
    mechanical_power_penalty_scale = -0.02
    position_target_error_penalty_scale = -0.05
    rot_dist_decreasing_reward_scale = 0.0
    no_rot_dist_decreasing_reward_thresh = 0.3
    max_consecutive_successes = 20
    update_last_targets = True


class QuadrupedPoseControlCustomController(_CustomControllerMixin, _QuadrupedTask):
    """tasks/quadruped_pose_control_tasks/quadruped_pose_control_custom_controller.py: class-default pose, base at z 0.18,
    goal fixed at yaw 1.57 (:67-78), dt 0.005, 4 in-task + 1 wrapper sub-steps per action, 500-step episodes."""
    min_roll, max_roll, min_pitch, max_pitch, min_yaw, max_yaw = 0.0, 0.0, 0.0, 0.0, 1.57, 1.57

    def __init__(self, sim_config, name="QuadrupedPoseControlCustomController", env=None, offset=None) -> None:
        self.robot_locomotion = QuadrupedRobotOVOmni()
        self.robot_locomotion.robot_description.control_mode = "effort"          # :127
        super().__init__(sim_config, name, env, offset)

    def engine_params(self) -> List[EngineParams]:
        return [self._loco_params(self.robot_locomotion)]

    def create_engine(self, engine_factory=None):
        e = super().create_engine(engine_factory); self.robot_locomotion.bind(e); return e

    @property
    def current_joint_position_targets_se(self): return self.engine.state[90:102].T
    @property
    def last_joint_position_targets(self): return self.engine.state[102:114].T


class QuadrupedManipulatePlateCustomController(_CustomControllerMixin, _QuadrupedTask):
    """tasks/quadruped_manipulate_plate/quadruped_manipulate_plate_custom_controller.py: plate dropped from z 0.18; its
    last_joint_position_targets are never refreshed after reset (only :378)."""
    default_obj_position = [0.0, 0.0, 0.18]
    update_last_targets = False

    def __init__(self, sim_config, name="QuadrupedManipulatePlateCustomController", env=None, offset=None) -> None:
        self.robot_manipulation = QuadrupedRobotOVFixedBaseOmni()
        rd = self.robot_manipulation.robot_description
        rd.default_quaternion = [0.0, 1.0, 0.0, 0.0]; rd.default_position = [0.0, 0.0, 0.0]; rd.control_mode = "effort"
        super().__init__(sim_config, name, env, offset)

    def engine_params(self) -> List[EngineParams]:
        return [self._mani_params(self.robot_manipulation, self.default_obj_position)]

    def create_engine(self, engine_factory=None):
        e = super().create_engine(engine_factory); self.robot_manipulation.bind(e); return e


class _PositionControlMixin:
    """Position-control task family: the PD actuator and swing / extension action space of the custom-controller
    tasks, a 64-wide observation whose last 24 entries are the scaled current / reset joint position targets, states_buf aliasing
    obs_buf, and the reward of the velocity-drive tasks (quadruped_pose_control_position_control.py:24-118,148-149,438-455).
    The two single-task files carry a live `actions[:] = 0.0` (:261, a debug leftover that freezes the targets); that line is not
    reproduced - set action_scale = 0 on a subclass to get it."""
    custom_controller = True
    controller_variant = 2
    cc_extras_keys = []
    _num_observations = 64
    _num_states = 64
    max_effort = 1.5
    control_kp = 4.5
    control_kd = 0.2
    joint_damping = 0.008
    joint_friction = 0.007            # not modelled
    action_scale = 0.1
    control_decimal = 4
    min_joint_pos_swing_ext = [-2.35, -0.78, -0.78, -2.35, -2.09, 0.52, 1.05, 0.52, 1.05, 0.52, -2.09, 0.52]
    max_joint_pos_swing_ext = [0.78, 2.35, 2.35, 0.78, -1.05, 2.09, 2.09, 2.09, 2.09, 2.09, -1.05, 2.09]
    init_joint_pos_swing_ext = [-1.57, 1.57, 1.57, -1.57, -1.57, 1.05, 1.57, 1.05, 1.57, 1.05, -1.57, 1.05]
    mechanical_power_penalty_scale = 0.0
    position_target_error_penalty_scale = 0.0
    rot_dist_decreasing_reward_scale = 0.0
    no_rot_dist_decreasing_reward_thresh = 0.3
    update_last_targets = False       # last_joint_position_targets is only written by reset_idx (:380)
    min_roll, max_roll, min_pitch, max_pitch, min_yaw, max_yaw = -0.5, 0.5, -0.5, 0.5, -3.14, 3.14
    success_thresh = 0.1

    @property
    def current_joint_position_targets_se(self): return self.engine.state[90:102].T
    @property
    def last_joint_position_targets(self): return self.engine.state[102:114].T


class QuadrupedPoseControlPositionControl(_PositionControlMixin, _QuadrupedTask):
    """tasks/quadruped_pose_control_tasks/quadruped_pose_control_position_control.py (dt 0.005, 4 + 1 sub-steps, 500-step episodes)."""

    def __init__(self, sim_config, name="QuadrupedPoseControlPositionControl", env=None, offset=None) -> None:
        self.robot_locomotion = QuadrupedRobotOVOmni()
        rd = self.robot_locomotion.robot_description
        rd.init_joint_pos = list(_TASK_INIT_Q); rd.default_position = [0.0, 0.0, 0.14]; rd.control_mode = "effort"      # :123-133
        super().__init__(sim_config, name, env, offset)

    def engine_params(self) -> List[EngineParams]:
        return [self._loco_params(self.robot_locomotion)]

    def create_engine(self, engine_factory=None):
        e = super().create_engine(engine_factory); self.robot_locomotion.bind(e); return e


class QuadrupedManipulatePlatePositionControl(_PositionControlMixin, _QuadrupedTask):
    """tasks/quadruped_manipulate_plate/quadruped_manipulate_plate_position_control.py: inverted robot fixed at z 0.3, plate
    dropped from z 0.44 (:122-124,177), 450-step episodes."""
    default_obj_position = [0.0, 0.0, 0.44]

    def __init__(self, sim_config, name="QuadrupedManipulatePlatePositionControl", env=None, offset=None) -> None:
        self.robot_manipulation = QuadrupedRobotOVFixedBaseOmni()
        rd = self.robot_manipulation.robot_description
        rd.default_quaternion = [0.0, 1.0, 0.0, 0.0]; rd.default_position = [0.0, 0.0, 0.3]; rd.control_mode = "effort"
        rd.init_joint_pos = list(_TASK_INIT_Q)
        super().__init__(sim_config, name, env, offset)

    def engine_params(self) -> List[EngineParams]:
        return [self._mani_params(self.robot_manipulation, self.default_obj_position)]

    def create_engine(self, engine_factory=None):
        e = super().create_engine(engine_factory); self.robot_manipulation.bind(e); return e


class JointLocomanipulationPositionControl(_PositionControlMixin, JointLocomanipulation):
    """tasks/joint_train_locomanipulation/joint_locomanipulation_position_control.py: co-training with the PD actuator
    (dt 1/240 s, control_decimal 7 + 1 sub-steps = 30 Hz control, :37-44,365-408), plate dropped from z 0.64 (:214)."""
    control_decimal = 7
    min_roll, max_roll, min_pitch, max_pitch, min_yaw, max_yaw = -0.4, 0.4, -0.4, 0.4, -1.57, 1.57      # :86-91
    joint_acc_scale = -0.001
    action_rate_scale = -0.03
    success_bonus = 5
    default_obj_position = [0.0, 0.0, 0.64]

    def _make_robots(self):          # :131-157
        JointLocomanipulation._make_robots(self)
        for rd in (self.robot_locomotion.robot_description, self.robot_manipulation.robot_description):
            rd.init_joint_pos = list(_TASK_INIT_Q); rd.control_mode = "effort"
        self.robot_locomotion.robot_description.default_position = [0.0, 0.0, 0.14]

    def __init__(self, sim_config, name="JointLocomanipulationPositionControl", env=None, offset=None) -> None:
        JointLocomanipulation.__init__(self, sim_config, name, env, offset)
