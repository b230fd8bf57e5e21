"""Robot parameter sets and the state facade the tasks read.

Mirrors robot/quadruped_robot.py:6-216 and robot/base/robot_description.py:1-98 (pure data: control mode
"velocity", kp 0, kd 100, velocity limit 3.0 rad/s, torque limit 1.5 / 2.0 N m, init joint pose, default base
pose) and the read-back attributes of robot/base/robot.py:276-321.  The USD paths of the reference are
replaced by the name of the compiled model asset.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

_H_INIT = [-1.2, 1.2, 1.2, -1.2, -1.22, -1.92, 1.92, 1.22, 1.92, 1.22, -1.22, -1.92] + [0.953, -0.953] * 4
_V_INIT = [0.0, 0.0, 0.0, 0.0] + [0.35, -0.35] * 4 + [0.95, -0.95] * 4


@dataclass
class RobotDescriptionOmni:
    robot_name: str = "Quadruped"
    model_asset: str = "quadruped_robot_v2"
    control_mode: str = "velocity"
    num_modules: int = 4
    module_prefix_list: List[str] = field(default_factory=lambda: ["a1", "a2", "a3", "a4"])
    init_joint_pos: List[float] = field(default_factory=lambda: list(_H_INIT))
    default_position: List[float] = field(default_factory=lambda: [0.0, 0.0, 0.18])
    default_quaternion: List[float] = field(default_factory=lambda: [1.0, 0.0, 0.0, 0.0])
    joint_kps: List[float] = field(default_factory=lambda: [0.0] * 12)
    joint_kds: List[float] = field(default_factory=lambda: [100.0] * 12)
    velocity_limits: List[float] = field(default_factory=lambda: [3.0] * 12)
    torque_limits: List[float] = field(default_factory=lambda: [1.5] * 12)
    fixed_base: bool = False


class RobotOmni:
    """State facade: attributes are views / gathers of the engine's SoA state (robot.py:276-321)."""
    num_dof_per_module = 3

    def __init__(self, robot_description: RobotDescriptionOmni):
        self.robot_description = robot_description
        self.robot_name = robot_description.robot_name
        self.num_modules = robot_description.num_modules
        self.num_dofs = self.num_modules * self.num_dof_per_module
        self._engine = None
        self._env_slice = slice(None)

    def bind(self, engine, env_slice=slice(None)):
        self._engine, self._env_slice = engine, env_slice

    def _rows(self, r0, n):
        return self._engine.state[r0:r0 + n, self._env_slice].T

    @property
    def joint_positions(self):
        return self._rows(13, 12)

    @property
    def joint_velocities(self):
        return self._rows(25, 12)

    @property
    def last_joint_velcoties(self):      # (sic) robot.py:291
        return self._rows(62, 12)

    @property
    def base_positions(self):
        return self._rows(0, 3)

    @property
    def base_quaternions(self):
        return self._rows(3, 4)

    @property
    def base_linear_velocities(self):
        return self._rows(7, 3)

    @property
    def base_angular_velocities(self):
        return self._rows(10, 3)

    def _fk(self):
        tips, knees = self._engine.forward_kinematics()
        return tips[self._env_slice], knees[self._env_slice]

    @property
    def tip_positions(self):
        return self._fk()[0]

    @property
    def knee_positions(self):
        return self._fk()[1]

    def update_all_states(self):          # state is always current: nothing to pull from a simulator
        return None


class QuadrupedRobotOmni(RobotOmni):
    def __init__(self):
        super().__init__(RobotDescriptionOmni(torque_limits=[2.0] * 12))


class QuadrupedRobotOVOmni(RobotOmni):
    def __init__(self):
        super().__init__(RobotDescriptionOmni())


class QuadrupedRobotVerticalOVOmni(RobotOmni):
    def __init__(self):
        super().__init__(RobotDescriptionOmni(model_asset="quadfinger", init_joint_pos=list(_V_INIT), default_position=[0.0, 0.0, 0.35]))


class QuadrupedRobotFixedBaseOmni(RobotOmni):
    def __init__(self):
        super().__init__(RobotDescriptionOmni(default_position=[0.0, 0.0, 0.3], torque_limits=[2.0] * 12, fixed_base=True))


class QuadrupedRobotOVFixedBaseOmni(RobotOmni):
    def __init__(self):
        super().__init__(RobotDescriptionOmni(robot_name="QuadrupedFixed", default_position=[0.0, 0.0, 0.3], fixed_base=True))


class QuadrupedRobotVerticalOVFixedOmni(RobotOmni):
    def __init__(self):
        super().__init__(RobotDescriptionOmni(robot_name="QuadrupedFixed", model_asset="quadfinger", init_joint_pos=list(_V_INIT),
                                              default_position=[0.0, 0.0, 0.0], fixed_base=True))
