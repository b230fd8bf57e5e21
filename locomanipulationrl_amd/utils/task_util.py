"""Task registry (utils/task_util.py:30-93 of the reference)."""
from __future__ import annotations

from .config import SimConfig


def task_map():
    from ..tasks.quadruped_tasks import (JointLocomanipulation, JointLocomanipulationPositionControl, JointLocomanipulationVertical,
                                          QuadrupedManipulatePlate, QuadrupedManipulatePlateCustomController,
                                          QuadrupedManipulatePlatePositionControl, QuadrupedManipulatePlateVertical, QuadrupedPoseControl,
                                          QuadrupedPoseControlCustomController, QuadrupedPoseControlPositionControl,
                                          QuadrupedPoseControlVertical)
    return {
        "JointLocomanipulation": JointLocomanipulation,
        "QuadrupedPoseControl": QuadrupedPoseControl,
        "QuadrupedManipulatePlate": QuadrupedManipulatePlate,
        "QuadrupedPoseControlVertical": QuadrupedPoseControlVertical,
        "QuadrupedManipulatePlateVertical": QuadrupedManipulatePlateVertical,
        "JointLocomanipulationVertical": JointLocomanipulationVertical,
        "QuadrupedPoseControlCustomController": QuadrupedPoseControlCustomController,
        "QuadrupedManipulatePlateCustomController": QuadrupedManipulatePlateCustomController,
        "QuadrupedPoseControlCustomControllerDR": QuadrupedPoseControlCustomController,      # same task, the DR YAML (its ..._dr.py adds record / replay only)
        "QuadrupedPoseControlPositionControl": QuadrupedPoseControlPositionControl,
        "QuadrupedManipulatePlatePositionControl": QuadrupedManipulatePlatePositionControl,
        "JointLocomanipulationPositionControl": JointLocomanipulationPositionControl,
    }


def initialize_task(config, env, init_sim=True, engine_factory=None):
    sim_config = SimConfig(config)
    cfg = sim_config.config
    tm = task_map()
    if cfg["task_name"] not in tm:
        raise KeyError(f"task {cfg['task_name']!r} is outside the hot path implemented here; available: {sorted(tm)}")
    task = tm[cfg["task_name"]](name=cfg["task_name"], sim_config=sim_config, env=env)
    env.set_task(task=task, sim_params=sim_config.get_physics_params(), backend="torch", init_sim=init_sim, engine_factory=engine_factory)
    return task
