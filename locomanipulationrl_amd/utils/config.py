"""Configuration loading: the reference composes cfg/config.yaml + cfg/task/<name>.yaml with Hydra and
wraps the result in SimConfig (utils/config_utils/sim_config.py:37-89).  Hydra / OmegaConf are not
required here: the same keys are read with PyYAML and returned as the plain dict the task classes expect
(cfg["task"]["env"]["numEnvs"], cfg["task"]["sim"]["dt"], cfg["sim_device"], cfg["rl_device"], ...)."""
from __future__ import annotations

import copy
import os
from typing import Any, Dict, Optional

import yaml

_CFG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cfg")


def _deep_update(dst: Dict, src: Dict) -> Dict:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _deep_update(dst[k], v)
        else:
            dst[k] = v
    return dst


def load_config(task_name: str = "QuadrupedPoseControl", num_envs: Optional[int] = None, overrides: Optional[Dict[str, Any]] = None,
                **top_level) -> Dict[str, Any]:
    """Return the config dict for ``task_name`` (equivalent of `python script.py task=<name> num_envs=<n>`)."""
    with open(os.path.join(_CFG, "config.yaml")) as f:
        cfg = yaml.safe_load(f)
    path = os.path.join(_CFG, "task", task_name + ".yaml")
    if not os.path.exists(path):
        raise FileNotFoundError(f"no task config for {task_name!r} under {os.path.dirname(path)}")
    with open(path) as f:
        cfg["task"] = yaml.safe_load(f)
    cfg["task_name"] = cfg["task"]["name"]
    if num_envs is not None and num_envs != "":
        cfg["task"]["env"]["numEnvs"] = int(num_envs)
        cfg["num_envs"] = int(num_envs)
    cfg.update(top_level)
    if overrides:
        _deep_update(cfg, copy.deepcopy(overrides))
    if cfg.get("sim_device") in ("gpu", "cuda"):
        cfg["sim_device"] = f"cuda:{cfg.get('device_id', 0)}"
    return cfg


class SimConfig:
    """Subset of the reference's SimConfig that the tasks on the hot path read."""

    def __init__(self, config: Dict[str, Any]):
        self._config = config
        self._cfg = config
        self._sim_params = copy.deepcopy(config["task"].get("sim", {}))

    @property
    def config(self):
        return self._config

    @property
    def task_config(self):
        return self._config["task"]

    @property
    def sim_params(self):
        return self._sim_params

    def get_physics_params(self):
        return {**self._sim_params}

    # inert hooks of the reference API (articulation / USD settings have no meaning here)
    def parse_actor_config(self, actor_name):
        return {}

    def apply_articulation_settings(self, name, prim, cfg):
        return None
