"""MI355X-native vectorised loco-manipulation physics-step engine.

Drop-in for the RLTask / VecEnvBase step()/reset()/get_observations() surface of
2361098148/LocoManipulationRL.  Importing the package does not touch the GPU.
"""
from .engine_config import EngineParams, loco_params, mani_params  # noqa: F401

__version__ = "0.1.0"


def make_env(task_name: str = "QuadrupedPoseControl", num_envs: int = 4096, headless: bool = True, engine_factory=None, **cfg):
    """Equivalent of the scripts' `load_omniverse_isaacgym_env(task_name=...)` (skrl_ppo_locomotion.py:55)."""
    from .envs.vec_env_rlgames import VecEnvRLGames
    from .utils.config import load_config
    from .utils.task_util import initialize_task
    config = load_config(task_name, num_envs=num_envs, **cfg)
    env = VecEnvRLGames(headless=headless, sim_device=config.get("device_id", 0))
    initialize_task(config, env, engine_factory=engine_factory)
    return env
