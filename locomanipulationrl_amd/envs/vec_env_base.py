"""VecEnvBase: what omni.isaac.gym.vec_env.VecEnvBase gives the reference's VecEnvRLGames
(attributes used by callers: SURVEY 8b) minus Isaac Sim: no SimulationApp boot, no USD stage."""
from __future__ import annotations


class _SimulationApp:
    """Stand-in for the Kit application handle (scripts/random_policy.py:53,65)."""

    def __init__(self):
        self._running = True

    def is_running(self):
        return self._running

    def close(self):
        self._running = False


class _PhysicsContext:
    prim_path = "/physicsScene"


class World:
    """`env._world`: step() advances the engine by one physics sub-step (vec_env_rlgames.py:64-66)."""

    def __init__(self, env):
        self._env = env
        self.current_time_step_index = 0

    def step(self, render: bool = False):
        self._env._task.physics_step(1)
        self.current_time_step_index += 1

    def reset(self, soft: bool = False):
        self._env._task.reset()
        self.current_time_step_index = 0

    def is_playing(self):
        return True

    def get_physics_context(self):
        return _PhysicsContext()


class VecEnvBase:
    def __init__(self, headless: bool = True, sim_device: int = 0, enable_livestream: bool = False, enable_viewport: bool = False) -> None:
        self._simulation_app = _SimulationApp()
        self._render = not headless
        self.sim_frame_count = 0
        self._sim_device = sim_device
        self._world = None
        self._task = None

    def set_task(self, task, backend="numpy", sim_params=None, init_sim=True, engine_factory=None) -> None:
        if backend != "torch":
            raise ValueError("only the torch backend exists on this path (utils/task_util.py:92 always passes 'torch')")
        self._task = task
        self._world = World(self)
        self._num_envs = task.num_envs
        self.observation_space = task.observation_space
        self.action_space = task.action_space
        if init_sim:
            task.create_engine(engine_factory)
            task.set_up_scene(None)
            task.post_reset()

    @property
    def num_envs(self):
        return self._num_envs

    def get_number_of_agents(self):
        return self._task.num_agents

    def close(self) -> None:
        if self._task is not None and getattr(self._task, "engine", None) is not None:
            self._task.engine.close()
        self._simulation_app.close()

    def seed(self, seed=-1):
        import torch
        if seed is not None and seed >= 0:
            torch.manual_seed(seed)
            if self._task is not None and getattr(self._task, "engine", None) is not None:
                self._task.engine.set_seed(seed)
        return seed
