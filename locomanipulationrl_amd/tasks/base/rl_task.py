"""RLTask: the task-side half of the drop-in boundary.

Mirrors RobotLearning/omniisaacgymenvs/tasks/base/rl_task.py:52-113,227-260 (buffers, spaces, reset flag,
post_physics_step sequencing) with the simulator replaced by the HIP engine: observation / reward /
termination / reset code that the reference runs as ~150 small torch ops per step
(quadruped_pose_control.py:200-633) executes inside the engine's fused step kernel, and every buffer the
reference exposes (obs_buf, states_buf, rew_buf, reset_buf, progress_buf, extras, goal_quaternions, ...)
is a zero-copy view of engine memory.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from ...engine_config import EngineParams
from ...utils.spaces import Box

EXTRAS_KEYS = ["env/rewards/orientation_rew", "env/rewards/translation_penalty", "env/rewards/joint_acc_penalty",
               "env/rewards/action_rate_penalty", "env/rewards/consecutive_successes_rew", "env/rewards/joint_limit_panelty",
               "env/rewards/fall_penalty", "env/success_rate"]     # ("panelty": sic, quadruped_pose_control.py:526)
COTRAIN_EXTRAS_KEYS = ["env/success_rate_loco", "env/success_rate_mani"]                     # joint_locomanipulation.py:857-859
CC_EXTRAS_KEYS = ["env/rewards/mechanical_power_penalty", "env/rewards/position_target_error_penalty",
                  "env/rewards/rot_dist_decreasing_reward"]                                      # …custom_controller.py:629-631


class RLTask:
    # sizes of the task family on this path
    _num_actions = 12
    _num_observations = 64
    _num_states = 93
    model_asset = "quadruped_robot_v2"

    def __init__(self, name: str, env=None, offset=None, sim_config=None) -> None:
        self.name = name
        self._env = env
        self._sim_config = sim_config
        self._cfg = sim_config.config
        self._task_cfg = sim_config.task_config
        self.test = self._cfg.get("test", False)
        self._device = self._cfg.get("sim_device", "cuda:0")
        self.rl_device = self._cfg.get("rl_device", "cuda:0")
        self._dt = self._task_cfg["sim"]["dt"]
        self._num_envs = int(self._task_cfg["env"]["numEnvs"])
        self._env_spacing = self._task_cfg["env"].get("envSpacing", 1.0)
        self._max_episode_length = self._task_cfg["sim"]["max_episode_length"]
        self.clip_obs = self._task_cfg["env"].get("clipObservations", np.inf)
        self.clip_actions = self._task_cfg["env"].get("clipActions", np.inf)
        self.control_frequency_inv = self._task_cfg["env"].get("controlFrequencyInv", 1)
        self._time_inv = self._dt * self.control_frequency_inv
        from ...utils.domain_randomization.randomize import Randomizer
        self._dr_randomizer = Randomizer(self._sim_config)          # rl_task.py:69-73
        self.randomize_actions = False
        self.randomize_observations = False
        if self._dr_randomizer.randomize:          # the reference does this in set_up_scene / post_reset (quadruped_pose_control.py:181-198)
            self._dr_randomizer.apply_on_startup_domain_randomization(self)
            self._dr_randomizer.set_up_domain_randomization(self)
        self._num_agents = 1
        self.action_space = Box(np.ones(self.num_actions, dtype=np.float32) * -1.0, np.ones(self.num_actions, dtype=np.float32) * 1.0)
        self.observation_space = Box(np.ones(self.num_observations, dtype=np.float32) * -np.inf, np.ones(self.num_observations, dtype=np.float32) * np.inf)
        self.state_space = Box(np.ones(self.num_states, dtype=np.float32) * -np.inf, np.ones(self.num_states, dtype=np.float32) * np.inf)
        self.extras: Dict[str, torch.Tensor] = {}
        self.engine = None
        self._goal_rng = self._task_cfg["env"].get("goalSampler", "engine")     # "engine" (in-kernel hash) | "torch" (torch.rand per step)
        self.current_actions: Optional[torch.Tensor] = None

    # ------------------------------------------------------------------ properties of the reference API
    @property
    def num_envs(self): return self._num_envs
    @property
    def num_actions(self): return self._num_actions
    @property
    def num_observations(self): return self._num_observations
    @property
    def num_states(self): return self._num_states
    @property
    def num_agents(self): return self._num_agents
    @property
    def device(self): return self._device

    # ------------------------------------------------------------------ engine plumbing
    def engine_params(self) -> List[EngineParams]:
        raise NotImplementedError

    def split_env(self) -> Optional[int]:
        return None

    def create_engine(self, engine_factory=None):
        """Instantiate the HIP engine (or an injected backend with the same interface, used by CPU tests)."""
        from ...model.robot_model import load_model
        params = self.engine_params()
        seed = int(self._cfg.get("seed", 42)) + int(self._cfg.get("rank", 0))
        if engine_factory is None:
            from ...lib import Engine
            self.engine = Engine(load_model(self.model_asset), params, self._num_envs, split_env=self.split_env(), seed=seed,
                                 device=self._device, clip_obs=float(self.clip_obs), clip_actions=float(self.clip_actions))
        else:
            self.engine = engine_factory(load_model(self.model_asset), params, self._num_envs, self.split_env(), seed,
                                         float(self.clip_obs), float(self.clip_actions))
        self.cleanup()
        return self.engine

    def cleanup(self) -> None:
        """rl_task.py:104-113 -- here the buffers are views of engine memory (reset_buf starts as ones)."""
        e = self.engine
        self.obs_buf = e.obs_buf
        self.states_buf = e.states_buf if self.num_states != self.num_observations else e.obs_buf
        self.rew_buf = e.rew_buf
        self.reset_buf = e.cnt[3]
        self.progress_buf = e.cnt[4]
        self.successes = e.cnt[0]
        self.consecutive_successes = e.cnt[1]
        self.goal_reset_buf = e.cnt[2]
        self.extras = {}

    # task-state views (quadruped_pose_control.py:123-154)
    @property
    def goal_quaternions(self): return self.engine.state[86:90].T
    @property
    def last_actions(self): return self.engine.state[50:62].T
    @property
    def last_base_tip_positions(self): return self.engine.state[74:86].T.reshape(self._num_envs, 4, 3)
    @property
    def num_successes(self): return self.engine.stats_i64[0]
    @property
    def num_resets(self): return self.engine.stats_i64[1]
    @property
    def success_rate(self): return self.engine.extras_buf[7]

    # ------------------------------------------------------------------ stepping
    def _goal_rand(self):
        if self._goal_rng == "torch":       # utils/math.py:184 semantics: torch's global generator on the sim device
            return torch.rand((self._num_envs, 3), device=self._device, dtype=torch.float32)
        return None

    def _alloc_outputs(self):
        N, dev = self._num_envs, self._device
        return (torch.empty((N, self.num_observations), device=dev), torch.empty((N, 93), device=dev), torch.empty((N,), device=dev),
                torch.empty((N,), dtype=torch.int64, device=dev), torch.empty((13,), device=dev))

    def _extras_layout(self):
        """(key, index into the 13-float extras block) pairs of this task, built once."""
        lay = getattr(self, "_extras_layout_cache", None)
        if lay is None:
            lay = [(k, i) for i, k in enumerate(EXTRAS_KEYS)]
            if self.split_env() is not None:
                lay += [(k, 8 + i) for i, k in enumerate(COTRAIN_EXTRAS_KEYS)]
            if getattr(self, "custom_controller", False):
                lay += [(k, 10 + i) for i, k in enumerate(self.cc_extras_keys)]
            self._extras_layout_cache = lay
        return lay

    def extras_dict(self, extras) -> Dict[str, torch.Tensor]:
        """The 13-float extras block of a step as the reference's `extras` dict of 0-d tensors (quadruped_pose_control.py:560,610,633);
        one unbind instead of one indexing op per key: this runs on every step() of the boundary."""
        parts = extras.unbind(0)
        return {k: parts[i] for k, i in self._extras_layout()}

    def state_dict(self):
        """Simulator checkpoint (engine state, counters, success windows, randomisation counters); see lib.Engine.state_dict."""
        return self.engine.state_dict()

    def load_state_dict(self, sd) -> None:
        self.engine.load_state_dict(sd)

    def make_rollout(self, policy: str, packed_params: torch.Tensor, log_std: torch.Tensor, T: int, noise_seed: int = 0):
        """A fused T-step rollout (policy forward -> sampling -> step, one hipGraph launch; include/lm_policy.h)."""
        from ...lib import POLICY_GNN, POLICY_MLP, Rollout
        if self._goal_rng != "engine":
            raise NotImplementedError("the fused rollout samples goals with the in-kernel generator (env.goalSampler: engine)")
        return Rollout(self.engine, {"mlp": POLICY_MLP, "gnn": POLICY_GNN}[policy], packed_params, log_std, T, noise_seed)

    def _publish(self, out):
        obs, states, rew, resets, extras = out
        self.extras = self.extras_dict(extras)
        if self.num_states == self.num_observations:
            states = obs
        return obs, states, rew, resets, self.extras

    def fused_step(self, actions: torch.Tensor):
        """pre_physics_step + controlFrequencyInv x world.step + post_physics_step in one launch
        (vec_env_rlgames.py:56-79).  Returned tensors are freshly allocated and clipped."""
        if actions.dtype != torch.float32 or str(actions.device) != str(self._device) or not actions.is_contiguous():
            actions = actions.to(self._device, dtype=torch.float32).contiguous()
        self.current_actions = actions
        out = self._alloc_outputs()
        self.engine.step(actions, self._goal_rand(), *out)
        return self._publish(out)

    # staged form (scripts/random_policy.py:57-61)
    def pre_physics_step(self, actions: torch.Tensor) -> None:
        if self._dr_randomizer.randomize:
            raise NotImplementedError("domain randomisation is sampled inside the fused lm_step launch; use env.step() when randomize is True")
        if self.engine_params()[0].drive_mode == 1:
            raise NotImplementedError("position control re-evaluates its PD torque every sub-step inside the fused launch; use env.step()")
        if getattr(self, "custom_controller", False):
            raise NotImplementedError("the custom-controller tasks run their physics inside pre_physics_step in the reference "
                                      "(…custom_controller.py:285-296); use env.step() (fused) for them")
        self.current_actions = torch.clamp(actions.to(self._device, dtype=torch.float32), -self.clip_actions, self.clip_actions).contiguous()
        self.engine.apply_resets(self._goal_rand())

    def physics_step(self, n: int = 1) -> None:
        act_scale = self.engine_params()[0].act_scale
        self.engine.substeps((self.current_actions * act_scale).contiguous(), n)

    def post_physics_step(self):
        out = self._alloc_outputs()
        self.engine.post_physics(self.current_actions, *out)
        obs, states, rew, resets, extras = self._publish(out)
        self._last_states = states
        return obs, rew, resets, extras

    def get_states(self):
        return self._last_states

    def reset(self):
        """Flags all envs for reset (rl_task.py:227-230)."""
        self.engine.reset_all()

    # hooks of the reference API that have nothing to do here
    def set_up_scene(self, scene=None) -> None: return None
    def post_reset(self) -> None: return None
    def get_observations(self): return {"obs": self.obs_buf}
    def calculate_metrics(self) -> None: return None
    def is_done(self) -> None: return None
