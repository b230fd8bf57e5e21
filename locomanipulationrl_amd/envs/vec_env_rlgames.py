"""VecEnvRLGames: the drop-in boundary (RobotLearning/omniisaacgymenvs/envs/vec_env_rlgames.py:38-90).

Same signature and return contract: ``step(actions) -> ({"obs","states"}, rew, resets, extras)`` with
freshly allocated tensors on rl_device, obs/states clamped to +-clipObservations, actions clamped to
+-clipActions; ``reset()`` flags every env and takes one zero-action step.  The clamp/clone traffic of
``_process_data`` (:41-46) is produced by the step kernel itself, which writes the clipped copies straight
into the tensors handed back to the caller.
"""
from __future__ import annotations

from datetime import datetime

import torch

from .vec_env_base import VecEnvBase


class VecEnvRLGames(VecEnvBase):

    def set_task(self, task, backend="numpy", sim_params=None, init_sim=True, engine_factory=None) -> None:
        super().set_task(task, backend, sim_params, init_sim, engine_factory)
        self.num_states = self._task.num_states
        self.state_space = self._task.state_space

    def step(self, actions):
        if self._task.randomize_actions:
            actions = self._task._dr_randomizer.apply_actions_randomization(actions=actions, reset_buf=self._task.reset_buf)
        # clamp(+-clip_actions), the reset scatter, controlFrequencyInv sub-steps and post_physics_step are one launch
        self._obs, self._states, self._rew, self._resets, self._extras = self._task.fused_step(actions)
        self.sim_frame_count += self._task.control_frequency_inv
        if self._task.randomize_observations:
            self._obs = self._task._dr_randomizer.apply_observations_randomization(
                observations=self._obs.to(device=self._task.rl_device), reset_buf=self._task.reset_buf)
        rl = self._task.rl_device
        if str(self._obs.device) != str(torch.device(rl)):
            self._obs, self._states = self._obs.to(rl), self._states.to(rl)
            self._rew, self._resets = self._rew.to(rl), self._resets.to(rl)
        self._extras = self._extras.copy()
        obs_dict = {"obs": self._obs, "states": self._states}
        return obs_dict, self._rew, self._resets, self._extras

    def reset(self):
        """ Resets the task and applies default zero actions to recompute observations and states. """
        now = datetime.now().strftime("%Y-%m-%d %H:%M:%S")
        print(f"[{now}] Running RL reset")
        self._task.reset()
        actions = torch.zeros((self.num_envs, self._task.num_actions), device=self._task.rl_device)
        obs_dict, _, _, _ = self.step(actions)
        return obs_dict
