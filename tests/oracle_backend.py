"""Oracle-backed stand-in for lib.Engine (CPU tensors).  TEST USE ONLY: lets the host-side boundary code
(VecEnvRLGames / RLTask / task classes) run its logic without a GPU.  Never importable from the package."""
from __future__ import annotations

import numpy as np
import torch

from oracle.lmo import Oracle


class OracleEngine:
    def __init__(self, robot_model, params, num_envs, split_env, seed, clip_obs, clip_actions, precision="f64"):
        self.N = self.num_envs = int(num_envs)
        self.params = list(params)
        self.split = int(split_env) if (split_env and len(self.params) == 2) else self.N
        self.oracles = [Oracle(robot_model, p, precision) for p in self.params]
        self.seed, self.clip_obs, self.clip_actions = seed, clip_obs, clip_actions
        o = self.oracles[0]
        self.phys, self.task, self.cntv = o.new_state(self.N)
        self.state = torch.zeros((115, self.N), dtype=torch.float32)
        self.cnt = torch.zeros((6, self.N), dtype=torch.int64)
        self.num_obs = self.params[0].num_obs
        self.obs_buf = torch.zeros((self.N, self.num_obs)); self.states_buf = torch.zeros((self.N, 93)); self.rew_buf = torch.zeros(self.N)
        self.extras_buf = torch.zeros(13); self.terms = torch.zeros((11, self.N)); self.stats_i64 = torch.zeros(6, dtype=torch.int64)
        self._sr = [0.0, 0.0, 0.0]
        self.drc = o.new_dr_counters(self.N); self.dr_enabled = bool(self.params[0].dr_enabled)
        self._sync_out()

    def _halves(self):
        if len(self.oracles) == 1:
            return [(self.oracles[0], slice(0, self.N))]
        return [(self.oracles[0], slice(0, self.split)), (self.oracles[1], slice(self.split, self.N))]

    def _sync_out(self):
        self.state[:50] = torch.as_tensor(self.phys.T.astype(np.float32)); self.state[50:] = torch.as_tensor(self.task.T.astype(np.float32))
        self.cnt[:] = torch.as_tensor(self.cntv.T.copy())

    def _sync_in(self):
        # reset flags may have been written through the torch views (RLTask.reset)
        self.cntv[:, 3] = self.cnt[3].numpy()

    def _finish(self, obs, states, rew, terms, outs):
        self.obs_buf[:] = torch.as_tensor(obs.astype(np.float32)); self.states_buf[:] = torch.as_tensor(states.astype(np.float32))
        self.rew_buf[:] = torch.as_tensor(rew.astype(np.float32)); self.terms[:] = torch.as_tensor(terms.T.astype(np.float32))
        means = terms[:, :7].mean(0)
        for w, sl in enumerate((slice(0, self.N), slice(0, self.split), slice(self.split, self.N))):
            ns, nr = int(self.stats_i64[2 * w]), int(self.stats_i64[2 * w + 1])
            if nr > self.params[0].max_reset_counts:
                self._sr[w] = ns / nr; ns = nr = 0
            ns += int(round(terms[sl, 7].sum())); nr += int(self.cntv[sl, 3].sum())
            self.stats_i64[2 * w] = ns; self.stats_i64[2 * w + 1] = nr
            self.extras_buf[7 + w] = self._sr[w]
        self.extras_buf[:7] = torch.as_tensor(means.astype(np.float32))
        self.extras_buf[10:13] = torch.as_tensor(terms[:, 8:11].mean(0).astype(np.float32))
        self._sync_out()
        out_obs, out_states, out_rew, out_resets, out_extras = outs
        c = self.clip_obs
        if out_obs is not None: out_obs[:] = self.obs_buf.clamp(-c, c)
        if out_states is not None: out_states[:] = self.states_buf.clamp(-c, c)
        if out_rew is not None: out_rew[:] = self.rew_buf
        if out_resets is not None: out_resets[:] = self.cnt[3]
        if out_extras is not None: out_extras[:] = self.extras_buf

    def step(self, actions, goal_rand=None, out_obs=None, out_states=None, out_rew=None, out_resets=None, out_extras=None):
        self._sync_in()
        raw = actions.detach().cpu().numpy().astype(np.float64)
        a = np.clip(raw, -self.clip_actions, self.clip_actions)
        gr = None if goal_rand is None else goal_rand.detach().cpu().numpy().astype(np.float64)
        obs = np.zeros((self.N, self.num_obs)); states = np.zeros((self.N, 93)); rew = np.zeros(self.N); terms = np.zeros((self.N, 11))
        for o, sl in self._halves():
            ph, tk, ct = self.phys[sl].copy(), self.task[sl].copy(), self.cntv[sl].copy()
            # env ids feed the hash RNG: the oracle numbers envs from 0 inside each call, so sample goals here for parity
            g = gr[sl] if gr is not None else np.stack([o.hash_uniform3(self.seed, e, int(self.cntv[e, 5])) for e in range(sl.start, sl.stop)])
            if self.dr_enabled:
                dc = self.drc[sl].copy()
                ob, st, rw, tr, _, _ = o.step_dr(ph, tk, ct, dc, raw[sl], clip_actions=self.clip_actions, goal_rand=g, seed=self.seed, env_offset=sl.start)
                self.drc[sl] = dc
            else:
                ob, st, rw, tr = o.step(ph, tk, ct, a[sl], goal_rand=g, seed=self.seed)
            self.phys[sl], self.task[sl], self.cntv[sl] = ph, tk, ct
            obs[sl], states[sl], rew[sl], terms[sl] = ob, st, rw, tr
        self._finish(obs, states, rew, terms, (out_obs, out_states, out_rew, out_resets, out_extras))

    def apply_resets(self, goal_rand=None):
        self._sync_in()
        gr = None if goal_rand is None else goal_rand.detach().cpu().numpy().astype(np.float64)
        for o, sl in self._halves():
            ph, tk, ct = self.phys[sl].copy(), self.task[sl].copy(), self.cntv[sl].copy()
            g = gr[sl] if gr is not None else np.stack([o.hash_uniform3(self.seed, e, int(self.cntv[e, 5])) for e in range(sl.start, sl.stop)])
            o.reset(ph, tk, ct, goal_rand=g, seed=self.seed)
            self.phys[sl], self.task[sl], self.cntv[sl] = ph, tk, ct
        self._sync_out()

    def substeps(self, targets, n=1):
        t = targets.detach().cpu().numpy().astype(np.float64)
        for o, sl in self._halves():
            ph = self.phys[sl].copy()
            for _ in range(n):
                o.substep(ph, t[sl])
            self.phys[sl] = ph
        self._sync_out()

    def post_physics(self, actions, out_obs=None, out_states=None, out_rew=None, out_resets=None, out_extras=None):
        self._sync_in()
        a = np.clip(actions.detach().cpu().numpy().astype(np.float64), -self.clip_actions, self.clip_actions)
        obs = np.zeros((self.N, self.num_obs)); states = np.zeros((self.N, 93)); rew = np.zeros(self.N); terms = np.zeros((self.N, 11))
        for o, sl in self._halves():
            ph, tk, ct = self.phys[sl].copy(), self.task[sl].copy(), self.cntv[sl].copy()
            n = ph.shape[0]; rb = np.zeros((n, 99)); mode = o._ep.mode
            rb[:, 0:12] = ph[:, 13:25]; rb[:, 12:24] = ph[:, 25:37]; rb[:, 24:36] = (ph[:, 25:37] - tk[:, 12:24]) / o._ep.ctrl_dt
            tk[:, 12:24] = ph[:, 25:37]
            rb[:, 36:49] = ph[:, 0:13] if mode == 0 else ph[:, 37:50]
            tips, knees = o.fk(ph); rb[:, 49:61] = tips.reshape(n, 12); rb[:, 61:85] = knees.reshape(n, 24)
            ob, st, rw, tr = o.task_eval(rb, a[sl], tk, ct)
            self.task[sl], self.cntv[sl] = tk, ct
            obs[sl], states[sl], rew[sl], terms[sl] = ob, st, rw, tr
        self._finish(obs, states, rew, terms, (out_obs, out_states, out_rew, out_resets, out_extras))

    def reset_all(self):
        self.cnt[3] = 1; self.cntv[:, 3] = 1

    def forward_kinematics(self):
        tips = np.zeros((self.N, 4, 3)); knees = np.zeros((self.N, 8, 3))
        for o, sl in self._halves():
            tips[sl], knees[sl] = o.fk(self.phys[sl])
        return torch.as_tensor(tips.astype(np.float32)), torch.as_tensor(knees.astype(np.float32))

    def set_seed(self, seed):
        self.seed = seed

    def close(self):
        pass


def oracle_engine_factory(robot_model, params, num_envs, split_env, seed, clip_obs, clip_actions):
    return OracleEngine(robot_model, params, num_envs, split_env, seed, clip_obs, clip_actions)
