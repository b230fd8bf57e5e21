"""Compact PPO with the hyper-parameters and semantics the reference's skrl scripts configure
(scripts/skrl_ppo_locomotion.py:86-112: rollouts 48, 5 epochs, 1 mini-batch, gamma 0.99, lambda 0.95, lr 3e-4 with
KL-adaptive schedule (threshold 0.012), grad-norm clip 1.0, ratio clip 0.2, value clip 0.2 with clipped predictions,
entropy scale 0, value scale 1, RunningStandardScaler on observations and values).

skrl itself is third-party and not installed here; this trainer exists so that the engine can be accepted
*behaviourally* (the task's success rate rises under the reference's PPO recipe) and to exercise the multi-GPU
exchange of SURVEY 8(e): rollout returns / advantages are all-gathered over RCCL for global advantage normalisation and
the gradients are all-reduced.  Rollouts use the matrix-core policy forward (csrc/lm_policy.hip); updates use autograd.
"""
from __future__ import annotations

import math
import time
from typing import Dict

import torch
import torch.distributed as dist

from .. import distributed as D


class RunningStandardScaler:
    """skrl.resources.preprocessors.torch.RunningStandardScaler: running mean/variance, (x-mean)/(sqrt(var)+eps), clip 5."""

    def __init__(self, size: int, device, epsilon: float = 1e-8, clip: float = 5.0):
        self.mean = torch.zeros(size, device=device, dtype=torch.float64)
        self.var = torch.ones(size, device=device, dtype=torch.float64)
        self.count = torch.ones((), device=device, dtype=torch.float64)
        self.eps, self.clip = epsilon, clip

    def update(self, x: torch.Tensor):
        x = x.reshape(-1, self.mean.numel()).double()
        n = torch.tensor(float(x.shape[0]), device=x.device, dtype=torch.float64)
        s, ss = x.sum(0), (x * x).sum(0)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            buf = torch.cat((s, ss, n.view(1))); dist.all_reduce(buf); s, ss, n = buf[:s.numel()], buf[s.numel():-1], buf[-1]
        bmean = s / n; bvar = (ss / n - bmean * bmean).clamp_min(0) * n / (n - 1).clamp_min(1)
        delta = bmean - self.mean; tot = self.count + n
        m2 = self.var * self.count + bvar * n + delta * delta * self.count * n / tot
        self.mean = self.mean + delta * n / tot; self.var = m2 / tot; self.count = tot

    def __call__(self, x: torch.Tensor, inverse: bool = False):
        mean, std = self.mean.float(), self.var.float().sqrt()
        if inverse:
            return std * torch.clamp(x, -self.clip, self.clip) + mean
        return torch.clamp((x - mean) / (std + self.eps), -self.clip, self.clip)


class PPO:
    def __init__(self, env, model, rollouts=48, learning_epochs=5, mini_batches=1, gamma=0.99, lam=0.95, lr=3e-4, kl_threshold=0.012,
                 grad_norm_clip=1.0, ratio_clip=0.2, value_clip=0.2, value_loss_scale=1.0, entropy_loss_scale=0.0, hip_inference=True,
                 fused_rollout=True, freeze_obs_scaler=False, max_lr=1e-2, min_log_std=None):
        self.env, self.model = env, model
        self.dev = next(model.parameters()).device
        self.N = env.num_envs
        self.T, self.epochs, self.mb = rollouts, learning_epochs, mini_batches
        self.gamma, self.lam, self.lr, self.kl_thr = gamma, lam, lr, kl_threshold
        self.gclip, self.rclip, self.vclip, self.vscale, self.escale = grad_norm_clip, ratio_clip, value_clip, value_loss_scale, entropy_loss_scale
        self.opt = torch.optim.Adam(model.parameters(), lr=lr)
        self.freeze_obs_scaler = freeze_obs_scaler          # diagnostics: identity observation scaler (mean 0, var 1)
        self.max_lr, self.min_log_std = max_lr, min_log_std  # diagnostics: cap of the KL-adaptive rate (skrl: 1e-2); floor of log_std (skrl: -20)
        self.n_obs = int(env.observation_space.shape[0])          # 88 for the custom-controller tasks
        self.obs_scaler = RunningStandardScaler(self.n_obs, self.dev); self.val_scaler = RunningStandardScaler(1, self.dev)
        # MFMA forward kernels: MLP on the 64- and 88-wide observations, GNN on the 64-wide one
        self.hip = hip_inference and hasattr(model, "act_inference") and torch.cuda.is_available() and \
            (self.n_obs == 64 or (self.n_obs == 88 and type(model).__name__ == "SharedMLP"))
        z = lambda *s, dt=torch.float32: torch.zeros(*s, device=self.dev, dtype=dt)
        self.b_obs, self.b_act, self.b_logp = z(self.T, self.N, self.n_obs), z(self.T, self.N, 12), z(self.T, self.N)
        self.b_val, self.b_rew, self.b_done = z(self.T, self.N), z(self.T, self.N), z(self.T, self.N)
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        # fused rollout: forward -> sampling -> step x T as one hipGraph launch, buffers shared with the update
        self.rollout = None
        if self.hip and fused_rollout and hasattr(env, "_task") and hasattr(env._task, "make_rollout"):
            kind = "gnn" if type(model).__name__ == "GraphPolicy" else "mlp"
            model.refresh(self.dev, self.obs_scaler.mean.float(), self.obs_scaler.var.float(), self.obs_scaler.eps, self.obs_scaler.clip)
            self._packed = model._packed.clone(); self._log_std = model.log_std_parameter.detach().clone().float().contiguous()
            rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
            self.rollout = env._task.make_rollout(kind, self._packed, self._log_std, self.T, noise_seed=1000 + rank)
            ro = self.rollout
            self._ro_mode = "auto"      # one persistent kernel per rollout where the engine supports it and it pays, else the hipGraph replay
            self.b_obs, self.b_act, self.b_logp, self.b_rew = ro.obs[:self.T], ro.actions, ro.logp, ro.rewards

    # ------------------------------------------------------------------ policy evaluation
    def _policy(self, obs_raw):
        """mean, log_std, value (de-normalised) for raw observations; rollouts go through the MFMA forward."""
        with torch.no_grad():
            if self.hip:
                self.model.refresh(self.dev, self.obs_scaler.mean.float(), self.obs_scaler.var.float(), self.obs_scaler.eps, self.obs_scaler.clip)
                mean, log_std, v = self.model.act_inference(obs_raw.contiguous())
            else:
                mean, log_std, v = self.model(self.obs_scaler(obs_raw))
            return mean, log_std.detach(), self.val_scaler(v, inverse=True).squeeze(-1)

    @staticmethod
    def _logp(mean, log_std, act):
        return (-0.5 * ((act - mean) / log_std.exp()) ** 2 - log_std - 0.5 * math.log(2 * math.pi)).sum(-1)

    # ------------------------------------------------------------------ one iteration = T env steps + update
    def collect(self, obs_raw):
        if self.rollout is not None:
            ro = self.rollout
            self.model.refresh(self.dev, self.obs_scaler.mean.float(), self.obs_scaler.var.float(), self.obs_scaler.eps, self.obs_scaler.clip)
            self._packed.copy_(self.model._packed); self._log_std.copy_(self.model.log_std_parameter.detach())
            ro.obs[0].copy_(obs_raw)
            ro.run(self._ro_mode)
            v = self.val_scaler(ro.values.reshape(-1, 1), inverse=True).reshape(self.T + 1, self.N)
            self.b_val, self.b_done = v[:self.T], ro.dones.float()
            return ro.obs[self.T], v[self.T], self.env._task.extras_dict(ro.extras[self.T - 1])
        for t in range(self.T):
            mean, log_std, value = self._policy(obs_raw)
            act = mean + log_std.exp() * torch.randn_like(mean)
            o, rew, done, extras = self.env.step(act)
            self.b_obs[t], self.b_act[t], self.b_logp[t] = obs_raw, act, self._logp(mean, log_std, act)
            self.b_val[t], self.b_rew[t], self.b_done[t] = value, rew, done.float()
            obs_raw = o["obs"]
        _, _, last_value = self._policy(obs_raw)
        return obs_raw, last_value, extras

    def update(self, last_value) -> Dict[str, float]:
        ret, adv = D.compute_gae(self.b_rew, self.b_val, self.b_done, last_value, self.gamma, self.lam)
        g_ret, g_adv = D.all_gather_rollout(ret, adv)                 # RCCL all-gather over xGMI when world > 1
        adv = (adv - g_adv.mean()) / (g_adv.std() + 1e-8)             # global advantage normalisation
        obs, act = self.b_obs.reshape(-1, self.n_obs), self.b_act.reshape(-1, 12)
        old_logp, old_val = self.b_logp.reshape(-1), self.b_val.reshape(-1)
        ret, adv = ret.reshape(-1), adv.reshape(-1)
        if not self.freeze_obs_scaler: self.obs_scaler.update(obs)
        self.val_scaler.update(ret.unsqueeze(-1))      # preprocessors train on the first epoch's data
        obs_n = self.obs_scaler(obs); ret_n = self.val_scaler(ret.unsqueeze(-1)).squeeze(-1); old_val_n = self.val_scaler(old_val.unsqueeze(-1)).squeeze(-1)
        n = obs.shape[0]; stats = {}
        for epoch in range(self.epochs):
            perm = torch.randperm(n, device=self.dev) if self.mb > 1 else None
            kls = []
            for i in range(self.mb):
                idx = slice(None) if perm is None else perm[i * n // self.mb:(i + 1) * n // self.mb]
                mean, log_std, v = self.model(obs_n[idx])
                logp = self._logp(mean, log_std, act[idx])
                ratio_log = logp - old_logp[idx]
                with torch.no_grad():
                    kls.append(((ratio_log.exp() - 1) - ratio_log).mean())
                ratio = ratio_log.exp()
                surr = torch.min(adv[idx] * ratio, adv[idx] * ratio.clamp(1 - self.rclip, 1 + self.rclip))
                v = v.squeeze(-1); v = old_val_n[idx] + (v - old_val_n[idx]).clamp(-self.vclip, self.vclip)
                loss_pi = -surr.mean(); loss_v = self.vscale * torch.nn.functional.mse_loss(ret_n[idx], v)
                ent = (log_std + 0.5 + 0.5 * math.log(2 * math.pi)).sum()
                loss = loss_pi + loss_v - self.escale * ent
                self.opt.zero_grad(set_to_none=True); loss.backward()
                if self.world > 1:
                    flat = torch.cat([p.grad.reshape(-1) for p in self.model.parameters() if p.grad is not None])
                    dist.all_reduce(flat); flat /= self.world; o = 0
                    for p in self.model.parameters():
                        if p.grad is not None:
                            p.grad.copy_(flat[o:o + p.numel()].view_as(p)); o += p.numel()
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.gclip)
                self.opt.step()
                if self.min_log_std is not None:
                    with torch.no_grad(): self.model.log_std_parameter.clamp_(min=self.min_log_std)
            kl = torch.stack(kls).mean()
            if self.world > 1:
                dist.all_reduce(kl); kl /= self.world
            kl = float(kl)                                                        # KLAdaptiveRL (skrl): lr /= 1.5 above 2*thr, *= 1.5 below thr/2
            if self.kl_thr > 0:                                                   # kl_threshold 0 = fixed learning rate
                if kl > self.kl_thr * 2: self.lr = max(self.lr / 1.5, 1e-6)
                elif kl < self.kl_thr / 2: self.lr = min(self.lr * 1.5, self.max_lr)
            for gp in self.opt.param_groups: gp["lr"] = self.lr
            stats = {"kl": kl, "loss_pi": float(loss_pi), "loss_v": float(loss_v), "lr": self.lr}
        return stats

    def train(self, timesteps: int, log_every: int = 10, log=print):
        obs = self.env.reset()["obs"]; history = []; t0 = time.perf_counter()
        for it in range(timesteps // self.T):
            obs, last_value, extras = self.collect(obs)
            st = self.update(last_value)
            if it % log_every == 0 or it == timesteps // self.T - 1:
                ex = D.global_extras(extras, self.N)
                rec = {"iteration": it, "timesteps": (it + 1) * self.T, "wall_s": time.perf_counter() - t0, "mean_reward": float(self.b_rew.mean()),
                       "success_rate": float(ex["env/success_rate"]), "std": float(self.model.log_std_parameter.exp().mean()), **st}
                history.append(rec); log(rec)
        return history
