"""MLP policy of the reference's PPO scripts (scripts/skrl_ppo_locomotion.py:25-52): shared trunk 64-256-128-64 (ELU),
Gaussian mean head (12), value head (1), log-std parameter; plus the matrix-core forward for rollouts with skrl's
RunningStandardScaler observation preprocessor folded in (:96-99)."""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn


class SharedMLP(nn.Module):
    def __init__(self, num_observations=64, num_actions=12):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(num_observations, 256), nn.ELU(), nn.Linear(256, 128), nn.ELU(), nn.Linear(128, 64), nn.ELU())
        self.mean_layer = nn.Linear(64, num_actions)
        self.log_std_parameter = nn.Parameter(torch.zeros(num_actions))
        self.value_layer = nn.Linear(64, 1)
        self._packed = None

    def forward(self, obs):
        h = self.net(obs)
        return self.mean_layer(h), self.log_std_parameter, self.value_layer(h)

    @torch.no_grad()
    def act_inference(self, obs, obs_mean=None, obs_var=None, eps=1e-8, clip=5.0):
        if self._packed is None or self._packed.device != obs.device:
            self.refresh(obs.device, obs_mean, obs_var, eps, clip)
        mean, value = mlp_forward_hip(obs, self._packed)
        return mean, self.log_std_parameter, value

    def refresh(self, device=None, obs_mean=None, obs_var=None, eps=1e-8, clip=5.0):
        self._packed = pack_mlp_params(self, obs_mean, obs_var, eps, clip).to(device or self.log_std_parameter.device)


def _permute(W: torch.Tensor, natural_k: bool) -> torch.Tensor:
    """(out, in) weight -> [out/16][in/4][64] in the lane order v_mfma_f32_16x16x4_f32 reads its A operand:
    lane l supplies row 16*mb + (l & 15); its k index is 4*step + (l >> 4) for an input in natural order, and
    16*(step >> 2) + 4*(l >> 4) + (step & 3) for an input that is a previous layer's accumulator tile."""
    out_f, in_f = W.shape
    lane = torch.arange(64); n, g = lane & 15, lane >> 4
    mb = torch.arange(out_f // 16).view(-1, 1, 1); st = torch.arange(in_f // 4).view(1, -1, 1)
    row = 16 * mb + n.view(1, 1, -1)
    col = (4 * st + g.view(1, 1, -1)) if natural_k else (16 * (st // 4) + 4 * g.view(1, 1, -1) + (st % 4))
    return W[row.expand(-1, in_f // 4, -1), col.expand(out_f // 16, -1, -1)].reshape(-1)


def pack_mlp_params(m: SharedMLP, obs_mean=None, obs_var=None, eps=1e-8, clip=5.0) -> torch.Tensor:
    dev = m.log_std_parameter.device
    nobs = m.net[0].in_features
    assert nobs in (64, 88), "the MFMA forward is built for the 64- and 88-wide observations"
    mean = torch.zeros(nobs, device=dev) if obs_mean is None else obs_mean.detach().float().to(dev)
    # RunningStandardScaler: (x - mean) / (sqrt(var) + eps), then clamp(+-clip); identity statistics when absent
    istd = torch.ones(nobs, device=dev) if obs_var is None else 1.0 / (obs_var.detach().float().to(dev).sqrt() + eps)
    clipv = torch.tensor([clip if obs_var is not None else 3.0e38, 0, 0, 0], device=dev, dtype=torch.float32)
    l1, l2, l3 = m.net[0], m.net[2], m.net[4]
    Wh = torch.zeros(16, 64, device=dev); Wh[:12] = m.mean_layer.weight.detach(); Wh[12] = m.value_layer.weight.detach()[0]
    bh = torch.zeros(16, device=dev); bh[:12] = m.mean_layer.bias.detach(); bh[12] = m.value_layer.bias.detach()[0]
    parts = [mean, istd, clipv, _permute(l1.weight.detach(), True), l1.bias.detach(), _permute(l2.weight.detach(), False), l2.bias.detach(),
             _permute(l3.weight.detach(), False), l3.bias.detach(), _permute(Wh, False), bh]
    return torch.cat([p.reshape(-1).float() for p in parts]).contiguous()


def mlp_forward_hip(obs: torch.Tensor, packed: torch.Tensor):
    from ..lib import load_library
    lib = load_library()
    assert obs.is_cuda and obs.dtype == torch.float32 and obs.is_contiguous() and obs.shape[1] in (64, 88)
    assert packed.device == obs.device and packed.dtype == torch.float32 and packed.is_contiguous() and packed.numel() == lib.lm_mlp_param_count_obs(obs.shape[1])
    B = obs.shape[0]
    mean = torch.empty((B, 12), device=obs.device); value = torch.empty((B, 1), device=obs.device)
    with torch.cuda.device(obs.device):            # the kernel launches on the calling thread's current device
        rc = lib.lm_mlp_forward_obs(C.c_void_p(obs.data_ptr()), B, obs.shape[1], C.c_void_p(packed.data_ptr()), C.c_void_p(mean.data_ptr()),
                                C.c_void_p(value.data_ptr()), C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"lm_mlp_forward failed ({rc})")
    return mean, value
