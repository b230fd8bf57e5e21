"""MLP policy of the reference's PPO scripts (scripts/skrl_ppo_locomotion.py:25-52): shared trunk 64-256-128-64 (ELU),
Gaussian mean head (12), value head (1), log-std parameter; plus the matrix-core forward for rollouts with skrl's
RunningStandardScaler observation preprocessor folded in (:96-99).  The forward computes its fp32 products on the fp16 matrix pipe with every
operand split into two fp16 halves (fp32-level accuracy): the weights are split here, on the host (`_pack_q`)."""
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


def _pack_q(W: torch.Tensor, natural_k: bool) -> torch.Tensor:
    """(out, in) fp32 weight -> the 32-bit words [out/16][KB][64 lanes][8] the MLP tile reads (csrc/lm_policy_dev.h): for output block mb,
    K-block kb (32 input features) and lane l = n + 16 g, the eight weights W[16 mb + n][col(kb, g, e)], e = 0..7, each split into two fp16
    halves (w = hi + lo, hi = fp16(w), lo = fp16(w - hi)): words 0..3 hold the hi halves (two per word, even e in the low 16 bits), words 4..7
    the lo halves.  col = 32 kb + 8 g + e for an input in natural order (the observation; columns beyond `in` are zero: 88 pads to 96), and
    16 (2 kb + (e >> 2)) + 4 g + (e & 3) for an input that is a previous layer's accumulator tile."""
    out_f, in_f = W.shape
    KB = (in_f + 31) // 32
    Wp = torch.zeros(out_f, 32 * KB, dtype=torch.float32, device=W.device); Wp[:, :in_f] = W.float()
    lane = torch.arange(64, device=W.device); n, g = lane & 15, lane >> 4
    mb = torch.arange(out_f // 16, device=W.device).view(-1, 1, 1, 1); kb = torch.arange(KB, device=W.device).view(1, -1, 1, 1)
    e = torch.arange(8, device=W.device).view(1, 1, 1, -1); nn, gg = n.view(1, 1, -1, 1), g.view(1, 1, -1, 1)
    row = (16 * mb + nn).expand(-1, KB, -1, 8)
    col = (32 * kb + 8 * gg + e) if natural_k else (16 * (2 * kb + (e >> 2)) + 4 * gg + (e & 3))
    w = Wp[row, col.expand(out_f // 16, -1, -1, -1)]                                   # [mb][kb][lane][e]
    if not bool(torch.isfinite(w).all()) or float(w.abs().max()) >= 6.0e4:
        raise ValueError("MLP weights must be finite and below the fp16 range (6e4) for the split-fp16 matrix products")
    hi = w.half(); lo = (w - hi.float()).half()                                          # round to nearest even, like the device's v_cvt_pk_f16_f32
    def pairs(h):                                                                        # fp16 [.., 8] -> int32 [.., 4]: even e in the low half
        u = h.view(torch.int16).to(torch.int32) & 0xFFFF
        return u[..., 0::2] | (u[..., 1::2] << 16)
    words = torch.cat([pairs(hi), pairs(lo)], dim=-1).contiguous()                       # [mb][kb][lane][8]
    return words.view(torch.float32).reshape(-1)                                         # bit patterns in a float32 tensor (never used as numbers)


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
    parts = [mean, istd, clipv, _pack_q(l1.weight.detach(), True), l1.bias.detach(), _pack_q(l2.weight.detach(), False), l2.bias.detach(),
             _pack_q(l3.weight.detach(), False), l3.bias.detach(), _pack_q(Wh, False), bh]
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
