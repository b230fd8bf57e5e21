"""Multi-GPU sharding: one process per GPU, environments split into contiguous blocks, no collective inside
step(); per PPO update the ranks exchange rollout returns / advantages with one all-gather over RCCL
(backend "nccl" on ROCm) -- SURVEY 8(e).  The reference has no collective call site (its only hook is
LOCAL_RANK -> device id, scripts/rlgames_train.py:93-96)."""
from __future__ import annotations

import os
from typing import Dict, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment; initialises the process group when
    WORLD_SIZE > 1.  backend defaults to nccl (= RCCL) when a GPU is visible, else gloo."""
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_envs(total_envs: int, rank: int, world_size: int, multiple: int = 16) -> Tuple[int, int]:
    """Contiguous block [start, start+count) of the global env range owned by ``rank``; every block is a
    multiple of ``multiple`` envs (16 = one wavefront; 32 keeps co-train halves wave-aligned)."""
    assert total_envs % (world_size * multiple) == 0, (total_envs, world_size, multiple)
    count = total_envs // world_size
    return rank * count, count


def all_gather_rollout(returns: torch.Tensor, advantages: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-gather (T, N_local) returns and advantages into (T, N_global) along the env axis with ONE collective
    (the two tensors are packed, so an 8-GPU xGMI mesh sees a single latency-bound exchange per update)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return returns, advantages
    world = dist.get_world_size()
    packed = torch.stack((returns, advantages)).contiguous()                       # (2, T, Nl)
    out = torch.empty((world * 2,) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(out, packed)                                       # concatenated along dim 0
    out = out.view(world, 2, packed.shape[1], packed.shape[2]).permute(1, 2, 0, 3).reshape(2, packed.shape[1], world * packed.shape[2])   # rank-major env order
    return out[0], out[1]


def global_extras(extras: Dict[str, torch.Tensor], local_envs: int) -> Dict[str, torch.Tensor]:
    """Env-weighted mean of the logged scalars over ranks so dashboards show global values."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return extras
    keys = sorted(extras)
    v = torch.stack([extras[k].float() for k in keys]) * float(local_envs)
    n = torch.tensor([float(local_envs)], device=v.device)
    buf = torch.cat((v, n))
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return {k: buf[i] / buf[-1] for i, k in enumerate(keys)}


def compute_gae(rewards: torch.Tensor, values: torch.Tensor, dones: torch.Tensor, last_value: torch.Tensor,
                gamma: float = 0.99, lam: float = 0.95) -> Tuple[torch.Tensor, torch.Tensor]:
    """GAE(gamma, lambda) as configured by the reference's PPO scripts (skrl_ppo_locomotion.py:86-112):
    rewards/values/dones are (T, N); returns (returns, advantages)."""
    T = rewards.shape[0]
    adv = torch.zeros_like(rewards); last = torch.zeros_like(last_value)
    nxt = last_value
    for t in range(T - 1, -1, -1):
        nd = 1.0 - dones[t].float()
        delta = rewards[t] + gamma * nxt * nd - values[t]
        last = delta + gamma * lam * nd * last
        adv[t] = last; nxt = values[t]
    return adv + values, adv
