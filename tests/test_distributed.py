"""N>1 path on CPU: world_size-2 gloo processes shard the env range, step independently (no collective in the
data path) and exchange rollout returns/advantages with one all-gather."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except Exception:      # reported to the parent instead of a silent non-zero exit (the parent retries a failed rendezvous once on a fresh port)
        import traceback
        q.put(("error", rank, traceback.format_exc()))
        raise


def _worker_body(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import locomanipulationrl_amd as lm
    from locomanipulationrl_amd import distributed as D
    from oracle_backend import oracle_engine_factory
    r, lr, w = D.init_from_env("gloo")
    start, count = D.shard_envs(64, r, w)
    env = lm.make_env("QuadrupedPoseControl", num_envs=count, engine_factory=oracle_engine_factory, sim_device="cpu", rl_device="cpu", rank=r)
    env.reset()
    g = torch.Generator().manual_seed(100 + r)
    T = 6
    rew = torch.zeros(T, count); done = torch.zeros(T, count); val = torch.zeros(T, count)
    for t in range(T):
        _, rw, rs, ex = env.step(torch.rand(count, 12, generator=g) * 2 - 1)
        rew[t], done[t] = rw, rs.float()
    ret, adv = D.compute_gae(rew, val, done, torch.zeros(count))
    gret, gadv = D.all_gather_rollout(ret, adv)
    gex = D.global_extras(ex, count)
    goal = env._task.goal_quaternions.clone()
    q.put((r, start, count, ret, gret, gadv, float(gex["env/rewards/orientation_rew"]), float(ex["env/rewards/orientation_rew"]), goal))
    dist.barrier(); dist.destroy_process_group()


def _run_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    try:
        import queue
        res = []
        for _ in range(2):
            try: x = q.get(timeout=240)
            except queue.Empty: return None, "a rank did not report within 240 s"
            if x[0] == "error": return None, x[2]      # (the other rank, blocked in the rendezvous, is terminated below)
            res.append(x)
        for p in procs: p.join(120)
        if not all(p.exitcode == 0 for p in procs):
            return None, f"exit codes {[p.exitcode for p in procs]}"
        return sorted(res, key=lambda x: x[0]), ""
    finally:
        for p in procs:
            if p.is_alive(): p.terminate()      # our own children, by handle


def test_two_rank_sharding_and_rollout_allgather():
    res, why = _run_two_ranks()
    if res is None:      # the free port can be taken between its probe and the rendezvous: one retry on a fresh port
        print("first attempt failed:", why)
        res, why = _run_two_ranks()
    assert res is not None, why
    (r0, s0, c0, ret0, g0, a0, ge0, e0, goal0), (r1, s1, c1, ret1, g1, a1, ge1, e1, goal1) = res
    assert (s0, c0, s1, c1) == (0, 32, 32, 32)
    assert g0.shape == (6, 64) and torch.equal(g0, g1) and torch.equal(a0, a1)           # every rank holds the global rollout
    assert torch.equal(g0[:, :32], ret0) and torch.equal(g0[:, 32:], ret1)               # rank-major env order
    assert abs(ge0 - 0.5 * (e0 + e1)) < 1e-6 and abs(ge0 - ge1) < 1e-7                   # env-weighted global mean
    assert not torch.equal(goal0, goal1)                                                  # per-rank RNG streams (seed + rank)


def test_shard_envs_and_gae_single_process():
    from locomanipulationrl_amd import distributed as D
    assert D.shard_envs(32768, 3, 8, multiple=32) == (3 * 4096, 4096)
    with pytest.raises(AssertionError):
        D.shard_envs(100, 0, 8)
    r = torch.ones(3, 2); v = torch.zeros(3, 2); d = torch.zeros(3, 2); d[1, 1] = 1
    ret, adv = D.compute_gae(r, v, d, torch.zeros(2), gamma=0.5, lam=1.0)
    assert torch.allclose(ret[:, 0], torch.tensor([1.75, 1.5, 1.0])) and torch.allclose(ret[:, 1], torch.tensor([1.5, 1.0, 1.0]))
    a, b = D.all_gather_rollout(ret, adv)
    assert a is ret and b is adv
