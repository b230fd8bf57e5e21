#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of VecEnv step() on the horizontal-locomotion task, 4096 envs per GPU.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one lm_step over the rank's 4096 environments = reset scatter + action clamp/scale + 4 physics
sub-steps + observations / reward / termination (reference: VecEnvRLGames.step, vec_env_rlgames.py:56-79).
Actions are fresh U(-1,1) draws per step (BASELINE config 2, like scripts/random_policy.py:57) taken from a pool
resident in HBM before the timed region.  N > 1 = weak scaling: every rank owns 4096 envs; every 48 steps
(the PPO rollout length, skrl_ppo_locomotion.py:86) the ranks all-gather a (2, 48, N_local) fp32 payload --
the size of the rollout's returns + advantages -- over RCCL, inside the timed region.

Rank 0 also reports, outside the timed region: the zero-action protocol of SURVEY 8(d) and the same envs with the reference's MLP
policy in the loop (BASELINE config 2 names one): 48-step rollouts as one persistent kernel (`config.mlp_policy_in_loop_...`).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
ROLLOUT = 48
BYTES_PER_ENV_STEP = 1488          # algorithmic HBM bytes per env-step (460 read + 1028 written)
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_PEAK_TFLOPS = 157.3           # vector fp32 peak
PMC_FILE = os.path.join(ROOT, "profiles", "r01_pmc_v6.json")   # rocprofv3 --pmc passes of this same command (FETCH/WRITE_SIZE, flop counters)


def cpu_baseline(steps: int = 150, envs: int = 4096, params=None):
    """The CPU oracle (float32 build, OpenMP over envs) timed on this host on a bounded sample of the same workload
    (params: engine parameters of another task family, default = the headline locomotion task)."""
    import numpy as np
    from locomanipulationrl_amd.engine_config import loco_params
    from locomanipulationrl_amd.model.robot_model import load_model
    from oracle.lmo import Oracle
    cores = os.cpu_count() or 1
    cores = min(cores, 64)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    try:      # a second call in one process: the OpenMP runtime is already up and ignores the environment
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)
    except OSError:
        pass
    o = Oracle(load_model("quadruped_robot_v2"), params or loco_params(), "f32")
    phys, task, cnt = o.new_state(envs)
    rng = np.random.default_rng(42)
    acts = rng.uniform(-1, 1, size=(steps + 2, envs, 12)).astype(np.float32)
    o.step(phys, task, cnt, acts[0], seed=42); o.step(phys, task, cnt, acts[1], seed=42)
    t0 = time.perf_counter()
    for t in range(steps):
        o.step(phys, task, cnt, acts[2 + t], seed=42)
    dt = time.perf_counter() - t0
    out = {"value": envs * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
           "sample": f"{envs} envs x {steps} steps of the same task on the CPU oracle (fp32 build, OpenMP over envs, {dt:.1f} s); "
                     "PhysX-CPU itself is unavailable (closed source, not installed)"}
    try:      # the same oracle on ONE core (SURVEY 8d asks for both)
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(1)
        n1, s1 = 1024, 12
        p1, t1, c1 = o.new_state(n1)
        o.step(p1, t1, c1, acts[0][:n1], seed=42)
        t0 = time.perf_counter()
        for t in range(s1):
            o.step(p1, t1, c1, acts[(2 + t) % len(acts)][:n1], seed=42)
        d1 = time.perf_counter() - t0
        out["value_1core"] = n1 * s1 / d1
        out["sample"] += f"; one core: {n1} envs x {s1} steps ({d1:.1f} s)"
    except OSError:
        pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-only", action="store_true", help="skip the event-timing and zero-action legs (counter collection runs: every k_step dispatch is then the timed protocol)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from locomanipulationrl_amd import distributed as D
    from locomanipulationrl_amd.engine_config import loco_params
    from locomanipulationrl_amd.lib import Engine
    from locomanipulationrl_amd.model.robot_model import load_model

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # LM_BENCH_BACKEND=gloo rehearses the multi-rank logic on a box with fewer GPUs than ranks (ranks then share devices); the
    # measured configuration is always nccl (= RCCL), one GPU per rank
    backend = os.environ.get("LM_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)                        # before the process group: RCCL binds to the current device
    dev = torch.device("cuda", dev_index)
    rank, local_rank, world = D.init_from_env(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    N = ENVS_PER_GPU
    eng = Engine(load_model("quadruped_robot_v2"), [loco_params()], N, seed=42 + rank, device=str(dev))
    gen = torch.Generator(device=dev).manual_seed(42 + rank)
    pool = [torch.rand(N, 12, device=dev, generator=gen) * 2 - 1 for _ in range(64)]
    out_obs = [torch.empty(N, 64, device=dev) for _ in range(2)]
    out_states = [torch.empty(N, 93, device=dev) for _ in range(2)]
    out_extras = torch.empty(13, device=dev)
    roll_rew = torch.zeros(ROLLOUT, N, device=dev)           # the kernel writes rewards / dones straight into the rollout slots
    roll_done = torch.zeros(ROLLOUT, N, dtype=torch.int64, device=dev)
    payload = torch.zeros(2, ROLLOUT, N, device=dev)
    gathered = torch.empty(2 * world, ROLLOUT, N, device=dev) if world > 1 else None

    def one_step(t):
        k = t % ROLLOUT
        eng.step(pool[t % 64], None, out_obs[t & 1], out_states[t & 1], roll_rew[k], roll_done[k], out_extras)
        if world > 1 and k == ROLLOUT - 1:
            payload[0].copy_(roll_rew); payload[1].copy_(roll_done)
            dist.all_gather_into_tensor(gathered, payload)

    def barrier():
        if world > 1:
            torch.cuda.synchronize(dev)
            dist.barrier(device_ids=[dev_index]) if backend == "nccl" else dist.barrier()
        torch.cuda.synchronize(dev)

    for t in range(args.warmup):
        one_step(t)
    barrier()
    t0 = time.perf_counter()
    for t in range(args.steps):
        one_step(args.warmup + t)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(out_obs[0]).all() and torch.isfinite(roll_rew).all()

    result = None
    if rank == 0:
        # dominant-kernel time: HIP events around each launch on the launch stream (outside the timed region)
        n_ev = 0 if args.timed_only else 200
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
        for i, (a, b) in enumerate(evs):
            a.record(); eng.step(pool[i % 64], None, out_obs[0], out_states[0], roll_rew[0], roll_done[0], out_extras); b.record()
        torch.cuda.synchronize(dev)
        k_ms = sorted(a.elapsed_time(b) for a, b in evs) or [elapsed / args.steps * 1e3]
        k_avg_ms = sum(k_ms) / len(k_ms)
        achieved = BYTES_PER_ENV_STEP * N / (k_avg_ms * 1e-3) / 1e9
        pmc = json.load(open(PMC_FILE)) if os.path.exists(PMC_FILE) else None
        traffic = pmc["per_launch"]["hbm_traffic_bytes"] if pmc else None
        flop_env = pmc["flop_per_env_step"] if pmc else 1.67e5
        value = world * N * args.steps / elapsed
        # second protocol of SURVEY 8(d): zero actions (standing robots, only the 300-step timeout resets); rank 0, untimed region
        zact = torch.zeros(N, 12, device=dev)
        nz = 0 if args.timed_only else 500
        for _ in range(nz // 10): eng.step(zact, None, out_obs[0], out_states[0], roll_rew[0], roll_done[0], out_extras)
        torch.cuda.synchronize(dev); tz = time.perf_counter()
        for _ in range(nz): eng.step(zact, None, out_obs[0], out_states[0], roll_rew[0], roll_done[0], out_extras)
        torch.cuda.synchronize(dev); zero_rate = N * nz / (time.perf_counter() - tz) if nz else None
        # BASELINE config 2 names an MLP policy: the same envs with the reference's 64-256-128-64 MLP (random init, seed 42) in the loop --
        # 48-step rollouts (forward on fp32 MFMA -> gaussian sampling -> step) as one persistent kernel; rank 0, untimed region
        mlp_rate = None
        if not args.timed_only:
            from locomanipulationrl_amd.lib import POLICY_MLP, Rollout
            from locomanipulationrl_amd.policies.mlp_model import SharedMLP, pack_mlp_params
            torch.manual_seed(42)
            pol = SharedMLP().to(dev); packed = pack_mlp_params(pol, None, None).to(dev)
            ro = Rollout(eng, POLICY_MLP, packed, torch.full((12,), -0.5, device=dev), ROLLOUT, noise_seed=42)
            eng.step(zact, None, ro.obs[0])
            for _ in range(3): ro.run("auto"); ro.obs[0].copy_(ro.obs[ROLLOUT])
            torch.cuda.synchronize(dev); tr = time.perf_counter()
            for _ in range(20): ro.run("auto"); ro.obs[0].copy_(ro.obs[ROLLOUT])
            torch.cuda.synchronize(dev); mlp_rate = N * ROLLOUT * 20 / (time.perf_counter() - tr)
            ro.close()
        result = {
            "metric": "env-steps/sec (whole node), horizontal-locomotion 4096 envs", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic" if backend == "nccl" else "synthetic (REHEARSAL over gloo, ranks share GPUs: not a measurement)",
            "config": {"workload": "QuadrupedPoseControl (horizontal locomotion), 4096 envs per GPU, actions U(-1,1) fresh each step, "
                                   "dt 0.0083 x 4 sub-steps, 8 PGS sweeps, obs 64 / states 93",
                       "envs_per_gpu": N, "global_envs": world * N, "physics_substeps_per_s": value * 4,
                       "zero_action_env_steps_per_s_rank0": zero_rate,
                       "mlp_policy_in_loop_env_steps_per_s_rank0": mlp_rate, "parallelism": f"env-sharded x{world}, all-gather(2,48,N) per 48 steps"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "k_step", "kernel_ms": k_avg_ms, "kernel_ms_median": k_ms[len(k_ms) // 2],
                         "note": "1488 algorithmic B/env-step x 4096 envs per launch; the path is fp32-VALU / latency bound, see 'valu'"},
            "valu": {"achieved": flop_env * N / (k_avg_ms * 1e-3) / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": flop_env * N / (k_avg_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, "waves_per_launch": N * 4 // 64,
                     "flop_per_env_step": flop_env},
        }
    eng.close()
    if world > 1:
        dist.barrier(device_ids=[dev_index]) if backend == "nccl" else dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline()
        print(json.dumps(result))


if __name__ == "__main__":
    main()
