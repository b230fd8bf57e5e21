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

Rank 0 also reports, outside the timed region: the same workload through the drop-in boundary itself (`VecEnvRLGames.step`,
`config.vec_env_step_env_steps_per_s`: fresh output tensors and the extras dict per call, as the reference's wrapper hands them out),
the zero-action protocol of SURVEY 8(d), the same envs with the reference's MLP policy in the loop (BASELINE config 2 names one):
48-step rollouts as one persistent kernel (`config.mlp_policy_in_loop_...`), BASELINE config 3 and config 4's per-GPU block (`config.manipulation_...`,
`config.cotrain_block_...`), one PD-actuator task family (`config.pd_family_...`) and BASELINE config 5 (`config.config5_...`: vertical co-training, 8192 envs, GNN in the loop).

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process starts `torch.distributed.run` with N ranks as a child BEFORE
touching the GPU and exits with its code.

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
# rocprofv3 --pmc passes of this same command (FETCH_SIZE / WRITE_SIZE / flop counters, tools/profile_round.sh): counters cannot be
# collected from inside the run, so `roofline.traffic` and `valu.flop_per_env_step` are READ FROM the newest of these files and labelled so
PMC_FILES = [os.path.join(ROOT, "profiles", f) for f in ("r04_pmc.json", "r03_pmc.json", "r02_pmc.json")]


def _omp_threads(n):
    os.environ["OMP_NUM_THREADS"] = str(n)
    try:      # a later call in one process: the OpenMP runtime is already up and ignores the environment
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(n)
    except OSError:
        pass


def _time_oracle(o, envs, steps, seed=42):
    import numpy as np
    phys, task, cnt = o.new_state(envs)
    rng = np.random.default_rng(seed)
    acts = rng.uniform(-1, 1, size=(steps + 2, envs, 12)).astype(np.float32)
    o.step(phys, task, cnt, acts[0], seed=seed); o.step(phys, task, cnt, acts[1], seed=seed)
    t0 = time.perf_counter()
    for t in range(steps):
        o.step(phys, task, cnt, acts[2 + t], seed=seed)
    return envs * steps / (time.perf_counter() - t0), time.perf_counter() - t0


def cpu_baseline(steps: int = 150, envs: int = 4096):
    """The CPU oracle (float32 build, OpenMP over envs) timed on this host on bounded samples of the same workloads: every hardware
    thread this process may use and one core, on BASELINE configs 2 (the headline), 3 and the per-GPU block of config 4."""
    from locomanipulationrl_amd.engine_config import loco_params, mani_params
    from locomanipulationrl_amd.model.robot_model import load_model
    from oracle.lmo import Oracle
    nproc = os.cpu_count() or 1
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else nproc
    rm = load_model("quadruped_robot_v2")
    o = Oracle(rm, loco_params(), "f32")
    # "all host cores" = all this process is ALLOWED to use: the affinity mask can name every hardware thread of the host while the
    # container's CPU quota is a fraction of it (oversubscribed OpenMP threads are then throttled: 256 threads ran 20x slower than 64 here).
    # Take the cgroup CPU quota when there is one; otherwise keep the thread count that is fastest on a short probe.
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else max(1, int(-(-int(q) // int(per))))
    except (OSError, ValueError):
        pass
    probe = {}
    if quota is not None:
        usable = min(quota, affinity)          # a short probe would ride the quota's burst allowance (64 threads probe 3x faster and then run 2x slower)
    else:
        for c in sorted({c for c in (8, 16, 32, 64, 128, affinity) if c <= affinity}):
            _omp_threads(c)
            probe[c] = _time_oracle(o, 2048, 4)[0]
        usable = max(probe, key=probe.get)
    _omp_threads(usable)
    rate, dt = _time_oracle(o, envs, steps)
    out = {"value": rate, "unit": "env-steps/s", "cores": usable, "nproc": nproc, "affinity": affinity, "cgroup_cpu_quota": quota, "kind": "port",
           "thread_probe_env_steps_per_s": {str(k): round(v) for k, v in probe.items()},
           "sample": f"{envs} envs x {steps} steps of the same task on the CPU oracle (fp32 build, OpenMP over envs, {usable} threads = the CPUs this process "
                     f"may use; nproc {nproc}, affinity {affinity}, cgroup quota {quota}; {dt:.1f} s); "
                     "PhysX-CPU itself is unavailable (closed source, not installed)"}
    om = Oracle(rm, mani_params(), "f32")
    out["config3_manipulation_4096"] = _time_oracle(om, envs, steps // 2)[0]
    # config 4's per-GPU block: 2048 locomotion + 2048 manipulation envs (two oracles, timed back to back)
    cl = __import__("dataclasses").replace
    lo = Oracle(rm, cl(loco_params(), init_q=[-1.2, 1.2, 1.2, -1.2, -1.22, -1.92, 1.92, 1.22, 1.92, 1.22, -1.22, -1.92], init_base_pos=[0, 0, 0.18]), "f32")
    mo = Oracle(rm, cl(mani_params(), init_q=[-1.2, 1.2, 1.2, -1.2, -1.22, -1.92, 1.92, 1.22, 1.92, 1.22, -1.22, -1.92], fixed_base_pos=[0, 0, 0.5], init_plate_pos=[0, 0, 0.68]), "f32")
    ra, rb = _time_oracle(lo, envs // 2, steps // 2)[0], _time_oracle(mo, envs // 2, steps // 2)[0]
    out["config4_cotrain_block_2048_2048"] = envs / (envs / 2 / ra + envs / 2 / rb)
    _omp_threads(1)      # the same oracle on ONE core
    n1, s1 = 1024, 12
    r1, d1 = _time_oracle(o, n1, s1)
    out["value_1core"] = r1
    out["sample"] += f"; one core: {n1} envs x {s1} steps ({d1:.1f} s)"
    return out


def _self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start the N ranks as children of a process that has not touched
    the GPU (never re-exec a process that has) and hand back their exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup)]
    cmd += ["--no-cpu-baseline"] if args.no_cpu_baseline else []
    cmd += ["--timed-only"] if args.timed_only else []
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-only", action="store_true", help="skip the event-timing and zero-action legs (counter collection runs: every k_step dispatch is then the timed protocol)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_launch(args))

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

    ag_events = []                                           # HIP events around each all-gather block of the timed region (world > 1)

    def one_step(t, timed=False):
        k = t % ROLLOUT
        eng.step(pool[t % 64], None, out_obs[t & 1], out_states[t & 1], roll_rew[k], roll_done[k], out_extras)
        if world > 1 and k == ROLLOUT - 1:
            if timed:
                ag_events.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))); ag_events[-1][0].record()
            payload[0].copy_(roll_rew); payload[1].copy_(roll_done)
            dist.all_gather_into_tensor(gathered, payload)
            if timed:
                ag_events[-1][1].record()

    def barrier():
        if world > 1:
            torch.cuda.synchronize(dev)
            dist.barrier(device_ids=[dev_index]) if backend == "nccl" else dist.barrier()
        torch.cuda.synchronize(dev)

    for t in range(args.warmup):
        one_step(t)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)      # HIP events on the launch stream, around the timed region
    ev0.record(); ev1.record(); torch.cuda.synchronize(dev)      # torch creates the HIP events at their first record: not inside the timed region
    t0 = time.perf_counter()
    ev0.record()
    for t in range(args.steps):
        one_step(args.warmup + t, timed=True)
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    region_ms = ev0.elapsed_time(ev1)
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(out_obs[0]).all() and torch.isfinite(roll_rew).all()

    result = None
    if rank == 0:
        # dominant-kernel time: the HIP events around the timed region / launches (k_step is the only kernel in it at N = 1; a launch every
        # `kernel_ms`, of which the kernel itself runs all but the ~3.5 us between dependent launches - rocprofv3's average duration of the same
        # command, profiles/r02_kernel_stats.csv, agrees to 1 %).  Events around single launches (outside the timed region) add their own
        # packets to every launch and read ~3 us more: reported as kernel_ms_single_launch_events.
        n_ev = 0 if args.timed_only else 200
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
        for i, (a, b) in enumerate(evs):
            a.record(); eng.step(pool[i % 64], None, out_obs[0], out_states[0], roll_rew[0], roll_done[0], out_extras); b.record()
        torch.cuda.synchronize(dev)
        k_ms = sorted(a.elapsed_time(b) for a, b in evs) or [region_ms / args.steps]
        step_period_ms = region_ms / args.steps                                            # launch period of the timed region, collectives included
        k_avg_ms = (region_ms - sum(a.elapsed_time(b) for a, b in ag_events)) / args.steps      # ... without the all-gather blocks: k_step's launch period
        achieved = BYTES_PER_ENV_STEP * N / (k_avg_ms * 1e-3) / 1e9
        pmc_file = next((f for f in PMC_FILES if os.path.exists(f)), None)
        pmc = json.load(open(pmc_file)) if pmc_file else None
        traffic = pmc["per_launch"]["hbm_traffic_bytes"] if pmc else None
        wave = (pmc or {}).get("wave")
        pmc_src = ("read from " + os.path.relpath(pmc_file, ROOT) + " (rocprofv3 --pmc passes of this command; not collected in this run)") if pmc else None
        value = world * N * args.steps / elapsed
        # second protocol of SURVEY 8(d): zero actions (standing robots, only the 300-step timeout resets); rank 0, untimed region
        zact = torch.zeros(N, 12, device=dev)
        nz = 0 if args.timed_only else 500
        for _ in range(nz // 10): eng.step(zact, None, out_obs[0], out_states[0], roll_rew[0], roll_done[0], out_extras)
        torch.cuda.synchronize(dev); tz = time.perf_counter()
        for _ in range(nz): eng.step(zact, None, out_obs[0], out_states[0], roll_rew[0], roll_done[0], out_extras)
        torch.cuda.synchronize(dev); zero_rate = N * nz / (time.perf_counter() - tz) if nz else None
        # the drop-in boundary itself: VecEnvRLGames.step (vec_env_rlgames.py:56-79) on the same workload -- fresh output tensors and the
        # extras dict per call; rank 0, untimed region
        vec_rate = None
        if not args.timed_only:
            import contextlib, io
            import locomanipulationrl_amd as lm
            env = lm.make_env("QuadrupedPoseControl", num_envs=N, seed=42, sim_device=str(dev), rl_device=str(dev))
            with contextlib.redirect_stdout(io.StringIO()):
                env.reset()
            for i in range(200): env.step(pool[i % 64])
            torch.cuda.synchronize(dev); tv = time.perf_counter()
            nv = 2000
            for i in range(nv): obs_d, rw_, rs_, ex_ = env.step(pool[i % 64])
            torch.cuda.synchronize(dev); vec_rate = N * nv / (time.perf_counter() - tv)
            assert obs_d["obs"].shape == (N, 64) and "env/success_rate" in ex_
            env.close()
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
        # other workloads of the path, same protocol (fresh U(-1,1) actions, 4096 envs), rank 0, untimed region: BASELINE config 3, config 4's
        # per-GPU block, one PD-actuator family
        def rate_of(params, split=None, obs=64, steps=600, n=N):
            e2 = Engine(load_model("quadruped_robot_v2"), params, n, seed=42, device=str(dev), **({} if split is None else dict(split_env=split)))
            oo, ss = torch.empty(n, obs, device=dev), torch.empty(n, 93, device=dev)
            acts = pool if n == N else torch.rand(16, n, 12, device=dev) * 2 - 1
            rew_, done_ = (roll_rew[0], roll_done[0]) if n == N else (torch.empty(n, device=dev), torch.empty(n, dtype=roll_done.dtype, device=dev))
            for i in range(100): e2.step(acts[i % len(acts)], None, oo, ss, rew_, done_, out_extras)
            torch.cuda.synchronize(dev); t_ = time.perf_counter()
            for i in range(steps): e2.step(acts[i % len(acts)], None, oo, ss, rew_, done_, out_extras)
            torch.cuda.synchronize(dev); r_ = n * steps / (time.perf_counter() - t_)
            e2.close()
            return r_
        extra_rates = {}
        if not args.timed_only:
            from locomanipulationrl_amd.engine_config import loco_cc_params, mani_params
            cq = [-1.2, 1.2, 1.2, -1.2, -1.22, -1.92, 1.92, 1.22, 1.92, 1.22, -1.22, -1.92]
            extra_rates = {
                "manipulation_env_steps_per_s": rate_of([mani_params()]),
                "cotrain_block_env_steps_per_s": rate_of([loco_params(init_q=cq, init_base_pos=[0, 0, 0.18]),
                                                          mani_params(init_q=cq, fixed_base_pos=[0, 0, 0.5], init_plate_pos=[0, 0, 0.68])], split=N // 2),
                "pd_family_env_steps_per_s": rate_of([loco_cc_params()], obs=88),
                # BASELINE config 4 whole (32 768 co-training envs, 16 384 + 16 384) on ONE GPU: two generations of 1024 wavefronts; the stand-in
                # for the 8-GPU case while no 8-GPU node exists (there each GPU runs the 2048 + 2048 block above)
                # the top of the throughput curve: 131 072 locomotion envs on one GPU (beyond 32 768 envs lm_step launches the two-wavefronts-per-SIMD
                # build of the step kernel)
                "locomotion_131072_envs_env_steps_per_s": rate_of([loco_params()], steps=100, n=131072),
                "config4_all_32768_envs_on_one_gpu_env_steps_per_s": rate_of([loco_params(init_q=cq, init_base_pos=[0, 0, 0.18]),
                                                          mani_params(init_q=cq, fixed_base_pos=[0, 0, 0.5], init_plate_pos=[0, 0, 0.68])], split=16384, steps=200, n=32768),
            }
            # BASELINE config 5: the vertical co-training task, 8192 envs, the reference's GNN policy (random init, seed 42) in the loop --
            # 48-step rollouts (GNN forward on the matrix cores, fp32 products as split fp16 -> sampling -> step) replayed as one hipGraph; rank 0, untimed region
            from locomanipulationrl_amd.lib import POLICY_GNN
            from locomanipulationrl_amd.policies.graph_model import GraphPolicy, pack_gnn_params
            from locomanipulationrl_amd.utils.config import SimConfig, load_config
            from locomanipulationrl_amd.utils.task_util import task_map
            N5 = 8192; name5 = "JointLocomanipulationVertical"
            t5 = task_map()[name5](name=name5, sim_config=SimConfig(load_config(name5, num_envs=N5)), env=None)
            e5 = Engine(load_model(t5.model_asset), t5.engine_params(), N5, split_env=t5.split_env(), seed=42, device=str(dev))
            torch.manual_seed(42); gm = GraphPolicy().to(dev)
            o5 = torch.empty(N5, e5.num_obs, device=dev); e5.step(torch.zeros(N5, 12, device=dev), None, o5)
            r5 = Rollout(e5, POLICY_GNN, pack_gnn_params(gm.net, gm.mean_layer, gm.value_layer).to(dev), torch.full((12,), -0.5, device=dev), ROLLOUT, noise_seed=42)
            r5.obs[0] = o5
            for _ in range(3): r5.run("auto"); r5.obs[0].copy_(r5.obs[ROLLOUT])
            torch.cuda.synchronize(dev); t_ = time.perf_counter()
            for _ in range(10): r5.run("auto"); r5.obs[0].copy_(r5.obs[ROLLOUT])
            torch.cuda.synchronize(dev); extra_rates["config5_vertical_cotrain_8192_gnn_in_loop_env_steps_per_s"] = N5 * ROLLOUT * 10 / (time.perf_counter() - t_)
            r5.close(); e5.close()
        result = {
            "metric": "env-steps/sec (whole node), horizontal-locomotion 4096 envs", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic" if backend == "nccl" else "synthetic (REHEARSAL over gloo, ranks share GPUs: not a measurement)",
            "config": {"workload": "QuadrupedPoseControl (horizontal locomotion), 4096 envs per GPU, actions U(-1,1) fresh each step, "
                                   "dt 0.0083 x 4 sub-steps, 8 PGS sweeps with cone friction, obs 64 / states 93",
                       "envs_per_gpu": N, "global_envs": world * N, "physics_substeps_per_s": value * 4,
                       # `value` divides by the wall clock around the timed region INCLUDING its closing barrier + synchronize (at 20 steps that
                       # tail is a fifth of the window); the same region by the HIP events on the launch stream, rank 0:
                       "event_timed_env_steps_per_s_rank0": N * args.steps / (region_ms * 1e-3),
                       "vec_env_step_env_steps_per_s": vec_rate,
                       "zero_action_env_steps_per_s_rank0": zero_rate,
                       "mlp_policy_in_loop_env_steps_per_s_rank0": mlp_rate, **extra_rates,
                       "drive_limit_reading": "max effort 1.5 read as an impulse limit per step (never binds; DESIGN.md 2.1 / 2.2: the reference's recordings rule the 1.5 N m torque clamp out - it is the replay test's negative control; its bench leg and YAML switch were removed in round 4)",
                       "all_gather_blocks_timed": len(ag_events),      # N > 1: the per-48-step all-gathers inside the timed region, rank 0
                       "all_gather_ms_per_block_rank0": (sum(a.elapsed_time(b) for a, b in ag_events) / len(ag_events)) if ag_events else None,
                       "parallelism": f"env-sharded x{world}, all-gather(2,48,N) per 48 steps"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": pmc_src, "kernel": "k_step",
                         # what plain streaming kernels reach on an MI355X of this pool (tools/microbench/stream_copy.hip; read from profiles/, not run here)
                         "peak_measured_stream": (lambda f: json.load(open(f)) if os.path.exists(f) else None)(os.path.join(ROOT, "profiles", "r04_stream_copy.json")),
                         "kernel_ms": k_avg_ms, "kernel_ms_definition": "HIP events around the timed region on the launch stream, minus the all-gather blocks, / steps = k_step's launch "
                                                                        "period (its duration + ~3.5 us between dependent launches; rocprofv3's average duration is in profiles/)",
                         "step_period_ms": step_period_ms, "kernel_ms_single_launch_events": k_ms[len(k_ms) // 2],
                         "note": "1488 algorithmic B/env-step x 4096 envs per launch; the path is fp32-VALU / latency bound, see 'valu'"},
            # one wavefront per SIMD at 4096 envs: the binding resource is the wavefront's own instruction stream (fp32 VALU issue slots +
            # exposed latency), reported from the PMC passes; `step_quad_cycles` = this run's kernel time in quad-cycles at 2.4 GHz
            "valu": None if wave is None else dict(wave, waves_per_launch=N * 4 // 64, simds=1024,
                                                   step_quad_cycles=k_avg_ms * 1e-3 * 2.4e9 / 4, source=pmc_src),
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
