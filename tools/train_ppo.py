#!/usr/bin/env python3
"""Behavioural acceptance run: train the reference's PPO recipe on the engine and log the success-rate curve.
    python tools/train_ppo.py --task QuadrupedPoseControl --num-envs 4096 --timesteps 4800 [--policy mlp|gnn] [--out profiles/x.json]
(multi-GPU: python -m torch.distributed.run --nproc-per-node N tools/train_ppo.py ...)"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import locomanipulationrl_amd as lm
from locomanipulationrl_amd import distributed as D
from locomanipulationrl_amd.train.ppo import PPO


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="QuadrupedPoseControl"); ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--timesteps", type=int, default=4800); ap.add_argument("--policy", default="mlp"); ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--out", default=""); ap.add_argument("--log-every", type=int, default=5)
    ap.add_argument("--randomize", action="store_true", help="switch the task YAML's domain_randomization block on (the reference's YAMLs ship it with randomize: False)")
    ap.add_argument("--fixed-lr", action="store_true", help="no KL-adaptive learning rate (diagnostics; not the reference recipe)")
    ap.add_argument("--gnn-env-order", action="store_true", help="diagnostic: route GNN node k's output to the joint whose state node k reads (the reference feeds node order straight to the env)")
    ap.add_argument("--no-obs-scaler", action="store_true", help="diagnostic: identity observation scaler")
    ap.add_argument("--no-hip", action="store_true", help="diagnostic: torch forward in the rollouts instead of the MFMA kernels")
    ap.add_argument("--no-fused", action="store_true", help="drive the rollout step by step from Python instead of the captured hipGraph")
    ap.add_argument("--max-lr", type=float, default=1e-2, help="diagnostic: cap of the KL-adaptive learning rate (skrl default 1e-2)")
    ap.add_argument("--min-log-std", type=float, default=None, help="diagnostic: floor of the log-std parameter (skrl clips at -20 only)")
    ap.add_argument("--engine", action="append", default=[], metavar="KEY=VALUE", help="diagnostic: a sim.engine override of the task YAML, e.g. friction_scale=1.0")
    a = ap.parse_args()
    # LM_DIST_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks then share devices)
    backend = os.environ.get("LM_DIST_BACKEND")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if backend == "gloo": local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    rank, _, world = D.init_from_env(backend)
    torch.manual_seed(a.seed + rank)
    env = lm.make_env(a.task, num_envs=a.num_envs, seed=a.seed, rank=rank, sim_device=f"cuda:{local}", rl_device=f"cuda:{local}",
                      overrides={"task": {**({"domain_randomization": {"randomize": True}} if a.randomize else {}),
                                          **({"sim": {"engine": {k: float(v) for k, v in (kv.split("=") for kv in a.engine)}}} if a.engine else {})}})
    if a.policy == "gnn":
        from locomanipulationrl_amd.policies.graph_model import GraphPolicy
        model = GraphPolicy().to(f"cuda:{local}"); hip = True
    else:
        from locomanipulationrl_amd.policies.mlp_model import SharedMLP
        model = SharedMLP(num_observations=env.observation_space.shape[0]).to(f"cuda:{local}"); hip = True
    if world > 1:
        for p in model.parameters(): torch.distributed.broadcast(p.data, 0)
    if a.gnn_env_order:
        inv = torch.empty(12, dtype=torch.long); 
        for k in range(12): inv[k if k < 4 else (4 + 2 * (k - 4) if k < 8 else 5 + 2 * (k - 8))] = k
        inv = inv.to(f"cuda:{local}"); _step = env.step
        env.step = lambda act: _step(act[:, inv].contiguous())
        a.no_fused = True
    ppo = PPO(env, model, hip_inference=hip and not a.no_hip, fused_rollout=not a.no_fused, **({"kl_threshold": 0.0} if a.fixed_lr else {}), freeze_obs_scaler=a.no_obs_scaler, max_lr=a.max_lr, min_log_std=a.min_log_std)
    hist = ppo.train(a.timesteps, log_every=a.log_every, log=(lambda r: print(json.dumps(r), flush=True)) if rank == 0 else (lambda r: None))
    if rank == 0 and a.out:
        json.dump({"task": a.task, "num_envs": a.num_envs, "world": world, "policy": a.policy, "history": hist}, open(a.out, "w"), indent=1)
    env.close()


if __name__ == "__main__":
    main()
