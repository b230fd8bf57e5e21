#!/usr/bin/env python3
"""Probe: BASELINE config 5 (vertical co-training, 8192 envs, GNN in the loop) as S env shards on ONE GPU -- S engines of 8192 / S envs, each with its
own captured rollout graph on its own HIP stream -- against the single 8192-env engine.  The shards are what each rank of DESIGN.md 7 runs; here they
share a GPU, so one shard's policy forward can overlap another shard's physics step.
    python tools/diag/sharded_rollout.py [--policy gnn] [--num-envs 8192] [--task JointLocomanipulationVertical]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from locomanipulationrl_amd.lib import Engine, Rollout, POLICY_GNN, POLICY_MLP
from locomanipulationrl_amd.model.robot_model import load_model
from locomanipulationrl_amd.utils.config import SimConfig, load_config
from locomanipulationrl_amd.utils.task_util import task_map


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--num-envs", type=int, default=8192); ap.add_argument("--policy", default="gnn")
    ap.add_argument("--task", default="JointLocomanipulationVertical"); ap.add_argument("--T", type=int, default=48); ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--mode", default="graph")
    a = ap.parse_args(); T = a.T; res = {"task": a.task, "num_envs": a.num_envs, "policy": a.policy, "T": T, "mode": a.mode}
    torch.manual_seed(42)
    if a.policy == "mlp":
        from locomanipulationrl_amd.policies.mlp_model import SharedMLP, pack_mlp_params
        kind = POLICY_MLP; mk = lambda nobs: pack_mlp_params(SharedMLP(num_observations=nobs).cuda(), None, None).cuda()
    else:
        from locomanipulationrl_amd.policies.graph_model import GraphPolicy, pack_gnn_params
        kind = POLICY_GNN
        def mk(nobs):
            m = GraphPolicy().cuda(); return pack_gnn_params(m.net, m.mean_layer, m.value_layer).cuda()
    log_std = torch.full((12,), -0.5, device="cuda")
    for S in (1, 2, 4):
        n = a.num_envs // S; shards = []
        for s in range(S):
            task = task_map()[a.task](name=a.task, sim_config=SimConfig(load_config(a.task, num_envs=n)), env=None)
            eng = Engine(load_model(task.model_asset), task.engine_params(), n, split_env=task.split_env(), seed=42 + s)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                o0 = torch.empty(n, eng.num_obs, device="cuda"); eng.step(torch.zeros(n, 12, device="cuda"), None, o0)
                ro = Rollout(eng, kind, mk(eng.num_obs), log_std, T, noise_seed=3 + s); ro.obs[0] = o0
            shards.append((eng, ro, st))
        torch.cuda.synchronize()
        mode = True if a.mode == "graph" else a.mode

        def sweep(k):
            for _ in range(k):
                for eng, ro, st in shards:
                    with torch.cuda.stream(st):
                        ro.run(use_graph=mode); ro.obs[0].copy_(ro.obs[T])
        sweep(3); torch.cuda.synchronize(); t0 = time.perf_counter(); sweep(a.reps); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.reps
        res[f"shards_{S}"] = {"envs_per_shard": n, "us_per_step": dt / T * 1e6, "env_steps_per_s": a.num_envs * T / dt}
        print(json.dumps({f"shards_{S}": res[f"shards_{S}"]}), flush=True)
        for eng, ro, st in shards: ro.close(); eng.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
