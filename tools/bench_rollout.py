#!/usr/bin/env python3
"""Rollout throughput with the policy in the loop: T = 48 steps of MLP / GNN forward -> sampling -> lm_step,
(a) one captured hipGraph launch, (b) the same 4T+1 kernels enqueued one by one from C, (c) driven step by step from Python.
    python tools/bench_rollout.py [--num-envs 4096] [--policy mlp|gnn] [--reps 20]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from locomanipulationrl_amd.engine_config import loco_params
from locomanipulationrl_amd.lib import Engine, Rollout, POLICY_GNN, POLICY_MLP, sample_actions
from locomanipulationrl_amd.model.robot_model import load_model


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--num-envs", type=int, default=4096); ap.add_argument("--policy", default="mlp")
    ap.add_argument("--reps", type=int, default=20); ap.add_argument("--T", type=int, default=48)
    ap.add_argument("--task", default="QuadrupedPoseControl", help="task class whose engine parameters / robot model are used (BASELINE config 5: "
                                                                   "--task JointLocomanipulationVertical --num-envs 8192 --policy gnn)")
    a = ap.parse_args()
    N, T = a.num_envs, a.T
    from locomanipulationrl_amd.utils.config import SimConfig, load_config
    from locomanipulationrl_amd.utils.task_util import task_map
    task = task_map()[a.task](name=a.task, sim_config=SimConfig(load_config(a.task, num_envs=N)), env=None)
    eng = Engine(load_model(task.model_asset), task.engine_params(), N, split_env=task.split_env(), seed=1)
    if a.policy == "mlp":
        from locomanipulationrl_amd.policies.mlp_model import SharedMLP, pack_mlp_params, mlp_forward_hip as fwd
        model = SharedMLP(num_observations=eng.num_obs).cuda(); packed = pack_mlp_params(model, None, None).cuda(); kind = POLICY_MLP
    else:
        from locomanipulationrl_amd.policies.graph_model import GraphPolicy, pack_gnn_params, gnn_forward_hip as fwd
        model = GraphPolicy().cuda(); packed = pack_gnn_params(model.net, model.mean_layer, model.value_layer).cuda(); kind = POLICY_GNN
    log_std = torch.full((12,), -0.5, device="cuda")
    o0 = torch.empty(N, eng.num_obs, device="cuda"); eng.step(torch.zeros(N, 12, device="cuda"), None, o0)
    ro = Rollout(eng, kind, packed, log_std, T, noise_seed=3); ro.obs[0] = o0
    res = {"task": a.task, "num_envs": N, "T": T, "policy": a.policy}
    modes = (("graph", True), ("enqueue", False), ("persistent", "persistent"), ("auto", "auto"))
    for name, graph in modes:
        for _ in range(3): ro.run(use_graph=graph); ro.obs[0].copy_(ro.obs[T])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.reps): ro.run(use_graph=graph); ro.obs[0].copy_(ro.obs[T])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.reps
        res[name] = {"ms_per_rollout": dt * 1e3, "us_per_step": dt / T * 1e6, "env_steps_per_s": N * T / dt}
    obs = ro.obs[T].clone(); o = torch.empty(N, eng.num_obs, device="cuda"); r = torch.empty(N, device="cuda"); d = torch.empty(N, dtype=torch.int64, device="cuda")
    def py_rollout(obs):
        for t in range(T):
            mean, value = fwd(obs, packed); act, logp = sample_actions(eng, mean, log_std, 3); eng.step(act, None, o, None, r, d); obs = o
        return obs
    for _ in range(3): py_rollout(obs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.reps): py_rollout(obs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.reps
    res["python_loop"] = {"ms_per_rollout": dt * 1e3, "us_per_step": dt / T * 1e6, "env_steps_per_s": N * T / dt}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
