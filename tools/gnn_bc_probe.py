#!/usr/bin/env python3
"""Isolate the GNN learner from PPO and the engine.  NOT an RL training run of the GNN: two supervised probes and one
gradient census, a minute of GPU time.

  1. behaviour cloning: a PPO-trained MLP teacher (horizontal co-training task, the recipe of tools/train_ppo.py) labels the observations of its own
     rollouts with its mean actions; GraphPolicy (the reference's GNN: 13 nodes, ONE 33-parameter action head shared by the 12 joint nodes) and a
     fresh MLP of the reference's shape are fitted to those labels by plain Adam regression.  FVU = held-out mean squared error / variance of the
     teacher's actions: if the GNN cannot bring it down, the gap of DESIGN.md 6.1 is architectural (the graph cannot express the controller the
     MLP finds); if it can, the gap is in the optimisation (PPO recipe x this model), not in what the model can represent.
  2. gradient census: the first PPO update of each model on identical fresh environments - L2 norm of the gradient per parameter group before the
     grad_norm_clip = 1.0 of the recipe and the clip's scale factor.

    python tools/gnn_bc_probe.py [--out profiles/r04_gnn_bc_probe.json]          (GPU box)
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import locomanipulationrl_amd as lm
from locomanipulationrl_amd.policies.graph_model import GraphPolicy
from locomanipulationrl_amd.policies.mlp_model import SharedMLP
from locomanipulationrl_amd.train.ppo import PPO


def groups_of(model):
    out = {}
    for name, p in model.named_parameters():
        key = name.rsplit(".", 1)[0] if name != "log_std_parameter" else name
        out.setdefault(key, []).append(p)
    return out


def grad_census(task, policy, dev, N):
    env = lm.make_env(task, num_envs=N, seed=42)
    torch.manual_seed(42)
    model = (GraphPolicy() if policy == "gnn" else SharedMLP()).to(dev)
    ppo = PPO(env, model); ppo.epochs = 1
    rec = {}
    orig = torch.nn.utils.clip_grad_norm_

    def spy(params, max_norm, *a, **k):
        g = groups_of(model)
        rec["per_group"] = {k_: float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in ps if p.grad is not None))) for k_, ps in g.items()}
        total = orig(params, max_norm, *a, **k)
        rec["total_before_clip"] = float(total); rec["clip_scale"] = min(1.0, float(max_norm) / (float(total) + 1e-6))
        return total
    torch.nn.utils.clip_grad_norm_ = spy
    try:
        obs = env.reset()["obs"]
        obs, last_value, _ = ppo.collect(obs); ppo.update(last_value)
    finally:
        torch.nn.utils.clip_grad_norm_ = orig
    env.close()
    rec["parameters"] = sum(p.numel() for p in model.parameters())
    return rec


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--out", default=""); ap.add_argument("--task", default="JointLocomanipulation"); ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--timesteps", type=int, default=14400); ap.add_argument("--fit-iters", type=int, default=4000)
    a = ap.parse_args(); dev = "cuda:0"; N = a.num_envs
    doc = {"source": "tools/gnn_bc_probe.py", "task": a.task, "envs": N}
    # ---- teacher
    env = lm.make_env(a.task, num_envs=N, seed=42); torch.manual_seed(42)
    teacher = SharedMLP().to(dev); ppo = PPO(env, teacher)
    hist = ppo.train(a.timesteps, log_every=1000, log=lambda r: None)
    doc["teacher"] = {"policy": "mlp", "ppo_timesteps": a.timesteps, "success_rate": hist[-1]["success_rate"], "action_std": hist[-1]["std"]}
    ppo.rollout = None
    X, Y = [], []
    obs = env.reset()["obs"]
    for t in range(144):
        mean, log_std, _ = ppo._policy(obs)
        X.append(ppo.obs_scaler(obs).clone()); Y.append(mean.clone())
        obs = env.step(mean + log_std.exp() * torch.randn_like(mean))[0]["obs"]
    env.close()
    X, Y = torch.cat(X), torch.cat(Y); n = X.shape[0]; perm = torch.randperm(n, device=dev); cut = n * 9 // 10
    tr, te = perm[:cut], perm[cut:]; var = float(Y[tr].var(dim=0).mean())
    doc["dataset"] = {"samples": n, "teacher_action_variance": var}
    # ---- students
    doc["behaviour_cloning"] = {}
    for name, mk in (("gnn", GraphPolicy), ("mlp", SharedMLP)):
        for lr in (1e-3, 3e-3):
            torch.manual_seed(0); st = mk().to(dev); opt = torch.optim.Adam(st.parameters(), lr=lr); curve = []
            for it in range(a.fit_iters):
                idx = tr[torch.randint(0, cut, (8192,), device=dev)]
                loss = ((st(X[idx])[0] - Y[idx]) ** 2).mean()
                opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
                if it % 500 == 0 or it == a.fit_iters - 1:
                    with torch.no_grad():
                        curve.append((it, float(((st(X[te])[0] - Y[te]) ** 2).mean()) / var))
            with torch.no_grad():
                per_joint = (((st(X[te])[0] - Y[te]) ** 2).mean(0) / Y[tr].var(dim=0)).tolist()
            row = {"parameters": sum(p.numel() for p in st.parameters()), "lr": lr, "iterations": a.fit_iters, "held_out_fvu": curve[-1][1],
                   "fvu_curve": [(i, round(v, 4)) for i, v in curve], "fvu_per_joint": [round(v, 3) for v in per_joint]}
            doc["behaviour_cloning"][f"{name}_lr{lr:g}"] = row
            print(json.dumps({f"{name}_lr{lr:g}": {"held_out_fvu": row["held_out_fvu"], "parameters": row["parameters"]}}), flush=True)
    # ---- gradient census of the first PPO update
    doc["first_update_gradients"] = {p: grad_census(a.task, p, dev, N) for p in ("mlp", "gnn")}
    print(json.dumps(doc["first_update_gradients"]), flush=True)
    if a.out:
        json.dump(doc, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
