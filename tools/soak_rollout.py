"""Soak of the persistent rollout kernel: ROLLOUTS x 48 steps with a random MLP policy in the loop (large exploration noise), state and
buffers checked every CHECK rollouts; a second engine replays a sample of the rollouts through the hipGraph path and must agree bit for bit.
    python tools/soak_rollout.py [rollouts=4000] [envs=4096] [task=JointLocomanipulation] [policy=mlp|gnn]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from locomanipulationrl_amd.lib import Engine, Rollout, POLICY_GNN, POLICY_MLP
from locomanipulationrl_amd.model.robot_model import load_model
from locomanipulationrl_amd.policies.mlp_model import SharedMLP, pack_mlp_params
from locomanipulationrl_amd.utils.config import SimConfig, load_config
from locomanipulationrl_amd.utils.task_util import task_map

kv = dict(a.split("=", 1) for a in sys.argv[1:])
R, N, name, T = int(kv.get("rollouts", 4000)), int(kv.get("envs", 4096)), kv.get("task", "JointLocomanipulation"), 48
task = task_map()[name](name=name, sim_config=SimConfig(load_config(name, num_envs=N)), env=None)
nobs = task.engine_params()[0].num_obs
torch.manual_seed(0)
if kv.get("policy", "mlp") == "gnn":
    from locomanipulationrl_amd.policies.graph_model import GraphPolicy, pack_gnn_params
    gm = GraphPolicy().cuda(); packed = pack_gnn_params(gm.net, gm.mean_layer, gm.value_layer).cuda(); POLICY = POLICY_GNN
else:
    model = SharedMLP(num_observations=nobs).cuda(); packed = pack_mlp_params(model, None, None).cuda(); POLICY = POLICY_MLP
log_std = torch.full((12,), 0.0, device="cuda")
engs, ros = [], []
for _ in range(2):
    e = Engine(load_model(task.model_asset), task.engine_params(), N, split_env=task.split_env(), seed=5)
    o0 = torch.empty(N, nobs, device="cuda"); e.step(torch.zeros(N, 12, device="cuda"), None, o0)
    r = Rollout(e, POLICY, packed, log_std, T, noise_seed=9); r.obs[0] = o0
    engs.append(e); ros.append(r)
t0 = time.time(); resets = 0; compared = 0
for i in range(R):
    ros[0].run("persistent"); ros[0].obs[0].copy_(ros[0].obs[T])
    if i < 50:        # the shadow engine follows the first rollouts through the graph path
        ros[1].run("graph"); ros[1].obs[0].copy_(ros[1].obs[T]); compared += 1
        for nm in ("obs", "actions", "logp", "values", "rewards", "dones", "extras"):
            assert torch.equal(getattr(ros[0], nm), getattr(ros[1], nm)), (i, nm)
        assert torch.equal(engs[0].state, engs[1].state) and torch.equal(engs[0].cnt, engs[1].cnt) and torch.equal(engs[0].stats_i64, engs[1].stats_i64), i
    resets += int(ros[0].dones.sum()) if (i % 100 == 0) else 0
    if i % 500 == 499 or i == R - 1:
        assert torch.isfinite(engs[0].state).all() and torch.isfinite(ros[0].obs).all() and torch.isfinite(ros[0].rewards).all() and torch.isfinite(ros[0].extras).all(), i
        print(f"rollout {i + 1}: ok, {time.time() - t0:.1f} s, blow-ups contained {engs[0].blowups}, success_rate {float(ros[0].extras[T - 1, 7]):.3f}", flush=True)
print(json.dumps({"task": name, "policy": kv.get("policy", "mlp"), "envs": N, "rollouts": R, "env_steps": N * T * R, "compared_with_graph": compared, "blowups_contained": engs[0].blowups,
                  "wall_s": time.time() - t0, "env_steps_per_s_incl_host_loop": N * T * R / (time.time() - t0)}))
