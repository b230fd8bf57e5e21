"""Diagnostic: is the GNN tile deterministic?  (1) k_gnn_forward on the same input, repeated; (2) graph replay twice from identical engines;
(3) persistent kernel twice from identical engines; (4) graph vs persistent: first differing step and buffer.
    LM_ENGINE_SO=tools/diag/liblm_engine_<x>.so python tools/diag/gnn_determinism.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from locomanipulationrl_amd.lib import Engine, Rollout, POLICY_GNN
from locomanipulationrl_amd.model.robot_model import load_model
from locomanipulationrl_amd.policies.graph_model import GraphPolicy, pack_gnn_params, gnn_forward_hip
from locomanipulationrl_amd.utils.config import SimConfig, load_config
from locomanipulationrl_amd.utils.task_util import task_map
torch.manual_seed(5)
model = GraphPolicy().cuda(); packed = pack_gnn_params(model.net, model.mean_layer, model.value_layer).cuda()
for B in (8192, 8197, 512):
    obs = torch.randn(B, 64, device="cuda") * 2
    m0, v0 = gnn_forward_hip(obs, packed); bad = 0
    for i in range(100):
        m, v = gnn_forward_hip(obs, packed)
        bad += int(not (torch.equal(m, m0) and torch.equal(v, v0)))
    print(f"forward B={B}: {bad} of 100 repeats differ from the first", flush=True)
task_name, N, T = "JointLocomanipulationVertical", 512, 20
task = task_map()[task_name](name=task_name, sim_config=SimConfig(load_config(task_name, num_envs=N)), env=None)
log_std = torch.full((12,), -0.3, device="cuda")
def make():
    e = Engine(load_model(task.model_asset), task.engine_params(), N, split_env=task.split_env(), seed=9)
    o0 = torch.empty(N, 64, device="cuda"); e.step(torch.zeros(N, 12, device="cuda"), None, o0)
    r = Rollout(e, POLICY_GNN, packed, log_std, T, noise_seed=21); r.obs[0] = o0
    return e, r
def first_diff(a, b):
    for name in ("actions", "obs", "values", "logp", "rewards"):
        x, y = getattr(a, name), getattr(b, name)
        if not torch.equal(x, y):
            d = (x.float() - y.float()).abs(); t = int((d.reshape(d.shape[0], -1).max(1).values > 0).nonzero()[0])
            return f"{name} first differs at step {t} (max diff there {float(d[t].max()):.3e}, {int((d[t] > 0).sum())} entries)"
    return "identical"
for trial in range(4):
    (e1, r1), (e2, r2), (e3, r3), (e4, r4) = make(), make(), make(), make()
    r1.run("graph"); r2.run("graph"); r3.run("persistent"); r4.run("persistent"); torch.cuda.synchronize()
    print(f"trial {trial}: graph vs graph: {first_diff(r1, r2)} | persistent vs persistent: {first_diff(r3, r4)} | graph vs persistent: {first_diff(r1, r3)}", flush=True)
    for r in (r1, r2, r3, r4): r.close()
    for e in (e1, e2, e3, e4): e.close()
