"""Diagnostic: period of back-to-back policy forwards (a captured graph of 100 launches) against the batch size - where the launch ramp of the
16-sample workgroups starts to show."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from locomanipulationrl_amd.policies.mlp_model import SharedMLP, pack_mlp_params, mlp_forward_hip
from locomanipulationrl_amd.policies.graph_model import GraphPolicy, gnn_forward_hip
torch.manual_seed(0); m = SharedMLP().cuda(); packed = pack_mlp_params(m).cuda(); pol = GraphPolicy().cuda(); pol.refresh(torch.device("cuda"))
res = {}
for name, fwd, par in (("gnn", gnn_forward_hip, pol._packed), ("mlp", mlp_forward_hip, packed)):
    row = {}
    for B in (16, 256, 1024, 2048, 4096, 8192, 16384, 32768):
        obs = torch.randn(B, 64, device="cuda")
        for _ in range(20): fwd(obs, par)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(100): fwd(obs, par)
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); g.replay(); b.record(); torch.cuda.synchronize()
        row[B] = round(a.elapsed_time(b) / 200 * 1e3, 2)
    res[name] = row
print(json.dumps({"so": os.path.basename(os.environ.get("LM_ENGINE_SO", "product")), "us_per_forward": res}))
