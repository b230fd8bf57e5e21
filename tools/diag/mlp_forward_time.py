"""Diagnostic: period of back-to-back lm_mlp_forward launches (a captured graph of 100) at 4096 / 8192 / 16384 samples."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from locomanipulationrl_amd.policies.mlp_model import SharedMLP, pack_mlp_params, mlp_forward_hip
torch.manual_seed(0); m = SharedMLP().cuda(); packed = pack_mlp_params(m).cuda(); out = []
for B in (4096, 8192, 16384):
    obs = torch.randn(B, 64, device="cuda")
    for _ in range(20): mlp_forward_hip(obs, packed)
    g = torch.cuda.CUDAGraph()          # 100 forwards in one graph: the period of back-to-back launches, not Python's launch overhead
    with torch.cuda.graph(g):
        for _ in range(100): mlp_forward_hip(obs, packed)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); g.replay(); b.record(); torch.cuda.synchronize()
    out.append(f"{B}: {a.elapsed_time(b) / 200 * 1e3:.2f} us")
print(os.path.basename(os.environ.get("LM_ENGINE_SO", "product")), "  ".join(out))
