"""Time the matrix-core GNN policy forward (BASELINE config 5: 8192 samples) against the eager torch modules."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from locomanipulationrl_amd.policies.graph_model import GraphPolicy, gnn_forward_hip

def main(B=8192, iters=200):
    pol = GraphPolicy().cuda(); pol.refresh(torch.device("cuda"))
    obs = torch.randn(B, 64, device="cuda")
    for _ in range(10): gnn_forward_hip(obs, pol._packed)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); gnn_forward_hip(obs, pol._packed); b.record()
    torch.cuda.synchronize()
    t_hip = sum(a.elapsed_time(b) for a, b in evs) / iters
    with torch.no_grad():
        for _ in range(5): pol(obs)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): pol(obs)
        torch.cuda.synchronize(); t_torch = (time.perf_counter() - t0) / 50 * 1e3
    flop = 2.0 * B * (16 * 32 + 12 * 4 * 32 + 3 * (13 * 32 * 64 + 24 * 32 * 32) + 13 * 32)
    print(json.dumps({"batch": B, "hip_ms": t_hip, "torch_eager_ms": t_torch, "speedup": t_torch / t_hip,
                      "mfma": {"achieved_tflops": flop / (t_hip * 1e-3) / 1e12, "peak_tflops_fp32_mfma": 157.3,
                               "frac": flop / (t_hip * 1e-3) / 1e12 / 157.3}}))

if __name__ == "__main__":
    main()
