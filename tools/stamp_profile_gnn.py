"""Per-phase shader cycles of the GNN policy tile (k_gnn_forward, 8192 samples) from a DIAGNOSTIC build (-DLM_GNN_STAMPS).
    python tools/stamp_profile_gnn.py --build   (here)      python tools/stamp_profile_gnn.py   (GPU box)"""
import ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "diag", "liblm_engine_gnnstamps.so")
if "--build" in sys.argv:
    sys.path.insert(0, ROOT)
    from locomanipulationrl_amd.lib import hipcc_command      # the product's own flags + the stamp switch
    subprocess.check_call(hipcc_command(extra=["-DLM_GNN_STAMPS"], out=SO))
    print("built", SO); sys.exit(0)
os.environ["LM_ENGINE_SO"] = SO
sys.path.insert(0, ROOT)
import numpy as np, torch
from locomanipulationrl_amd.lib import load_library
from locomanipulationrl_amd.policies.graph_model import GraphPolicy, gnn_forward_hip
lib = load_library()
NAMES = ["weights of layer 0 issued", "input layers", "stage 1 (x3): 32 MFMAs per owned node, P/Q to LDS", "next weights issued + barrier (x3)",
         "stage 2 (x3): ELU + 16 MFMAs per incoming edge, max", "barrier (x3)", "heads + sampling", "last barrier"]
out = {}
for B in (4096, 8192):
    pol = GraphPolicy().cuda(); pol.refresh(torch.device("cuda"))
    obs = torch.randn(B, 64, device="cuda")
    for _ in range(20): gnn_forward_hip(obs, pol._packed)
    torch.cuda.synchronize()
    buf = np.zeros(512 * 64, dtype=np.uint64)
    assert lib.lm_debug_gnn_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
    b = buf.reshape(512, 4, 16)[: B // 16].astype(np.float64)
    med = np.median(b, axis=0)            # [wave][bucket]
    out[f"samples_{B}"] = {"per_wave_cycles": {NAMES[k]: [round(float(med[w, k])) for w in range(4)] for k in range(8)},
                           "wave_total": [round(float(med[w, :8].sum())) for w in range(4)]}
print(json.dumps(out, indent=1))
