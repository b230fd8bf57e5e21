import ctypes as C, json, os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo") else os.getcwd()
os.environ["LM_ENGINE_SO"] = os.path.join(ROOT, "tools", "diag", "liblm_engine_gnnstamps.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
from locomanipulationrl_amd.lib import load_library
from locomanipulationrl_amd.policies.mlp_model import SharedMLP, pack_mlp_params, mlp_forward_hip
lib = load_library()
B = 4096
model = SharedMLP(num_observations=64).cuda(); packed = pack_mlp_params(model, None, None).cuda()
obs = torch.randn(B, 64, device="cuda")
for _ in range(20): mlp_forward_hip(obs, packed)
torch.cuda.synchronize()
buf = np.zeros(512 * 64, dtype=np.uint64)
assert lib.lm_debug_gnn_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
b = buf.reshape(512, 4, 16)[: B // 16].astype(np.float64)
med = np.median(b, axis=0)
names = ["biases + observation tile", "first chunks issued + barrier", "layer 1 compute", "barrier", "layer 2 compute", "barrier", "layer 3 compute", "barrier", "head + sampling"]
print(json.dumps({names[k]: [round(float(med[w, k])) for w in range(4)] for k in range(9)}), [round(float(med[w, :9].sum())) for w in range(4)])
