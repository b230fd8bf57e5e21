"""Where wavefront 0 of the persistent rollout kernel spends a step: policy tile / physics phases / closing barrier (diagnostic -DLM_STAMPS build).
    python tools/stamp_profile.py --build   (here; builds tools/diag/liblm_engine_stamps.so)      python tools/stamp_profile_rollout.py [mlp|gnn]   (GPU box)"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["LM_ENGINE_SO"] = os.path.join(ROOT, "tools", "diag", "liblm_engine_stamps.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
from locomanipulationrl_amd.lib import Engine, Rollout, POLICY_MLP, POLICY_GNN, load_library
from locomanipulationrl_amd.engine_config import loco_params
from locomanipulationrl_amd.model.robot_model import load_model
lib = load_library()
policy = sys.argv[1] if len(sys.argv) > 1 else "mlp"
N, T = 4096, 48
eng = Engine(load_model("quadruped_robot_v2"), [loco_params()], N, seed=1)
if policy == "mlp":
    from locomanipulationrl_amd.policies.mlp_model import SharedMLP, pack_mlp_params
    model = SharedMLP(num_observations=eng.num_obs).cuda(); packed = pack_mlp_params(model, None, None).cuda(); kind = POLICY_MLP
else:
    from locomanipulationrl_amd.policies.graph_model import GraphPolicy, pack_gnn_params
    model = GraphPolicy().cuda(); packed = pack_gnn_params(model.net, model.mean_layer, model.value_layer).cuda(); kind = POLICY_GNN
log_std = torch.full((12,), -0.5, device="cuda")
o0 = torch.empty(N, eng.num_obs, device="cuda"); eng.step(torch.zeros(N, 12, device="cuda"), None, o0)
ro = Rollout(eng, kind, packed, log_std, T, noise_seed=3); ro.obs[0] = o0
for _ in range(6):
    ro.run(use_graph="persistent"); ro.obs[0].copy_(ro.obs[T])
torch.cuda.synchronize()
buf = np.zeros(1024 * 64, dtype=np.uint64)
assert lib.lm_debug_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
b = buf.reshape(1024, 64)[: N // 16].astype(np.float64) / T
med = np.median(b, axis=0)
names = {13: "policy tile (wavefront 0)", 14: "closing barrier (+ tail of the outputs)"}
phys = float(med[:11].sum())
print(json.dumps({"policy": policy, "envs": N, "cycles_per_step": {"policy tile (wavefront 0)": round(float(med[13])), "physics step (all its phases)": round(phys),
                  "closing barrier": round(float(med[14]))}, "total": round(float(med[13] + phys + med[14])),
                  "policy_wavefronts_cycles_per_step": {f"P{w}": [round(float(med[16 * (w + 1) + k])) for k in range(10)] for w in range(3)},
                  "policy_buckets": ["wait physics + obs tile", "barrier", "layer 1", "barrier", "layer 2", "barrier", "layer 3", "barrier", "head + sampling", "barrier"]}))
