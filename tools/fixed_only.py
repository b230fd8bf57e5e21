"""k_step with zero sub-steps only (reset scatter + read-back FK + task layer + outputs + reductions): for a rocprofv3 kernel-duration capture."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locomanipulationrl_amd.engine_config import loco_params
from locomanipulationrl_amd.lib import Engine
from locomanipulationrl_amd.model.robot_model import load_model
N = 4096
eng = Engine(load_model("quadruped_robot_v2"), [loco_params()], N, seed=1)
a = torch.zeros(N, 12, device="cuda")
o = (torch.empty(N, 64, device="cuda"), torch.empty(N, 93, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, dtype=torch.int64, device="cuda"), torch.empty(13, device="cuda"))
eng.step(a, None, *o); eng.cnt[3].zero_()
for _ in range(1000): eng.post_physics(a, *o)
torch.cuda.synchronize()
