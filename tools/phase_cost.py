"""Where k_step's time goes (4096 envs, loco): the runtime parameters switch phases off, so the differences are phase costs.
    substeps 4 -> 1 : per-sub-step cost and the fixed part (load, reset, task layer, outputs)
    pgs_iters 0 / 8 / 16 / 32 : the contact solver sweeps (16 is the default on the ground);  tau_max 180.7 -> 1.5 : round 1's torque clamp, under which most sub-steps run the second active-set pass"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from locomanipulationrl_amd.engine_config import loco_params, loco_cc_params
from locomanipulationrl_amd.lib import Engine
from locomanipulationrl_amd.model.robot_model import load_model


def run(N, steps=400, warmup=50, family="velocity", **kw):
    ep = loco_cc_params(**kw) if family == "cc" else loco_params(**kw)
    eng = Engine(load_model("quadruped_robot_v2"), [ep], N, seed=1)
    g = torch.Generator(device="cuda").manual_seed(0)
    pool = [torch.rand(N, 12, device="cuda", generator=g) * 2 - 1 for _ in range(16)]
    o = (torch.empty(N, ep.num_obs, device="cuda"), torch.empty(N, 93, device="cuda"), torch.empty(N, device="cuda"),
         torch.empty(N, dtype=torch.int64, device="cuda"), torch.empty(13, device="cuda"))
    for t in range(warmup): eng.step(pool[t % 16], None, *o)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(steps): eng.step(pool[t % 16], None, *o)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    eng.close()
    return dt * 1e6


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    cases = {"default": {}, "pgs0": dict(pgs_iters=0), "pgs8": dict(pgs_iters=8), "pgs16": dict(pgs_iters=16), "pgs32": dict(pgs_iters=32),
             "torque_clamp_1.5": dict(tau_max=1.5), "torque_clamp_1.5_pgs0": dict(tau_max=1.5, pgs_iters=0),
             "sub1": dict(substeps=1), "sub1_pgs0": dict(substeps=1, pgs_iters=0), "sub8": dict(substeps=8),
             # the custom-controller family (PD actuator, 5 sub-steps at dt 0.005, obs 88): the limit lifted = no second pass at all
             "cc": dict(family="cc"), "cc_second_pass": dict(family="cc", pd_second_pass=1), "cc_nolimit": dict(family="cc", tau_max=1.0e3), "cc_pgs0": dict(family="cc", pgs_iters=0),
             "cc_sub1": dict(family="cc", substeps=1, acc_substeps=1), "cc_sub4": dict(family="cc", substeps=4)}
    if len(sys.argv) > 2:
        cases = {k: cases[k] for k in sys.argv[2].split(",")}
    res = {k: run(N, **v) for k, v in cases.items()}
    print(json.dumps({"envs": N, "library": os.environ.get("LM_ENGINE_SO", "product"), "us_per_step": res}))
