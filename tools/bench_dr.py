"""Cost of domain randomisation inside the step launch: k_step vs k_step_dr (YAML block of QuadrupedPoseControl.yaml) at 4096 envs."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import locomanipulationrl_amd as lm

res = {}
for name, ov in (("plain", None), ("randomised", {"task": {"domain_randomization": {"randomize": True}}})):
    env = lm.make_env("QuadrupedPoseControl", num_envs=4096, overrides=ov)
    e = env._task.engine; N = 4096
    g = torch.Generator(device="cuda").manual_seed(0)
    pool = [torch.rand(N, 12, device="cuda", generator=g) * 2 - 1 for _ in range(16)]
    o = (torch.empty(N, 64, device="cuda"), torch.empty(N, 93, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, dtype=torch.int64, device="cuda"), torch.empty(13, device="cuda"))
    for t in range(50): e.step(pool[t % 16], None, *o)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(500): e.step(pool[t % 16], None, *o)
    torch.cuda.synchronize(); res[name] = (time.perf_counter() - t0) / 500 * 1e6
    env.close()
print(json.dumps({"us_per_step": res, "overhead": res["randomised"] / res["plain"] - 1}))
