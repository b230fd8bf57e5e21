"""Soak run: N envs x STEPS random-action steps; every CHECK steps the state must be finite, quaternions unit, counters in range.
Reports how often the in-kernel blow-up guard had to contain an env (failure detection).
    python tools/soak.py [steps=200000] [envs=4096] [task=QuadrupedPoseControl]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from locomanipulationrl_amd.lib import Engine
from locomanipulationrl_amd.model.robot_model import load_model
from locomanipulationrl_amd.utils.config import SimConfig, load_config
from locomanipulationrl_amd.utils.task_util import task_map

kv = dict(a.split("=", 1) for a in sys.argv[1:])
steps, N, name = int(kv.get("steps", 200000)), int(kv.get("envs", 4096)), kv.get("task", "QuadrupedPoseControl")
task = task_map()[name](name=name, sim_config=SimConfig(load_config(name, num_envs=N)), env=None)
eng = Engine(load_model(task.model_asset), task.engine_params(), N, split_env=task.split_env(), seed=5)
g = torch.Generator(device="cuda").manual_seed(1)
pool = [torch.rand(N, 12, device="cuda", generator=g) * 2 - 1 for _ in range(97)]
fb = 0 if task.engine_params()[0].mode == 0 else 37
t0 = time.time(); resets = 0
rs = torch.empty(N, dtype=torch.int64, device="cuda"); acc = torch.zeros((), dtype=torch.int64, device="cuda")
for t in range(steps):
    eng.step(pool[t % 97], None, None, None, None, rs); acc += rs.sum()
    if t % 20000 == 19999 or t == steps - 1:
        s = eng.state
        assert torch.isfinite(s).all(), t
        assert (s[fb + 3:fb + 7].norm(dim=0) - 1).abs().max() < 1e-4 and (s[86:90].norm(dim=0) - 1).abs().max() < 1e-4, t
        assert int(eng.cnt[4].max()) < task.engine_params()[0].max_episode and int(eng.cnt[4].min()) >= 0, t
        print(f"step {t + 1}: ok, {time.time() - t0:.1f} s, resets so far {int(acc)}, blow-ups contained {eng.blowups}", flush=True)
print(json.dumps({"task": name, "envs": N, "steps": steps, "env_steps": N * steps, "resets": int(acc), "blowups_contained": eng.blowups,
                  "blowups_per_million_env_steps": eng.blowups / (N * steps / 1e6), "wall_s": time.time() - t0}))
