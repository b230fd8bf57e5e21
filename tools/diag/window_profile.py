#!/usr/bin/env python3
"""Where do the extra microseconds of a 20-step window go?  Events between the steps of a window that starts on an idle, synchronised GPU."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from locomanipulationrl_amd.engine_config import loco_params
from locomanipulationrl_amd.lib import Engine
from locomanipulationrl_amd.model.robot_model import load_model
dev = torch.device("cuda:0"); N = 4096
eng = Engine(load_model("quadruped_robot_v2"), [loco_params()], N, seed=42, device=str(dev))
pool = [torch.rand(N, 12, device=dev) * 2 - 1 for _ in range(64)]
o, s, r, d, x = torch.empty(N, 64, device=dev), torch.empty(N, 93, device=dev), torch.empty(N, device=dev), torch.empty(N, dtype=torch.int64, device=dev), torch.empty(13, device=dev)
for i in range(200): eng.step(pool[i % 64], None, o, s, r, d, x)
K = 20; rows = []
for rep in range(8):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    torch.cuda.synchronize(); time.sleep(0.002 * (rep % 2))          # odd repetitions: 2 ms of idle before the window
    t0 = time.perf_counter(); ev[0].record()
    for i in range(K): eng.step(pool[i % 64], None, o, s, r, d, x); ev[i + 1].record()
    t_enq = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter()
    per = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(K)]
    rows.append({"idle_ms_before": 2 * (rep % 2), "wall_us": (t1 - t0) * 1e6, "enqueue_us": (t_enq - t0) * 1e6, "events_total_us": ev[0].elapsed_time(ev[K]) * 1e3, "per_step_us": [round(p, 1) for p in per]})
    print(json.dumps(rows[-1]))
