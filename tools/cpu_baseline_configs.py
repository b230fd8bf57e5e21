"""SURVEY 8(d) "CPU baseline beside it": the CPU oracle (fp32 build, OpenMP over envs) on BASELINE configs 2 and 3 at all host cores and
at one core, plus a stream-copy measurement of the HBM rate this GPU actually sustains (the roofline's vendor figure is 8 TB/s).
    python tools/cpu_baseline_configs.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from locomanipulationrl_amd.engine_config import loco_params, mani_params

out = {"nproc": os.cpu_count()}
for name, prm in (("config2_loco", loco_params()), ("config3_mani", mani_params())):
    r = bench.cpu_baseline(steps=60, envs=4096, params=prm)
    out[name] = {"all_cores": r["value"], "cores": r["cores"], "one_core": r.get("value_1core")}
try:
    import torch
    if torch.cuda.is_available():
        n = 1 << 30                                  # 4 GiB of float32 -> 8 GiB moved per copy
        a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda"); a.normal_()
        for _ in range(3): b.copy_(a)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): b.copy_(a)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        out["stream_copy_GBps"] = 2 * 4 * n / dt / 1e9
except Exception as e:                                # noqa: BLE001
    out["stream_copy_error"] = str(e)
print(json.dumps(out))
