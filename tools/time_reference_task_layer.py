"""SURVEY 8(d), last row: the reference's OWN task-layer torch code (a4-a6, a9-a11: pre_physics_step with its reset scatter, get_observations,
calculate_metrics, is_done) timed on the CPU at 4096 envs - the part of the reference's step that is not PhysX, including the scipy round
trips of utils/math.py.  Runs in the build container only (it imports /root/reference through the placeholder harness of tools/gen_golden.py);
the number never runs on the GPU box and is no baseline of the engine - it bounds what the reference's Python adds to every step on a host.
    python -B tools/time_reference_task_layer.py > profiles/r02_reference_task_layer_cpu.json"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import gen_golden as G

assert os.path.isdir(G.REF_RL), "reference tree not present"
G._install_placeholders()
sys.path[:0] = [G.REF_RL, os.path.dirname(G.REF_RL)]
N, T = 4096, 40
acc = {}


def timed(obj, name):
    f = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc.setdefault(name, []).append(time.perf_counter() - t0); return r
    setattr(obj, name, w)


out = {"envs": N, "calls": T, "torch_threads": torch.get_num_threads(), "host_cpus": os.cpu_count(), "kinds": {}}
for kind in ("loco", "mani"):
    acc.clear()
    make = G.make_task
    def make_timed(k, n, make=make):
        t = make(k, n)
        for m in ("pre_physics_step", "get_observations", "calculate_metrics", "is_done"): timed(t, m)
        return t
    G.make_task = make_timed
    try: d = G.gen_task(kind, N=N, T=T)
    finally: G.make_task = make
    ms = {k: 1e3 * sorted(v)[len(v) // 2] for k, v in acc.items()}
    tot = sum(ms.values())
    out["kinds"][kind] = {"median_ms_per_call": {k: round(v, 3) for k, v in ms.items()}, "task_layer_ms_per_step": round(tot, 3),
                          "env_steps_per_s_task_layer_alone": round(N / (tot * 1e-3)), "resets_per_step_mean": float(d["reset_buf"].sum(1).mean())}
# the scipy round trips on their own (utils/math.py:33-193)
from utils.math import rand_quaternions, transform_vectors, rotate_orientations
g = torch.Generator().manual_seed(0)
q = G._rand_unit_quat(g, N); v = torch.randn(N, 3, generator=g); V = torch.randn(N, 4, 3, generator=g)
def med(f, n=20):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return 1e3 * sorted(ts)[n // 2]
out["utils_math_ms_at_4096"] = {"transform_vectors (4 points per env)": round(med(lambda: transform_vectors(q, v, V, "cpu")), 3),
                                "rotate_orientations": round(med(lambda: rotate_orientations(q, q, "cpu")), 3),
                                "rand_quaternions": round(med(lambda: rand_quaternions(N, -0.4, 0.4, -0.4, 0.4, -1.57, 1.57, "cpu")), 3)}
out["note"] = ("the reference's Python task layer alone, without PhysX, costs this much host time per step at 4096 envs; the engine runs the same "
               "arithmetic inside k_step (the whole step, physics included, takes 0.034 ms on the GPU)")
print(json.dumps(out, indent=1))
