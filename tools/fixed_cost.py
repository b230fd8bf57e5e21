import sys, time, torch
sys.path.insert(0, '/root/repo')
from locomanipulationrl_amd.engine_config import loco_params
from locomanipulationrl_amd.lib import Engine
from locomanipulationrl_amd.model.robot_model import load_model
N = 4096
eng = Engine(load_model("quadruped_robot_v2"), [loco_params()], N, seed=1)
a = torch.zeros(N, 12, device="cuda")
o = (torch.empty(N, 64, device="cuda"), torch.empty(N, 93, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, dtype=torch.int64, device="cuda"), torch.empty(13, device="cuda"))
def timeit(f, n=2000):
    for _ in range(100): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
eng.step(a, None, *o)
print("reset_all (trivial kernel)  %.2f us" % timeit(lambda: eng.reset_all()))
eng.cnt[3].zero_()
print("post_physics (no sub-steps) %.2f us" % timeit(lambda: eng.post_physics(a, *o)))
print("post_physics no outputs     %.2f us" % timeit(lambda: eng.post_physics(a)))
print("step                        %.2f us" % timeit(lambda: eng.step(a, None, *o)))
print("fk kernel                   %.2f us" % timeit(lambda: eng.forward_kinematics()))
# how much of the step period is the volume of output stores (the end-of-kernel write-back of the XCDs' L2s)?
print("step, no returned copies    %.2f us" % timeit(lambda: eng.step(a)))
print("step, returned obs only     %.2f us" % timeit(lambda: eng.step(a, None, o[0])))
