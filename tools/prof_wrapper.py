import cProfile, pstats, sys, io, torch
sys.path.insert(0, '/root/repo')
import locomanipulationrl_amd as lm
env = lm.make_env("QuadrupedPoseControl", num_envs=4096)
env.reset()
a = torch.rand(4096, 12, device="cuda") * 2 - 1
for _ in range(100): env.step(a)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): env.step(a)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:5000])
