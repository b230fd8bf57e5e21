#!/usr/bin/env python3
"""Random-policy driver: the call sequence of the reference's scripts/random_policy.py:41-63 on the MI355X engine.

    python -m locomanipulationrl_amd.scripts.random_policy task=QuadrupedPoseControl num_envs=4096 steps=1000

Arguments are `key=value` pairs as with the reference's Hydra command line (task, num_envs, steps, seed, staged).  With
`staged=True` the three phases are driven one by one exactly as the reference does (pre_physics_step -> world.step ->
post_physics_step); the default uses env.step(), which is the same work in one kernel launch."""
import sys
import time

import torch

from ..envs.vec_env_rlgames import VecEnvRLGames
from ..utils.config import load_config
from ..utils.task_util import initialize_task


def main(argv=None):
    kv = dict(a.split("=", 1) for a in (sys.argv[1:] if argv is None else argv))
    steps = int(kv.pop("steps", 1000)); staged = kv.pop("staged", "False").lower() in ("1", "true", "yes")
    cfg = load_config(kv.pop("task", "QuadrupedPoseControl"), num_envs=int(kv.pop("num_envs", 4096)), seed=int(kv.pop("seed", 42)))
    env = VecEnvRLGames(headless=True, sim_device=cfg.get("device_id", 0))
    task = initialize_task(cfg, env)
    env.reset()
    gen = torch.Generator(device=task.rl_device).manual_seed(cfg["seed"])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        actions = torch.rand((env.num_envs, task.num_actions), device=task.rl_device, generator=gen) * 2 - 1        # action_space.sample()
        if staged:
            env._task.pre_physics_step(actions)
            env._world.step(render=False)
            env.sim_frame_count += 1
            env._task.post_physics_step()
        else:
            env.step(actions)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{cfg['task_name']}: {env.num_envs} envs x {steps} steps in {dt:.3f} s = {env.num_envs * steps / dt / 1e6:.2f} M env-steps/s; "
          f"extras {{k: round(float(v), 4) for k, v in task.extras.items()}}".replace("{k: round(float(v), 4) for k, v in task.extras.items()}",
                                                                                 str({k: round(float(v), 4) for k, v in task.extras.items()})))
    env.close()


if __name__ == "__main__":
    main()
