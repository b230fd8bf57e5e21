#!/usr/bin/env python3
"""Time the reference's own task-layer torch code on CPU at N = 4096.

Build-container only (needs /root/reference; nothing here travels to the GPU box or is imported by the product).
Uses the placeholder harness of tools/gen_golden.py: the reference's pre_physics_step / get_observations /
calculate_metrics / is_done run unmodified on CPU tensors fed with a synthetic read-back state.  The number is
the cost of the part of the reference's step() that is NOT PhysX -- what the fused task layer of k_step replaces --
and is recorded in DESIGN.md section 6; PhysX itself cannot be timed here.

Run:  python -B tools/time_reference_task_layer.py [--envs 4096] [--iters 60] [--threads 8]
"""
import argparse
import os
import sys
import time

import torch

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--threads", type=int, default=0, help="torch intra-op threads (0 = torch default)")
    ap.add_argument("--kind", default="loco", choices=["loco", "mani", "loco_cc", "mani_cc", "loco_pc", "mani_pc"])
    a = ap.parse_args()
    assert os.path.isdir(G.REF_RL), "reference tree not present"
    if a.threads:
        torch.set_num_threads(a.threads)
    G._install_placeholders()
    sys.path[:0] = [G.REF_RL, os.path.dirname(G.REF_RL)]
    N = a.envs
    g = torch.Generator().manual_seed(1)
    torch.manual_seed(1)
    t = G.make_task(a.kind, N)
    loco = a.kind.startswith("loco")
    robot = t.robot_locomotion if loco else t.robot_manipulation
    q0 = t.default_joint_positions_loco[:, :12] if loco else t.default_joint_positions_mani[:, :12]
    acts = [(torch.rand(N, 12, generator=g) * 2 - 1) for _ in range(8)]

    def readback():
        robot.joint_positions = q0 + 0.25 * torch.randn(N, 12, generator=g)
        robot.joint_velocities = 2.0 * torch.randn(N, 12, generator=g)
        robot.joint_accelerations = 20.0 * torch.randn(N, 12, generator=g)
        robot.tip_positions = 0.15 * torch.randn(N, 4, 3, generator=g)
        robot.knee_positions = torch.cat((0.15 * torch.randn(N, 8, 2, generator=g), 0.10 + 0.03 * torch.randn(N, 8, 1, generator=g)), -1)
        pos = torch.cat((0.05 * torch.randn(N, 2, generator=g), 0.13 + 0.02 * torch.randn(N, 1, generator=g)), -1)
        quat = G._rand_unit_quat(g, N, small=0.25)
        lin, ang = 0.3 * torch.randn(N, 3, generator=g), torch.randn(N, 3, generator=g)
        if loco:
            robot.base_positions, robot.base_quaternions = pos, quat
            robot.base_linear_velocities, robot.base_angular_velocities = lin, ang
        else:
            flip = torch.tensor([0.0, 1.0, 0.0, 0.0]).repeat(N, 1)
            from omni.isaac.core.utils.torch.rotations import quat_mul
            t.obj.pos, t.obj.quat, t.obj.lin, t.obj.ang = pos, quat_mul(quat, flip), lin, ang

    def one(i):
        t.pre_physics_step(acts[i % 8].clone())
        t.progress_buf[:] += 1
        t.get_observations(); t.calculate_metrics(); t.is_done()

    for i in range(5):
        readback(); one(i)
    tot = 0.0
    for i in range(a.iters):
        readback()                               # synthetic state generation is outside the timed region
        t0 = time.perf_counter(); one(i); tot += time.perf_counter() - t0
    ms = tot / a.iters * 1e3
    print({"kind": a.kind, "envs": N, "iters": a.iters, "torch_threads": torch.get_num_threads(),
           "task_layer_ms_per_step": round(ms, 3), "env_steps_per_s_task_layer_only": round(N / ms * 1e3)})


if __name__ == "__main__":
    main()
