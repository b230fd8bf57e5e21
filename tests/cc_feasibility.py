#!/usr/bin/env python3
"""Bounded diagnosis of `QuadrupedPoseControlCustomController` = 0.00 (round-2 verdict, item 5): can the PD actuator of the custom-controller
family (kp 4.5, kd 0.2, torque clamped at +-1.5 N m, quadruped_pose_control_custom_controller.py:255-307) execute a turning gait at all?

The reference's PhysX recordings (tests/golden/npy_traj.npz) hold four successful locomotion episodes towards the fixed goal
(roll 0.2, pitch 0.2, yaw 0.785) - gaits that are known to work on the real simulator with the VELOCITY drive.  Their joint paths are
replayed here as position targets of the custom-controller task: same scene (class-default pose, base dropped from z 0.18), same goal, the
targets integrated from actions exactly as the task does (se <- clamp(se + 0.1 a), |a| <= 1: at most 4 rad/s), the recorded path resampled
from its 0.0332 s rows to the task's 0.025 s control steps, then held.  Outcome per file: the closest approach of rot_dist to the goal and
the worst joint tracking error - with the task's 1.5 N m clamp at the recorded speed and at a half / a third of it (the soft PD lags a gait
made for a rigid velocity drive), and with the clamp lifted as a control.

    python tests/cc_feasibility.py [out.json]        (CPU oracle; test infrastructure)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa: F401
import npy_replay as R
from locomanipulationrl_amd.engine_config import loco_cc_params
from locomanipulationrl_amd.model.robot_model import load_model
from oracle.lmo import Oracle


def to_se(q):
    """driven joint angles (12) -> the task's swing / extension coordinates (…custom_controller.py:267-276, inverted)"""
    se = np.zeros(12); se[:4] = q[:4]
    for l in range(4):
        se[4 + 2 * l] = 0.5 * (q[4 + 2 * l] + q[5 + 2 * l]); se[5 + 2 * l] = q[4 + 2 * l] - q[5 + 2 * l]
    return se


def replay(rm, rec, hold=40, stretch=1.0, **kw):
    ep = loco_cc_params(goal_lo=list(R.GOAL), goal_hi=list(R.GOAL), **kw)
    o = Oracle(rm, ep); phys, task, cnt = o.new_state(1)
    t_rec = np.arange(len(rec)) * 0.0332 * stretch; dt_c = ep.dt * ep.substeps      # stretch > 1: the same path, slower
    n = int(t_rec[-1] / dt_c) + hold
    rd, terr, done = [], [], None
    for k in range(n):
        tq = np.array([np.interp(min(k * dt_c, t_rec[-1]), t_rec, rec[:, j]) for j in range(12)])
        se = task[0, 40:52].astype(np.float64) if k else np.array(ep.init_se)
        a = np.clip((to_se(tq) - se) / ep.act_scale_se, -1, 1)
        obs, st, rew, terms = o.step(phys, task, cnt, a[None], seed=0)
        rd.append(R.rot_dist(obs[0])); terr.append(float(np.abs(phys[0, 13:25] - tq).max()))
        if cnt[0, 3]:
            done = k; break
    return dict(min_rot_dist=round(float(min(rd)), 3), first_rot_dist=round(float(rd[0]), 3), worst_joint_error=round(float(max(terr[3:] or [0])), 3),
                terminated_on_step=done, success=bool(done is not None and cnt[0, 2]))


def main():
    rm = load_model("quadruped_robot_v2"); rec = R.load()
    out = {"source": "tests/cc_feasibility.py (CPU oracle fp64)", "rows": []}
    for name in [n for n in R.GOAL_KNOWN if R.kind_of(n) == "loco"]:
        for label, kw in (("task actuator: kp 4.5, kd 0.2, clamp 1.5 N m", {}), ("task actuator, path at half speed", dict(stretch=2.0)),
                          ("task actuator, path at a third of the speed", dict(stretch=3.0)), ("control: clamp lifted (1e3 N m)", dict(tau_max=1e3)),
                          ("control: clamp lifted, stiff PD (kp 45, kd 2)", dict(tau_max=1e3, pd_kp=45.0, kd=2.0))):
            r = replay(rm, rec[name], **kw); r.update(file=name, actuator=label); out["rows"].append(r)
            print(f"{name:30s} {label:48s} rot_dist {r['first_rot_dist']:.2f} -> min {r['min_rot_dist']:.3f}  worst joint error {r['worst_joint_error']:.3f} rad  "
                  f"ended on step {r['terminated_on_step']} {'(success)' if r['success'] else ''}", flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
