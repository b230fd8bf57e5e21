"""Replay of the reference's recorded PhysX joint trajectories (tests/golden/npy_traj.npz) -- shared by the CPU (oracle) and GPU (HIP
engine) tests.  TEST INFRASTRUCTURE.

What the recordings are (RobotLearning/omniisaacgymenvs/tasks/joint_train_locomanipulation/joint_locomanipulation.py:36-46,556-563,
861-874): the co-training task with its committed FIXED goal (roll 0.2, pitch 0.2, yaw 0.785, :61-66) appends the 12 driven joint positions
of env 0 of the locomotion half and of the manipulation robot view in every get_observations() call, i.e. one row per control step
starting with the reset step (zero action), until that half's reset flag is raised; the episode that ends the file is therefore
recorded up to and including its terminal step.  Scene of that task: class-default joint pose (robot/quadruped_robot.py:45-52),
locomotion base dropped from z 0.18 (:53), inverted manipulation robot fixed at z 0.5 with the plate dropped from z 0.68 (:139,198).

What can be recovered: PhysX's velocity drive tracks its target within a control period, so the action sequence is
a_t = clip((q[t+1] - q[t]) / (3.0 rad/s * 0.0332 s), +-1).  Replaying it open loop from the same reset pins the physics of row a7
(drive + contact + articulated dynamics) against the only simulator-derived numbers the reference holds.

Which goal a file was recorded under is known only for the committed code (0.2, 0.2, 0.785); the three `04roll-*` files carry
another goal in their name, so orientation statistics are taken from the `mlp_*` files and `test` only.
"""
import os

import numpy as np

from conftest import GOLDEN
from locomanipulationrl_amd.engine_config import loco_params, mani_params

INIT_Q = [-1.2, 1.2, 1.2, -1.2, -1.22, -1.92, 1.92, 1.22, 1.92, 1.22, -1.22, -1.92]      # robot/quadruped_robot.py:45-52
GOAL = [0.2, 0.2, 0.785]                                                                 # joint_locomanipulation.py:61-66
FULL = 3.0 * 0.0332                                                                      # joint displacement of a saturated action

# the two duplicated recordings (byte-identical arrays) are replayed once
FILES = ["04roll_loco_from_mani", "04roll_loco_from_scratch", "04roll_mani_from_scratch", "mlp_joint_loco", "mlp_joint_loco_from_mani",
         "mlp_joint_loco_from_scratch", "mlp_joint_mani", "mlp_joint_mani_from_scratch", "mlp_loco_from_mani", "mlp_mani_from_loco", "test"]
GOAL_KNOWN = [f for f in FILES if f.startswith("mlp_")]


def kind_of(name):
    return "mani" if "mani" in name.split("from")[0] else "loco"


def load():
    g = np.load(os.path.join(GOLDEN, "npy_traj.npz"))
    return {k: g[k].astype(np.float64) for k in g.files}


def cotrain_params(kind, **kw):
    """The two parameter blocks of JointLocomanipulation with the committed fixed goal."""
    if kind == "loco":
        return loco_params(**{**dict(init_q=list(INIT_Q), init_base_pos=[0, 0, 0.18], goal_lo=GOAL, goal_hi=GOAL), **kw})
    return mani_params(**{**dict(init_q=list(INIT_Q), fixed_base_pos=[0, 0, 0.5], init_plate_pos=[0, 0, 0.68], goal_lo=GOAL, goal_hi=GOAL), **kw})


def recovered_actions(rec):
    return np.clip(np.diff(rec, axis=0) / FULL, -1.0, 1.0)


def rot_dist(obs_row):
    return 2.0 * np.arcsin(min(float(np.linalg.norm(obs_row[7:10])), 1.0))


def replay(rec, step, servo=False, until_done=False):
    """step(action (12,)) -> (q (12,), obs (64,), reset flag, goal_reset flag, reward) advances one control step; the first call gets the
    zero action of VecEnvRLGames.reset().  servo=True steers the joints back onto the recording every step (actions still clipped to +-1).
    The replay stops on the engine's own reset flag or after the recording's last row (until_done: holds the last action's successor at
    zero and keeps stepping up to 2 more rows to see a success streak that started a row or two late complete)."""
    T = rec.shape[0]
    q, obs, rst, _, rew = step(np.zeros(12))
    out = dict(T=T, row0_err=float(np.abs(q - rec[0]).max()), row0=q.copy(), rows=[q.copy()], rd=[rot_dist(obs)], rew=[rew], done_at=None, goal=0)
    acts = recovered_actions(rec)
    derr = []
    for t in range(T - 1):
        a = np.clip((rec[t + 1] - q) / FULL, -1, 1) if servo else acts[t]
        q0 = q
        q, obs, rst, goal, rew = step(a)
        out["rows"].append(q.copy()); out["rd"].append(rot_dist(obs)); out["rew"].append(rew); derr.append((q - q0) - (rec[t + 1] - rec[t]))
        if rst:
            out["done_at"], out["goal"] = t + 1, int(goal)
            break
    out["rd_rec"] = np.array(out["rd"])                      # rot_dist over the recorded rows only
    if until_done and out["done_at"] is None:
        for k in range(2):                                   # the recording is over: hold still (zero velocity targets) for at most two more rows
            q, obs, rst, goal, rew = step(np.zeros(12))
            out["rd"].append(rot_dist(obs)); out["rew"].append(rew)
            if rst:
                out["done_at"], out["goal"] = T + k, int(goal)
                break
    rows = np.array(out["rows"]); derr = np.array(derr); n = len(rows)
    out.update(rows=rows, rd=np.array(out["rd"]), rew=np.array(out["rew"]), tracked=float((np.abs(derr) < 1e-3).mean()), qerr=float(np.abs(rows - rec[:n]).max()),
               early=float(np.abs(derr[:4]).mean() / FULL), succ_row=T - 17)
    first = np.nonzero(out["rd_rec"] <= 0.15)[0]
    out["first_succ"] = int(first[0]) if len(first) else None
    # rows of PhysX's success window [T - 17, T - 1] on which this replay is inside the 0.15 rad window too (rows it did not reach count as outside)
    win = out["rd_rec"][T - 17:T]
    out["in_window"] = int((win <= 0.15).sum())
    return out


def oracle_stepper(robot_model, ep, precision="f64"):
    from oracle.lmo import Oracle
    o = Oracle(robot_model, ep, precision)
    phys, task, cnt = o.new_state(1)

    def step(a):
        obs, st, rew, terms = o.step(phys, task, cnt, np.asarray(a, dtype=np.float64)[None], seed=0)
        return phys[0, 13:25].astype(np.float64).copy(), obs[0].astype(np.float64), int(cnt[0, 3]), int(cnt[0, 2]), float(rew[0])
    return step


def summary_line(name, r):
    return (f"{name:30s} T-1={r['T'] - 1:3d} row0={r['row0_err']:.4f} tracked={r['tracked']:.3f} qerr={r['qerr']:.3f} early={r['early']:.4f} "
            f"done_at={r['done_at']}{'G' if r['goal'] else ''} first<=0.15:{r['first_succ']} (PhysX {r['succ_row']}) min rd={r['rd_rec'].min():.3f} last rd={r['rd_rec'][-1]:.3f} in-window {r['in_window']}/17")


def sweep_convergence(robot_model, ep, counts, N=128, steps=24, every=4, seed=0):
    """Relative error of the contact velocity change of ONE sub-step (base / plate twist + 12 joint rates) against the 128-sweep solve, from
    identical states sampled along a random-action rollout of `ep` with random drive targets: {count: array over the envs with an active
    contact}.  The criterion behind engine_config.PGS_ITERS_*: median <= 1 %, 90th percentile <= 20 %, at least 4 sweeps."""
    from dataclasses import replace
    from oracle.lmo import Oracle
    o = Oracle(robot_model, ep); phys, task, cnt = o.new_state(N); rng = np.random.default_rng(seed)
    o.step(phys, task, cnt, np.zeros((N, 12)), seed=0)
    solvers = {k: Oracle(robot_model, replace(ep, pgs_iters=k)) for k in set(counts) | {0, 128}}
    acc = {k: [] for k in counts}
    for t in range(steps):
        o.step(phys, task, cnt, rng.uniform(-1, 1, (N, 12)), seed=0)
        if t % every:
            continue
        tg = ep.act_scale * rng.uniform(-1, 1, (N, 12)); res = {}
        for k, ok in solvers.items():
            ph = phys.copy(); ok.substep(ph, tg)
            res[k] = np.concatenate([ph[:, 7:13] if ep.mode == 0 else ph[:, 44:50], ph[:, 25:37]], 1)
        den = np.linalg.norm(res[128] - res[0], axis=1); m = den > 1e-3
        for k in counts:
            acc[k].append(np.linalg.norm(res[k] - res[128], axis=1)[m] / den[m])
    return {k: np.concatenate(v) for k, v in acc.items()}


def pd_actuator_trajectories(robot_model, ep0, scheme, K, targets):
    """The PD-actuator law  tau = clamp(kp (q* - q) - kd qd, +-max_effort)  of the custom-controller tasks under three discretisations, same scene,
    same joint position targets (targets: (steps, N, 12), one row per control step); returns the joint positions after every control step.
      "explicit"   the reference's own scheme (quadruped_pose_control_custom_controller.py:289-293): the torque is evaluated on the state before the
                   sub-step and held over it (the oracle's effort mode with that torque; viscous joint damping added explicitly)
      "implicit"   the shipped scheme: joints outside the limit by the pre-step torque get the limit torque, the others the law
                   with the end-of-step velocity; "implicit2" = with pd_second_pass
    K divides the sub-step: "explicit" at K = 16 stands for the continuous-time actuator both schemes discretise."""
    from dataclasses import replace
    from oracle.lmo import Oracle
    kp, kd, cj, tm = ep0.pd_kp, ep0.kd, ep0.joint_damping, ep0.tau_max
    if scheme == "explicit":
        ep = replace(ep0, variant=0, num_obs=64, drive_mode=2, dt=ep0.dt / K, substeps=ep0.substeps * K, tau_max=1e9, act_scale=1.0, kd=100.0)
    else:
        ep = replace(ep0, dt=ep0.dt / K, substeps=ep0.substeps * K, pd_second_pass=1 if scheme == "implicit2" else 0)
    o = Oracle(robot_model, ep); phys, _, _ = Oracle(robot_model, ep0).new_state(targets.shape[1]); out = []
    for qs in targets:
        for _ in range(ep.substeps):
            q, qd = phys[:, 13:25], phys[:, 25:37]
            o.substep(phys, (np.clip(kp * (qs - q) - kd * qd, -tm, tm) - cj * qd) if scheme == "explicit" else kp / kd * (qs - q))
        out.append(phys[:, 13:25].copy())
    return np.array(out)


def pd_targets(robot_model, ep0, N, steps, seed=1):
    """A random walk of joint position targets, 0.1 rad per control step as the task's action scale allows, within 0.6 rad of the reset pose."""
    from oracle.lmo import Oracle
    rng = np.random.default_rng(seed); ph, _, _ = Oracle(robot_model, ep0).new_state(N); q0 = ph[:, 13:25].copy(); q = q0.copy(); tg = []
    for _ in range(steps):
        q = np.clip(q + 0.1 * rng.uniform(-1, 1, (N, 12)), q0 - 0.6, q0 + 0.6); tg.append(q.copy())
    return np.array(tg)
