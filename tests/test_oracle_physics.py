"""Weak pins of the physics specification ("what pins the physics instead"): conservation laws with
first-order convergence, analytic free fall, standing stability, and the envelope of the reference's recorded
PhysX joint trajectories (row 0 of RL/tasks/joint_train_locomanipulation/*.npy, extracted into
tests/golden/npy_row0.npz).  PhysX itself cannot be run: parity against it is UNPINNED."""
import os
from dataclasses import replace

import numpy as np
import pytest

from conftest import GOLDEN
from locomanipulationrl_amd.engine_config import loco_params, mani_params
from oracle.lmo import Oracle


def quat2mat(q):
    w, x, y, z = q
    return np.array([[1-2*(y*y+z*z), 2*(x*y-w*z), 2*(x*z+w*y)], [2*(x*y+w*z), 1-2*(x*x+z*z), 2*(y*z-w*x)], [2*(x*z-w*y), 2*(y*z+w*x), 1-2*(x*x+y*y)]])


def momentum_energy(o, ph):
    M, h = o.dyn_terms(ph); R = quat2mat(ph[3:7])
    u = np.concatenate([R.T @ ph[10:13], R.T @ ph[7:10], ph[25:37]])
    hb = M[:6] @ u; n, f = R @ hb[:3], R @ hb[3:]
    return np.concatenate([n + np.cross(ph[:3], f), f]), 0.5 * u @ M @ u


def test_free_flight_conserves_momentum_and_energy_first_order(robot_model):
    errs = []
    for dt in (1e-3, 1e-4):
        o = Oracle(robot_model, loco_params(dt=dt, kd=0.0, gravity=0.0))
        phys, task, cnt = o.new_state(1); o.reset(phys, task, cnt)
        rng = np.random.default_rng(0)
        phys[0, 2] = 5.0; phys[0, 7:13] = rng.normal(size=6) * 0.5; phys[0, 25:37] = rng.normal(size=12) * 2
        m0, e0 = momentum_energy(o, phys[0])
        for _ in range(int(0.2 / dt)):
            o.substep(phys, np.zeros((1, 12)))
        m1, e1 = momentum_energy(o, phys[0])
        errs.append((np.abs(m1 - m0).max(), abs(e1 - e0) / e0))
    assert errs[0][0] < 2e-3 and errs[0][1] < 5e-4
    assert errs[1][0] < errs[0][0] / 5 and errs[1][1] < errs[0][1] / 5            # O(dt) convergence


def test_free_fall_is_analytic(robot_model):
    o = Oracle(robot_model, loco_params())
    phys, task, cnt = o.new_state(1); o.reset(phys, task, cnt); phys[0, 2] = 10.0
    n = 40
    for _ in range(n):
        o.substep(phys, np.zeros((1, 12)))
    t = n * 0.0083
    assert abs(phys[0, 9] + 9.81 * t) < 1e-6                                       # v_z = -g t exactly (semi-implicit Euler)
    assert abs(phys[0, 2] - (10.0 - 9.81 * 0.0083 ** 2 * n * (n + 1) / 2)) < 1e-6
    assert np.abs(phys[0, 3:7] - [1, 0, 0, 0]).max() < 1e-6                        # zero-velocity drive holds the limbs: no tumbling


def test_standing_is_stable(robot_model):
    ep = loco_params(); o = Oracle(robot_model, ep)
    phys, task, cnt = o.new_state(4); o.reset(phys, task, cnt)
    for _ in range(300):
        o.substep(phys, np.zeros((4, 12)))
    tips, knees = o.fk(phys)
    feet = np.stack([robot_model.foot_centres(ph[13:25], quat2mat(ph[3:7]), ph[:3]) for ph in phys])
    assert np.abs(feet[:, :, 2] - ep.tip_radius).max() < 1e-3                      # the foot spheres rest on the ground
    assert 0.12 < phys[0, 2] < 0.14 and np.abs(phys[:, 7:13]).max() < 0.02        # base settled ~0.131 m
    assert np.abs(phys[:, 13:25] - np.array(ep.init_q)).max() < 0.03               # velocity servo creeps slowly under load
    assert knees[:, :, 2].min() > 0.04 and np.abs(phys[:, 3:7] - [1, 0, 0, 0]).max() < 0.01


def test_ground_reaction_balances_weight(robot_model):
    """Static equilibrium: the normal impulses the contact solver finds for a settled robot add up to m g dt (and the plate's weight is
    carried by the four tips in the manipulation scene) - a known answer for the contact model that needs no PhysX."""
    import ctypes as C
    for ep, total in ((loco_params(), float(np.sum(robot_model.mass))), (mani_params(), mani_params().plate_mass)):
        o = Oracle(robot_model, ep)
        phys, task, cnt = o.new_state(1); o.reset(phys, task, cnt)
        for _ in range(400):
            o.substep(phys, np.zeros((1, 12)))
        W = np.zeros((12, 12)); vf = np.zeros(12); bn = np.zeros(4); lam = np.zeros(12)
        o.lib.lmo_contact_problem(C.byref(o.model), C.byref(o.params), o._p(phys[0]), o._p(np.zeros(12)), o._p(W), o._p(vf), o._p(bn), o._p(lam))
        fn = lam[0::3].sum() / ep.dt
        assert abs(fn - total * ep.gravity) < 0.03 * total * ep.gravity, (fn, total * ep.gravity)
        assert (lam[0::3] > 0).all()                                                    # all four tips carry load
        assert (np.hypot(lam[1::3], lam[2::3]) <= ep.mu * lam[0::3] + 1e-12).all()     # every contact's friction impulse inside its cone


def test_plate_rests_on_inverted_robot(robot_model):
    ep = mani_params(); o = Oracle(robot_model, ep)
    phys, task, cnt = o.new_state(2); o.reset(phys, task, cnt)
    for _ in range(300):
        o.substep(phys, np.zeros((2, 12)))
    feet = np.stack([robot_model.foot_centres(ph[13:25], quat2mat(ep.fixed_base_quat), np.array(ep.fixed_base_pos)) for ph in phys])
    # flipped plate (quat [0,1,0,0]): its lower world face is plate-frame z = 0.008
    assert np.abs((phys[:, 39] - 0.008) - (feet[:, :, 2].mean(1) + ep.tip_radius)).max() < 1.5e-3
    assert np.abs(phys[:, 44:50]).max() < 0.02 and np.abs(phys[:, 37:39]).max() < 5e-3


def test_reference_npy_row0_envelope(robot_model):
    """One control period after reset the reference's PhysX joints sit within 4.2e-3 rad (loco) / 1.3e-2 rad (mani) of
    init_joint_pos.  The recordings were made under an unknown first policy action, so this is an envelope:
    ours under zero action must stay inside it and near the recorded values."""
    g = np.load(os.path.join(GOLDEN, "npy_row0.npz"))
    init = np.array([-1.2, 1.2, 1.2, -1.2, -1.22, -1.92, 1.92, 1.22, 1.92, 1.22, -1.22, -1.92])
    loco = g["mlp_joint_loco"]; mani = g["mlp_joint_mani"]
    assert np.abs(loco - init).max() < 4.5e-3 and np.abs(mani - init).max() < 1.3e-2
    # co-train poses: loco base z 0.18 (quadruped_robot.py:53), mani base z 0.5 inverted + plate z 0.68 (joint_locomanipulation.py:139,198)
    pl = loco_params(init_q=list(init), init_base_pos=[0, 0, 0.18])
    pm = mani_params(init_q=list(init), fixed_base_pos=[0, 0, 0.5], init_plate_pos=[0, 0, 0.68])
    for ep, ref, tol in ((pl, loco, 5e-3), (pm, mani, 2.5e-2)):
        o = Oracle(robot_model, ep)
        phys, task, cnt = o.new_state(1)
        o.step(phys, task, cnt, np.zeros((1, 12)), seed=0)
        assert np.abs(phys[0, 13:25] - init).max() < tol
        assert np.abs(phys[0, 13:25] - ref).max() < tol + 3e-3
        assert np.abs(phys[0, 13:25]).max() < np.pi


def test_saturated_drive_respects_torque_limit(robot_model):
    """A target far from the joint velocity saturates the drive: the velocity change per sub-step is bounded by
    tau_max * dt / (smallest joint-space inertia), not by the 100 N m s/rad gain."""
    ep = loco_params(tau_max=1.5); o = Oracle(robot_model, ep)      # a real 1.5 N m clamp (the PD-actuator families; `sim.engine.tau_max: 1.5`)
    phys, task, cnt = o.new_state(1); o.reset(phys, task, cnt); phys[0, 2] = 5.0
    M, _ = o.dyn_terms(phys[0])
    o.substep(phys, np.full((1, 12), 3.0))
    dv = np.abs(phys[0, 25:37])
    assert dv.max() < 3.0                                                      # did not jump to the target (it would unsaturated)
    Minv = np.linalg.inv(M)
    assert dv.max() <= 1.5 * 0.0083 * np.abs(Minv[6:, 6:]).sum(1).max() * 1.05
    o2 = Oracle(robot_model, replace(ep, tau_max=1e9))
    p2, t2, c2 = o2.new_state(1); o2.reset(p2, t2, c2); p2[0, 2] = 5.0
    o2.substep(p2, np.full((1, 12), 3.0))
    assert np.abs(p2[0, 25:37] - 3.0).max() < 0.02                             # unsaturated: servo reaches the target in one step


def test_pd_actuator_clamp_follows_the_reference_rule(robot_model):
    """PD-actuator families: the reference clamps the PD torque evaluated on the state BEFORE the sub-step
    (quadruped_pose_control_custom_controller.py:289-293).  Every joint whose explicit torque kp (q* - q) - kd qd leaves +-1.5 N m gets exactly the
    limit torque; the others get the implicit form kd (v* - qd_end) in the same single pass.  The implicit torque of such a joint can leave the
    limit - in fewer than 1e-3 of the joint-sub-steps of this (rougher than the task's) random walk of targets, by at most kd x the velocity change of
    the sub-step - and `pd_second_pass` puts those
    joints on the limit too and solves again: then the applied torque never exceeds it (a third-order leftover is tolerated at 1e-4)."""
    from locomanipulationrl_amd.engine_config import loco_cc_params
    N = 64
    for second in (0, 1):
        ep = loco_cc_params(pd_second_pass=second); o = Oracle(robot_model, ep)
        phys, task, cnt = o.new_state(N); rng = np.random.default_rng(0)
        o.step(phys, task, cnt, np.zeros((N, 12)), seed=0)
        qstar = phys[:, 13:25].copy(); n = n_sat = n_over = 0; worst = 0.0
        for t in range(30):
            qstar = np.clip(qstar + 0.1 * rng.uniform(-1, 1, (N, 12)), phys[:, 13:25] - 1.0, phys[:, 13:25] + 1.0)
            for s in range(ep.substeps):
                tg = ep.pd_kp / ep.kd * (qstar - phys[:, 13:25]); tau_e = ep.kd * (tg - phys[:, 25:37])
                tau = o.substep_tau(phys, tg); S = np.abs(tau_e) > ep.tau_max
                assert np.allclose(tau[S], np.sign(tau_e[S]) * ep.tau_max)
                assert (np.abs(tau) <= ep.tau_max * (1 + 1e-12)).all()            # the LOGGED torque is clipped like the reference's (:289-293), whatever was applied
                on_limit = np.abs(tau) >= ep.tau_max * (1 - 1e-12)                # joints the solve put on the limit (pre-step rule, or the second pass)
                applied = np.where(on_limit & (S | bool(second)), tau, ep.kd * (tg - phys[:, 25:37]))      # the others: the implicit torque on the end-of-step velocity
                n += S.size; n_sat += int(S.sum()); n_over += int((np.abs(applied) > ep.tau_max * (1 + 1e-9)).sum()); worst = max(worst, float(np.abs(applied).max()))
        assert n_sat > 0.01 * n, n_sat / n
        if second:
            assert n_over <= 1e-4 * n, n_over / n
        else:
            assert 0 < n_over <= 1e-3 * n and worst < ep.tau_max + ep.kd * 2 * ep.max_joint_vel, (n_over / n, worst)      # the excess is the implicit damping of one sub-step's velocity change


def test_pd_actuator_scheme_follows_the_continuous_law(robot_model):
    """The custom-controller tasks' actuator.  The reference evaluates  clamp(kp (q* - q) - kd qd)  explicitly per sub-step; at its
    own dt = 0.005 that discretisation is far from the law it stands for (kd dt / I is of order 1 on these light links): against the same explicit
    scheme at dt / 16 - the continuous-time actuator - it is off by tenths of a radian within a second, the shipped implicit form by a hundredth.
    The shipped scheme is therefore pinned to the fine-step explicit law, not to the reference's coarse one."""
    import npy_replay as R
    from locomanipulationrl_amd.engine_config import loco_cc_params
    ep = loco_cc_params(); tg = R.pd_targets(robot_model, ep, N=8, steps=16)
    truth = R.pd_actuator_trajectories(robot_model, ep, "explicit", 16, tg)
    shipped = R.pd_actuator_trajectories(robot_model, ep, "implicit", 1, tg)
    coarse = R.pd_actuator_trajectories(robot_model, ep, "explicit", 1, tg)
    rms = lambda a: float(np.sqrt(np.nanmean((a - truth) ** 2)))
    assert np.isfinite(truth).all() and np.isfinite(shipped).all()
    assert rms(shipped) < 0.02 and np.abs(shipped - truth).max() < 0.15, (rms(shipped), np.abs(shipped - truth).max())
    assert (not np.isfinite(coarse).all()) or rms(coarse) > 5 * rms(shipped), (rms(coarse), rms(shipped))


def test_shipped_sweep_counts_meet_the_convergence_criterion(robot_model):
    """engine_config.PGS_ITERS_*: the contact velocity change of a sub-step is within 1 % (median) / 20 % (90th percentile) of the 128-sweep
    solve's with the shipped counts - 8 on the ground under the rigid velocity drives, 4 on the plate and for the soft PD actuators - and
    half the ground's count is NOT (the feet couple through the base: 6 % / 33 %), so the ground's count is not padded either.  States from
    random-action rollouts; the full table is `convergence` in profiles/r03_npy_replay_evidence.json."""
    import npy_replay as R
    from locomanipulationrl_amd.engine_config import loco_cc_params, loco_params, mani_cc_params, mani_params
    for name, ep in (("ground", loco_params()), ("plate", mani_params()), ("ground, PD actuator", loco_cc_params()), ("plate, PD actuator", mani_cc_params())):
        e = R.sweep_convergence(robot_model, ep, [ep.pgs_iters], N=64, steps=16)[ep.pgs_iters]
        assert len(e) > 100 and np.median(e) <= 0.01 and np.percentile(e, 90) <= 0.20, (name, ep.pgs_iters, np.median(e), np.percentile(e, 90))
    e = R.sweep_convergence(robot_model, loco_params(), [4], N=64, steps=16)[4]
    assert np.median(e) > 0.02 and np.percentile(e, 90) > 0.20


def test_effort_and_position_control_modes(robot_model):
    """RobotOmni.take_action's other two control modes (robot/base/robot.py:444-461), golden = the reference's own scaling
    (tests/golden/take_action.npz).  Effort: the generalised force the oracle applies to the 12 driven joints is exactly the golden effort
    (free flight, no gravity, from rest: M du/dt on the joint rows = tau).  Position: the PD drive (robot_description.py:37-41 gains
    kp 5, kd 1) pulls every joint onto the golden position target."""
    import math
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "take_action.npz"))
    a = g["actions"].astype(np.float64); N = a.shape[0]
    # ---- effort (1e-3 of the 1.5 N m scale: a full-scale torque on these links would run into the 450 deg/s joint speed limit within the sub-step)
    ep = loco_params(drive_mode=2, act_scale=1.5e-3, gravity=0.0)
    o = Oracle(robot_model, ep)
    phys, task, cnt = o.new_state(N); o.reset(phys, task, cnt); phys[:, 2] = 5.0
    M = [o.dyn_terms(phys[e])[0] for e in range(N)]
    o.substep(phys, a * ep.act_scale)
    for e in range(N):
        R = quat2mat([1, 0, 0, 0])
        du = np.concatenate([R.T @ phys[e, 10:13], R.T @ phys[e, 7:10], phys[e, 25:37]]) / ep.dt          # from rest: u = du
        f = M[e] @ du
        assert np.abs(f[6:] - 1e-3 * g["effort"][e, :12]).max() < 1e-7 and np.abs(f[:6]).max() < 5e-6, (e, np.abs(f[6:] - 1e-3 * g["effort"][e, :12]).max())
    # ---- position (targets inside the admissible range of the loop closure: the golden targets span +-pi, scale them into +-0.4 rad about the pose)
    ep = loco_params(drive_mode=1, act_scale=math.pi, pd_kp=5.0, kd=1.0, gravity=0.0, tau_max=1e9)
    o = Oracle(robot_model, ep)
    phys, task, cnt = o.new_state(N); o.step(phys, task, cnt, np.zeros((N, 12)), seed=0)
    phys[:, 2] = 5.0; phys[:, 7:13] = 0
    q0 = np.array(ep.init_q)
    act = (q0 + 0.12 * g["position"].astype(np.float64)) / math.pi          # q* = init pose + 0.12 * golden position target
    for _ in range(150):
        phys[:, 2] = 5.0
        o.step(phys, task, cnt, act, seed=0); cnt[:, 3] = 0
    assert np.abs(phys[:, 13:25] - act * math.pi).max() < 2e-3, np.abs(phys[:, 13:25] - act * math.pi).max()
