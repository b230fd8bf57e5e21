"""BASELINE.json configs 3, 4 (its per-GPU block) and 5 at their REAL sizes, through size-independent properties
(config 2 is tests/test_gpu_parity.py::test_full_size_invariants_4096; the oracle comparisons run at 16-256 envs).

  config 3  QuadrupedManipulatePlate, 4096 envs: plate-pose sanity on top of the generic invariants
  config 4  JointLocomanipulation, 2048 locomotion + 2048 manipulation envs per GPU: per-half extras and success windows
  config 5  JointLocomanipulationVertical, 8192 envs with the GNN policy in the loop (48-step fused rollouts)

Invariants: finite state / outputs, unit quaternions, clipped observations, reward inside the reference's assert window (> -50,
quadruped_pose_control.py:556-558) and below bonus + best orientation reward, counters inside their ranges, returned resets == reset_buf,
extras == means of the per-env terms (the fused cross-wavefront reduction is complete), resets do occur, no contained blow-up.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def generic_invariants(task, out, half=None):
    obs_d, rew, resets, extras = out
    e = task.engine; s = e.state; c = e.cnt
    obs = obs_d["obs"]
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all() and torch.isfinite(s).all()
    assert obs.abs().max() <= 5.0 and float(rew.min()) > -50 and float(rew.max()) < 610
    assert (s[86:90].norm(dim=0) - 1).abs().max() < 1e-5                                        # goal quaternions
    assert int(c[4].max()) <= int(task._max_episode_length) - 1 and int(c[4].min()) >= 1 and int(c[1].max()) <= 17 and int(c[1].min()) >= 0
    assert torch.equal(resets, c[3]) and set(torch.unique(c[3]).tolist()) <= {0, 1}
    tm = e.terms[:7].double().mean(dim=1)
    ex = torch.stack([extras[k] for k in list(extras)[:7]]).double()
    assert (ex - tm).abs().max() < 1e-5 * max(1.0, float(tm.abs().max()))
    assert e.blowups == 0


def test_config3_manipulation_4096():
    import locomanipulationrl_amd as lm
    N = 4096
    env = lm.make_env("QuadrupedManipulatePlate", num_envs=N, seed=42)
    env.reset(); task = env._task
    task.engine.terms                                   # ask for the per-env reward terms: the engine writes them from now on (include/lm_engine.h, lm_ptr_kind)
    g = torch.Generator(device="cuda").manual_seed(42)
    total = 0
    for t in range(300):
        out = env.step(torch.rand(N, 12, device="cuda", generator=g) * 2 - 1)
        total += int(out[2].sum())
        if t % 50 == 49:
            generic_invariants(task, out)
            s = task.engine.state
            assert (s[40:44].norm(dim=0) - 1).abs().max() < 1e-5                                # plate quaternion
            # the plate stays near the inverted robot: an env whose plate slides off or is thrown resets (plate-frame height tests,
            # quadruped_manipulate_plate.py:576-603) before it can leave this box.  Lower z bound = a plate's half-width below the base plane
            # (z 0): the resets are written in the PLATE frame, so a tilted plate sliding past the frame edge can be centimetres below the base
            # plane for the few steps until the corner / knee test fires (-0.059 seen with round 3's friction cone; same reason as config 4's bound)
            assert float(s[37:39].abs().max()) < 0.6 and float(s[39].min()) > -0.25 and float(s[39].max()) < 0.6
            assert float(s[44:50].abs().max()) < 50.0
            assert (task.plate_pos_ground - s[37:40].T).abs().max() == 0                        # the reference-named views are the engine's memory
    assert total > 0
    env.close()


def test_config4_cotrain_block_2048_2048():
    import locomanipulationrl_amd as lm
    N = 4096; h = N // 2
    env = lm.make_env("JointLocomanipulation", num_envs=N, seed=42)
    obs = env.reset(); task = env._task
    task.engine.terms                                   # ask for the per-env reward terms
    assert obs["obs"].shape == (N, 64) and obs["states"].data_ptr() == obs["obs"].data_ptr()   # states alias obs (joint_locomanipulation.py:548)
    g = torch.Generator(device="cuda").manual_seed(7)
    res_l = res_m = 0
    for t in range(300):
        out = env.step(torch.rand(N, 12, device="cuda", generator=g) * 2 - 1)
        res_l += int(out[2][:h].sum()); res_m += int(out[2][h:].sum())
        if t % 50 == 49:
            generic_invariants(task, out)
            s = task.engine.state
            assert (s[3:7, :h].norm(dim=0) - 1).abs().max() < 1e-5 and (s[40:44, h:].norm(dim=0) - 1).abs().max() < 1e-5
            # locomotion half: bases near their drop point; manipulation half: plates around the inverted robots fixed at z 0.5.
            # Why the plate bound is 0.3 and not the robots' 0.5: a plate that random actions have thrown off the feet is reset by the task's
            # own tests, all of which are written in the PLATE frame (quadruped_manipulate_plate.py:576-603: robot base / corners / knees
            # against the plate's plane), not by its world height - a tilted plate sliding past the frame edge can be 5-20 cm below the base
            # plane for the few steps until the corner or knee test fires (0.449 was seen in round 2).  0.3 = a plate's half-width below the base.
            assert float(s[2, :h].min()) > 0.0 and float(s[2, :h].max()) < 0.3 and float(s[39, h:].min()) > 0.3 and float(s[39, h:].max()) < 1.0
            ex = out[3]
            for k in ("env/success_rate", "env/success_rate_loco", "env/success_rate_mani"):
                assert 0.0 <= float(ex[k]) <= 1.0
            # {successes, resets} x {all, first half, second half}.  The three windows are NOT additive ("halves add up to the whole" was
            # asserted in round 2's first draft and is wrong): the reference keeps three independent counters and empties each one on its own
            # when IT passes max_reset_counts = 2048 (joint_locomanipulation.py:795-830: success_rate, success_rate_loco, success_rate_mani
            # are reset in three separate `if` blocks), so after the first roll-over they cover different spans of steps.
            st = task.engine.stats_i64
            assert all(0 <= int(st[2 * k]) <= int(st[2 * k + 1]) <= 2048 + N for k in range(3))
    assert res_l > 0 and res_m > 0
    env.close()


def test_config5_vertical_gnn_8192_rollout():
    import locomanipulationrl_amd as lm
    from locomanipulationrl_amd.policies.graph_model import GraphPolicy, pack_gnn_params
    N, T = 8192, 48
    env = lm.make_env("JointLocomanipulationVertical", num_envs=N, seed=42)
    first = env.reset(); task = env._task
    torch.manual_seed(42)
    pol = GraphPolicy().cuda()
    ro = task.make_rollout("gnn", pack_gnn_params(pol.net, pol.mean_layer, pol.value_layer).cuda(), torch.full((12,), -0.7, device="cuda"), T, noise_seed=1)
    ro.obs[0].copy_(first["obs"])
    resets = 0
    for it in range(6):
        ro.run("auto"); torch.cuda.synchronize()
        assert torch.isfinite(ro.obs).all() and torch.isfinite(ro.actions).all() and torch.isfinite(ro.rewards).all() and torch.isfinite(ro.values).all()
        assert ro.obs.abs().max() <= 5.0 and float(ro.rewards.min()) > -50 and float(ro.rewards.max()) < 610
        assert set(torch.unique(ro.dones).tolist()) <= {0, 1}
        resets += int(ro.dones.sum())
        # the policy really is in the loop: actions are the GNN's gaussian samples (not constant, inside a few sigma of a bounded mean)
        assert float(ro.actions.std()) > 0.1 and float(ro.actions.abs().max()) < 50
        with torch.no_grad():
            mean, _, _ = pol(ro.obs[3].clamp(-5, 5))
        assert float((ro.actions[3] - mean).std()) == pytest.approx(float(np.exp(-0.7)), rel=0.05)
        e = task.engine; s = e.state; h = N // 2
        assert torch.isfinite(s).all() and (s[3:7, :h].norm(dim=0) - 1).abs().max() < 1e-5 and (s[40:44, h:].norm(dim=0) - 1).abs().max() < 1e-5
        assert int(e.cnt[4].max()) <= 299 and e.blowups == 0
        ro.obs[0].copy_(ro.obs[T])
    assert resets > 0
    ro.close(); env.close()


def test_extras_reduction_beyond_131040_wavefronts():
    """The counted accumulator words of the extras reduction take at most 4095 arrivals: above 32 x 4095 wavefronts lm_create
    doubles the first-level rows.  2.1 M envs = 131 251 wavefronts -> 64 rows; the means and the success-window counters must still be exact."""
    from locomanipulationrl_amd.engine_config import loco_params
    from locomanipulationrl_amd.lib import Engine
    from locomanipulationrl_amd.model.robot_model import load_model
    N = 2_100_010                                       # not a multiple of 16: the last wavefront is ragged too
    ep = loco_params(); eng = Engine(load_model("quadruped_robot_v2"), [ep], N, seed=3)
    g = torch.Generator(device="cuda").manual_seed(3)
    ex = torch.empty(13, device="cuda"); eng.terms      # (asked for before the steps whose values are read)
    ns = nr = 0
    for t in range(4):
        a = torch.rand(N, 12, device="cuda", generator=g) * 2 - 1
        if t == 1: eng.cnt[4][::3] = ep.max_episode - 2      # a third of the envs time out in this step: 700 k resets, the window rolls over in the next
        eng.step(a, out_extras=ex)
        tm = eng.terms[:7].double().mean(dim=1)
        assert (ex[:7].double() - tm).abs().max() < 1e-5 * max(1.0, float(tm.abs().max()))
        # the window of quadruped_pose_control.py:618-633 restated on the host
        if nr > ep.max_reset_counts: ns = nr = 0
        ns += int(eng.cnt[2].sum()); nr += int(eng.cnt[3].sum())
        assert int(eng.stats_i64[0]) == ns and int(eng.stats_i64[1]) == nr
    assert eng.blowups == 0 and 0 < int(eng.stats_i64[1]) < 700_000      # resets were counted, and the window has rolled over
    eng.close()


@pytest.mark.parametrize("case", ["loco", "vertical", "mani", "cotrain_ragged"])
def test_two_wavefront_step_kernel_is_bit_identical(case, monkeypatch):
    """Beyond 32 768 envs lm_step launches k_step_w2 on locomotion engines, the step kernel compiled for two wavefronts per SIMD
    (csrc/lm_engine_w2.hip).  Same arithmetic in the same order: forced at a test size (LM_W2_MIN_ENVS, read by lm_create) it must
    return the bits of k_step - outputs, extras and the whole simulator state - over 60 random-action steps with their resets.  Manipulation and
    co-training engines stay on k_step at every size (the cases check that the switch leaves them alone)."""
    from locomanipulationrl_amd.engine_config import loco_params, mani_params
    from locomanipulationrl_amd.lib import Engine
    from locomanipulationrl_amd.model.robot_model import load_model
    rm = load_model("quadruped_robot_v2")
    cq = [-1.2, 1.2, 1.2, -1.2, -1.22, -1.92, 1.92, 1.22, 1.92, 1.22, -1.22, -1.92]
    if case == "loco": N, eps, kw = 4096, [loco_params()], {}
    elif case == "vertical":      # the other robot model (quadfinger.urdf) through the task class's own parameters
        from locomanipulationrl_amd.utils.config import SimConfig, load_config
        from locomanipulationrl_amd.utils.task_util import task_map
        name = "QuadrupedPoseControlVertical"; N = 2064
        task = task_map()[name](name=name, sim_config=SimConfig(load_config(name, num_envs=N)), env=None)
        rm, eps, kw = load_model(task.model_asset), task.engine_params(), {}
    elif case == "mani": N, eps, kw = 4096, [mani_params()], {}
    else: N, eps, kw = 4090, [loco_params(init_q=cq, init_base_pos=[0, 0, 0.18]), mani_params(init_q=cq, fixed_base_pos=[0, 0, 0.5], init_plate_pos=[0, 0, 0.68])], dict(split_env=2048)
    monkeypatch.delenv("LM_W2_MIN_ENVS", raising=False)
    a = Engine(rm, eps, N, seed=5, **kw)
    monkeypatch.setenv("LM_W2_MIN_ENVS", "16")
    b = Engine(rm, eps, N, seed=5, **kw)
    monkeypatch.delenv("LM_W2_MIN_ENVS", raising=False)
    g = torch.Generator(device="cuda").manual_seed(5)
    outs = [[torch.zeros(N, 64, device="cuda"), torch.zeros(N, 93, device="cuda"), torch.zeros(N, device="cuda"), torch.zeros(N, dtype=torch.int64, device="cuda"), torch.zeros(13, device="cuda")] for _ in range(2)]
    resets = 0
    for t in range(60):
        act = torch.rand(N, 12, device="cuda", generator=g) * 2 - 1
        a.step(act, None, *outs[0]); b.step(act, None, *outs[1])
        for x, y in zip(outs[0], outs[1]): assert torch.equal(x, y), (case, t)
        resets += int(outs[0][3].sum())
    assert torch.equal(a.state, b.state) and torch.equal(a.cnt, b.cnt) and resets > 0
    a.close(); b.close()
