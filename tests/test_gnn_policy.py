"""GNN policy: numpy oracle and torch modules pinned to the reference's own GraphNet outputs
(tests/golden/gnn.npz); the matrix-core HIP forward against both on the GPU."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from locomanipulationrl_amd.policies.graph_model import (Action_Layer, GraphNet, GraphPolicy, Value_Layer, create_edge_index,
                                                         pack_gnn_params)
from oracle.gnn_ref import gnn_forward


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(GOLDEN, "gnn.npz"))


def state_dict(g):
    return {k: g[k] for k in g.files if k.startswith(("net.", "mean_layer.", "value_layer."))}


def load_modules(g):
    net, act, val = GraphNet(32, 32), Action_Layer(32, 12), Value_Layer(32)
    net.load_state_dict({k[4:]: torch.as_tensor(g[k]) for k in g.files if k.startswith("net.")})
    act.load_state_dict({k[len("mean_layer."):]: torch.as_tensor(g[k]) for k in g.files if k.startswith("mean_layer.")})
    val.load_state_dict({k[len("value_layer."):]: torch.as_tensor(g[k]) for k in g.files if k.startswith("value_layer.")})
    return net, act, val


def test_numpy_oracle_matches_reference(golden):
    h, mean, value = gnn_forward(golden["obs"], state_dict(golden))
    assert np.abs(h - golden["h"]).max() < 2e-6 and np.abs(mean - golden["mean"]).max() < 2e-6 and np.abs(value - golden["value"]).max() < 2e-6
    assert np.array_equal(create_edge_index().numpy(), golden["edge_index"])


def test_torch_modules_are_state_dict_compatible_with_reference(golden):
    net, act, val = load_modules(golden)
    with torch.no_grad():
        h = net(torch.as_tensor(golden["obs"]))
        assert (h - torch.as_tensor(golden["h"])).abs().max() < 1e-6
        assert (act(h) - torch.as_tensor(golden["mean"])).abs().max() < 1e-6
        assert (val(h) - torch.as_tensor(golden["value"])).abs().max() < 1e-6
    # trainable: gradients flow to every parameter
    pol = GraphPolicy()
    mean, log_std, value = pol(torch.randn(8, 64))
    (mean.sum() + value.sum()).backward()
    assert all(p.grad is not None for n, p in pol.named_parameters() if n != "log_std_parameter")
    assert mean.shape == (8, 12) and value.shape == (8, 1) and log_std.shape == (12,)


def test_param_packing_size(golden):
    net, act, val = load_modules(golden)
    assert pack_gnn_params(net, act, val).numel() == 704 + 3 * 3136 + 66 + 129


@pytest.mark.gpu
@pytest.mark.parametrize("B", [40, 8192, 8197])          # golden batch, BASELINE config 5 size, ragged last wavefront
def test_hip_forward_matches_reference_and_oracle(golden, B):
    from locomanipulationrl_amd.lib import build_library
    from locomanipulationrl_amd.policies.graph_model import gnn_forward_hip
    build_library()
    net, act, val = load_modules(golden)
    packed = pack_gnn_params(net, act, val).cuda()
    if B == 40:
        obs = torch.as_tensor(golden["obs"]).cuda()
        ref_mean, ref_value = golden["mean"], golden["value"]
    else:
        obs = (torch.randn(B, 64, generator=torch.Generator().manual_seed(B)) * 1.5).cuda()
        _, ref_mean, ref_value = gnn_forward(obs.cpu().numpy(), state_dict(golden))
    mean, value = gnn_forward_hip(obs.contiguous(), packed)
    torch.cuda.synchronize()
    # fp32 MFMA = an fmaf chain (exact fp32 products): tolerance 1e-5 on O(1) outputs
    assert np.abs(mean.cpu().numpy() - ref_mean).max() < 1e-5
    assert np.abs(value.cpu().numpy() - ref_value).max() < 1e-5
    with torch.no_grad():
        h = net.cuda()(obs)
        assert (act.cuda()(h) - mean).abs().max() < 1e-5 and (val.cuda()(h) - value).abs().max() < 1e-5
        # with the observation scaler folded into the kernel
        mu, var = obs.mean(0), obs.var(0)
        m2, v2 = gnn_forward_hip(obs.contiguous(), pack_gnn_params(net, act, val, mu, var, 1e-8, 5.0))
        h2 = net(torch.clamp((obs - mu) / (var.sqrt() + 1e-8), -5, 5))
        assert (act(h2) - m2).abs().max() < 1e-5 and (val(h2) - v2).abs().max() < 1e-5


@pytest.mark.gpu
def test_gnn_policy_drives_vertical_env():
    """BASELINE config 5: vertical configuration + GNN policy in the loop (actions from the policy, not zeroed)."""
    import locomanipulationrl_amd as lm
    env = lm.make_env("QuadrupedPoseControlVertical", num_envs=512)
    pol = GraphPolicy().cuda()
    obs = env.reset()["obs"]
    for _ in range(20):
        mean, log_std, value = pol.act_inference(obs.contiguous())
        o, rew, resets, _ = env.step(mean.clamp(-1, 1))
        obs = o["obs"]
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all() and torch.isfinite(mean).all()
    env.close()


def test_mlp_policy_shape_and_param_count():
    from locomanipulationrl_amd.policies.mlp_model import SharedMLP, pack_mlp_params
    m = SharedMLP()
    assert sum(p.numel() for p in m.parameters()) == 58649                # SURVEY Appendix F (trunk + heads + 12 log_std)
    mean, log_std, value = m(torch.randn(5, 64))
    assert mean.shape == (5, 12) and value.shape == (5, 1)
    assert pack_mlp_params(m).numel() == 132 + 256 * 64 + 256 + 128 * 256 + 128 + 64 * 128 + 64 + 16 * 64 + 16


@pytest.mark.gpu
@pytest.mark.parametrize("B", [4096, 4101])
def test_mlp_hip_forward_matches_torch(B):
    """MFMA fp32 forward vs the eager fp32 torch modules, with and without the observation scaler (tolerance 2e-5)."""
    from locomanipulationrl_amd.lib import build_library
    from locomanipulationrl_amd.policies.mlp_model import SharedMLP, mlp_forward_hip, pack_mlp_params
    build_library()
    torch.manual_seed(0)
    m = SharedMLP().cuda()
    obs = (torch.randn(B, 64, generator=torch.Generator().manual_seed(1)) * 2).cuda()
    with torch.no_grad():
        ref_mean, _, ref_value = m(obs)
    mean, value = mlp_forward_hip(obs, pack_mlp_params(m))
    assert (mean - ref_mean).abs().max() < 2e-5 and (value - ref_value).abs().max() < 2e-5
    mu, var = obs.mean(0), obs.var(0)
    with torch.no_grad():
        xn = torch.clamp((obs - mu) / (var.sqrt() + 1e-8), -5, 5)
        ref_mean, _, ref_value = m(xn)
    mean, value = mlp_forward_hip(obs, pack_mlp_params(m, mu, var, 1e-8, 5.0))
    assert (mean - ref_mean).abs().max() < 2e-5 and (value - ref_value).abs().max() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("policy", ["mlp", "gnn"])
def test_fused_rollout_graph_equals_stepwise(policy):
    """T steps of forward -> sampling -> lm_step as ONE hipGraph launch give bit-identical buffers to the same kernels
    driven step by step from Python, and to the un-captured enqueue; sampled actions follow N(mean, std) with the stated log-prob."""
    import math
    import numpy as np
    from locomanipulationrl_amd.engine_config import loco_params
    from locomanipulationrl_amd.lib import Engine, Rollout, POLICY_GNN, POLICY_MLP, sample_actions
    from locomanipulationrl_amd.model.robot_model import load_model
    from locomanipulationrl_amd.policies.graph_model import GraphPolicy, pack_gnn_params, gnn_forward_hip
    from locomanipulationrl_amd.policies.mlp_model import SharedMLP, pack_mlp_params, mlp_forward_hip
    torch.manual_seed(3)
    N, T = 512, 12
    rm = load_model("quadruped_robot_v2")
    if policy == "mlp":
        model = SharedMLP().cuda(); packed = pack_mlp_params(model, None, None).cuda(); fwd = mlp_forward_hip; kind = POLICY_MLP
    else:
        model = GraphPolicy().cuda(); packed = pack_gnn_params(model.net, model.mean_layer, model.value_layer).cuda(); fwd = gnn_forward_hip; kind = POLICY_GNN
    log_std = torch.full((12,), -0.7, device="cuda")
    engines = [Engine(rm, [loco_params()], N, seed=4) for _ in range(3)]
    outs0 = []
    for e in engines:      # the reset step, to have observations to start from
        o = torch.empty(N, 64, device="cuda"); e.step(torch.zeros(N, 12, device="cuda"), None, o); outs0.append(o)
    ro_graph = Rollout(engines[0], kind, packed, log_std, T, noise_seed=77); ro_graph.obs[0] = outs0[0]
    ro_plain = Rollout(engines[1], kind, packed, log_std, T, noise_seed=77); ro_plain.obs[0] = outs0[1]
    ro_graph.run(use_graph=True); ro_plain.run(use_graph=False)
    # step-by-step reference on the third engine with the same kernels
    e = engines[2]; obs = outs0[2]
    ref = dict(obs=[obs.clone()], act=[], logp=[], val=[], rew=[], done=[])
    for t in range(T):
        mean, value = fwd(obs.contiguous(), packed)
        act, logp = sample_actions(e, mean, log_std, 77)
        o = torch.empty(N, 64, device="cuda"); r = torch.empty(N, device="cuda"); d = torch.empty(N, dtype=torch.int64, device="cuda")
        e.step(act, None, o, None, r, d)
        ref["act"].append(act); ref["logp"].append(logp); ref["val"].append(value.reshape(-1)); ref["rew"].append(r); ref["done"].append(d); ref["obs"].append(o.clone())
        obs = o
    _, vlast = fwd(obs.contiguous(), packed); ref["val"].append(vlast.reshape(-1))
    torch.cuda.synchronize()
    for ro in (ro_graph, ro_plain):
        assert torch.equal(ro.obs, torch.stack(ref["obs"])) and torch.equal(ro.actions, torch.stack(ref["act"]))
        assert torch.equal(ro.logp, torch.stack(ref["logp"])) and torch.equal(ro.values, torch.stack(ref["val"]))
        assert torch.equal(ro.rewards, torch.stack(ref["rew"])) and torch.equal(ro.dones, torch.stack(ref["done"]))
    assert torch.equal(engines[0].state, engines[2].state) and torch.equal(engines[0].cnt, engines[2].cnt)
    # a second replay of the same graph continues the trajectory (fresh noise: the counters moved on)
    ro_graph.obs[0] = ro_graph.obs[T]; a_prev = ro_graph.actions.clone(); ro_graph.run(use_graph=True); torch.cuda.synchronize()
    assert not torch.equal(a_prev, ro_graph.actions) and torch.isfinite(ro_graph.obs).all()
    # sampling statistics and log-prob
    mean_all = []
    obs_t = ro_plain.obs[:T].reshape(-1, 64)
    m, _ = fwd(obs_t.contiguous(), packed)
    eps = ((ro_plain.actions.reshape(-1, 12) - m) / log_std.exp()).cpu().numpy()
    assert abs(eps.mean()) < 0.02 and abs(eps.std() - 1.0) < 0.02 and abs((eps ** 4).mean() - 3.0) < 0.2
    c = np.corrcoef(eps[:, :4].T); assert np.abs(c - np.eye(4)).max() < 0.05
    lp = (-0.5 * eps ** 2 - log_std.cpu().numpy() - 0.5 * math.log(2 * math.pi)).sum(1)
    assert np.abs(lp - ro_plain.logp.reshape(-1).cpu().numpy()).max() < 2e-4
    for r in (ro_graph, ro_plain): r.close()
    for e in engines: e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nobs", [64, 88])
def test_mlp_forward_matches_torch_for_both_observation_widths(nobs):
    """MLP trunk 64-256-128-64 + heads on fp32 MFMA (4 wavefronts per 16-sample tile) against the torch modules, with the running
    observation scaler folded in; 88 = the custom-controller observation (…custom_controller.py:432-455)."""
    from locomanipulationrl_amd.policies.mlp_model import SharedMLP, pack_mlp_params, mlp_forward_hip
    torch.manual_seed(nobs)
    m = SharedMLP(num_observations=nobs).cuda()
    mean, var = torch.randn(nobs, device="cuda") * 0.3, torch.rand(nobs, device="cuda") + 0.2
    packed = pack_mlp_params(m, mean, var, 1e-8, 5.0)
    for B in (16, 4096, 4101):
        obs = torch.randn(B, nobs, device="cuda") * 2
        mu, v = mlp_forward_hip(obs, packed)
        with torch.no_grad():
            x = torch.clamp((obs - mean) / (var.sqrt() + 1e-8), -5, 5)
            mu_t, _, v_t = m(x)
        assert (mu - mu_t).abs().max() < 2e-5 and (v - v_t).abs().max() < 2e-5


@pytest.mark.gpu
def test_fused_rollout_on_a_custom_controller_task():
    """88-wide observations: the fused rollout runs the custom-controller task with the MLP on MFMA (f-1 x f-2)."""
    import locomanipulationrl_amd as lm
    from locomanipulationrl_amd.policies.mlp_model import SharedMLP, pack_mlp_params, mlp_forward_hip
    env = lm.make_env("QuadrupedPoseControlCustomController", num_envs=256)
    obs = env.reset()["obs"]
    m = SharedMLP(num_observations=88).cuda(); packed = pack_mlp_params(m).cuda(); log_std = torch.full((12,), -1.0, device="cuda")
    ro = env._task.make_rollout("mlp", packed, log_std, T=8, noise_seed=5)
    assert ro.obs.shape == (9, 256, 88)
    ro.obs[0].copy_(obs); ro.run(); torch.cuda.synchronize()
    mu, v = mlp_forward_hip(ro.obs[3].contiguous(), packed)
    assert torch.equal(v.reshape(-1), ro.values[3]) and torch.isfinite(ro.obs).all() and float(ro.actions.abs().max()) > 0
    eps = (ro.actions[3] - mu) / log_std.exp()
    assert abs(float(eps.mean())) < 0.1 and abs(float(eps.std()) - 1.0) < 0.1
    ro.close(); env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("task_name,N,policy", [("QuadrupedPoseControl", 512, "mlp"), ("QuadrupedPoseControl", 500, "mlp"), ("QuadrupedManipulatePlate", 256, "mlp"),
                                                ("JointLocomanipulation", 512, "mlp"), ("QuadrupedPoseControlCustomController", 256, "mlp"),
                                                ("JointLocomanipulationPositionControl", 256, "mlp"),
                                                ("JointLocomanipulationVertical", 512, "gnn"), ("QuadrupedPoseControl", 500, "gnn"),
                                                ("JointLocomanipulation", 8192, "mlp")])      # more blocks than compute units: two generations of blocks
def test_persistent_rollout_kernel_equals_graph_replay(task_name, N, policy):
    """SURVEY 8 f-2, one-kernel form: every block keeps its 16 envs for the T steps (wavefront 0 steps them exactly like k_step, all four
    wavefronts run the policy tile -- MLP or GNN -- on the observations left in LDS) and the extras are published afterwards from per-step accumulators.  All
    rollout buffers, the extras, the simulator state, the counters and the success windows must equal the hipGraph replay bit for bit;
    two consecutive rollouts, so that the accumulators are checked to come back clean."""
    from locomanipulationrl_amd.lib import Engine, Rollout, POLICY_GNN, POLICY_MLP
    from locomanipulationrl_amd.model.robot_model import load_model
    from locomanipulationrl_amd.policies.graph_model import GraphPolicy, pack_gnn_params
    from locomanipulationrl_amd.policies.mlp_model import SharedMLP, pack_mlp_params
    from locomanipulationrl_amd.utils.config import SimConfig, load_config
    from locomanipulationrl_amd.utils.task_util import task_map
    torch.manual_seed(5)
    T = 20
    task = task_map()[task_name](name=task_name, sim_config=SimConfig(load_config(task_name, num_envs=N)), env=None)
    nobs = task.engine_params()[0].num_obs
    if policy == "mlp":
        model = SharedMLP(num_observations=nobs).cuda(); packed = pack_mlp_params(model, None, None).cuda(); kind = POLICY_MLP
    else:
        model = GraphPolicy().cuda(); packed = pack_gnn_params(model.net, model.mean_layer, model.value_layer).cuda(); kind = POLICY_GNN
    log_std = torch.full((12,), -0.3, device="cuda")          # large noise: resets, saturation and goal changes inside the window
    engs, ros = [], []
    for _ in range(2):
        e = Engine(load_model(task.model_asset), task.engine_params(), N, split_env=task.split_env(), seed=9)
        o0 = torch.empty(N, nobs, device="cuda"); e.step(torch.zeros(N, 12, device="cuda"), None, o0)
        r = Rollout(e, kind, packed, log_std, T, noise_seed=21); r.obs[0] = o0
        engs.append(e); ros.append(r)
    for rep in range(2):
        ros[0].run("graph"); ros[1].run("persistent"); torch.cuda.synchronize()
        for name in ("obs", "actions", "logp", "values", "rewards", "dones", "extras"):
            a, b = getattr(ros[0], name), getattr(ros[1], name)
            assert torch.equal(a, b), (task_name, policy, rep, name, float((a.float() - b.float()).abs().max()))
        assert torch.equal(engs[0].state, engs[1].state) and torch.equal(engs[0].cnt, engs[1].cnt)
        assert torch.equal(engs[0].stats_i64, engs[1].stats_i64) and torch.equal(engs[0].extras_buf, engs[1].extras_buf)
        assert torch.equal(engs[0].obs_buf, engs[1].obs_buf) and torch.equal(engs[0].rew_buf, engs[1].rew_buf)
        for r in ros: r.obs[0].copy_(r.obs[T])
    assert int(ros[0].dones.sum()) > 0                        # episodes did end inside the window
    for r in ros: r.close()
    for e in engs: e.close()


def _plain_graph_layer(layer, h, edge_index):
    """GraphLayer restated without scatter_reduce: per target node, the messages of its incoming edges are stacked and reduced by torch.max(dim) -
    a different autograd path (index-select backward of the max instead of amax's mask backward) through the same arithmetic."""
    src, tgt = edge_index[0].tolist(), edge_index[1].tolist()
    out = []
    for i in range(h.shape[-2]):
        inc = [e for e in range(len(tgt)) if tgt[e] == i]
        m = torch.stack([layer.elu2(layer.linear2(layer.elu1(layer.linear1(torch.cat([h[..., tgt[e], :], h[..., src[e], :]], -1))))) for e in inc], -2)
        out.append(m.max(dim=-2).values)
    return torch.stack(out, -2)


def test_gnn_backward_is_the_plain_per_node_backward():
    """the GNN forward is pinned to the reference's outputs, its BACKWARD (what PPO trains through: scatter_reduce 'amax')
    was not.  In float64: (1) finite differences agree with autograd on GraphLayer and on the whole GraphPolicy (gradcheck); (2) the gradients of
    every parameter and of the input agree to 1e-12 with a plain per-node restatement that reduces with torch.max over stacked messages - no
    scatter, another backward formula; (3) at an exact tie amax splits the gradient evenly where max(dim) picks one index (the sums agree): a
    measure-zero case for continuous observations.  So the learner's gradient path is the textbook one; the GNN's slow PPO curve (DESIGN.md 6.1:
    not reproduced) is not a backward bug."""
    from locomanipulationrl_amd.policies.graph_model import GraphLayer, GraphPolicy, create_edge_index
    torch.manual_seed(0)
    ei = create_edge_index()
    layer = GraphLayer(8, 8, 8).double()
    h = torch.randn(3, 13, 8, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda x: layer(x, ei), (h,), eps=1e-6, atol=1e-7)
    a = layer(h, ei); b = _plain_graph_layer(layer, h, ei)
    assert float((a - b).abs().max()) < 1e-14          # (batched against per-edge matmuls: summation order only)
    w = torch.randn_like(a)
    ga = torch.autograd.grad((a * w).sum(), [h] + list(layer.parameters()))
    gb = torch.autograd.grad((b * w).sum(), [h] + list(layer.parameters()))
    assert max(float((x - y).abs().max()) for x, y in zip(ga, gb)) < 1e-12
    # the whole policy: mean and value heads through three layers (64-wide observation, batch 2)
    pol = GraphPolicy().double()
    obs = torch.randn(2, 64, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda o: torch.cat([pol(o)[0], pol(o)[2]], -1), (obs,), eps=1e-6, atol=1e-6)
    # an exact tie: two incoming messages of node 5 made identical (sources 1 and 9 carry the same features)
    ht = torch.randn(1, 13, 8, dtype=torch.float64); ht[0, 9] = ht[0, 1]; ht.requires_grad_(True)
    ta = layer(ht, ei); tb = _plain_graph_layer(layer, ht, ei)
    assert float((ta - tb).abs().max()) < 1e-14
    g1, = torch.autograd.grad(ta[0, 5].sum(), ht, retain_graph=True); g2, = torch.autograd.grad(tb[0, 5].sum(), ht)
    assert float((g1[0, 1] + g1[0, 9] - g2[0, 1] - g2[0, 9]).abs().max()) < 1e-12          # same total; amax halves it between the tied sources
    assert float((g1[0, 1] - g1[0, 9]).abs().max()) < 1e-12


def test_mlp_weight_packing_is_a_lossless_enough_permutation():
    """pack_mlp_params (round 4): every weight matrix becomes 32-bit words [out/16][K-block][lane][hi 4 | lo 4] in the operand order of
    v_mfma_f32_16x16x32_f16, each weight split into two fp16 halves.  Unpacking with the column map of csrc/lm_policy_dev.h gives back W to 2^-21
    relative (the split carries 22 bits), for the natural-order first layer (88 columns padded to 96 with zeros) and for an accumulator-order layer;
    block sizes are those the kernel's offset table assumes."""
    from locomanipulationrl_amd.policies.mlp_model import SharedMLP, _pack_q, pack_mlp_params
    torch.manual_seed(3)
    for out_f, in_f, natural in ((256, 88, True), (256, 64, True), (128, 256, False), (16, 64, False)):
        W = torch.randn(out_f, in_f) * 0.3
        q = _pack_q(W, natural); KB = (in_f + 31) // 32
        assert q.numel() == (out_f // 16) * KB * 64 * 8 and q.dtype == torch.float32
        words = q.view(torch.int32).reshape(out_f // 16, KB, 64, 8)
        def halves(u):      # int32 [.., 4] -> fp16 values [.., 8]: even k-slot in the low half
            lo16 = (u & 0xFFFF).to(torch.int16); hi16 = ((u >> 16) & 0xFFFF).to(torch.int16)
            return torch.stack([lo16, hi16], -1).reshape(*u.shape[:-1], 8).view(torch.float16).float()
        val = halves(words[..., :4]) + halves(words[..., 4:])                                # hi + lo
        back = torch.zeros(out_f, 32 * KB)
        for mb in range(out_f // 16):
            for kb in range(KB):
                for lane in range(64):
                    n, g = lane & 15, lane >> 4
                    for e in range(8):
                        col = 32 * kb + 8 * g + e if natural else 16 * (2 * kb + (e >> 2)) + 4 * g + (e & 3)
                        back[16 * mb + n, col] = val[mb, kb, lane, e]
        assert (back[:, :in_f] - W).abs().max() <= 2.0 ** -21 * W.abs().max() and (back[:, in_f:] == 0).all()
    for nobs, kp in ((64, 64), (88, 96)):
        assert pack_mlp_params(SharedMLP(num_observations=nobs)).numel() == 2 * nobs + 4 + 256 * kp + 256 + 128 * 256 + 128 + 64 * 128 + 64 + 16 * 64 + 16
    m = SharedMLP();
    with torch.no_grad(): m.net[0].weight[0, 0] = 7.0e4
    with pytest.raises(ValueError, match="fp16 range"):
        pack_mlp_params(m)


@pytest.mark.gpu
def test_policy_tiles_on_the_fp16_matrix_pipe_are_fp32_accurate():
    """Round 4: the GNN and MLP tiles compute every fp32 product as four fp16 half products on v_mfma_f32_16x16x32_f16 (operands split hi + lo).
    Against the torch modules evaluated in float64 their error is that of an fp32 evaluation: within 4 x torch's own fp32 error + 2e-7 on 8192
    random observations, also with weights scaled by 4 (outputs in the thousands) and observations at the scaler's clip."""
    from locomanipulationrl_amd.policies.graph_model import GraphPolicy, gnn_forward_hip, pack_gnn_params
    from locomanipulationrl_amd.policies.mlp_model import SharedMLP, mlp_forward_hip, pack_mlp_params
    torch.manual_seed(0)
    for kind in ("gnn", "mlp"):
        for scale_w, scale_x in ((1.0, 1.0), (1.0, 5.0), (4.0, 1.0)):
            model = (GraphPolicy() if kind == "gnn" else SharedMLP()).cuda()
            with torch.no_grad():
                for p in model.parameters(): p.mul_(scale_w)
            m64 = (GraphPolicy() if kind == "gnn" else SharedMLP()).cuda().double(); m64.load_state_dict({k: v.double() for k, v in model.state_dict().items()})
            obs = (torch.randn(8192, 64, device="cuda") * scale_x).clamp(-5.0, 5.0)
            if kind == "gnn": m, v = gnn_forward_hip(obs, pack_gnn_params(model.net, model.mean_layer, model.value_layer).cuda())
            else: m, v = mlp_forward_hip(obs, pack_mlp_params(model).cuda())
            with torch.no_grad():
                mr, _, vr = m64(obs.double()); m32, _, v32 = model(obs)
            ref_err = max(float((m32.double() - mr).abs().max()), float((v32.double() - vr).abs().max()))
            hip_err = max(float((m.double() - mr).abs().max()), float((v.double() - vr).abs().max()))
            scale = max(1.0, float(mr.abs().max()), float(vr.abs().max()))
            assert hip_err <= 4.0 * ref_err + 2e-7 * scale, (kind, scale_w, scale_x, hip_err, ref_err)
