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
