"""GNN policy: torch modules with the reference's names / parameter names (trainable with autograd, state_dict
compatible with scripts/graph_model_orebot_ov.py) plus the matrix-core forward pass for rollouts.

Reference: RobotLearning/omniisaacgymenvs/scripts/graph_model_orebot_ov.py (GraphLayer :11-80, GraphNet :82-159,
Action_Layer :215-226, Value_Layer :228-241); instantiated with hidden_features = out_features = 32
(scripts/skrl_ppo_locomanipulation_vertical.py:44-49).  torch_scatter is not required: the max aggregation over the
fixed 24-edge graph is an `amax` over each node's incoming messages.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn


def create_edge_index(device=None) -> torch.Tensor:
    """(2, 24): row 0 = source, row 1 = target (graph_model_orebot_ov.py:142-159)."""
    e1 = torch.tensor([[0] * 4, list(range(1, 5))], dtype=torch.long, device=device)
    e2 = torch.tensor([list(range(1, 5)), list(range(5, 9))], dtype=torch.long, device=device)
    e3 = torch.tensor([list(range(5, 9)), list(range(9, 13))], dtype=torch.long, device=device)
    e = torch.cat([e1, e2, e3], dim=1)
    return torch.cat([e, torch.cat([e[1:2, :], e[0:1, :]], dim=0)], dim=1)


def create_features(inp: torch.Tensor):
    """obs (B,64) -> object feature (B,16), joint features (B,12,4) (graph_model_orebot_ov.py:112-140)."""
    obj = inp[:, 0:16]
    cols = list(range(4)) + [4 + 2 * i for i in range(4)] + [5 + 2 * i for i in range(4)]
    idx = torch.tensor(cols, device=inp.device)
    joint = torch.stack([inp[:, 16 + idx], inp[:, 28 + idx], inp[:, 40 + idx], inp[:, 52 + idx]], dim=-1)
    return obj, joint


class GraphLayer(nn.Module):
    def __init__(self, in_features, hidden_features, out_features):
        super().__init__()
        self.linear1 = nn.Linear(in_features * 2, hidden_features)
        self.elu1 = nn.ELU()
        self.linear2 = nn.Linear(hidden_features, out_features)
        self.elu2 = nn.ELU()

    def forward(self, h, edge_index):
        src, tgt = edge_index[0], edge_index[1]
        m = torch.cat([h.index_select(-2, tgt), h.index_select(-2, src)], dim=-1)      # [h_i (target) || h_j (source)]
        m = self.elu2(self.linear2(self.elu1(self.linear1(m))))
        n = h.shape[-2]
        out = torch.zeros(h.shape[:-2] + (n, m.shape[-1]), dtype=m.dtype, device=m.device)
        index = tgt.view(1, -1, 1).expand(m.shape)
        return out.scatter_reduce(-2, index, m, reduce="amax", include_self=False)     # = torch_scatter.scatter(reduce='max')


class GraphNet(nn.Module):
    def __init__(self, hidden_features=32, out_features=32):
        super().__init__()
        self.input_features = self.hidden_features = hidden_features
        self.input_layer1 = nn.Linear(16, hidden_features)
        self.input_layer2 = nn.Linear(4, hidden_features)
        self.graph_layer1 = GraphLayer(hidden_features, hidden_features, hidden_features)
        self.graph_layer2 = GraphLayer(hidden_features, hidden_features, hidden_features)
        self.graph_layer3 = GraphLayer(hidden_features, hidden_features, out_features)
        self.register_buffer("edge_index", create_edge_index(), persistent=False)

    def forward(self, inp):
        obj, joint = create_features(inp)
        h = torch.cat([self.input_layer1(obj).unsqueeze(1), self.input_layer2(joint)], dim=1)
        h = self.graph_layer1(h, self.edge_index)
        h = self.graph_layer2(h, self.edge_index)
        return self.graph_layer3(h, self.edge_index)


class Action_Layer(nn.Module):
    def __init__(self, hidden_features=32, num_actions=12):
        super().__init__()
        self.action_layer = nn.Linear(hidden_features, 1)
        self.num_actions = num_actions

    def forward(self, inp):
        return self.action_layer(inp[:, 1:13]).squeeze(-1)


class Value_Layer(nn.Module):
    def __init__(self, hidden_features=32):
        super().__init__()
        self.hidden_features = hidden_features
        self.action_layer = nn.Linear(hidden_features, 1)

    def forward(self, inp):
        return self.action_layer(torch.max(inp, dim=1).values)


def pack_gnn_params(net: GraphNet, mean_layer: Action_Layer, value_layer: Value_Layer, obs_mean=None, obs_var=None, eps=1e-8, clip=5.0) -> torch.Tensor:
    """Flatten the parameters in the order include/lm_policy.h documents (optionally with the observation scaler folded in)."""
    assert net.hidden_features == 32, "the matrix-core kernel is built for hidden_features = 32"
    parts = [net.input_layer1.weight, net.input_layer1.bias, net.input_layer2.weight, net.input_layer2.bias]
    for gl in (net.graph_layer1, net.graph_layer2, net.graph_layer3):
        parts += [gl.linear1.weight, gl.linear1.bias, gl.linear2.weight, gl.linear2.bias]
    parts += [mean_layer.action_layer.weight, mean_layer.action_layer.bias, value_layer.action_layer.weight, value_layer.action_layer.bias]
    for q in parts:      # the tile splits every fp32 operand into two fp16 halves (csrc/lm_policy_dev.h): weights must be inside the fp16 range
        if not bool(torch.isfinite(q).all()) or float(q.detach().abs().max()) >= 6.0e4:
            raise ValueError("GNN weights must be finite and below the fp16 range (6e4) for the split-fp16 matrix products")
    dev = net.input_layer1.weight.device
    parts += [torch.zeros(64, device=dev) if obs_mean is None else obs_mean.to(dev),
              torch.ones(64, device=dev) if obs_var is None else 1.0 / (obs_var.to(dev).float().sqrt() + eps),
              torch.tensor([clip if obs_var is not None else 3.0e38], device=dev)]
    return torch.cat([p.detach().reshape(-1).float() for p in parts]).contiguous()


def gnn_forward_hip(obs: torch.Tensor, packed: torch.Tensor):
    """Policy forward on the GPU's matrix cores: obs (B,64) cuda float32 -> (mean (B,12), value (B,1))."""
    from ..lib import load_library
    lib = load_library()
    assert obs.is_cuda and obs.dtype == torch.float32 and obs.is_contiguous() and obs.shape[1] == 64
    assert packed.device == obs.device and packed.dtype == torch.float32 and packed.is_contiguous() and packed.numel() == lib.lm_gnn_param_count()
    B = obs.shape[0]
    mean = torch.empty((B, 12), device=obs.device); value = torch.empty((B, 1), device=obs.device)
    with torch.cuda.device(obs.device):            # the kernel launches on the calling thread's current device
        rc = lib.lm_gnn_forward(C.c_void_p(obs.data_ptr()), B, C.c_void_p(packed.data_ptr()), C.c_void_p(mean.data_ptr()),
                                C.c_void_p(value.data_ptr()), C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"lm_gnn_forward failed ({rc})")
    return mean, value


class GraphPolicy(nn.Module):
    """Shared policy/value model of scripts/skrl_ppo_locomanipulation_vertical.py:25-59 with `use_graph = True`."""

    def __init__(self, num_actions=12):
        super().__init__()
        self.net = GraphNet(32, 32)
        self.mean_layer = Action_Layer(32, num_actions)
        self.log_std_parameter = nn.Parameter(torch.zeros(num_actions))
        self.value_layer = Value_Layer(32)
        self._packed = None

    def forward(self, obs):
        h = self.net(obs)
        return self.mean_layer(h), self.log_std_parameter, self.value_layer(h)

    @torch.no_grad()
    def act_inference(self, obs):
        """Rollout-time forward on the HIP kernel (call `refresh()` after every optimiser step)."""
        if self._packed is None or self._packed.device != obs.device:
            self.refresh(obs.device)
        mean, value = gnn_forward_hip(obs, self._packed)
        return mean, self.log_std_parameter, value

    def refresh(self, device=None, obs_mean=None, obs_var=None, eps=1e-8, clip=5.0):
        self._packed = pack_gnn_params(self.net, self.mean_layer, self.value_layer, obs_mean, obs_var, eps, clip).to(device or self.log_std_parameter.device)
