"""numpy float64 restatement of the reference's GNN policy (scripts/graph_model_orebot_ov.py:11-241).
TEST INFRASTRUCTURE ONLY.  Pinned by tests/golden/gnn.npz, which tools/gen_golden.py produced by running the
reference's own GraphNet / Action_Layer / Value_Layer (with a stand-in for the absent torch_scatter)."""
import numpy as np


def elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))


def edges():
    src = [0] * 4 + list(range(1, 5)) + list(range(5, 9))
    tgt = list(range(1, 5)) + list(range(5, 9)) + list(range(9, 13))
    return np.array(src + tgt), np.array(tgt + src)           # :142-159: forward edges then their reverses


def gnn_forward(obs, sd):
    """obs (B,64); sd: dict of arrays with the reference's state_dict keys ('net.input_layer1.weight', ...)."""
    obs = np.asarray(obs, dtype=np.float64); B = obs.shape[0]
    W = {k: np.asarray(v, dtype=np.float64) for k, v in sd.items()}
    cols = list(range(4)) + [4 + 2 * i for i in range(4)] + [5 + 2 * i for i in range(4)]        # :115-126
    joint = np.stack([obs[:, 16 + np.array(cols)], obs[:, 28 + np.array(cols)], obs[:, 40 + np.array(cols)], obs[:, 52 + np.array(cols)]], -1)
    h = np.zeros((B, 13, 32))
    h[:, 0] = obs[:, :16] @ W["net.input_layer1.weight"].T + W["net.input_layer1.bias"]
    h[:, 1:] = joint @ W["net.input_layer2.weight"].T + W["net.input_layer2.bias"]
    src, tgt = edges()
    for l in (1, 2, 3):
        p = f"net.graph_layer{l}."
        m = np.concatenate([h[:, tgt], h[:, src]], -1)                                            # [h_i || h_j], :53-59
        m = elu(elu(m @ W[p + "linear1.weight"].T + W[p + "linear1.bias"]) @ W[p + "linear2.weight"].T + W[p + "linear2.bias"])
        hn = np.full((B, 13, 32), -np.inf)
        for e in range(24):
            hn[:, tgt[e]] = np.maximum(hn[:, tgt[e]], m[:, e])                                    # scatter(reduce='max') onto the target
        h = np.where(np.isfinite(hn), hn, 0.0)
    mean = (h[:, 1:13] @ W["mean_layer.action_layer.weight"].T + W["mean_layer.action_layer.bias"])[..., 0]
    value = h.max(1) @ W["value_layer.action_layer.weight"].T + W["value_layer.action_layer.bias"]
    return h, mean, value
