"""torch-CPU fp32 restatement of the keras layer stack the hot path uses (TEST INFRASTRUCTURE).

Layers are plain dicts: {"kind": "rescale", "div": 255.0} | {"kind": "conv", "filters", "kernel",
"stride", "act"} | {"kind": "flatten"} | {"kind": "dense", "units", "act"}.  Parameters are a list
of torch tensors in keras order (kernel HWIO / [in,out], then bias) so they can be copied to and
from agents_amd's flat buffer.  Arithmetic mirrors keras Conv2D(padding='valid') / Dense on NHWC
inputs (tf_agents/examples/dqn/mnih15/dqn_train_eval_atari.py:80-112;
tf_agents/networks/q_network.py:46-158, encoding_network.py:222-359).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _act(x, act):
    if act in (None, "none", "linear"):
        return x
    if act == "relu":
        return torch.relu(x)
    if act == "tanh":
        return torch.tanh(x)
    raise ValueError(act)


def param_shapes(layers, input_shape):
    """[(kernel_shape, bias_shape), ...] per parametrised layer; also returns output shape."""
    shape = tuple(input_shape)
    out = []
    for l in layers:
        k = l["kind"]
        if k == "rescale":
            pass
        elif k == "conv":
            H, W, C = shape
            kh, kw = l["kernel"]
            s = l["stride"]
            out.append(((kh, kw, C, l["filters"]), (l["filters"],)))
            shape = ((H - kh) // s + 1, (W - kw) // s + 1, l["filters"])
        elif k == "flatten":
            shape = (int(np.prod(shape)),)
        elif k == "dense":
            out.append(((shape[-1], l["units"]), (l["units"],)))
            shape = shape[:-1] + (l["units"],)
        else:
            raise ValueError(k)
    return out, shape


def init_params(layers, input_shape, seed=0, scale=2.0):
    """VarianceScaling(scale, fan_in, truncated_normal)-like init; deterministic numpy."""
    rng = np.random.RandomState(seed)
    shapes, _ = param_shapes(layers, input_shape)
    params = []
    for ks, bs in shapes:
        fan_in = int(np.prod(ks[:-1]))
        std = math.sqrt(scale / fan_in)
        params.append(torch.tensor(rng.randn(*ks).astype(np.float32) * np.float32(std)))
        params.append(torch.tensor((rng.randn(*bs) * 0.1).astype(np.float32)))
    return params


def forward(layers, params, x, dtype=torch.float32):
    """x: [B, *input_shape] (uint8 or float32).  Returns [B, out].  `dtype=torch.float64` (with
    float64 params) is the arbiter the parity tests rank two fp32 implementations against."""
    it = iter(params)
    h = x
    for l in layers:
        k = l["kind"]
        if k == "rescale":
            h = h.to(dtype) / l["div"]
        elif k == "conv":
            w = next(it)
            b = next(it)
            h = h.to(dtype)
            y = F.conv2d(h.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), b, stride=l["stride"])
            h = _act(y.permute(0, 2, 3, 1), l["act"])
        elif k == "flatten":
            h = h.reshape(h.shape[0], -1)
        elif k == "dense":
            w = next(it)
            b = next(it)
            h = _act(h.to(dtype) @ w + b, l["act"])
    return h


ATARI_LAYERS = [
    {"kind": "rescale", "div": 255.0},
    {"kind": "conv", "filters": 32, "kernel": (8, 8), "stride": 4, "act": "relu"},
    {"kind": "conv", "filters": 64, "kernel": (4, 4), "stride": 2, "act": "relu"},
    {"kind": "conv", "filters": 64, "kernel": (3, 3), "stride": 1, "act": "relu"},
    {"kind": "flatten"},
    {"kind": "dense", "units": 512, "act": "relu"},
]


def atari_q_layers(num_actions):
    return ATARI_LAYERS + [{"kind": "dense", "units": num_actions, "act": None}]


def mlp_q_layers(fc, num_actions, act="relu"):
    return [{"kind": "dense", "units": u, "act": act} for u in fc] + \
           [{"kind": "dense", "units": num_actions, "act": None}]


def forward_branch(layers, params, x, masks=None, dtype=torch.float64):
    """The same stack as `forward`, returning (output, [pre-activation of every parametrised
    layer]).  With `masks` (one 0/1 tensor per parametrised layer, None for layers without an
    activation) every ReLU is replaced by a multiplication with the given mask: the network is
    evaluated -- and differentiated by autograd -- on the linear branch some OTHER implementation
    took.  The float64 arbiter of tests/test_gpu_bench_config.py uses this to separate the two
    ways in which fp32 implementations of a ReLU network differ: rounding of the sums (small,
    every step) and units whose exact pre-activation is within rounding of zero landing on
    different sides of the ReLU (rare, but each one removes a whole term from a gradient sum)."""
    it = iter(params)
    h = x
    pre = []
    pi = 0
    for l in layers:
        k = l["kind"]
        if k == "rescale":
            h = h.to(dtype) / l["div"]
            continue
        if k == "flatten":
            h = h.reshape(h.shape[0], -1)
            continue
        w = next(it)
        b = next(it)
        h = h.to(dtype)
        if k == "conv":
            z = F.conv2d(h.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), b,
                         stride=l["stride"]).permute(0, 2, 3, 1)
        else:
            z = h @ w + b
        pre.append(z)
        if masks is not None and masks[pi] is not None and l["act"] == "relu":
            h = z * masks[pi].to(dtype)
        else:
            h = _act(z, l["act"])
        pi += 1
    return h, pre
