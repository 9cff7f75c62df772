"""GPU: the fused small-MLP kernels (csrc/mlp_small.hip; one forward launch, one backward launch
for a whole <=64-wide Dense stack) against torch fp64 autograd and against the per-layer GEMM path.
Tolerance 2e-5 relative to the largest reference value (fp32 summation order)."""
import numpy as np
import pytest
import torch

from agents_amd.networks import layers as L
from agents_amd.networks import sequential
from agents_amd.specs import tensor_spec

pytestmark = pytest.mark.gpu


def close(got, ref, tol=2e-5):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    scale = max(ref.abs().max().item(), 1e-30)
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


def build(n0, widths, acts, seed):
    layers = [L.Dense(w, a) for w, a in zip(widths, acts)]
    net = sequential.Sequential(layers, input_spec=tensor_spec.TensorSpec((n0,), torch.float32),
                                seed=seed)
    net.create_variables()
    return net


def ref_forward_backward(net, x, dy):
    ws = [torch.from_numpy(w.copy()).double().requires_grad_(True) for w in net.get_weights()]
    h = x.double().cpu().requires_grad_(True)
    cur = h
    for i, l in enumerate(net._param_layers):
        cur = cur @ ws[2 * i] + ws[2 * i + 1]
        if l.activation == "relu":
            cur = torch.relu(cur)
        elif l.activation == "tanh":
            cur = torch.tanh(cur)
    (cur * dy.double().cpu()).sum().backward()
    return cur.detach(), [w.grad for w in ws], h.grad


CASES = [(4096, 17, (64, 64, 6), ("tanh", "tanh", None)),
         (4096, 17, (64, 64, 1), ("tanh", "tanh", None)),
         (100, 4, (64, 2), ("relu", None)),
         (1, 3, (5, 7, 1), ("tanh", "relu", None)),
         (777, 64, (64, 64, 64, 64), ("relu", "tanh", "relu", "tanh")),
         (65, 11, (33,), (None,))]


@pytest.mark.parametrize("B,n0,widths,acts", CASES)
def test_fused_mlp_matches_autograd_and_gemm_path(dev, B, n0, widths, acts):
    g = torch.Generator().manual_seed(B + n0)
    x = torch.randn(B, n0, generator=g).to(dev)
    dy = torch.randn(B, widths[-1], generator=g).to(dev)
    net = build(n0, widths, acts, seed=5)
    assert net._fused_small_ok()
    y = net.forward(x, slot="f", need_grad=True).clone()
    dx = torch.full((B, n0), float("nan"), device=dev)
    net.flat_grads.fill_(float("nan"))
    net.backward(dy, slot="f", input_grad=dx)
    y_ref, g_ref, dx_ref = ref_forward_backward(net, x, dy)
    close(y, y_ref)
    for got, ref in zip(net.gradients, g_ref):
        close(got, ref)
    close(dx, dx_ref)
    assert torch.isfinite(net.flat_grads).all()          # alignment padding is zeroed, not NaN
    # the per-layer GEMM path gives the same numbers up to summation order
    fused = [t.clone() for t in net.gradients]
    sequential.FUSED_SMALL_MLP = False
    try:
        net._fused_ok = None
        y2 = net.forward(x, slot="g", need_grad=True).clone()
        net.backward(dy, slot="g")
        close(y, y2, tol=1e-5)
        for a, b in zip(fused, net.gradients):
            close(a, b, tol=1e-5)
    finally:
        sequential.FUSED_SMALL_MLP = True
        net._fused_ok = None


def test_fused_mlp_is_deterministic(dev):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1000, 17, generator=g).to(dev)
    dy = torch.randn(1000, 6, generator=g).to(dev)
    net = build(17, (64, 64, 6), ("tanh", "tanh", None), seed=2)
    outs = []
    for _ in range(3):
        net.forward(x, slot="f", need_grad=True)
        net.backward(dy, slot="f")
        outs.append(net.flat_grads.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_wide_networks_keep_the_gemm_path(dev):
    net = build(17, (100, 2), ("relu", None), seed=1)
    assert not net._fused_small_ok()
