"""CPU: the float64 arbiter (oracle/arbiter.py) behaves as tests/test_gpu_bench_config.py relies on:
it holds a correct fp32 implementation to rounding on its own branch, it explains a boundary flip
(a gradient difference of 1e-3 .. 1e-2 between two correct implementations), and it rejects an
implementation that drops work or takes a branch the exact function is nowhere near."""
import numpy as np
import pytest
import torch

from oracle import arbiter, nets

LAYERS = [{"kind": "rescale", "div": 255.0},
          {"kind": "conv", "filters": 8, "kernel": (4, 4), "stride": 2, "act": "relu"},
          {"kind": "conv", "filters": 8, "kernel": (3, 3), "stride": 1, "act": "relu"},
          {"kind": "flatten"},
          {"kind": "dense", "units": 16, "act": "relu"},
          {"kind": "dense", "units": 3, "act": None}]
SHAPE = (12, 12, 2)


def _setup(B=16, seed=0):
    params = nets.init_params(LAYERS, SHAPE, seed=seed)
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(0, 256, (B,) + SHAPE, dtype=torch.uint8, generator=g)
    dq = torch.randn(B, 3, generator=g) / B
    return params, x, dq


def _fp32_impl(params, x, dq, masks=None):
    """An fp32 implementation: autograd on the fp32 forward (optionally on an imposed branch)."""
    p = [t.clone().requires_grad_(True) for t in params]
    q, pre = nets.forward_branch(LAYERS, p, x, masks=masks, dtype=torch.float32)
    grads = torch.autograd.grad((q * dq).sum(), p)
    acts = [None if l["act"] != "relu" else (z.detach() > 0)
            for z, l in zip(pre, [l for l in LAYERS if l["kind"] in ("conv", "dense")])]
    return [g.detach() for g in grads], acts if masks is None else masks


def test_correct_implementation_is_at_rounding_on_its_branch():
    params, x, dq = _setup()
    grads, masks = _fp32_impl(params, x, dq)
    errs = arbiter.gradient_errors(LAYERS, params, x, grads, masks, dq)
    assert max(errs) < 5e-6, errs
    _, pre = arbiter.natural(LAYERS, params, x)
    n, worst = arbiter.check_branch(pre, masks, 1e-5)
    assert n <= 2 and worst <= 1e-5
    assert arbiter.fp32_reference_branch(LAYERS, params, x)[0].equal(masks[0])


def test_boundary_flip_is_explained_not_hidden():
    """Flip the unit closest to zero: the plain gradient comparison jumps by orders of magnitude,
    the arbiter still sees rounding-level error and reports one boundary flip."""
    params, x, dq = _setup(seed=3)
    # move a bias so that one first-layer unit sits exactly on the kink
    _, pre = arbiter.natural(LAYERS, params, x)
    z = pre[0]
    idx = np.unravel_index(int(z.abs().argmin()), z.shape)
    params[1][idx[-1]] -= z[idx].float()
    _, pre = arbiter.natural(LAYERS, params, x)
    assert float(pre[0][idx].abs()) < 1e-6
    g_a, m_a = _fp32_impl(params, x, dq)
    m_b = [None if m is None else m.clone() for m in m_a]
    m_b[0][idx] = ~m_b[0][idx]                       # the other side of the kink
    g_b, _ = _fp32_impl(params, x, dq, masks=m_b)
    plain = max(float((a - b).norm() / b.norm()) for a, b in zip(g_a, g_b))
    e_a = arbiter.gradient_errors(LAYERS, params, x, g_a, m_a, dq)
    e_b = arbiter.gradient_errors(LAYERS, params, x, g_b, m_b, dq)
    assert max(e_a) < 5e-6 and max(e_b) < 5e-6
    assert plain > 50 * max(max(e_a), max(e_b)), (plain, e_a, e_b)
    na, _ = arbiter.check_branch(pre, m_a, 1e-5)
    nb, _ = arbiter.check_branch(pre, m_b, 1e-5)
    assert abs(na - nb) == 1


def test_defects_are_rejected():
    params, x, dq = _setup(seed=5)
    grads, masks = _fp32_impl(params, x, dq)
    bad = [g.clone() for g in grads]
    bad[2][:, :, :, :2] = 0                          # a dropped tile of conv2's weight gradient
    errs = arbiter.gradient_errors(LAYERS, params, x, bad, masks, dq)
    assert errs[2] > 1e-2 and max(errs[:2]) < 5e-6
    _, pre = arbiter.natural(LAYERS, params, x)
    wrong = [None if m is None else m.clone() for m in masks]
    far = np.unravel_index(int(pre[1].abs().argmax()), pre[1].shape)
    wrong[1][far] = ~wrong[1][far]                   # a unit far from the kink on the wrong side
    with pytest.raises(AssertionError, match="not a boundary flip"):
        arbiter.check_branch(pre, wrong, 1e-5)
