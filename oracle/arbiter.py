"""float64 arbiter for gradient parity of ReLU networks (TEST INFRASTRUCTURE; oracle/__init__.py).

Two fp32 implementations of the same network (the HIP kernels and the torch-CPU oracle) differ in
two ways that a plain "gradient A vs gradient B" tolerance cannot tell apart:

  rounding   every sum is rounded differently: relative 1e-7 .. 1e-6 on a gradient tensor, at
             every step -- a systematic defect would show here;
  branches   a unit whose exact pre-activation is within rounding of zero lands on different sides
             of the ReLU.  The gradient of a weight tensor is a sum of ~1e4 .. 1e5 terms of random
             sign (norm ~ sqrt(N) x one term), so ONE term that appears / vanishes moves the
             tensor by 1/sqrt(N) ~ 1e-3 of its norm, not 1/N.  Rare (P(|z| < ulp) x 5.5 M units
             ~ 0.4 per step on the Atari net) and legitimate: both results are exact gradients of
             the piecewise-linear function on the two sides of a kink.

This module evaluates the network in float64 (oracle/nets.py: forward_branch) three times per
check: on its NATURAL branch (exact activation pattern), and on the branch each implementation
took (its own activation pattern imposed as 0/1 masks).  Against the float64 gradient OF ITS OWN
BRANCH an implementation is held to rounding at every step and every tensor; its branch may
differ from the natural one only at units whose exact pre-activation is numerically zero.  The
upstream gradient dL/dq is taken as given from each implementation (it is checked separately on
the implementation's own q values: the TD error q - y is a difference of nearly equal numbers, and
feeding two implementations' dq through one comparison would only measure that cancellation).
"""
import numpy as np
import torch

from oracle import nets


def _rel_l2(a, b):
    return float((a - b).norm() / max(float(b.norm()), 1e-300))


def relu_masks(activations):
    """0/1 masks from post-ReLU activations (None entries stay None)."""
    return [None if a is None else (torch.as_tensor(a) > 0) for a in activations]


def branch_gradients(layers, params, x, masks, dq):
    """float64 gradient of sum(q * dq) wrt `params` on the branch `masks` (None = natural).
    Returns (grads, q, pre-activations)."""
    p64 = [torch.as_tensor(p).detach().double().requires_grad_(True) for p in params]
    q, pre = nets.forward_branch(layers, p64, x, masks=masks, dtype=torch.float64)
    grads = torch.autograd.grad((q * torch.as_tensor(dq).double()).sum(), p64)
    return [g.detach() for g in grads], q.detach(), [z.detach() for z in pre]


def check_branch(natural_pre, masks, flip_tol):
    """Units where `masks` disagrees with the exact activation pattern.  Returns
    (number of flips, largest |z_exact| / max|z_exact of the layer| among them); every flipped
    unit must sit within `flip_tol` of zero on that scale, else the implementation took a branch
    the exact function is nowhere near."""
    n_flip, worst = 0, 0.0
    for z, m in zip(natural_pre, masks):
        if m is None:
            continue
        diff = (z > 0) != m.to(torch.bool)
        k = int(diff.sum())
        if k:
            n_flip += k
            worst = max(worst, float(z[diff].abs().max() / z.abs().max()))
    assert worst <= flip_tol, \
        f"{n_flip} activation(s) differ from the exact pattern, one at |z| = {worst:.2e} of the " \
        f"layer's scale (> {flip_tol:.0e}): not a boundary flip"
    return n_flip, worst


def gradient_errors(layers, params, x, impl_grads, impl_masks, impl_dq):
    """Relative L2 error, per parameter tensor, of an implementation's gradients against the
    float64 gradient of the branch it took, given its own upstream gradient."""
    ref, _, _ = branch_gradients(layers, params, x, impl_masks, impl_dq)
    return [_rel_l2(torch.as_tensor(g).detach().double().reshape(r.shape), r)
            for g, r in zip(impl_grads, ref)]


def fp32_reference_branch(layers, params, x):
    """The torch-CPU fp32 oracle's own activation pattern (what its autograd differentiates)."""
    with torch.no_grad():
        _, pre = nets.forward_branch(layers, [torch.as_tensor(p).detach() for p in params], x,
                                     dtype=torch.float32)
    return [None if l_act != "relu" else (z > 0)
            for z, l_act in zip(pre, [l["act"] for l in layers
                                      if l["kind"] in ("conv", "dense")])]


def natural(layers, params, x):
    """Exact (float64) q values and pre-activations."""
    with torch.no_grad():
        p64 = [torch.as_tensor(p).detach().double() for p in params]
        q, pre = nets.forward_branch(layers, p64, x, dtype=torch.float64)
    return q, pre
