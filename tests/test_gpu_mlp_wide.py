"""csrc/mlp_wide.hip: whole <=256-wide MLPs forward in one launch, backward in two, several networks
per launch -- against a float64 torch reference of the same stack (keras Dense arithmetic:
y = act(x W + b); networks/critic_network.py:150-170, actor_distribution_network.py of the
reference build these stacks) and against the per-layer GEMM path of the same Sequential."""
import numpy as np
import pytest
import torch

from agents_amd.networks import layers as L
from agents_amd.networks import sequential
from agents_amd.specs import tensor_spec

pytestmark = pytest.mark.gpu


@pytest.fixture
def dev():
    return torch.device("cuda", 0)


def _net(d0, widths, acts, seed, dev):
    net = sequential.Sequential([L.Dense(w, activation=a) for w, a in zip(widths, acts)], seed=seed)
    net.create_variables(tensor_spec.TensorSpec((d0,), torch.float32, "x"), device=dev)
    return net


def _ref(net, x, dout, acts, gpu_ys=None):
    """float64 forward / backward of the stack with the network's own parameters.  With gpu_ys the
    ReLU masks are the kernel's own (y > 0 of its float32 outputs): a pre-activation within
    rounding of zero may land on the other side in float32, which moves a whole column of a weight
    gradient -- that is a property of the input, not an error (oracle/arbiter.py); such units must
    be within 1e-5 of the boundary."""
    ws = [v.detach().double().cpu().requires_grad_(True) for v in net.variables]
    h = x.double().cpu().requires_grad_(True)
    cur = h
    ys = []
    for i, a in enumerate(acts):
        cur = cur @ ws[2 * i] + ws[2 * i + 1]
        if a == "relu":
            if gpu_ys is None:
                cur = torch.relu(cur)
            else:
                mask = (gpu_ys[i].detach().cpu() > 0)
                flips = mask != (cur.detach() > 0)
                assert float(cur.detach()[flips].abs().max() if flips.any() else 0.0) < 1e-5
                cur = cur * mask.double()
        elif a == "tanh":
            cur = torch.tanh(cur)
        ys.append(cur)
    cur.backward(dout.double().cpu())
    return ys, [w.grad for w in ws], h.grad


CASES = [
    # d0, widths, acts, B, split (None = single input tensor)
    (393, (256, 256, 1), ("relu", "relu", None), 256, 376),     # SAC critic, [obs | action]
    (376, (256, 256, 34), ("relu", "relu", None), 256, None),   # SAC actor trunk + projection
    (23, (256, 256, 1), ("relu", "relu", None), 64, 17),        # HalfCheetah critic
    (17, (256, 256, 12), ("tanh", "relu", None), 7, None),      # ragged batch, tanh
    (5, (33, 200, 6), ("relu", "tanh", "tanh"), 130, 2),        # odd widths: scalar column path
    (1000, (128,), (None,), 9, None),                           # one layer, wide input
    (70, (96, 80, 72, 100), ("relu",) * 4, 33, None),           # four layers
]


@pytest.mark.parametrize("d0,widths,acts,B,split", CASES)
@pytest.mark.parametrize("n_nets", [1, 2, 4])
def test_wide_mlp_matches_float64_reference(dev, d0, widths, acts, B, split, n_nets):
    g = torch.Generator().manual_seed(B * 7 + d0)
    nets = [_net(d0, widths, acts, seed=11 + i, dev=dev) for i in range(n_nets)]
    assert nets[0].wide_ok(B)
    xs = [torch.randn(B, d0, generator=g) for _ in range(n_nets)]
    douts = [torch.randn(B, widths[-1], generator=g) for _ in range(n_nets)]
    xd = [x.to(dev) for x in xs]
    if split is None:
        outs = sequential.forward_wide(nets, xd, slot="t", need_grad=True)
    else:
        # the two halves live in tensors of their own, with padded row strides
        a = [torch.zeros(B, split + 3, device=dev) for _ in xs]
        b = [torch.zeros(B, d0 - split + 5, device=dev) for _ in xs]
        for i, x in enumerate(xd):
            a[i][:, :split] = x[:, :split]
            b[i][:, :d0 - split] = x[:, split:]
        outs = sequential.forward_wide(nets, [t[:, :split] for t in a], slot="t", need_grad=True,
                                       x2s=[t[:, :d0 - split] for t in b])
    lo, hi = (split, d0) if split is not None else (0, d0)
    dxs = [torch.full((B, d0), float("nan"), device=dev) for _ in nets]
    for net in nets:
        net.flat_grads.fill_(float("nan"))
    sequential.backward_wide(nets, [d.to(dev) for d in douts], slot="t", input_grads=dxs,
                             input_grad_cols=(lo, hi))
    torch.cuda.synchronize()
    for i, net in enumerate(nets):
        slot = net._slots[("t", B)]
        ys, wg, xg = _ref(net, xs[i], douts[i], acts, gpu_ys=slot.ys)
        for y, yr in zip(slot.ys, ys):
            np.testing.assert_allclose(y.cpu().numpy(), yr.detach().numpy(), rtol=2e-5, atol=2e-5)
        assert outs[i].data_ptr() == slot.ys[-1].data_ptr()
        for v, gr in zip(net.gradients, wg):
            scale = float(gr.abs().max()) + 1e-30
            np.testing.assert_allclose(v.cpu().numpy(), gr.numpy(), rtol=1e-4, atol=2e-6 * scale)
        got = dxs[i][:, lo:hi].cpu().numpy()
        scale = float(xg.abs().max()) + 1e-30
        np.testing.assert_allclose(got, xg[:, lo:hi].numpy(), rtol=1e-4, atol=2e-6 * scale)
        if lo > 0:      # columns outside the requested range are not written
            assert torch.isnan(dxs[i][:, :lo]).all()


def test_wide_path_of_sequential_equals_the_layer_path(dev, monkeypatch):
    """Sequential.forward / backward take the wide path on their own; against the same network run
    layer by layer through the GEMM kernels (AA_FUSED_WIDE_MLP=0's path): same results up to the
    summation order, and the chain-only backward (param_grads=False) leaves the gradients alone."""
    B, d0 = 256, 393
    net = _net(d0, (256, 256, 1), ("relu", "relu", None), seed=5, dev=dev)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, d0, generator=g).to(dev)
    dout = torch.randn(B, 1, generator=g).to(dev)
    dx_w = torch.zeros(B, d0, device=dev)
    y_w = net.forward(x, slot="a", need_grad=True).clone()
    assert net._slots[("a", B)].wide_in is not None
    net.backward(dout, slot="a", input_grad=dx_w)
    g_w = [v.clone() for v in net.gradients]
    net.flat_grads.fill_(7.0)
    net.backward(dout, slot="a", param_grads=False, input_grad=dx_w, input_grad_cols=(376, 393))
    assert bool((net.flat_grads == 7.0).all())
    monkeypatch.setattr(sequential, "FUSED_WIDE_MLP", False)
    dx_l = torch.zeros(B, d0, device=dev)
    y_l = net.forward(x, slot="b", need_grad=True).clone()
    assert net._slots[("b", B)].wide_in is None
    net.backward(dout, slot="b", input_grad=dx_l)
    torch.cuda.synchronize()
    np.testing.assert_allclose(y_w.cpu().numpy(), y_l.cpu().numpy(), rtol=2e-5, atol=2e-5)
    for a, b in zip(g_w, net.gradients):      # (the flat buffer also holds alignment padding)
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4,
                                   atol=2e-6 * float(b.abs().max()))
    np.testing.assert_allclose(dx_w.cpu().numpy(), dx_l.cpu().numpy(), rtol=1e-4,
                               atol=2e-6 * float(dx_l.abs().max()))


def test_wide_launches_are_deterministic(dev):
    """Fixed summation orders everywhere: two runs are bit-identical (HIP-graph replays equal
    eager steps)."""
    B, d0 = 200, 393
    net = _net(d0, (256, 256, 1), ("relu", "relu", None), seed=9, dev=dev)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, d0, generator=g).to(dev)
    dout = torch.randn(B, 1, generator=g).to(dev)
    res = []
    for _ in range(2):
        dx = torch.zeros(B, d0, device=dev)
        y = net.forward(x, slot="d", need_grad=True).clone()
        net.backward(dout, slot="d", input_grad=dx)
        res.append((y, net.flat_grads.clone(), dx))
    for a, b in zip(*res):
        assert torch.equal(a, b)
