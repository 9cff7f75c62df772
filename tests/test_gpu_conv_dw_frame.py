"""GPU parity: the per-frame conv weight gradient on the bf16 matrix cores (csrc/conv_dw_frame_x6.hip
through ops.conv_dw) vs float64 references and vs the fp32-MFMA GEMM path it replaces -- the same
accuracy class (three exact bf16 pieces per operand, six products, fp32 accumulation), bit-exact on
small-integer operands, refusal of shapes outside its limits (the GEMM path takes over)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from agents_amd import _lib, ops

pytestmark = pytest.mark.gpu


def _ref(x, dz, KH, KW, s):
    """float64 weight / bias gradient of a VALID NHWC conv via autograd on the CPU."""
    xd = x.double().permute(0, 3, 1, 2)
    Cin, Cout = x.shape[3], dz.shape[3]
    w = torch.zeros(Cout, Cin, KH, KW, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(xd, w, b, stride=s)
    y.backward(dz.double().permute(0, 3, 1, 2))
    return w.grad.permute(2, 3, 1, 0).contiguous(), b.grad       # HWIO


def _run(x, dz, KH, KW, s, x6):
    dev = x.device
    Cin, Cout = x.shape[3], dz.shape[3]
    old = ops.CONV_DW_X6
    ops.CONV_DW_X6 = x6
    ops._DW_X6_WS.clear()
    try:
        g = torch.full((KH, KW, Cin, Cout), float("nan"), device=dev)
        bg = torch.full((Cout,), float("nan"), device=dev)
        ops.conv_dw(x, dz.reshape(-1, Cout), (KH, KW, Cin, Cout), s, g, a_div=1.0, bias_grad=bg)
        torch.cuda.synchronize()
        return g, bg
    finally:
        ops.CONV_DW_X6 = old
        ops._DW_X6_WS.clear()


def _supported(shape, KH, KW, s, Cout):
    n, H, W, C = shape
    d = ops._dxf_desc(shape, (KH, KW, C, Cout), s)
    return int(_lib.load().aa_conv_dw_frame_x6_workspace_bytes(ctypes.byref(d))) > 0


SHAPES = [
    # n, H, W, Cin, K, stride, Cout
    (256, 20, 20, 32, 4, 2, 64),     # DQN conv2
    (256, 9, 9, 64, 3, 1, 64),       # DQN conv3
    (37, 11, 13, 16, 4, 1, 64),      # ragged group count, 16-channel input, 80 pixels
    (5, 12, 12, 64, 2, 2, 64),       # KW*Cin/16 = 8, 36 pixels, fewer frames than CUs
    (64, 16, 10, 32, 2, 3, 64),      # stride 3, non-square
    (1, 9, 9, 64, 3, 1, 64),         # one frame
]


@pytest.mark.parametrize("n,H,W,C,K,s,Cout", SHAPES)
def test_conv_dw_x6_vs_float64(dev, n, H, W, C, K, s, Cout):
    assert _supported((n, H, W, C), K, K, s, Cout)
    g = torch.Generator().manual_seed(n + H + C)
    x = torch.relu(torch.randn(n, H, W, C, generator=g)).to(dev)      # a ReLU output, as in the net
    OH, OW = (H - K) // s + 1, (W - K) // s + 1
    dz = torch.randn(n, OH, OW, Cout, generator=g).to(dev)
    got, gb = _run(x, dz, K, K, s, True)
    ref, rb = _ref(x.cpu(), dz.cpu(), K, K, s)
    scale = float(ref.abs().max())
    assert float((got.cpu().double() - ref).abs().max()) <= 2e-6 * scale
    assert float((gb.cpu().double() - rb).abs().max()) <= 2e-6 * float(rb.abs().max() + 1)
    # same accuracy class as the fp32-MFMA GEMM path it replaces
    old, ob = _run(x, dz, K, K, s, False)
    assert float((old.cpu().double() - ref).abs().max()) <= 2e-5 * scale
    assert float((got - old).abs().max()) <= 2e-5 * scale
    assert float((gb - ob).abs().max()) <= 2e-5 * float(rb.abs().max() + 1)


@pytest.mark.parametrize("n,H,W,C,K,s,Cout", SHAPES[:4])
def test_conv_dw_x6_exact_on_small_integers(dev, n, H, W, C, K, s, Cout):
    """Integer operands: every piece product and every partial sum is an integer below 2^24, so the
    six-product scheme, the fp32 accumulation and the slab sum must reproduce float64 bit for bit."""
    g = torch.Generator().manual_seed(3)
    x = torch.randint(-3, 4, (n, H, W, C), generator=g).float().to(dev)
    OH, OW = (H - K) // s + 1, (W - K) // s + 1
    dz = torch.randint(-2, 3, (n, OH, OW, Cout), generator=g).float().to(dev)
    got, gb = _run(x, dz, K, K, s, True)
    ref, rb = _ref(x.cpu(), dz.cpu(), K, K, s)
    assert float(ref.abs().max()) < 2 ** 24
    assert torch.equal(got.cpu().double(), ref)
    assert torch.equal(gb.cpu().double(), rb)


def test_conv_dw_x6_is_deterministic_and_leaves_no_state(dev):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(256, 20, 20, 32, generator=g).to(dev)
    dz = torch.randn(256, 9, 9, 64, generator=g).to(dev)
    a, ab = _run(x, dz, 4, 4, 2, True)
    dz2 = torch.randn(256, 9, 9, 64, generator=g).to(dev)
    _run(x, dz2, 4, 4, 2, True)               # other data through the same scratch
    b, bb = _run(x, dz, 4, 4, 2, True)
    assert torch.equal(a, b) and torch.equal(ab, bb)


def test_conv_dw_x6_refuses_what_it_cannot_do(dev):
    lib = _lib.load()
    for shape, K, s, Cout in [((8, 20, 20, 32), 4, 2, 32),      # 32 output channels
                              ((8, 20, 20, 24), 4, 2, 64),      # Cin not a power of two
                              ((8, 84, 84, 16), 8, 4, 64),      # 400 output pixels
                              ((8, 20, 20, 32), 3, 1, 64)]:     # KW*Cin/16 = 6: not 4 row groups
        assert not _supported(shape, K, K, s, Cout)
        n, H, W, C = shape
        OH, OW = (H - K) // s + 1, (W - K) // s + 1
        x = torch.randn(n, H, W, C, device=dev)
        dz = torch.randn(n, OH, OW, Cout, device=dev)
        got, gb = _run(x, dz, K, K, s, True)      # falls back to the GEMM path: still correct
        ref, rb = _ref(x.cpu(), dz.cpu(), K, K, s)
        assert float((got.cpu().double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    # the C entry itself says no
    d = ops._dxf_desc((8, 20, 20, 32), (4, 4, 32, 32), 2)
    z = torch.zeros(1 << 16, device=dev)
    assert lib.aa_conv_dw_frame_x6(ctypes.byref(d), z.data_ptr(), z.data_ptr(), None, z.data_ptr(),
                                   z.numel() * 4, _lib.stream_ptr()) != 0
    # too small a workspace / missing pointers
    d = ops._dxf_desc((8, 9, 9, 64), (3, 3, 64, 64), 1, dz=z)
    assert lib.aa_conv_dw_frame_x6(ctypes.byref(d), z.data_ptr(), z.data_ptr(), None, z.data_ptr(),
                                   1024, _lib.stream_ptr()) != 0
    assert lib.aa_conv_dw_frame_x6(ctypes.byref(d), None, z.data_ptr(), None, z.data_ptr(),
                                   z.numel() * 4, _lib.stream_ptr()) != 0
    torch.cuda.synchronize()


def test_sequential_backward_with_and_without_dw_x6(dev):
    """The Atari Q-network's gradients with the per-frame weight-gradient kernel vs the GEMM path."""
    from agents_amd.networks import sequential, layers as L
    from agents_amd.specs import tensor_spec
    spec = tensor_spec.TensorSpec((84, 84, 4), torch.uint8)
    g = torch.Generator().manual_seed(5)
    x = torch.randint(0, 256, (64, 84, 84, 4), dtype=torch.uint8, generator=g).to(dev)
    dq = torch.randn(64, 6, generator=g).to(dev)
    res = {}
    for x6 in (True, False):
        old = ops.CONV_DW_X6
        ops.CONV_DW_X6 = x6
        ops._DW_X6_WS.clear()
        try:
            net = sequential.Sequential([
                L.Rescale(255.0), L.Conv2D(32, 8, 4, activation="relu"),
                L.Conv2D(64, 4, 2, activation="relu"), L.Conv2D(64, 3, 1, activation="relu"),
                L.Flatten(), L.Dense(512, activation="relu"), L.Dense(6)], input_spec=spec, seed=3)
            net.create_variables(spec, device=dev)
            net.forward(x, slot="t", need_grad=True)
            net.backward(dq, slot="t", side_stream=ops.new_side_stream(dev))
            torch.cuda.synchronize()
            res[x6] = net.flat_grads.clone()
        finally:
            ops.CONV_DW_X6 = old
            ops._DW_X6_WS.clear()
    scale = float(res[False].abs().max())
    assert float((res[True] - res[False]).abs().max()) <= 2e-5 * scale
    assert not torch.equal(res[True], res[False])     # the other kernel did run
