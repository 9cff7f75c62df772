"""Gather-form conv input gradient (csrc/conv_dx_frame.hip) against float64 autograd and against
the GEMM + col2im path."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from agents_amd import ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda", 0)


def rnd(rng, *shape):
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32))


def close(got, ref, tol=2e-5):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape
    scale = max(ref.abs().max().item(), 1e-30)
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


def actgrad(y, act):
    if act == "relu":
        return (y > 0).double()
    if act == "tanh":
        return 1.0 - y.double() ** 2
    return torch.ones_like(y, dtype=torch.float64)


CASES = [  # (B, H, W, Cin, KH, KW, stride, Cout)
    (8, 20, 20, 32, 4, 4, 2, 64),     # Atari conv2: four sub-pixel classes of 100 pixels
    (8, 9, 9, 64, 3, 3, 1, 64),       # Atari conv3: one class, 81 pixels
    (3, 21, 22, 16, 4, 4, 2, 32),     # rows / columns the VALID conv never reads get zero gradient
    (2, 17, 14, 16, 5, 3, 3, 32),     # stride 3, 5x3 kernel: classes with 2 and 1 taps per axis
    (700, 9, 9, 64, 3, 3, 1, 64),     # more frames than workgroups
    (1, 3, 3, 16, 3, 3, 1, 32),       # single output pixel
]


@pytest.fixture(params=[True, False], ids=["bf16x6", "fp32mfma"])
def x6(request):
    """Both implementations: csrc/conv_dx_frame_x6.hip (default where the shape fits) and
    csrc/conv_dx_frame.hip."""
    prev = ops.CONV_DX_X6
    ops.CONV_DX_X6 = request.param
    ops._DXF_X6_WS.clear()
    yield request.param
    ops.CONV_DX_X6 = prev
    ops._DXF_X6_WS.clear()


def test_atari_shapes_take_the_bf16_kernel(dev):
    import ctypes
    from agents_amd import _lib
    lib = _lib.load()
    for x_shape, w_shape, s in (((256, 20, 20, 32), (4, 4, 32, 64), 2),
                                ((256, 9, 9, 64), (3, 3, 64, 64), 1)):
        d = ops._dxf_desc(x_shape, w_shape, s)
        assert lib.aa_conv_dx_frame_x6_workspace_bytes(ctypes.byref(d)) > 0
    assert ops.CONV_DX_X6


def test_conv_dx_x6_exact_on_bf16_representable_operands(dev):
    """Small-integer operands are single bf16 pieces and every partial sum is exact in fp32: the
    kernel must reproduce float64 bit for bit (any fragment / K-order mismatch shows)."""
    rng = np.random.default_rng(3)
    for (B, H, W, C, KH, KW, s, Fo) in CASES[:4]:
        w = torch.from_numpy(rng.integers(-3, 4, (KH, KW, C, Fo)).astype(np.float32))
        OH, OW = ops.conv_out_hw(H, W, KH, KW, s)
        dz = torch.from_numpy(rng.integers(-4, 5, (B, OH, OW, Fo)).astype(np.float32))
        out = torch.full((B, H, W, C), float("nan"), device=dev)
        ops.conv_dx_frame(dz.to(dev).view(-1, Fo), w.to(dev), (B, H, W, C), s, out)
        xd = torch.zeros(B, H, W, C, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(xd.permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), None, stride=s)
        g, = torch.autograd.grad(y, xd, dz.double().permute(0, 3, 1, 2))
        assert torch.equal(out.cpu().double(), g)


@pytest.mark.parametrize("cfg", CASES)
@pytest.mark.parametrize("act", [None, "relu", "tanh"])
def test_conv_dx_frame(dev, cfg, act, x6):
    B, H, W, C, KH, KW, s, Fo = cfg
    rng = np.random.default_rng(sum(cfg))
    x = torch.tanh(rnd(rng, B, H, W, C))           # a plausible activation output (mask source)
    w = rnd(rng, KH, KW, C, Fo) * 0.2
    OH, OW = ops.conv_out_hw(H, W, KH, KW, s)
    dz = rnd(rng, B, OH, OW, Fo)
    assert ops.conv_dx_frame_supported((B, H, W, C), w.shape, s)
    out = torch.full((B, H, W, C), float("nan"), device=dev)
    ops.conv_dx_frame(dz.to(dev).view(-1, Fo), w.to(dev), (B, H, W, C), s, out,
                      mask_src=x.to(dev) if act else None, mask_act=act)
    xd = x.double().requires_grad_(True)
    y = F.conv2d(xd.permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), None, stride=s)
    g, = torch.autograd.grad(y, xd, dz.double().permute(0, 3, 1, 2))
    ref = g * actgrad(x, act)
    close(out, ref)
    # the GEMM + col2im path agrees
    if B <= 16:
        old = torch.empty(B, H, W, C, device=dev)
        dcol = torch.empty(B * OH * OW * KH * KW * C, device=dev)
        saved = ops.CONV_DX_FRAME
        ops.CONV_DX_FRAME = False
        try:
            ops.conv_dx(dz.to(dev).view(-1, Fo), w.to(dev), (B, H, W, C), s, dcol, old,
                        mask_src=x.to(dev) if act else None, mask_act=act)
        finally:
            ops.CONV_DX_FRAME = saved
        close(out, old.cpu(), tol=5e-5)


def test_conv_dx_frame_deterministic_and_dispatch(dev):
    rng = np.random.default_rng(2)
    x_shape = (6, 20, 20, 32)
    w = (rnd(rng, 4, 4, 32, 64) * 0.1).to(dev)
    dz = rnd(rng, 6 * 81, 64).to(dev)
    outs = []
    for _ in range(2):
        out = torch.empty(x_shape, device=dev)
        dcol = torch.empty(1, device=dev)      # not needed by the frame kernel
        with pytest.raises(ValueError):
            ops.conv_dx(dz, w, x_shape, 2, dcol, out)          # the size check stays
        dcol = torch.empty(6 * 81 * 512, device=dev)
        ops.conv_dx(dz, w, x_shape, 2, dcol, out)              # dispatches to the frame kernel
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1])


def test_conv_dx_frame_unsupported(dev):
    assert not ops.conv_dx_frame_supported((4, 84, 84, 4), (8, 8, 4, 32), 4)     # Cin % 16
    assert not ops.conv_dx_frame_supported((4, 40, 40, 32), (3, 3, 32, 64), 1)   # 1600 pixels
    assert not ops.conv_dx_frame_supported((4, 9, 9, 64), (3, 3, 64, 48), 1)     # Cout % 32
