"""GPU parity of the fp32 MFMA GEMM family (csrc/gemm.hip) against a plain torch fp32/fp64 CPU
reference of the same op.  Tolerance: |err| <= 2e-5 * max|ref| (fp32 accumulation-order noise;
north star: 1e-5 relative on losses).  Covers every loader mode, every tile config, split-K,
ragged edges, unaligned (scalar-path) operands, strided batches and the fused epilogues."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from agents_amd import _lib, ops

pytestmark = pytest.mark.gpu
TOL = 2e-5


def close(got, ref, tol=TOL):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert got.shape == ref.shape
    scale = max(ref.abs().max().item(), 1e-30)
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, f"max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.3e})"


def rnd(rng, *shape):
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32))


def act_ref(x, act):
    return {None: x, "relu": torch.relu(x), "tanh": torch.tanh(x)}[act]


def actgrad_ref(y, act):
    return {None: torch.ones_like(y), "relu": (y > 0).to(y.dtype), "tanh": 1 - y * y}[act]


SHAPES = [(256, 512, 3136), (256, 6, 512), (64, 100, 4), (1, 2, 100), (37, 45, 53),
          (300, 33, 64), (256, 64, 17), (128, 128, 32), (2048, 64, 64), (5, 7, 3)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("act", [None, "relu", "tanh"])
def test_dense_forward(dev, M, N, K, act):
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    x, w, b = rnd(rng, M, K), rnd(rng, K, N) * 0.1, rnd(rng, N)
    out = torch.empty(M, N, device=dev)
    ops.dense_forward(x.to(dev), w.to(dev), b.to(dev), act, out)
    close(out, act_ref(x.double() @ w.double() + b.double(), act))


ALL_CFGS = [1, 2, 3, 4, 5, 6, 7]  # 128x64, 128x32, 64x64, 128x128, 64x32(k2), 32x64(k2), 32x32(k4)


@pytest.mark.parametrize("cfg", ALL_CFGS)
@pytest.mark.parametrize("splits", [1, 3])
def test_dense_forward_forced_configs(dev, cfg, splits):
    rng = np.random.default_rng(cfg * 10 + splits)
    M, N, K = 200, 70, 300
    x, w, b = rnd(rng, M, K), rnd(rng, K, N) * 0.1, rnd(rng, N)
    out = torch.empty(M, N, device=dev)
    ops.dense_forward(x.to(dev), w.to(dev), b.to(dev), "relu", out, force_cfg=cfg,
                      force_splits=splits)
    close(out, torch.relu(x.double() @ w.double() + b.double()))


def test_dense_forward_strided_rows(dev):
    """A operand = experience.observation[:, 0] of a [B, T, K] tensor (row pitch T*K)."""
    rng = np.random.default_rng(5)
    B, T, K, N = 64, 2, 4, 100
    x3 = rnd(rng, B, T, K).to(dev)
    w, b = rnd(rng, K, N), rnd(rng, N)
    out = torch.empty(B, N, device=dev)
    ops.dense_forward(x3[:, 1], w.to(dev), b.to(dev), None, out)
    close(out, x3[:, 1].cpu().double() @ w.double() + b.double())


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_dense_dw(dev, M, N, K):
    rng = np.random.default_rng(M + N + K)
    x, dz = rnd(rng, M, K), rnd(rng, M, N)
    out = torch.empty(K, N, device=dev)
    ops.dense_dw(x.to(dev), dz.to(dev), out)
    close(out, x.double().T @ dz.double())


@pytest.mark.parametrize("cfg", ALL_CFGS)
@pytest.mark.parametrize("splits", [1, 2, 5])
def test_dense_dw_fused_bias_grad(dev, cfg, splits):
    """bias_grad = column sums of dz, produced by the dW GEMM (all tiles, with and without
    split-K; ragged M/N/K so edge tiles and the zero-padded K tail are covered)."""
    rng = np.random.default_rng(cfg * 100 + splits)
    M, N, K = 333, 70, 150  # x[M,K], dz[M,N]
    x, dz = rnd(rng, M, K), rnd(rng, M, N)
    out = torch.empty(K, N, device=dev)
    bg = torch.full((N,), float("nan"), device=dev)
    ops.dense_dw(x.to(dev), dz.to(dev), out, force_cfg=cfg, force_splits=splits, bias_grad=bg)
    close(out, x.double().T @ dz.double())
    close(bg, dz.double().sum(0), tol=5e-6)


@pytest.mark.parametrize("cfg", ALL_CFGS)
def test_dense_dx_forced_configs(dev, cfg):
    rng = np.random.default_rng(cfg)
    M, N, K = 130, 40, 210
    dz, w = rnd(rng, M, N), rnd(rng, K, N) * 0.1
    y = torch.tanh(rnd(rng, M, K))
    out = torch.empty(M, K, device=dev)
    ops.dense_dx(dz.to(dev), w.to(dev), out, mask_src=y.to(dev), mask_act="tanh", force_cfg=cfg)
    close(out, (dz.double() @ w.double().T) * actgrad_ref(y.double(), "tanh"))


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("mask_act", [None, "relu", "tanh"])
def test_dense_dx(dev, M, N, K, mask_act):
    rng = np.random.default_rng(M * 3 + N + K * 5)
    dz, w = rnd(rng, M, N), rnd(rng, K, N) * 0.1
    y = torch.tanh(rnd(rng, M, K))
    out = torch.empty(M, K, device=dev)
    ops.dense_dx(dz.to(dev), w.to(dev), out, mask_src=y.to(dev) if mask_act else None,
                 mask_act=mask_act)
    close(out, (dz.double() @ w.double().T) * actgrad_ref(y.double(), mask_act))


CONVS = [  # (B, H, W, C, KH, KW, stride, F, dtype)
    (4, 84, 84, 4, 8, 8, 4, 32, torch.uint8),      # Atari conv1
    (3, 20, 20, 32, 4, 4, 2, 64, torch.float32),   # Atari conv2
    (5, 9, 9, 64, 3, 3, 1, 64, torch.float32),     # Atari conv3
    (2, 12, 10, 8, 3, 2, 1, 20, torch.float32),    # ragged N, rectangular kernel
    (2, 36, 36, 4, 8, 8, 4, 32, torch.uint8),
    (7, 11, 11, 4, 5, 5, 2, 16, torch.float32),
]


def conv_ref(x, w, b, stride, div):
    xf = x.double() / div if x.dtype == torch.uint8 else x.double()
    y = F.conv2d(xf.permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), b.double(),
                 stride=stride)
    return y.permute(0, 2, 3, 1).contiguous()


def make_conv(rng, cfg):
    B, H, W, C, KH, KW, s, Fo, dt = cfg
    if dt == torch.uint8:
        x = torch.from_numpy(rng.integers(0, 256, size=(B, H, W, C), dtype=np.uint8))
    else:
        x = rnd(rng, B, H, W, C)
    w = rnd(rng, KH, KW, C, Fo) * 0.1
    b = rnd(rng, Fo)
    return x, w, b


@pytest.mark.parametrize("cfg", CONVS)
def test_conv_forward(dev, cfg):
    rng = np.random.default_rng(sum(cfg[:8]))
    x, w, b = make_conv(rng, cfg)
    B, H, W, C, KH, KW, s, Fo, dt = cfg
    OH, OW = ops.conv_out_hw(H, W, KH, KW, s)
    out = torch.empty(B, OH, OW, Fo, device=dev)
    ops.conv_forward(x.to(dev), w.to(dev), b.to(dev), s, "relu", out, a_div=255.0)
    close(out, torch.relu(conv_ref(x, w, b, s, 255.0)))


def test_conv_forward_strided_batch(dev):
    """x = observation[:, 1] of a [B, T, H, W, C] replay sample (image pitch T*H*W*C)."""
    rng = np.random.default_rng(11)
    x5 = torch.from_numpy(rng.integers(0, 256, size=(6, 2, 84, 84, 4), dtype=np.uint8)).to(dev)
    w, b = rnd(rng, 8, 8, 4, 32) * 0.1, rnd(rng, 32)
    out = torch.empty(6, 20, 20, 32, device=dev)
    ops.conv_forward(x5[:, 1], w.to(dev), b.to(dev), 4, None, out, a_div=255.0)
    close(out, conv_ref(x5[:, 1].cpu(), w, b, 4, 255.0))


@pytest.mark.parametrize("cfg", CONVS[:3])
@pytest.mark.parametrize("tile", ALL_CFGS)
def test_conv_forward_forced_configs(dev, cfg, tile):
    rng = np.random.default_rng(sum(cfg[:8]) + tile)
    x, w, b = make_conv(rng, cfg)
    B, H, W, C, KH, KW, s, Fo, dt = cfg
    OH, OW = ops.conv_out_hw(H, W, KH, KW, s)
    out = torch.empty(B, OH, OW, Fo, device=dev)
    ops.conv_forward(x.to(dev), w.to(dev), b.to(dev), s, "relu", out, a_div=255.0,
                     force_cfg=tile, force_splits=2)
    close(out, torch.relu(conv_ref(x, w, b, s, 255.0)))


def test_conv_u8_division_is_exact(dev):
    """(float)u8 / 255 inside the conv1 loader must be the IEEE quotient (the reference's
    tf.cast(obs, float32) / 255.): a 1x1 'conv' with identity weights returns it unchanged."""
    x = torch.arange(256, dtype=torch.uint8).repeat(16).reshape(1, 16, 16, 16).contiguous()
    w = torch.eye(16).reshape(1, 1, 16, 16).contiguous()
    for div in (255.0, 3.0, 1.0, 127.5):
        out = torch.empty(1, 16, 16, 16, device=dev)
        ops.conv_forward(x.to(dev), w.to(dev), None, 1, None, out, a_div=div)
        want = x.float() / np.float32(div)
        assert torch.equal(out.cpu(), want), div


@pytest.mark.parametrize("cfg", CONVS)
@pytest.mark.parametrize("tile,splits", [(0, 0), (1, 4), (5, 1), (7, 3), (4, 2)])
def test_conv_dw(dev, cfg, tile, splits):
    rng = np.random.default_rng(sum(cfg[:8]) + 1)
    x, w, b = make_conv(rng, cfg)
    B, H, W, C, KH, KW, s, Fo, dt = cfg
    OH, OW = ops.conv_out_hw(H, W, KH, KW, s)
    dz = rnd(rng, B, OH, OW, Fo)
    out = torch.empty(KH, KW, C, Fo, device=dev)
    bg = torch.full((Fo,), float("nan"), device=dev)
    ops.conv_dw(x.to(dev), dz.to(dev).view(-1, Fo), (KH, KW, C, Fo), s, out, a_div=255.0,
                force_cfg=tile, force_splits=splits, bias_grad=bg)
    close(bg, dz.double().sum((0, 1, 2)), tol=5e-6)
    xf = (x.double() / 255.0 if dt == torch.uint8 else x.double()).requires_grad_(False)
    wd = w.double().requires_grad_(True)
    y = F.conv2d(xf.permute(0, 3, 1, 2), wd.permute(3, 2, 0, 1), None, stride=s)
    g, = torch.autograd.grad(y, wd, dz.double().permute(0, 3, 1, 2))
    close(out, g)


@pytest.mark.parametrize("cfg", [c for c in CONVS if c[8] == torch.float32])
@pytest.mark.parametrize("mask_act", [None, "relu"])
def test_conv_dx(dev, cfg, mask_act):
    rng = np.random.default_rng(sum(cfg[:8]) + 2)
    x, w, b = make_conv(rng, cfg)
    B, H, W, C, KH, KW, s, Fo, dt = cfg
    OH, OW = ops.conv_out_hw(H, W, KH, KW, s)
    dz = rnd(rng, B, OH, OW, Fo)
    dcol = torch.empty(B * OH * OW * KH * KW * C, device=dev)
    out = torch.empty(B, H, W, C, device=dev)
    ops.conv_dx(dz.to(dev).view(-1, Fo), w.to(dev), (B, H, W, C), s, dcol, out,
                mask_src=x.to(dev) if mask_act else None, mask_act=mask_act)
    xd = x.double().requires_grad_(True)
    y = F.conv2d(xd.permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), None, stride=s)
    g, = torch.autograd.grad(y, xd, dz.double().permute(0, 3, 1, 2))
    close(out, g * actgrad_ref(x.double(), mask_act))


@pytest.mark.parametrize("M,N", [(102400, 32), (12544, 64), (256, 512), (256, 6), (7, 3),
                                 (1000, 100)])
def test_colsum(dev, M, N):
    rng = np.random.default_rng(M + N)
    x = rnd(rng, M, N)
    out = torch.empty(N, device=dev)
    ops.colsum(x.to(dev), out)
    close(out, x.double().sum(0), tol=5e-6)


def test_mfma_layout_transpose_detecting(dev):
    """A = I with an ASYMMETRIC B: catches row/column swaps in the MFMA C layout."""
    n = 96
    eye = torch.eye(n)
    b = torch.arange(n * 40, dtype=torch.float32).reshape(n, 40) / 7.0
    out = torch.empty(n, 40, device=dev)
    ops.dense_forward(eye.to(dev), b.to(dev), None, None, out)
    assert torch.equal(out.cpu(), b)


# ---- LDS-DMA main loop (gemm_dma.h): ragged but 16-byte-regular shapes, every tile, both loops ----
@pytest.fixture
def both_loops():
    """Runs the body twice: LDS-DMA main loop (default) and register-staged loop."""
    yield
    ops.FORCE_NO_DMA = False


DMA_SHAPES = [(200, 72, 300), (33, 36, 4), (260, 132, 68), (64, 64, 32), (1, 4, 4)]


@pytest.mark.parametrize("M,N,K", DMA_SHAPES)
@pytest.mark.parametrize("cfg", [0] + ALL_CFGS)
@pytest.mark.parametrize("splits", [0, 3])
def test_dma_dense_all_modes(dev, both_loops, M, N, K, cfg, splits):
    """Forward (A_ROW x B_ROW), dX (A_ROW x B_COL) and dW (A_COL x B_ROW, fused bias gradient)
    through the LDS-DMA loop with ragged M / N / K edges (zero fill must come from the DMA range
    check), and the register-staged loop on the same inputs."""
    if (cfg == 0) != (splits == 0):
        pytest.skip("auto plan only with auto splits")
    rng = np.random.default_rng(M * 5 + N * 3 + K + cfg)
    x, w, b = rnd(rng, M, K), rnd(rng, K, N) * 0.1, rnd(rng, N)
    dz = rnd(rng, M, N)
    y = torch.tanh(rnd(rng, M, K))
    res = {}
    for no_dma in (False, True):
        ops.FORCE_NO_DMA = no_dma
        out = torch.full((M, N), float("nan"), device=dev)
        ops.dense_forward(x.to(dev), w.to(dev), b.to(dev), "relu", out, force_cfg=cfg,
                          force_splits=splits)
        close(out, torch.relu(x.double() @ w.double() + b.double()))
        dx = torch.full((M, K), float("nan"), device=dev)
        ops.dense_dx(dz.to(dev), w.to(dev), dx, mask_src=y.to(dev), mask_act="tanh",
                     force_cfg=cfg, force_splits=splits)
        close(dx, (dz.double() @ w.double().T) * actgrad_ref(y.double(), "tanh"))
        dw = torch.full((K, N), float("nan"), device=dev)
        bg = torch.full((N,), float("nan"), device=dev)
        ops.dense_dw(x.to(dev), dz.to(dev), dw, force_cfg=cfg, force_splits=splits, bias_grad=bg)
        close(dw, x.double().T @ dz.double())
        close(bg, dz.double().sum(0), tol=5e-6)
        res[no_dma] = (out.cpu(), dx.cpu(), dw.cpu())
    # the two loops agree to fp32 summation-order noise
    for a, b2 in zip(res[False], res[True]):
        close(a, b2.double(), tol=1e-5)


@pytest.mark.parametrize("cfg", CONVS)
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 5, 6, 7])
def test_dma_conv_forward_and_dw(dev, both_loops, cfg, tile):
    rng = np.random.default_rng(sum(cfg[:8]) + 7 * tile)
    x, w, b = make_conv(rng, cfg)
    B, H, W, C, KH, KW, s, Fo, dt = cfg
    OH, OW = ops.conv_out_hw(H, W, KH, KW, s)
    dz = rnd(rng, B, OH, OW, Fo)
    xf = x.double() / 255.0 if dt == torch.uint8 else x.double()
    wd = w.double().requires_grad_(True)
    yref = F.conv2d(xf.permute(0, 3, 1, 2), wd.permute(3, 2, 0, 1), None, stride=s)
    gref, = torch.autograd.grad(yref, wd, dz.double().permute(0, 3, 1, 2))
    for no_dma in (False, True):
        ops.FORCE_NO_DMA = no_dma
        out = torch.full((B, OH, OW, Fo), float("nan"), device=dev)
        ops.conv_forward(x.to(dev), w.to(dev), b.to(dev), s, "relu", out, a_div=255.0,
                         force_cfg=tile, force_splits=0 if tile == 0 else 2)
        close(out, torch.relu(conv_ref(x, w, b, s, 255.0)))
        g = torch.full((KH, KW, C, Fo), float("nan"), device=dev)
        bg = torch.full((Fo,), float("nan"), device=dev)
        ops.conv_dw(x.to(dev), dz.to(dev).view(-1, Fo), (KH, KW, C, Fo), s, g, a_div=255.0,
                    force_cfg=tile, force_splits=0 if tile == 0 else 3, bias_grad=bg)
        close(g, gref)
        close(bg, dz.double().sum((0, 1, 2)), tol=5e-6)


# ---- small-N dense kernels (Q / value heads) vs the MFMA GEMM path and the fp64 reference -------
@pytest.mark.parametrize("M,N,K", [(256, 6, 512), (1, 2, 100), (37, 16, 53), (300, 1, 64),
                                   (2048, 6, 64), (5, 7, 3), (256, 4, 3136)])
@pytest.mark.parametrize("act", [None, "tanh"])
def test_dense_small_n(dev, M, N, K, act):
    rng = np.random.default_rng(M + 31 * N + K)
    x, w, b = rnd(rng, M, K), rnd(rng, K, N) * 0.1, rnd(rng, N)
    dz, h = rnd(rng, M, N), torch.relu(rnd(rng, M, K))
    assert ops.USE_SMALL_N and N <= ops.SMALL_N
    out = torch.full((M, N), float("nan"), device=dev)
    ops.dense_forward(x.to(dev), w.to(dev), b.to(dev), act, out)
    close(out, act_ref(x.double() @ w.double() + b.double(), act))
    dx = torch.full((M, K), float("nan"), device=dev)
    ops.dense_dx(dz.to(dev), w.to(dev), dx, mask_src=h.to(dev), mask_act="relu")
    close(dx, (dz.double() @ w.double().T) * (h > 0).double())
    dw = torch.full((K, N), float("nan"), device=dev)
    db = torch.full((N,), float("nan"), device=dev)
    ops.dense_dw(x.to(dev), dz.to(dev), dw, bias_grad=db)
    close(dw, x.double().T @ dz.double())
    close(db, dz.double().sum(0), tol=5e-6)
    # dX and dW (+ db) of the head in one launch: the same bits
    if ops.dense_small_backward_ok(x.to(dev), dz.to(dev), h.to(dev)):
        dx2 = torch.full((M, K), float("nan"), device=dev)
        dw2 = torch.full((K, N), float("nan"), device=dev)
        db2 = torch.full((N,), float("nan"), device=dev)
        ops.dense_small_backward(x.to(dev), dz.to(dev), w.to(dev), dx2, dw2, mask_src=h.to(dev),
                                 mask_act="relu", bias_grad=db2)
        assert torch.equal(dx2, dx) and torch.equal(dw2, dw) and torch.equal(db2, db)
    # and against the GEMM path (same math, different summation order)
    ops.USE_SMALL_N = False
    try:
        out2 = torch.empty(M, N, device=dev)
        ops.dense_forward(x.to(dev), w.to(dev), b.to(dev), act, out2)
    finally:
        ops.USE_SMALL_N = True
    close(out, out2.cpu(), tol=1e-5)


# ---- hidden layer + head with the split-K sum in the head's prologue ------------------------------
@pytest.mark.parametrize("M,K,H,N", [(256, 3136, 512, 6), (256, 3136, 512, 1), (32, 3136, 512, 6),
                                     (7, 1000, 64, 16), (256, 64, 32, 3), (100, 2052, 260, 5)])
@pytest.mark.parametrize("act1", ["relu", "tanh", None])
def test_dense_tail_is_bit_identical_to_the_two_launch_pair(dev, M, K, H, N, act1):
    """aa_gemm_f32_slabs + aa_dense_small_forward_slabs (csrc/gemm.hip, dense_small.hip) must give
    the SAME bits as aa_gemm_f32 (main loop + reduce launch) followed by aa_dense_small_forward --
    same slab order, same head accumulation order -- and both agree with float64."""
    rng = np.random.default_rng(M + K + H + N)
    x = torch.relu(rnd(rng, M, K)).to(dev)
    w1, b1 = (rnd(rng, K, H) * 0.05).to(dev), rnd(rng, H).to(dev)
    w2, b2 = (rnd(rng, H, N) * 0.1).to(dev), rnd(rng, N).to(dev)
    assert ops.dense_tail_supported(x, w1, w2)
    h_a = torch.full((M, H), float("nan"), device=dev)
    y_a = torch.full((M, N), float("nan"), device=dev)
    ops.dense_tail_forward(x, w1, b1, act1, h_a, w2, b2, None, y_a)
    h_b = torch.full((M, H), float("nan"), device=dev)
    y_b = torch.full((M, N), float("nan"), device=dev)
    ops.dense_forward(x, w1, b1, act1, h_b)
    ops.dense_forward(h_b, w2, b2, None, y_b)
    assert torch.equal(h_a, h_b)
    assert torch.equal(y_a, y_b)
    href = act_ref(x.cpu().double() @ w1.cpu().double() + b1.cpu().double(), act1)
    close(h_a, href)
    close(y_a, href @ w2.cpu().double() + b2.cpu().double(), tol=2e-5)


def test_dense_tail_refuses_bad_arguments(dev):
    import ctypes
    lib = _lib.load()
    z = torch.zeros(4, 8, device=dev)
    w = torch.zeros(8, 2, device=dev)
    y = torch.zeros(4, 2, device=dev)
    st = _lib.stream_ptr()
    ok = lib.aa_dense_small_forward_slabs(z.data_ptr(), 1, 4, 8, None, 0, z.data_ptr(), 8,
                                          w.data_ptr(), None, 0, 2, y.data_ptr(), st)
    assert ok == 0
    assert lib.aa_dense_small_forward_slabs(z.data_ptr(), 0, 4, 8, None, 0, z.data_ptr(), 8,
                                            w.data_ptr(), None, 0, 2, y.data_ptr(), st) != 0
    assert lib.aa_dense_small_forward_slabs(z.data_ptr(), 1, 4, 6, None, 0, z.data_ptr(), 8,
                                            w.data_ptr(), None, 0, 2, y.data_ptr(), st) != 0
    assert lib.aa_dense_small_forward_slabs(z.data_ptr(), 1, 4, 8, None, 0, z.data_ptr(), 8,
                                            w.data_ptr(), None, 0, 17, y.data_ptr(), st) != 0
    # masks belong to the input-gradient contractions, which keep their reduce launch (a fused
    # column sum is accepted since ABI 13: its rows follow the slabs, tests/test_gpu_opt_slabs.py)
    d = ops.gemm_desc(A=z.data_ptr(), B=w.data_ptr(), C=y.data_ptr(), M=4, N=2, K=8, lda=8, ldb=2,
                      ldc=2, a_mode=_lib.AA_A_ROW, b_mode=_lib.AA_B_ROW, mask_src=y.data_ptr(),
                      ldm=2, mask_kind=_lib.AA_ACT_RELU)
    sp = ctypes.c_int32(0)
    assert lib.aa_gemm_f32_slabs(ctypes.byref(d), None, 0, ctypes.byref(sp), st) != 0
    assert lib.aa_gemm_f32_slabs(None, None, 0, ctypes.byref(sp), st) != 0
    torch.cuda.synchronize()


def test_conv1_dw_whole_m_tile(dev):
    """The Atari conv1 weight gradient (uint8 frames, 8x8x4 patches, 32 filters) takes the 256x32
    tile with the pixels split over the workgroups; forced and automatic plans agree with the
    32x32-tile plan and the fp64 reference."""
    rng = np.random.default_rng(5)
    B = 32
    x = torch.from_numpy(rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8))
    dz = rnd(rng, B * 400, 32)
    res = {}
    for name, cfg, splits in [("auto", 0, 0), ("c8", 8, 7), ("c7", 7, 16)]:
        g = torch.full((8, 8, 4, 32), float("nan"), device=dev)
        gb = torch.full((32,), float("nan"), device=dev)
        ops.conv_dw(x.to(dev), dz.to(dev), (8, 8, 4, 32), 4, g, a_div=255.0, force_cfg=cfg,
                    force_splits=splits, bias_grad=gb)
        res[name] = (g.cpu(), gb.cpu())
    xf = x.double() / 255.0
    wd = torch.zeros(8, 8, 4, 32, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(xf.permute(0, 3, 1, 2), wd.permute(3, 2, 0, 1), None, stride=4)
    ref, = torch.autograd.grad(y, wd, dz.double().view(B, 20, 20, 32).permute(0, 3, 1, 2))
    for name, (g, gb) in res.items():
        close(g, ref)
        close(gb, dz.double().sum(0), tol=5e-6)


# ---- uint8 conv forward on the bf16 matrix cores (csrc/conv_u8_bf16.h) -------------------------
U8_BF16_CONVS = [  # (B, H, W, C, KH, KW, stride, F)
    (4, 84, 84, 4, 8, 8, 4, 32),     # Atari conv1: unrolled 8-chunk path, 1,600 pixels (ragged)
    (3, 36, 36, 4, 8, 8, 4, 32),     # 192 pixels: one partial super-tile
    (2, 20, 36, 4, 4, 8, 4, 32),     # 4 chunks per patch: runtime-loop path
    (2, 24, 40, 4, 3, 16, 4, 32),    # 64-byte patch rows (two chunks per row), odd KH
    (1, 8, 8, 4, 8, 8, 4, 32),       # a single pixel
]


@pytest.mark.parametrize("cfg", U8_BF16_CONVS)
@pytest.mark.parametrize("act", [None, "relu"])
def test_conv_u8_bf16x3_forward(dev, cfg, act):
    """Three bf16 pieces of the fp32 filter bank x exact bf16 bytes: same accuracy class as the
    fp32 MFMA loader (both checked against float64), and the auto plan takes it."""
    B, H, W, C, KH, KW, s, Fo = cfg
    rng = np.random.default_rng(sum(cfg))
    x = torch.from_numpy(rng.integers(0, 256, size=(B, H, W, C), dtype=np.uint8))
    w = rnd(rng, KH, KW, C, Fo) * 0.1
    b = rnd(rng, Fo)
    OH, OW = ops.conv_out_hw(H, W, KH, KW, s)
    ref = conv_ref(x, w, b, s, 255.0)
    if act == "relu":
        ref = torch.relu(ref)
    outs = {}
    for name, force in (("bf16x3", 9), ("auto", 0), ("fp32", 2)):
        out = torch.full((B, OH, OW, Fo), float("nan"), device=dev)
        ops.conv_forward(x.to(dev), w.to(dev), b.to(dev), s, act, out, a_div=255.0,
                         force_cfg=force)
        outs[name] = out.cpu()
        close(out, ref)
    assert torch.equal(outs["auto"], outs["bf16x3"])
    err = lambda o: (o.double() - ref).abs().max().item()
    assert err(outs["bf16x3"]) <= 2.0 * err(outs["fp32"]) + 1e-7


def test_conv_u8_bf16x3_exact_cases(dev):
    """Integer-valued weights: every product and partial sum is an integer < 2^24, so the result
    must be the IEEE quotient sum / a_div exactly (no approximation anywhere in the kernel); and
    weights needing all 24 significand bits survive the three-piece split."""
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.integers(0, 256, size=(2, 36, 36, 4), dtype=np.uint8))
    w = torch.from_numpy(rng.integers(-7, 8, size=(8, 8, 4, 32)).astype(np.float32))
    out = torch.empty(2, 8, 8, 32, device=dev)
    ops.conv_forward(x.to(dev), w.to(dev), None, 4, None, out, a_div=255.0, force_cfg=9)
    acc = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), stride=4)
    want = (acc.permute(0, 2, 3, 1).float() / np.float32(255.0))
    assert torch.equal(out.cpu(), want)
    # one non-zero tap with a full 24-bit significand: out = fl(u * w) / 255 exactly
    w2 = torch.zeros(8, 8, 4, 32)
    w2[3, 5, 2, :] = torch.tensor(np.float32(1.0) + np.float32(2.0 ** -23) * np.arange(32, dtype=np.float32) * 99991 % 8388608 * 0 + np.float32(1.2345678))
    w2[3, 5, 2, :] += torch.arange(32) * np.float32(2.0 ** -22)
    ops.conv_forward(x.to(dev), w2.to(dev), None, 4, None, out, a_div=255.0, force_cfg=9)
    u = x.reshape(2, 36, 36, 4)[:, 3:3 + 32:4, 5:5 + 32:4, 2].float()       # [2, 8, 8]
    want2 = (u[..., None] * w2[3, 5, 2, :]) / np.float32(255.0)
    assert torch.equal(out.cpu(), want2)


def test_conv_u8_bf16x3_strided_batch_and_ineligible(dev):
    rng = np.random.default_rng(12)
    x5 = torch.from_numpy(rng.integers(0, 256, size=(5, 2, 84, 84, 4), dtype=np.uint8)).to(dev)
    w, b = rnd(rng, 8, 8, 4, 32) * 0.1, rnd(rng, 32)
    out = torch.empty(5, 20, 20, 32, device=dev)
    ops.conv_forward(x5[:, 1], w.to(dev), b.to(dev), 4, "relu", out, a_div=255.0, force_cfg=9)
    close(out, torch.relu(conv_ref(x5[:, 1].cpu(), w, b, 4, 255.0)))
    # 16 filters / fp32 input: the bf16x3 kernel must refuse instead of computing something else
    w16 = rnd(rng, 8, 8, 4, 16)
    with pytest.raises(Exception):
        ops.conv_forward(x5[:, 1], w16.to(dev), None, 4, None,
                         torch.empty(5, 20, 20, 16, device=dev), force_cfg=9)


@pytest.mark.parametrize("cfg", U8_BF16_CONVS)
@pytest.mark.parametrize("with_bias", [False, True])
def test_conv_u8_bf16x3_dw(dev, cfg, with_bias):
    """Weight gradient over uint8 frames with dZ split into three exact bf16 pieces: same accuracy
    class as the fp32 MFMA plan (both against float64); the automatic plan takes it."""
    B, H, W, C, KH, KW, s, Fo = cfg
    rng = np.random.default_rng(sum(cfg) + 3)
    x = torch.from_numpy(rng.integers(0, 256, size=(B, H, W, C), dtype=np.uint8))
    OH, OW = ops.conv_out_hw(H, W, KH, KW, s)
    dz = rnd(rng, B * OH * OW, Fo)
    xf = x.double() / 255.0
    wd = torch.zeros(KH, KW, C, Fo, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(xf.permute(0, 3, 1, 2), wd.permute(3, 2, 0, 1), None, stride=s)
    ref, = torch.autograd.grad(y, wd, dz.double().view(B, OH, OW, Fo).permute(0, 3, 1, 2))
    res = {}
    for name, force in (("bf16x3", 9), ("auto", 0), ("fp32", 7)):
        g = torch.full((KH, KW, C, Fo), float("nan"), device=dev)
        gb = torch.full((Fo,), float("nan"), device=dev) if with_bias else None
        ops.conv_dw(x.to(dev), dz.to(dev), (KH, KW, C, Fo), s, g, a_div=255.0, force_cfg=force,
                    bias_grad=gb)
        close(g, ref)
        if with_bias:
            close(gb, dz.double().sum(0), tol=5e-6)
        res[name] = g.cpu()
    assert torch.equal(res["auto"], res["bf16x3"])
    err = lambda o: (o.double() - ref).abs().max().item()
    assert err(res["bf16x3"]) <= 2.0 * err(res["fp32"]) + 1e-7


def test_conv_u8_bf16x3_dw_exact_and_deterministic(dev):
    """Integer-valued dZ: every partial sum is an integer below 2^24, so each slab is an exact IEEE
    quotient and the result equals the float64 one rounded once per slab; two runs are bit-equal."""
    rng = np.random.default_rng(9)
    B = 3
    x = torch.from_numpy(rng.integers(0, 256, size=(B, 36, 36, 4), dtype=np.uint8))
    dz = torch.from_numpy(rng.integers(-3, 4, size=(B * 64, 32)).astype(np.float32))
    outs = []
    for _ in range(2):
        g = torch.empty(8, 8, 4, 32, device=dev)
        gb = torch.empty(32, device=dev)
        ops.conv_dw(x.to(dev), dz.to(dev), (8, 8, 4, 32), 4, g, a_div=1.0, force_cfg=9,
                    bias_grad=gb)
        outs.append((g.cpu(), gb.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    wd = torch.zeros(8, 8, 4, 32, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.double().permute(0, 3, 1, 2), wd.permute(3, 2, 0, 1), None, stride=4)
    ref, = torch.autograd.grad(y, wd, dz.double().view(B, 8, 8, 32).permute(0, 3, 1, 2))
    assert torch.equal(outs[0][0].double(), ref)          # a_div = 1: integers, exact
    assert torch.equal(outs[0][1].double(), dz.double().sum(0))


def test_sequential_dense_tail_equals_layer_by_layer(dev):
    """The Atari Q-network with the head summing fc1's split-K slabs: outputs, stored activations and
    gradients are bit-identical to the layer-by-layer forward (ops.FUSE_DENSE_TAIL off)."""
    from agents_amd.networks import sequential, layers as L
    from agents_amd.specs import tensor_spec
    spec = tensor_spec.TensorSpec((84, 84, 4), torch.uint8)

    def build():
        net = sequential.Sequential([
            L.Rescale(255.0), L.Conv2D(32, 8, 4, activation="relu"),
            L.Conv2D(64, 4, 2, activation="relu"), L.Conv2D(64, 3, 1, activation="relu"),
            L.Flatten(), L.Dense(512, activation="relu"), L.Dense(6)], input_spec=spec, seed=3)
        net.create_variables(spec, device=dev)
        return net

    g = torch.Generator().manual_seed(2)
    x = torch.randint(0, 256, (64, 84, 84, 4), dtype=torch.uint8, generator=g).to(dev)
    dq = torch.randn(64, 6, generator=g).to(dev)
    res = {}
    for fuse in (True, False):
        ops.FUSE_DENSE_TAIL = fuse
        try:
            net = build()
            q = net.forward(x, slot="t", need_grad=True).clone()
            net.backward(dq, slot="t")
            torch.cuda.synchronize()
            res[fuse] = (q, net.flat_grads.clone())
        finally:
            ops.FUSE_DENSE_TAIL = True
    assert torch.equal(res[True][0], res[False][0])
    assert torch.equal(res[True][1], res[False][1])
