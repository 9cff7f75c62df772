"""The numerical claims behind the bf16 matrix-core kernels and the gather-form conv input gradient,
checked on the CPU (oracle/numerics.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import numerics


def test_three_bf16_pieces_are_exact():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(200_000, generator=g) * torch.logspace(-20, 20, 200_000)
    hi, mid, lo = numerics.split3(x)
    for p in (hi, mid, lo):
        assert torch.equal(numerics.bf16_round(p), p)              # each piece is a bf16 number
    assert torch.equal((hi.double() + mid.double() + lo.double()).float(), x)
    assert torch.equal(hi.double() + mid.double() + lo.double(), x.double())   # exactly, not rounded
    nz = hi != 0
    assert (mid[nz].abs() <= hi[nz].abs() * 2.0 ** -8).all()
    assert (lo[nz].abs() <= hi[nz].abs() * 2.0 ** -16).all()


def test_byte_times_piece_is_exact_in_fp32():
    g = torch.Generator().manual_seed(1)
    w = torch.randn(4096, generator=g)
    u = torch.arange(256, dtype=torch.float32)
    for piece in numerics.split3(w):
        prod32 = u[:, None] * piece[None, :]
        assert torch.equal(prod32.double(), u.double()[:, None] * piece.double()[None, :])
        assert torch.equal(numerics.bf16_round(u), u)              # bytes are bf16 numbers


def test_u8_contraction_matches_fp32_accuracy():
    """Atari conv1 as a GEMM (K = 256 patch bytes, 32 filters): the three-piece bf16 contraction
    is as close to float64 as the plain fp32 one (only accumulation roundings in both)."""
    g = torch.Generator().manual_seed(2)
    u8 = torch.randint(0, 256, (2048, 256), dtype=torch.uint8, generator=g)
    w = torch.randn(256, 32, generator=g) * 0.1
    ref = u8.double() @ w.double()
    got = numerics.u8_dot_bf16x3(u8, w)
    plain = u8.float() @ w
    scale = ref.abs().max()
    e_got = (got.double() - ref).abs().max() / scale
    e_plain = (plain.double() - ref).abs().max() / scale
    assert e_got <= 2.0 * e_plain + 1e-7 and e_got < 2e-6
    # integer-valued filters: every partial sum is an integer < 2^24 -> exact
    wi = torch.randint(-7, 8, (256, 32), generator=g).float()
    assert torch.equal(numerics.u8_dot_bf16x3(u8, wi).double(), u8.double() @ wi.double())


def test_u8_weight_gradient_contraction():
    """conv1 dW: reduction over 6,400 pixels of uint8 patch bytes x fp32 dZ split in three pieces."""
    g = torch.Generator().manual_seed(4)
    u8 = torch.randint(0, 256, (6400, 256), dtype=torch.uint8, generator=g)
    dz = torch.randn(6400, 32, generator=g) * 1e-3
    ref = u8.double().t() @ dz.double()
    got = numerics.u8t_dot_bf16x3(u8, dz)
    plain = u8.float().t() @ dz
    scale = ref.abs().max()
    assert (got.double() - ref).abs().max() / scale <= 2.0 * (plain.double() - ref).abs().max() / scale + 1e-7
    dzi = torch.randint(-3, 4, (6400, 32), generator=g).float()
    assert torch.equal(numerics.u8t_dot_bf16x3(u8, dzi).double(), u8.double().t() @ dzi.double())


def test_six_product_contraction_has_fp32_class_error():
    g = torch.Generator().manual_seed(3)
    x, w = torch.randn(512, 512, generator=g), torch.randn(512, 64, generator=g)
    ref = x.double() @ w.double()
    e6 = (numerics.dot_bf16x6(x, w).double() - ref).abs().max()
    e32 = ((x @ w).double() - ref).abs().max()
    assert e6 <= 2.0 * e32 + 1e-7
    # the three dropped cross products are below fp32 resolution of each product
    x1, x2, x3 = numerics.split3(x)
    w1, w2, w3 = numerics.split3(w)
    dropped = (x2.double() @ w3.double() + x3.double() @ w2.double() + x3.double() @ w3.double())
    assert dropped.abs().max() < 2.0 ** -20 * ref.abs().max()


@pytest.mark.parametrize("cfg", [
    (2, 20, 20, 8, 4, 4, 2, 6),     # Atari conv2 geometry
    (2, 9, 9, 5, 3, 3, 1, 4),       # Atari conv3 geometry
    (1, 21, 22, 3, 4, 4, 2, 2),     # rows / columns the VALID conv never reads
    (2, 17, 14, 2, 5, 3, 3, 3),     # stride 3, classes with different tap counts
    (1, 3, 3, 1, 3, 3, 1, 1),
])
def test_gather_form_input_gradient_equals_autograd(cfg):
    B, H, W, Cin, KH, KW, s, Cout = cfg
    rng = np.random.default_rng(sum(cfg))
    w = rng.standard_normal((KH, KW, Cin, Cout))
    OH, OW = (H - KH) // s + 1, (W - KW) // s + 1
    dz = rng.standard_normal((B, OH, OW, Cout))
    x = torch.zeros(B, H, W, Cin, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.permute(0, 3, 1, 2), torch.from_numpy(w).permute(3, 2, 0, 1), stride=s)
    ref, = torch.autograd.grad(y, x, torch.from_numpy(dz).permute(0, 3, 1, 2))
    got = numerics.conv_dx_gather(dz, w, (B, H, W, Cin), s)
    np.testing.assert_allclose(got, ref.numpy(), rtol=1e-12, atol=1e-12)


def test_fma_refined_quotient_is_the_ieee_quotient():
    """The conv1 kernels divide each sum by a_div with q0 = s * (1/d), q = fma(fma(-d, q0, s), 1/d,
    q0).  In exact rational arithmetic with one rounding per fma this equals RN(s / d) for the
    divisors in use (255 and the test divisors), over six decades of s."""
    import random
    random.seed(0)
    for d in (255.0, 3.0, 127.5, 7.0):
        for _ in range(400):
            s = random.uniform(-1, 1) * 10.0 ** random.uniform(-3, 5)
            q, want = numerics.markstein_quotient(s, d)
            assert q == want, (s, d)
