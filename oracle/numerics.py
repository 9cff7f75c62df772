"""CPU restatement of the two numerical schemes the HIP kernels rely on (TEST INFRASTRUCTURE; only
tests/ may import this).

1. Exact bf16 pieces (csrc/conv_u8_bf16.h, csrc/gemm_bf16x6.h): an fp32 value is the sum of three
   round-to-nearest bf16 numbers, a byte is a bf16 number, and a product of two bf16 numbers is an
   fp32 number.  So a contraction of uint8 frames with fp32 filters (the Atari conv1 of
   tf_agents/examples/dqn/mnih15/dqn_train_eval_atari.py:80-112) can run on bf16 matrix cores with
   fp32 accumulation and lose nothing but the accumulation roundings the fp32 path has as well.
2. Gather-form input gradient of a VALID strided convolution by sub-pixel classes
   (csrc/conv_dx_frame.hip; what tf.GradientTape returns for keras Conv2D,
   tf_agents/agents/dqn/dqn_agent.py:412-426).
"""
import numpy as np
import torch


def bf16_round(x):
    """fp32 tensor -> nearest-even bf16, returned as fp32 (torch's cast is RNE)."""
    return x.to(torch.bfloat16).to(torch.float32)


def split3(x):
    """x (fp32) -> (hi, mid, lo), each exactly representable in bf16, hi + mid + lo == x."""
    hi = bf16_round(x)
    r1 = x - hi                       # exact: the residual of a rounding is representable
    mid = bf16_round(r1)
    r2 = r1 - mid
    lo = bf16_round(r2)
    return hi, mid, lo


def u8_dot_bf16x3(u8, w):
    """[M,K] uint8 x [K,N] fp32 -> fp32, the way the conv1 kernel forms it: three chains of exact
    byte x piece products accumulated in fp32, small pieces summed first."""
    a = u8.to(torch.float32)
    hi, mid, lo = split3(w)
    return (a @ lo + a @ mid) + a @ hi


def u8t_dot_bf16x3(u8, dz):
    """[P,K] uint8 frames-as-patches, [P,N] fp32 dZ -> [K,N]: the conv1 weight gradient, reduction
    over pixels, dZ in three exact pieces, small pieces summed first (one chain per piece)."""
    a = u8.to(torch.float32).t()
    hi, mid, lo = split3(dz)
    return (a @ lo + a @ mid) + a @ hi


def dot_bf16x6(x, w):
    """[M,K] fp32 x [K,N] fp32 with both operands in pieces: the six largest cross products."""
    x1, x2, x3 = split3(x)
    w1, w2, w3 = split3(w)
    small = x3 @ w1 + x1 @ w3 + x2 @ w2 + x2 @ w1 + x1 @ w2
    return x1 @ w1 + small


def conv_dx_gather(dz, w, x_shape, stride):
    """dz [B,OH,OW,Cout], w [KH,KW,Cin,Cout] -> dx [B,H,W,Cin] of a VALID conv, class by class:
    input pixels with (iy % s, ix % s) == (py, px) only meet the taps ky = py + s ty,
    kx = px + s tx, and for them dx is a stride-1 correlation of zero-padded dz."""
    B, H, W, Cin = x_shape
    KH, KW, _, Cout = w.shape
    OH, OW = dz.shape[1], dz.shape[2]
    s = stride
    TY, TX = -(-KH // s), -(-KW // s)
    pad = np.zeros((B, OH + 2 * TY, OW + 2 * TX, Cout), dtype=np.float64)
    pad[:, TY:TY + OH, TX:TX + OW] = dz
    dx = np.zeros((B, H, W, Cin), dtype=np.float64)
    for py in range(min(s, H)):
        for px in range(min(s, W)):
            ny, nx = -(-(H - py) // s), -(-(W - px) // s)
            acc = np.zeros((B, ny, nx, Cin), dtype=np.float64)
            for ty in range(-(-(KH - py) // s)):
                for tx in range(-(-(KW - px) // s)):
                    tap = w[py + s * ty, px + s * tx]                      # [Cin, Cout]
                    win = pad[:, TY - ty:TY - ty + ny, TX - tx:TX - tx + nx]   # dz[y'-ty, x'-tx]
                    acc += win @ tap.T
            dx[:, py::s, px::s] = acc
    return dx


# ---- the fma-refined quotient the conv1 kernels apply to each sum (Lambda(x / 255)) -------------
def _rn32(x):
    """Exact rational -> nearest-even fp32 (normal range), as a Fraction."""
    import math
    from fractions import Fraction
    if x == 0:
        return Fraction(0)
    sgn = -1 if x < 0 else 1
    a = abs(x)
    e = math.floor(math.log2(float(a))) - 23
    two = Fraction(2)
    while a / two ** e >= 2 ** 24:
        e += 1
    while a / two ** e < 2 ** 23:
        e -= 1
    m = a / two ** e
    fl = m.numerator // m.denominator
    rem = m - fl
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and fl % 2 == 1):
        fl += 1
    return sgn * fl * two ** e


def markstein_quotient(s, d):
    """q0 = RN(s * RN(1/d)); q = RN(q0 + RN(s - d q0) * RN(1/d)) with every fma rounded once
    (exact rational arithmetic in between).  Returns (q, RN(s / d)) as Fractions."""
    from fractions import Fraction
    s, d = Fraction(float(np.float32(s))), Fraction(float(np.float32(d)))
    rcp = _rn32(1 / d)
    q0 = _rn32(s * rcp)
    r = _rn32(s - d * q0)
    return _rn32(q0 + r * rcp), _rn32(s / d)
