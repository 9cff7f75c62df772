"""Tensor-level wrappers over the C ABI (include/agents_amd.h).

These take torch device tensors, validate shape / dtype / contiguity, and enqueue the HIP kernels on
torch's current stream.  No arithmetic happens in torch; there is no CPU fallback.
"""
import ctypes
import itertools

import torch

from agents_amd import _lib
from agents_amd._lib import (AA_A_COL, AA_A_PATCH, AA_A_PATCH_T, AA_A_PATCH_T_U8, AA_A_PATCH_U8,
                             AA_A_ROW, AA_ACT_NONE, AA_ACT_RELU, AA_ACT_TANH, AA_B_COL, AA_B_ROW,
                             AA_LOSS_HUBER, AA_LOSS_SQUARED, GemmDesc, check, ptr, require_cuda,
                             stream_ptr)

ACT = {None: AA_ACT_NONE, "none": AA_ACT_NONE, "linear": AA_ACT_NONE, "relu": AA_ACT_RELU,
       "tanh": AA_ACT_TANH}


class _Workspace:
    """Grow-only scratch buffer per (device, scope, line of execution): split-K slabs, column-sum
    partials.  GEMMs enqueued on a side line (`side_line`) run concurrently with the caller's, so
    each line owns its scratch; so does each `workspace_scope`.  Lines are named by the code that
    forks them, not by the stream handle: torch hands out stream handles from a small pool, so a
    graph-capture stream may well share its handle with some agent's side stream."""

    def __init__(self):
        self._buf = {}
        self._retired = []  # outgrown buffers stay alive: captured HIP graphs may point at them

    def get(self, nbytes, device):
        key = (device.type, device.index, _SCOPE, _LINE)
        buf = self._buf.get(key)
        if buf is None or buf.numel() < nbytes:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.AgentsAmdError(
                    "workspace would grow during graph capture; run one eager step first")
            if buf is not None:
                self._retired.append(buf)
            buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            self._buf[key] = buf
        return buf

    def release(self, tag):
        """Drops the buffers of scope `tag` (its owner's graphs are gone or parked for good)."""
        for key in [k for k in self._buf if k[2] == tag]:
            del self._buf[key]

    def reserve(self, tag, device):
        """Gives scope `tag` its own buffer, as large as the main line's current one."""
        main = self._buf.get((device.type, device.index, None, None))
        key = (device.type, device.index, tag, None)
        need = main.numel() if main is not None else 1 << 20
        if key not in self._buf or self._buf[key].numel() < need:
            if key in self._buf:
                self._retired.append(self._buf[key])
            self._buf[key] = torch.empty(need, dtype=torch.uint8, device=device)


_SCOPE = None
_LINE = None
_LINE_IDS = itertools.count(1)


class workspace_scope:
    """Kernels launched (or captured into a HIP graph) inside this context take their split-K /
    column-sum scratch from a buffer private to `tag` instead of the caller's: a graph that will
    replay concurrently with other GEMM work (the collect graph next to the train graphs,
    agents_amd/utils/graph.py: Lanes) must not share slabs with it."""

    def __init__(self, tag, device):
        self._tag, self._device = tag, torch.device(device)
        if self._device.type == "cuda" and self._device.index is None:
            self._device = torch.device("cuda", torch.cuda.current_device())

    def __enter__(self):
        global _SCOPE
        self._prev = _SCOPE
        if not torch.cuda.is_current_stream_capturing():
            _WS.reserve(self._tag, self._device)
            _WS2.reserve(self._tag, self._device)
            _WS3.reserve(self._tag, self._device)
        _SCOPE = self._tag
        return self

    def __exit__(self, *exc):
        global _SCOPE
        _SCOPE = self._prev
        return False


def release_scope(tag):
    """Forgets the scratch of a `workspace_scope` whose owner is going away (utils/graph.py:
    scopes are named after `id(owner)`, which the interpreter hands out again)."""
    for ws in (_WS, _WS2, _WS3):
        ws.release(tag)


class SideStream(torch.cuda.Stream):
    """A stream for work that overlaps with the caller's stream (fork / join by wait_stream); enter
    it with `side_line` so that its kernels take their scratch from the line's own buffers."""


def new_side_stream(device):
    st = SideStream(device=device)
    st.aa_line = next(_LINE_IDS)
    return st


class side_line:
    """`with side_line(stream):` = `with torch.cuda.stream(stream):` + the workspaces of that line."""

    def __init__(self, stream):
        self._stream = stream
        self._ctx = torch.cuda.stream(stream)

    def __enter__(self):
        global _LINE
        self._prev = _LINE
        line = getattr(self._stream, "aa_line", None)
        _LINE = line if line is not None else ("handle", self._stream.cuda_stream)
        self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        global _LINE
        self._ctx.__exit__(*exc)
        _LINE = self._prev
        return False


_WS = _Workspace()
# Separate scratch for column-sum partials so a bias-grad launch never aliases a live split-K slab
_WS2 = _Workspace()
# split filter planes of the bf16x6 conv pair (read by a whole launch: never shared with slabs)
_WS3 = _Workspace()


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError(f"{name} must be a contiguous float32 tensor, got {t.dtype} "
                         f"contiguous={t.is_contiguous()}")


def _rows_ok(t, name, dtype=torch.float32):
    """2-D tensor whose rows are contiguous (outer stride free): returns the row pitch."""
    if t.dtype != dtype or t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError(f"{name} must be a 2-D {dtype} tensor with contiguous rows")
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def _img_pitch(x):
    """NHWC tensor with dense images but a free batch stride: returns elements between images."""
    Bn, H, W, C = x.shape
    if x.stride(3) != 1 or x.stride(2) != C or x.stride(1) != W * C:
        raise ValueError("conv input images must be dense NHWC (only the batch dim may be strided)")
    return x.stride(0) if Bn > 1 else H * W * C


def _bias_grad_ptr(bias_grad, n):
    if bias_grad is None:
        return None
    require_cuda(bias_grad)
    _f32c(bias_grad, "bias_grad")
    if bias_grad.numel() != n:
        raise ValueError(f"bias_grad must have {n} elements, got {bias_grad.numel()}")
    return ptr(bias_grad)


def gemm_desc(**kw):
    d = GemmDesc()
    for k, v in kw.items():
        setattr(d, k, v)
    return d


# Test / tuning knob: True forces the register-staged GEMM main loop (aa_gemm_desc.no_dma) so the
# two main loops can be compared on identical inputs.  The default is the LDS-DMA loop.
FORCE_NO_DMA = False
# Dense layers with at most SMALL_N output units (Q / value heads) take the dedicated small-N
# kernels (csrc/dense_small.hip) instead of an MFMA GEMM launch; USE_SMALL_N = False is a test knob.
SMALL_N = 16
SMALL_DW_MAX_M = 512
USE_SMALL_N = True
# a_mode values whose contractions take the LDS-DMA loop (None = all that qualify; tuning tools set
# it to a set of modes)
import os as _os
_DMA_MODES = None


def gemm(desc, device):
    lib = _lib.load()
    if FORCE_NO_DMA or (_DMA_MODES is not None and desc.a_mode not in _DMA_MODES):
        desc.no_dma = 1
    need = lib.aa_gemm_f32_workspace_bytes(ctypes.byref(desc))
    if need < 0:
        raise ValueError("aa_gemm_f32: invalid descriptor (M,N,K must be positive)")
    ws = _WS.get(need, device) if need > 0 else None
    check(lib.aa_gemm_f32(ctypes.byref(desc), ptr(ws), ws.numel() if ws is not None else 0,
                          stream_ptr()), "aa_gemm_f32")


# ---- Dense ---------------------------------------------------------------------------------
def dense_forward(x, w, bias, act, out, a_div=None, force_cfg=0, force_splits=0):
    """out[M,N] = act(x[M,K] @ w[K,N] + bias)."""
    require_cuda(x, w, out)
    lda = _rows_ok(x, "x"); _f32c(w, "w"); _f32c(out, "out")
    M, K = x.shape
    K2, N = w.shape
    if K != K2 or tuple(out.shape) != (M, N):
        raise ValueError(f"dense_forward shape mismatch x{tuple(x.shape)} w{tuple(w.shape)} "
                         f"out{tuple(out.shape)}")
    if N <= SMALL_N and not force_cfg and not force_splits and USE_SMALL_N:
        check(_lib.load().aa_dense_small_forward(ptr(x), lda, ptr(w), ptr(bias), ACT[act], M, K, N,
                                                 ptr(out), stream_ptr()), "aa_dense_small_forward")
        return out
    d = gemm_desc(A=ptr(x), B=ptr(w), C=ptr(out), M=M, N=N, K=K, lda=lda, ldb=N, ldc=N,
                  a_mode=AA_A_ROW, b_mode=AA_B_ROW, bias=ptr(bias), act=ACT[act],
                  force_cfg=force_cfg, force_splits=force_splits)
    gemm(d, x.device)
    return out


# Dense(hidden) -> Dense(<= SMALL_N units): the head sums the hidden layer's split-K slabs itself
# (one launch less per forward pass and no round trip of the hidden activation); test / A-B knob
FUSE_DENSE_TAIL = True


def dense_tail_supported(x, w1, w2):
    K1, H = w1.shape
    return (FUSE_DENSE_TAIL and USE_SMALL_N and w2.shape[0] == H and w2.shape[1] <= SMALL_N
            and H > SMALL_N and H % 4 == 0 and x.dtype == torch.float32)


def dense_tail_forward(x, w1, b1, act1, h, w2, b2, act2, y):
    """h[M,H] = act1(x[M,K] @ w1 + b1); y[M,N] = act2(h @ w2 + b2), N <= SMALL_N.  Two launches
    when the first contraction is split-K (GEMM main loop, then the head summing the slabs in its
    prologue), otherwise the plain pair; bit-identical to dense_forward twice either way."""
    require_cuda(x, w1, h, w2, y)
    lda = _rows_ok(x, "x"); _f32c(w1, "w1"); _f32c(h, "h"); _f32c(w2, "w2"); _f32c(y, "y")
    M, K = x.shape
    K2, H = w1.shape
    H2, N = w2.shape
    if K != K2 or H != H2 or tuple(h.shape) != (M, H) or tuple(y.shape) != (M, N):
        raise ValueError("dense_tail_forward shape mismatch")
    lib = _lib.load()
    d = gemm_desc(A=ptr(x), B=ptr(w1), C=ptr(h), M=M, N=H, K=K, lda=lda, ldb=H, ldc=H,
                  a_mode=AA_A_ROW, b_mode=AA_B_ROW, bias=ptr(b1), act=ACT[act1])
    if FORCE_NO_DMA or (_DMA_MODES is not None and d.a_mode not in _DMA_MODES):
        d.no_dma = 1
    need = lib.aa_gemm_f32_workspace_bytes(ctypes.byref(d))
    if need < 0:
        raise ValueError("aa_gemm_f32: invalid descriptor (M,N,K must be positive)")
    ws = _WS.get(need, x.device) if need > 0 else None
    splits = ctypes.c_int32(0)
    check(lib.aa_gemm_f32_slabs(ctypes.byref(d), ptr(ws), ws.numel() if ws is not None else 0,
                                ctypes.byref(splits), stream_ptr()), "aa_gemm_f32_slabs")
    if splits.value > 1:
        check(lib.aa_dense_small_forward_slabs(
            ptr(ws), splits.value, M, H, ptr(b1), ACT[act1], ptr(h), H, ptr(w2), ptr(b2),
            ACT[act2], N, ptr(y), stream_ptr()), "aa_dense_small_forward_slabs")
    else:
        check(lib.aa_dense_small_forward(ptr(h), H, ptr(w2), ptr(b2), ACT[act2], M, H, N, ptr(y),
                                         stream_ptr()), "aa_dense_small_forward")
    return y


def dense_dx(dz, w, out, mask_src=None, mask_act=None, force_cfg=0, force_splits=0):
    """out[M,K] = (dz[M,N] @ w[K,N]^T) * act'(mask_src[M,K])."""
    require_cuda(dz, w, out, mask_src)
    _f32c(dz, "dz"); _f32c(w, "w"); _f32c(out, "out")
    M, N = dz.shape
    K, N2 = w.shape
    if N != N2 or tuple(out.shape) != (M, K):
        raise ValueError("dense_dx shape mismatch")
    if N <= SMALL_N and not force_cfg and not force_splits and USE_SMALL_N and \
            (mask_src is None or mask_src.is_contiguous()):
        check(_lib.load().aa_dense_small_dx(ptr(dz), ptr(w), ptr(mask_src),
                                            ACT[mask_act] if mask_src is not None else 0, M, K, N,
                                            ptr(out), stream_ptr()), "aa_dense_small_dx")
        return out
    d = gemm_desc(A=ptr(dz), B=ptr(w), C=ptr(out), M=M, N=K, K=N, lda=N, ldb=N, ldc=K,
                  a_mode=AA_A_ROW, b_mode=AA_B_COL, mask_src=ptr(mask_src), ldm=K,
                  mask_kind=ACT[mask_act] if mask_src is not None else 0,
                  force_cfg=force_cfg, force_splits=force_splits)
    gemm(d, dz.device)
    return out


def dense_small_backward_ok(x, dz, mask_src):
    M, N = dz.shape
    return (USE_SMALL_N and N <= SMALL_N and M <= SMALL_DW_MAX_M and
            (mask_src is None or mask_src.is_contiguous()))


def dense_small_backward(x, dz, w, dx, dw, mask_src=None, mask_act=None, bias_grad=None):
    """dense_dx + dense_dw (+ bias gradient) of a head with N <= SMALL_N units in one launch."""
    require_cuda(x, dz, w, dx, dw, mask_src)
    lda = _rows_ok(x, "x"); _f32c(dz, "dz"); _f32c(w, "w"); _f32c(dx, "dx"); _f32c(dw, "dw")
    M, K = x.shape
    M2, N = dz.shape
    if M != M2 or tuple(w.shape) != (K, N) or tuple(dx.shape) != (M, K) or \
            tuple(dw.shape) != (K, N):
        raise ValueError("dense_small_backward shape mismatch")
    check(_lib.load().aa_dense_small_backward(
        ptr(x), lda, ptr(dz), ptr(w), ptr(mask_src),
        ACT[mask_act] if mask_src is not None else 0, M, K, N, ptr(dx), ptr(dw),
        _bias_grad_ptr(bias_grad, N), stream_ptr()), "aa_dense_small_backward")
    return dx


def dense_dw(x, dz, out, force_cfg=0, force_splits=0, bias_grad=None):
    """out[K,N] = x[M,K]^T @ dz[M,N]; bias_grad[N] = sum_m dz[m, :] (fused, optional)."""
    require_cuda(x, dz, out)
    lda = _rows_ok(x, "x"); _f32c(dz, "dz"); _f32c(out, "out")
    M, K = x.shape
    M2, N = dz.shape
    if M != M2 or tuple(out.shape) != (K, N):
        raise ValueError("dense_dw shape mismatch")
    # (one workgroup per 64 weight rows walks all M samples: right for the DQN head's M = 256,
    # 100 us at PPO's M = 4,096 where the split-K GEMM takes 13 us)
    if N <= SMALL_N and M <= SMALL_DW_MAX_M and not force_cfg and not force_splits and USE_SMALL_N:
        check(_lib.load().aa_dense_small_dw(ptr(x), lda, ptr(dz), M, K, N, ptr(out),
                                            _bias_grad_ptr(bias_grad, N), stream_ptr()),
              "aa_dense_small_dw")
        return out
    d = gemm_desc(A=ptr(x), B=ptr(dz), C=ptr(out), M=K, N=N, K=M, lda=lda, ldb=N, ldc=N,
                  a_mode=AA_A_COL, b_mode=AA_B_ROW, force_cfg=force_cfg,
                  force_splits=force_splits, colsum_out=_bias_grad_ptr(bias_grad, N))
    gemm(d, x.device)
    return out


# ---- Conv2D (NHWC, VALID) -------------------------------------------------------------------
def conv_out_hw(H, W, KH, KW, stride):
    return (H - KH) // stride + 1, (W - KW) // stride + 1


def _conv_common(x, w, stride):
    if x.dim() != 4 or w.dim() != 4:
        raise ValueError("conv expects NHWC input [B,H,W,C] and HWIO kernel [KH,KW,Cin,Cout]")
    if not w.is_contiguous():
        raise ValueError("conv kernel must be contiguous")
    Bn, H, W, C = x.shape
    KH, KW, Cin, Cout = w.shape
    if Cin != C:
        raise ValueError("conv channel mismatch")
    OH, OW = conv_out_hw(H, W, KH, KW, stride)
    return Bn, H, W, C, KH, KW, Cout, OH, OW


def conv_forward(x, w, bias, stride, act, out, a_div=255.0, force_cfg=0, force_splits=0):
    """out[B,OH,OW,Cout] = act(conv_valid(x / a_div if uint8 else x, w) + bias)."""
    require_cuda(x, w, out)
    Bn, H, W, C, KH, KW, Cout, OH, OW = _conv_common(x, w, stride)
    _f32c(w, "w"); _f32c(out, "out")
    if x.dtype == torch.uint8:
        mode = AA_A_PATCH_U8
    elif x.dtype == torch.float32:
        mode = AA_A_PATCH
    else:
        raise ValueError("conv input must be uint8 or float32")
    if out.numel() != Bn * OH * OW * Cout:
        raise ValueError("conv_forward: bad output size")
    Kp = KH * KW * C
    d = gemm_desc(A=ptr(x), B=ptr(w), C=ptr(out), M=Bn * OH * OW, N=Cout, K=Kp, lda=0, ldb=Cout,
                  ldc=Cout, a_mode=mode, b_mode=AA_B_ROW, n_img=Bn, H=H, W=W, Cin=C, KH=KH, KW=KW,
                  stride=stride, img_pitch=_img_pitch(x), a_div=float(a_div), bias=ptr(bias),
                  act=ACT[act], force_cfg=force_cfg, force_splits=force_splits)
    gemm(d, x.device)
    return out


def _pair_descs(w1, b1, s1, act1, y1, w2, b2, s2, act2, y2):
    d = []
    for w, b, st, act, y in ((w1, b1, s1, act1, y1), (w2, b2, s2, act2, y2)):
        KH, KW, _, Cout = w.shape
        d.append(_lib.ConvLayerDesc(w=ptr(w), bias=ptr(b), y=ptr(y), KH=KH, KW=KW, stride=st,
                                    Cout=Cout, act=ACT[act]))
    return d


_PAIR_OK = {}
_PAIR_X6_WS = {}
# AA_CONV_PAIR_X6=0: keep the fused conv pair on the fp32 MFMA kernel (A/B measurements)
CONV_PAIR_X6 = _os.environ.get("AA_CONV_PAIR_X6", "1") != "0"


def conv_pair_supported(x_shape, w1, s1, w2, s2):
    """True when two consecutive VALID convs over fp32 NHWC frames of x_shape fit the fused
    one-workgroup-per-frame kernel (csrc/conv_pair.hip)."""
    key = (tuple(x_shape), tuple(w1.shape), s1, tuple(w2.shape), s2)
    ok = _PAIR_OK.get(key)
    if ok is None:
        Bn, H, W, C = x_shape
        if len(w1.shape) != 4 or len(w2.shape) != 4 or w1.shape[2] != C or \
                w2.shape[2] != w1.shape[3]:
            ok = False
        else:
            d = []
            for w, st in ((w1, s1), (w2, s2)):
                d.append(_lib.ConvLayerDesc(w=None, bias=None, y=None, KH=w.shape[0],
                                            KW=w.shape[1], stride=st, Cout=w.shape[3], act=0))
            ok = bool(_lib.load().aa_conv_pair_supported(Bn, H, W, C, ctypes.byref(d[0]),
                                                         ctypes.byref(d[1])))
        _PAIR_OK[key] = ok
    return ok


def conv_pair_prepare_bytes(x_shape, w1, s1, w2, s2):
    """Bytes of filter-plane scratch of the bf16x6 pair for these shapes; 0 when the pair takes the
    fp32 kernel (nothing to prepare)."""
    Bn, H, W, C = x_shape
    key = (Bn, H, W, C, tuple(w1.shape), s1, tuple(w2.shape), s2)
    ws_bytes = _PAIR_X6_WS.get(key)
    if ws_bytes is None:
        d = []
        for w, st in ((w1, s1), (w2, s2)):
            d.append(_lib.ConvLayerDesc(w=None, bias=None, y=None, KH=w.shape[0], KW=w.shape[1],
                                        stride=st, Cout=w.shape[3], act=0))
        ws_bytes = int(_lib.load().aa_conv_pair_x6_workspace_bytes(
            Bn, H, W, C, ctypes.byref(d[0]), ctypes.byref(d[1]))) if CONV_PAIR_X6 else 0
        _PAIR_X6_WS[key] = ws_bytes
    return ws_bytes


def conv_pair_prepare(x_shape, w1, s1, w2, s2, ws):
    """The weights-only half of conv_pair_forward (filter banks split into `ws`): issue it early,
    on another stream, and pass `prepared=ws` to conv_pair_forward."""
    require_cuda(w1, w2, ws)
    Bn, H, W, C = x_shape
    d = _pair_descs(w1, None, s1, None, None, w2, None, s2, None, None)
    with torch.cuda.device(ws.device):
        check(_lib.load().aa_conv_pair_x6_phase(None, 0, Bn, H, W, C, ctypes.byref(d[0]),
                                                ctypes.byref(d[1]), ptr(ws), ws.numel(), 1,
                                                _lib.stream_ptr()), "aa_conv_pair_x6_phase(1)")


def conv_pair_forward(x, w1, b1, s1, act1, y1, w2, b2, s2, act2, y2, prepared=None):
    """y1 = act1(conv(x, w1) + b1), y2 = act2(conv(y1, w2) + b2) in one launch; both are written
    (y1 = None on the bf16x6 kernel: the middle activation stays in LDS -- a forward no backward
    pass follows).  prepared: scratch that conv_pair_prepare filled for these weights."""
    require_cuda(x, w1, w2, y2)
    if x.dtype != torch.float32:
        raise ValueError("conv_pair_forward needs a float32 NHWC input")
    Bn, H, W, C, KH1, KW1, C1, OH1, OW1 = _conv_common(x, w1, s1)
    KH2, KW2, Cin2, C2 = w2.shape
    OH2, OW2 = conv_out_hw(OH1, OW1, KH2, KW2, s2)
    if Cin2 != C1 or not w2.is_contiguous():
        raise ValueError("conv_pair_forward: second kernel does not match the first layer")
    _f32c(w1, "w1"); _f32c(w2, "w2"); _f32c(y2, "y2")
    if y1 is not None:
        _f32c(y1, "y1")
    elif conv_pair_prepare_bytes((Bn, H, W, C), w1, s1, w2, s2) <= 0:
        raise ValueError("conv_pair_forward: y1=None needs the bf16x6 kernel")
    if (y1 is not None and y1.numel() != Bn * OH1 * OW1 * C1) or \
            y2.numel() != Bn * OH2 * OW2 * C2:
        raise ValueError("conv_pair_forward: bad output sizes")
    d = _pair_descs(w1, b1, s1, act1, y1, w2, b2, s2, act2, y2)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        ws_bytes = conv_pair_prepare_bytes((Bn, H, W, C), w1, s1, w2, s2)
        if ws_bytes > 0:
            # bf16 matrix cores, fp32 accuracy (csrc/conv_pair_x6.hip); the split filter planes
            # live in the calling line's scratch (concurrent forwards on other streams -- target
            # network, collect graph -- own theirs) unless the caller prepared them already
            ws = prepared if prepared is not None else _WS3.get(ws_bytes, x.device)
            check(lib.aa_conv_pair_x6_phase(ptr(x), _img_pitch(x), Bn, H, W, C,
                                            ctypes.byref(d[0]), ctypes.byref(d[1]), ptr(ws),
                                            ws.numel(), 2 if prepared is not None else 3,
                                            _lib.stream_ptr()), "aa_conv_pair_x6_phase")
        else:
            check(lib.aa_conv_pair_forward(ptr(x), _img_pitch(x), Bn, H, W, C,
                                           ctypes.byref(d[0]), ctypes.byref(d[1]),
                                           _lib.stream_ptr()), "aa_conv_pair_forward")
    return y2


# AA_CONV_DW_X6=0: keep the conv weight gradients of fp32 layers on the fp32 MFMA GEMM (A/B)
CONV_DW_X6 = _os.environ.get("AA_CONV_DW_X6", "1") != "0"
_DW_X6_WS = {}
# slabs of weight gradients whose reduce is deferred (one buffer per pending layer and line)
_WS_DW_DEFER = [_Workspace() for _ in range(4)]
# AA_CONV_DW_MERGE_REDUCE=0: every conv weight gradient sums its own slabs (A/B measurements)
CONV_DW_MERGE_REDUCE = True


class PendingDwReduce:
    """Conv weight gradients whose per-frame-group slabs are written but not summed yet
    (`conv_dw(..., defer=pending)`); `conv_dw_flush(pending)` sums them all in ONE launch
    (csrc/splitk_reduce.h: aa_splitk_reduce_multi_kernel) -- the backward pass of a conv stack
    used to pay one reduce launch per layer on its side stream."""

    def __init__(self, keep=False, kept_ws=None):
        self.items = []      # (desc, slabs, out, bias_grad)
        # kept slabs outlive the backward pass (until the optimizer launch): they live in buffers
        # of their OWNER (`kept_ws`: four _Workspace objects held by the network), not in the
        # module-wide per-line scratch another network's backward on the same line would overwrite
        self.kept_ws = kept_ws
        # keep=True: nobody sums the slabs -- the optimizer launch reads its gradients from them
        # (`grad_slabs`, csrc/optim.hip: aa_rmsprop_step_slabs); `kept` then also lists the slabs
        # of GEMM-path weight gradients: (slabs, splits, M * N, N, out, bias_grad)
        self.keep = keep
        self.kept = []


def new_kept_workspaces():
    """Four scratch buffers (one per kept layer) for an owner of keep=True weight-gradient slabs."""
    return [_Workspace() for _ in range(4)]


def _slabs_deep(splits, mn, n_tail):
    """The slab shapes the 16-z-lane reduce (and aa_rmsprop_step_slabs) takes."""
    return splits >= 32 and (mn + n_tail) // 4 <= 65536 and mn % 4 == 0 and n_tail % 4 == 0


def _follows(out, bias_grad, mn):
    """bias_grad is the tensor right behind `out` in one flat buffer (kernel, bias order)."""
    return bias_grad is not None and bias_grad.dtype == torch.float32 and \
        bias_grad.data_ptr() == out.data_ptr() + 4 * mn


def grad_slabs(pending, flat_grads):
    """aa_grad_slabs of a keep=True PendingDwReduce relative to `flat_grads` (None: nothing kept)."""
    if pending is None or not pending.kept:
        return None
    kept = sorted(pending.kept, key=lambda k: k[4].data_ptr())
    g = _lib.GradSlabs()
    g.n = len(kept)
    for i, (ws, splits, mn, n_tail, out, _bg) in enumerate(kept):
        off = (out.data_ptr() - flat_grads.data_ptr()) // 4
        if off < 0 or off + mn + n_tail > flat_grads.numel():
            raise ValueError("grad_slabs: a weight gradient is not a view of flat_grads")
        g.splits[i], g.mn[i], g.n_tail[i], g.offset[i], g.slab[i] = splits, mn, n_tail, off, ptr(ws)
    g._keepalive = [k[0] for k in kept]
    return g


def conv_dw_flush(pending):
    if pending.keep:
        return
    items, pending.items = pending.items, []
    if not items:
        return
    n = len(items)
    descs = (ctypes.c_void_p * n)(*[ctypes.addressof(it[0]) for it in items])
    wss = (ctypes.c_void_p * n)(*[ptr(it[1]) for it in items])
    dws = (ctypes.c_void_p * n)(*[ptr(it[2]) for it in items])
    dbs = (ctypes.c_void_p * n)(*[_bias_grad_ptr(it[3], it[0].Cout) for it in items])
    with torch.cuda.device(items[0][2].device):
        check(_lib.load().aa_conv_dw_frame_x6_reduce(
            n, ctypes.cast(descs, ctypes.c_void_p), ctypes.cast(wss, ctypes.c_void_p),
            ctypes.cast(dws, ctypes.c_void_p), ctypes.cast(dbs, ctypes.c_void_p), stream_ptr()),
            "aa_conv_dw_frame_x6_reduce")


def conv_dw(x, dz, w_shape, stride, out, a_div=255.0, force_cfg=0, force_splits=0,
            bias_grad=None, defer=None):
    """out[KH,KW,Cin,Cout] = patches(x)^T @ dz[B*OH*OW, Cout]; bias_grad[Cout] = column sums of
    dz (fused, optional).  defer: a PendingDwReduce -- when the per-frame kernel takes the layer,
    only its slabs are written and `conv_dw_flush(defer)` finishes `out` / `bias_grad` later."""
    require_cuda(x, dz, out)
    KH, KW, Cin, Cout = w_shape
    Bn, H, W, C = x.shape
    OH, OW = conv_out_hw(H, W, KH, KW, stride)
    _f32c(dz, "dz"); _f32c(out, "out")
    mode = AA_A_PATCH_T_U8 if x.dtype == torch.uint8 else AA_A_PATCH_T
    Kp = KH * KW * C
    if dz.numel() != Bn * OH * OW * Cout or out.numel() != Kp * Cout:
        raise ValueError("conv_dw: bad sizes")
    # (a_div rescales uint8 inputs only)
    if CONV_DW_X6 and x.dtype == torch.float32 and not force_cfg and not force_splits and \
            x.is_contiguous() and x.data_ptr() % 16 == 0 and \
            dz.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0:
        key = (Bn, H, W, C, KH, KW, stride, Cout)
        ws_bytes = _DW_X6_WS.get(key)
        dd = _dxf_desc((Bn, H, W, C), (KH, KW, C, Cout), stride, dz=dz)
        if ws_bytes is None:
            ws_bytes = int(_lib.load().aa_conv_dw_frame_x6_workspace_bytes(ctypes.byref(dd)))
            _DW_X6_WS[key] = ws_bytes
        if ws_bytes > 0:
            # per-frame kernel on the bf16 matrix cores at fp32 accuracy (csrc/conv_dw_frame_x6.hip);
            # its slabs live in the calling line's scratch like the GEMM's
            groups = ws_bytes // (4 * (Kp * Cout + Cout))
            keep = defer is not None and defer.keep and len(defer.kept) < 4 and \
                _follows(out, bias_grad, Kp * Cout) and _slabs_deep(groups, Kp * Cout, Cout)
            if keep or (defer is not None and not defer.keep and CONV_DW_MERGE_REDUCE and
                        len(defer.items) < 4):
                n_pending = len(defer.kept) if keep else len(defer.items)
                pool = defer.kept_ws if (keep and defer.kept_ws is not None) else _WS_DW_DEFER
                ws = pool[n_pending].get(ws_bytes, x.device)
                with torch.cuda.device(x.device):
                    check(_lib.load().aa_conv_dw_frame_x6_slabs(
                        ctypes.byref(dd), ptr(x), 1 if bias_grad is not None else 0, ptr(ws),
                        ws.numel(), stream_ptr()), "aa_conv_dw_frame_x6_slabs")
                if keep:
                    defer.kept.append((ws, groups, Kp * Cout, Cout, out, bias_grad))
                else:
                    defer.items.append((dd, ws, out, bias_grad))
                return out
            ws = _WS.get(ws_bytes, x.device)
            with torch.cuda.device(x.device):
                check(_lib.load().aa_conv_dw_frame_x6(
                    ctypes.byref(dd), ptr(x), ptr(out), _bias_grad_ptr(bias_grad, Cout), ptr(ws),
                    ws.numel(), stream_ptr()), "aa_conv_dw_frame_x6")
            return out
    d = gemm_desc(A=ptr(x), B=ptr(dz), C=ptr(out), M=Kp, N=Cout, K=Bn * OH * OW, lda=0, ldb=Cout,
                  ldc=Cout, a_mode=mode, b_mode=AA_B_ROW, n_img=Bn, H=H, W=W, Cin=C, KH=KH, KW=KW,
                  stride=stride, img_pitch=_img_pitch(x), a_div=float(a_div),
                  force_cfg=force_cfg, force_splits=force_splits,
                  colsum_out=_bias_grad_ptr(bias_grad, Cout))
    if defer is not None and defer.keep and len(defer.kept) < 4 and \
            _follows(out, bias_grad, Kp * Cout):
        # GEMM-path weight gradient (the uint8 first layer): slabs + column-sum rows only when the
        # plan is split the way the optimizer's slab walk takes; splits = bytes / slab bytes
        lib = _lib.load()
        need = int(lib.aa_gemm_f32_workspace_bytes(ctypes.byref(d)))
        splits = need // (4 * (Kp * Cout + Cout)) if need > 0 else 1
        if _slabs_deep(splits, Kp * Cout, Cout):
            pool = defer.kept_ws if defer.kept_ws is not None else _WS_DW_DEFER
            ws = pool[len(defer.kept)].get(need, x.device)
            got = ctypes.c_int32(0)
            with torch.cuda.device(x.device):
                check(lib.aa_gemm_f32_slabs(ctypes.byref(d), ptr(ws), ws.numel(), ctypes.byref(got),
                                            stream_ptr()), "aa_gemm_f32_slabs")
            if got.value != splits:
                raise RuntimeError(f"conv_dw: plan has {got.value} slabs, workspace says {splits}")
            defer.kept.append((ws, splits, Kp * Cout, Cout, out, bias_grad))
            return out
    gemm(d, x.device)
    return out


def conv_dx(dz, w, x_shape, stride, dcol, out, mask_src=None, mask_act=None, prepared=None):
    """Input gradient of a VALID conv: dcol = dz @ w^T (GEMM), out = col2im(dcol) * act'(mask).
    prepared: scratch filled by conv_dx_prepare for these weights (gather-form bf16x6 path)."""
    require_cuda(dz, w, dcol, out, mask_src)
    lib = _lib.load()
    Bn, H, W, C = x_shape
    KH, KW, Cin, Cout = w.shape
    OH, OW = conv_out_hw(H, W, KH, KW, stride)
    Kp = KH * KW * C
    M = Bn * OH * OW
    _f32c(dz, "dz"); _f32c(w, "w"); _f32c(dcol, "dcol"); _f32c(out, "out")
    if dcol.numel() < M * Kp or out.numel() != Bn * H * W * C:
        raise ValueError("conv_dx: bad sizes")
    if CONV_DX_FRAME and (mask_src is None or mask_src.is_contiguous()) and \
            dz.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0 and \
            conv_dx_frame_supported(tuple(x_shape), tuple(w.shape), stride):
        return conv_dx_frame(dz, w, x_shape, stride, out, mask_src=mask_src, mask_act=mask_act,
                             prepared=prepared)
    d = gemm_desc(A=ptr(dz), B=ptr(w), C=ptr(dcol), M=M, N=Kp, K=Cout, lda=Cout, ldb=Cout,
                  ldc=Kp, a_mode=AA_A_ROW, b_mode=AA_B_COL)
    gemm(d, dz.device)
    check(lib.aa_col2im_f32(ptr(dcol), Bn, H, W, C, KH, KW, stride, ptr(out), ptr(mask_src),
                            ACT[mask_act] if mask_src is not None else 0, stream_ptr()),
          "aa_col2im_f32")
    return out


CONV_DX_FRAME = True   # gather-form conv input gradient
_DXF_OK = {}
_DXF_X6_WS = {}
# AA_CONV_DX_X6=0: keep the gather-form conv input gradient on the fp32 MFMA kernel (A/B)
CONV_DX_X6 = _os.environ.get("AA_CONV_DX_X6", "1") != "0"


def _dxf_desc(x_shape, w_shape, stride, dz=None, w=None, mask_src=None, mask_act=None, out=None):
    Bn, H, W, C = x_shape
    KH, KW, Cin, Cout = w_shape
    return _lib.ConvDxDesc(dz=ptr(dz), w=ptr(w), mask_src=ptr(mask_src), dx=ptr(out), n_img=Bn,
                           H=H, W=W, Cin=C, KH=KH, KW=KW, stride=stride, Cout=Cout,
                           mask_kind=ACT[mask_act] if mask_src is not None else 0)


def conv_dx_frame_supported(x_shape, w_shape, stride):
    key = (tuple(x_shape), tuple(w_shape), stride)
    ok = _DXF_OK.get(key)
    if ok is None:
        d = _dxf_desc(x_shape, w_shape, stride)
        ok = x_shape[3] == w_shape[2] and bool(
            _lib.load().aa_conv_dx_frame_supported(ctypes.byref(d)))
        _DXF_OK[key] = ok
    return ok


def conv_dx_prepare_bytes(x_shape, w, stride):
    """Bytes of scratch the bf16x6 gather-form input gradient prepares from the weights for these
    shapes; 0 when conv_dx would not take that path (nothing to prepare)."""
    if not (CONV_DX_FRAME and CONV_DX_X6 and w.data_ptr() % 16 == 0 and
            conv_dx_frame_supported(tuple(x_shape), tuple(w.shape), stride)):
        return 0
    key = (tuple(x_shape), tuple(w.shape), stride)
    ws_bytes = _DXF_X6_WS.get(key)
    if ws_bytes is None:
        d = _dxf_desc(x_shape, w.shape, stride)
        ws_bytes = int(_lib.load().aa_conv_dx_frame_x6_workspace_bytes(ctypes.byref(d)))
        _DXF_X6_WS[key] = ws_bytes
    return ws_bytes


def conv_dx_prepare(x_shape, w, stride, ws):
    """The weights-only half of the bf16x6 input gradient (filter fragments + k-step tables into
    `ws`): issue it early, on another stream, and pass `prepared=ws` to conv_dx."""
    require_cuda(w, ws)
    d = _dxf_desc(x_shape, w.shape, stride, w=w)
    with torch.cuda.device(ws.device):
        check(_lib.load().aa_conv_dx_frame_x6_phase(ctypes.byref(d), ptr(ws), ws.numel(), 1,
                                                    stream_ptr()), "aa_conv_dx_frame_x6_phase(1)")


def conv_dx_frame(dz, w, x_shape, stride, out, mask_src=None, mask_act=None, prepared=None):
    """Input gradient of a VALID conv, one workgroup per frame (csrc/conv_dx_frame.hip)."""
    require_cuda(dz, w, out, mask_src)
    Bn, H, W, C = x_shape
    KH, KW, Cin, Cout = w.shape
    OH, OW = conv_out_hw(H, W, KH, KW, stride)
    _f32c(dz, "dz"); _f32c(w, "w"); _f32c(out, "out")
    if mask_src is not None:
        _f32c(mask_src, "mask_src")
        if mask_src.numel() != Bn * H * W * C:
            raise ValueError("conv_dx_frame: bad mask size")
    if dz.numel() != Bn * OH * OW * Cout or out.numel() != Bn * H * W * C or Cin != C:
        raise ValueError("conv_dx_frame: bad sizes")
    d = _dxf_desc(x_shape, w.shape, stride, dz, w, mask_src, mask_act, out)
    lib = _lib.load()
    with torch.cuda.device(dz.device):
        key = (tuple(x_shape), tuple(w.shape), stride)
        ws_bytes = _DXF_X6_WS.get(key)
        if ws_bytes is None:
            ws_bytes = int(lib.aa_conv_dx_frame_x6_workspace_bytes(ctypes.byref(d))) \
                if CONV_DX_X6 else 0
            _DXF_X6_WS[key] = ws_bytes
        if ws_bytes > 0:
            # bf16 matrix cores, fp32 accuracy (csrc/conv_dx_frame_x6.hip); split filter planes in
            # the calling stream's scratch
            ws = prepared if prepared is not None else _WS3.get(ws_bytes, dz.device)
            check(lib.aa_conv_dx_frame_x6_phase(ctypes.byref(d), ptr(ws), ws.numel(),
                                                2 if prepared is not None else 3, stream_ptr()),
                  "aa_conv_dx_frame_x6_phase")
        else:
            check(lib.aa_conv_dx_frame(ctypes.byref(d), stream_ptr()), "aa_conv_dx_frame")
    return out


def colsum(x2d, out):
    """out[N] = sum_m x2d[m, :]."""
    require_cuda(x2d, out)
    lib = _lib.load()
    _f32c(x2d, "x"); _f32c(out, "out")
    M, N = x2d.shape
    need = lib.aa_colsum_workspace_bytes(M, N)
    ws = _WS2.get(need, x2d.device)
    check(lib.aa_colsum_f32(ptr(x2d), N, M, N, ptr(out), ptr(ws), ws.numel(), stream_ptr()),
          "aa_colsum_f32")
    return out


def copy_segments(pairs):
    """[(src, dst), ...] (at most 8): 2-D views [rows, cols_i] of 4-byte-element tensors whose
    last axis is contiguous (rows may be strided: slices of a time axis, column blocks of a wider
    buffer); every dst gets its src in ONE launch (csrc/ppo.hip: aa_copy_segments)."""
    n = len(pairs)
    if not 1 <= n <= 8:
        raise ValueError("copy_segments takes 1..8 (src, dst) pairs")
    rows = pairs[0][0].shape[0]
    src = (ctypes.c_void_p * n)()
    dst = (ctypes.c_void_p * n)()
    sp = (ctypes.c_int64 * n)()
    dp = (ctypes.c_int64 * n)()
    cols = (ctypes.c_int32 * n)()
    for i, (a, b) in enumerate(pairs):
        require_cuda(a, b)
        if a.dim() != 2 or b.dim() != 2 or a.shape != b.shape or a.shape[0] != rows or \
                a.dtype != b.dtype or a.element_size() != 4 or \
                (a.shape[1] > 1 and (a.stride(1) != 1 or b.stride(1) != 1)):
            raise ValueError(f"copy_segments: bad pair {i}: {tuple(a.shape)} {a.dtype} "
                             f"strides {a.stride()} -> {tuple(b.shape)} {b.dtype} {b.stride()}")
        src[i], dst[i] = a.data_ptr(), b.data_ptr()
        sp[i], dp[i], cols[i] = max(a.stride(0), a.shape[1]), max(b.stride(0), b.shape[1]), a.shape[1]
    check(_lib.load().aa_copy_segments(src, dst, sp, dp, cols, n, rows, stream_ptr()),
          "aa_copy_segments")


# ---- DQN loss -------------------------------------------------------------------------------
# TD loss + backward of the Q head in one launch (csrc/dqn.hip: aa_dqn_loss_head_backward); A/B knob
FUSE_LOSS_HEAD = True


def dqn_td_loss(q_online, q_next_target, q_next_select, next_mask, actions, reward, discount,
                step_type, weights, gamma, reward_scale, loss_kind, global_batch, loss_out,
                td_loss_out, td_error_out, dq_out, gamma_loss=None, field_sums_out=None,
                head=None):
    """field_sums_out (optional float32[2]) receives sum(td_loss), sum(td_error).
    head = dict(x, w, dx, dw, mask_src, mask_act, bias_grad): also runs the backward pass of the
    Q head (the Dense layer that produced q_online from x) in the same launch -- same results as
    this call without `head` followed by dense_small_backward(x, dq_out, w, dx, dw, ...)."""
    if gamma_loss is None:
        gamma_loss = gamma
    require_cuda(q_online, q_next_target, actions, reward, discount, step_type)
    lib = _lib.load()
    B, A = q_online.shape
    T = reward.shape[1]
    for t, n in ((q_online, "q_online"), (q_next_target, "q_next_target"), (reward, "reward"),
                 (discount, "discount")):
        _f32c(t, n)
    if step_type.dtype != torch.int32 or not step_type.is_contiguous():
        raise ValueError("step_type must be contiguous int32 [B,T]")
    if actions.dtype not in (torch.int32, torch.int64):
        raise ValueError("actions must be int32 or int64")
    if actions.dim() == 2:
        action_stride = actions.stride(0)
    else:
        action_stride = actions.stride(0) if actions.numel() > 1 else 1
    if field_sums_out is not None:
        require_cuda(field_sums_out)
        _f32c(field_sums_out, "field_sums_out")
        if field_sums_out.numel() < 2:
            raise ValueError("field_sums_out needs two elements")
    common_ = (ptr(q_online), ptr(q_next_target), ptr(q_next_select), ptr(next_mask), ptr(actions),
               1 if actions.dtype == torch.int64 else 0, action_stride, ptr(reward), ptr(discount),
               ptr(step_type), ptr(weights), B, T, A, float(gamma), float(gamma_loss),
               float(reward_scale), int(loss_kind), float(global_batch), ptr(loss_out),
               ptr(td_loss_out), ptr(td_error_out), ptr(dq_out), ptr(field_sums_out))
    if head is None:
        check(lib.aa_dqn_td_loss_sums(*common_, stream_ptr()), "aa_dqn_td_loss_sums")
        return
    x, w, dx, dw = head["x"], head["w"], head["dx"], head["dw"]
    mask_src = head.get("mask_src")
    require_cuda(x, w, dx, dw, mask_src)
    lda = _rows_ok(x, "x"); _f32c(w, "w"); _f32c(dx, "dx"); _f32c(dw, "dw")
    K = x.shape[1]
    if x.shape[0] != B or tuple(w.shape) != (K, A) or tuple(dx.shape) != (B, K) or \
            tuple(dw.shape) != (K, A) or not dense_loss_head_ok(B, A, mask_src):
        raise ValueError("dqn_td_loss(head=...): shapes do not describe the Q head")
    check(lib.aa_dqn_loss_head_backward(
        *common_, ptr(x), lda, ptr(w), ptr(mask_src),
        ACT[head.get("mask_act")] if mask_src is not None else 0, K, ptr(dx), ptr(dw),
        _bias_grad_ptr(head.get("bias_grad"), A), stream_ptr()), "aa_dqn_loss_head_backward")


def dense_loss_head_ok(B, A, mask_src):
    return (FUSE_LOSS_HEAD and USE_SMALL_N and A <= SMALL_N and B <= SMALL_DW_MAX_M and
            (mask_src is None or mask_src.is_contiguous()))


__all__ = [n for n in dir() if not n.startswith("_")]
