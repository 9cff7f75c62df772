"""Sequential: a feed-forward stack (Rescale / Conv2D / Flatten / Dense) executed by HIP kernels.

API follows tf_agents/networks/sequential.py:294 (a Network built from a list of layers, used by
agents/dqn/examples/v2/train_eval.py:168,364 and examples/dqn/mnih15/dqn_train_eval_atari.py:100).
There is no autograd: `forward` caches each layer's output, `backward` walks the stack in reverse
issuing the weight-gradient / input-gradient GEMMs of csrc/gemm.hip (what tf.GradientTape does for
keras Dense / Conv2D in tf_agents/agents/dqn/dqn_agent.py:412-426).

Memory: all parameters of the network live in ONE flat fp32 buffer (`flat_params`), gradients in a
second one of identical layout (`flat_grads`) -- kernel then bias per layer, every tensor starting
on a 16-byte boundary -- so the optimizer, the target-network update and the RCCL gradient
all-reduce are each a single pass over one contiguous range.
"""
import ctypes

import numpy as np
import os

import torch

from agents_amd import _lib, ops
from agents_amd.networks import layers as L
from agents_amd.networks import network
from agents_amd.utils import nest_utils


SMALL_HEAD_ON_MAIN = True
FUSE_HEAD_BACKWARD = True   # dX + dW of a small head: one launch
FUSED_SMALL_MLP = True   # whole <=64-wide MLPs in one forward / one backward launch
# whole <=256-wide MLPs (SAC's actor / critics) at batch <= 1024: one forward launch, two backward
# launches, several networks of one layout per launch (csrc/mlp_wide.hip).  AA_FUSED_WIDE_MLP=0:
# one GEMM launch per layer and direction instead (A/B measurements)
FUSED_WIDE_MLP = True
DX_FIRST = True   # record a layer's input-gradient launch before its weight-gradient launch
# the first layer's weight gradient on the main stream (AA_LAST_DW_ON_MAIN=0: on the side stream)
LAST_DW_ON_MAIN = True


# conv -> conv over LDS-sized fp32 frames in one launch (csrc/conv_pair.hip); AA_FUSE_CONV_PAIRS=0
# selects the layer-by-layer kernels (A/B measurements)
FUSE_CONV_PAIRS = True
# The bf16x6 convolutions split their filter banks in a pre-pass that depends on the weights only.
# It can be issued on the network's own side stream ahead of the kernel that needs it:
#   AA_HOIST_PREP=3 (default)  backward only: the two input-gradient pre-passes run next to the
#                              dense layers' backward instead of on the dX chain
#   AA_HOIST_PREP=1            forward too (next to the layer in front of the pair)
#   AA_HOIST_PREP=0            nowhere (each kernel call splits for itself)
# Measured inside the DQN iteration on MI355X (same box, alternating runs): 3 = 0.421 ms,
# 0 = 0.461 ms, 1 = 0.502 ms -- the forward fork adds a graph branch at the point where three
# forward chains (collect, online, target) already compete for the four hardware queues.
_HOIST = "3"
HOIST_PREP = _HOIST != "0"
_HOIST_FWD = _HOIST in ("1", "2")
_HOIST_BWD = _HOIST in ("1", "3")
# AA_DX_PREP_LATE=1 (default): the backward pre-passes fork from the start of the backward pass but
# are RECORDED behind the first input-gradient launch, so that under graph capture the chain's
# first kernel -- not a pre-pass -- is the first child of the loss node and stays on its queue
# (rocprofv3 timeline: fc1's dX ran on another hardware queue than the loss before it and conv3's
# dX after it, ~10 us of cross-queue hand-over each way)
DX_PREP_LATE = True
# (Tried on top of it and not kept: conv3's weight gradient waiting for the fork of conv2's -- one
# fork of the weight-gradient branch from the input-gradient chain for the pair instead of one per
# layer, a superset of its dependencies: 0.3158 vs 0.3128 ms, three alternating pairs.)
# (The pre-passes on the weight-gradient side stream instead of a stream of their own: 0.427 vs
# 0.360 ms -- the join in front of the first conv dX then also waits for fc1's weight gradient.)


# Prepared weights (opt-in per network, `Sequential.enable_prepared_weights`): the weights-only
# pre-passes of the bf16x6 convolutions (filter split of a fused pair, fragments + tables of the
# conv input gradients) are run by whoever WRITES the weights -- the agent's optimizer step, a
# target update, a restore -- into buffers owned by the network, and every forward / backward that
# reads those weights (policy forward of the collect graph, online forward, backward) skips its
# own pre-pass.  Round 2 refreshed them with the pre-pass LAUNCHES after the optimizer step and lost
# 12 % (0.404 vs 0.359 ms, same box): three launches on the one point of the iteration that every
# lane waits for (theta_k+1), against per-call pre-passes that run as first nodes of three parallel
# branches.  Round 3: the optimizer kernel itself writes the three bf16 pieces of every new filter
# value to its place in every plane set (`plane_scatter`, csrc/optim.hip: aa_*_step_planes) -- no
# launch at all, five fewer kernels and one fewer graph branch per DQN iteration -- so DqnAgent now
# opts in by default (AA_PREPARED_WEIGHTS=0: every call splits for itself again).
PREPARED_WEIGHTS = True
# Which plane sets are prepared: "pair" = the fused conv pair's forward filters (default), "dx" =
# the conv input gradients' fragments (opt-in, AA_PW_KINDS=pair,dx).  Measured in the DQN iteration
# on MI355X, alternating runs on one box (tools/ab_matrix.py): none 0.3745 ms, pair 0.3739,
# pair+dx 0.4003, dx 0.4035 -- with the three streams of the default loop the two backward
# pre-passes are worth keeping as a side BRANCH of the train graph (without it the HIP-graph
# executor schedules the weight-gradient branch worse), although they are pure overhead on a
# single stream (`bench.py --no-overlap`: none 0.4058, pair+dx 0.3703).  The forward pair planes
# remove three launches (and 7 us of host time per iteration) at no cost either way.
_PW_KINDS = ("pair",)
_PREPARED_NETS = []      # weak references to the networks that opted in


def ensure_prepared():
    """Re-runs the pre-passes of every opted-in network whose weights were written behind its back
    (torch in-place ops bump the parameter tensor's version).  Called by the HIP-graph wrappers
    before a replay: a captured forward reads the prepared buffers unconditionally."""
    dead = False
    for ref in _PREPARED_NETS:
        net = ref()
        if net is None:
            dead = True
        elif net._pw is not None and net.flat_params._version != net._pw["torch_version"]:
            net.refresh_prepared()
    if dead:
        _PREPARED_NETS[:] = [r for r in _PREPARED_NETS if r() is not None]


def _align4(n):
    return (n + 3) // 4 * 4


class _Slot:
    """Per-(slot, batch) activation and gradient buffers."""

    def __init__(self):
        self.xs = []      # input of each parametrised layer (tensor views)
        self.ys = []      # output (post-activation) of each parametrised layer
        self.dxs = []     # gradient wrt input of each parametrised layer (None for the first)
        self.dcol = None  # column-gradient scratch for conv input gradients
        self.dz_top = None
        self.pair_prep = None   # {param index of a fused pair's first conv: filter-plane scratch}
        self.prep_issued = None  # pairs whose pre-pass prepare_forward already put on the prep stream
        self.dx_prep = None     # {param index of a conv: input-gradient filter scratch}
        self.dzs = None         # wide-MLP path: d loss / d pre-activation of each layer
        self.wide_in = None     # wide-MLP path: (x, x2) the last forward read its input from


class Sequential(network.Network):
    selected = False     # the last forward's head launch also selected the actions (forward(select=))

    def __init__(self, layers, input_spec=None, name=None, seed=None):
        super().__init__(input_tensor_spec=input_spec, state_spec=(), name=name or "Sequential")
        if not layers:
            raise ValueError("`layers` must not be empty")
        for l in layers:
            if not isinstance(l, L.Layer):
                raise TypeError(
                    f"Sequential layers must be agents_amd.networks.layers objects, got {l!r}")
        self._layers = list(layers)
        self._seed = seed
        self._slots = {}
        self.flat_params = None
        self.flat_grads = None
        self._param_layers = [l for l in self._layers if l.has_params]
        self._shapes = None     # [(kernel_shape, bias_shape)]
        self._offsets = None    # [(k_off, b_off)]
        self._kviews = self._bviews = self._gkviews = self._gbviews = None
        self._reg_scratch = None
        self._pw = None         # prepared weights (enable_prepared_weights)
        self._kept_ws = None    # scratch of kept weight-gradient slabs (backward(keep_dw_slabs=True))
        # forward slots that take the three-conv launch (None = every slot; an empty set = none):
        # the launch owns a CU for its whole duration (153 KB of LDS), which costs a forward that
        # runs beside other lanes' kernels more than one that runs alone -- DqnAgent decides

    # ---- construction -------------------------------------------------------------------------
    @property
    def layers(self):
        return list(self._layers)

    def _infer(self, input_shape):
        """Walks the stack: per param layer (kernel shape, bias shape, in shape, out shape)."""
        shape = tuple(input_shape)
        info = []
        for l in self._layers:
            if isinstance(l, L.Rescale):
                continue
            if isinstance(l, L.Flatten):
                shape = (int(np.prod(shape)),)
            elif isinstance(l, L.Conv2D):
                if len(shape) != 3:
                    raise ValueError(f"Conv2D needs an [H,W,C] input, got {shape}")
                H, W, C = shape
                kh, kw = l.kernel_size
                oh, ow = ops.conv_out_hw(H, W, kh, kw, l.stride)
                info.append(((kh, kw, C, l.filters), (l.filters,), shape, (oh, ow, l.filters)))
                shape = (oh, ow, l.filters)
            elif isinstance(l, L.Dense):
                k = int(np.prod(shape))
                info.append(((k, l.units), (l.units,), (k,), (l.units,)))
                shape = (l.units,)
        return info, shape

    def create_variables(self, input_tensor_spec=None, device=None, **kwargs):
        if self._built:
            return self._output_spec_shape
        if input_tensor_spec is not None:
            self._input_tensor_spec = input_tensor_spec
        spec = self._input_tensor_spec
        if spec is None:
            raise ValueError("create_variables needs an input_tensor_spec")
        if nest_utils.is_nested(spec):
            raise NotImplementedError("Sequential takes a single-tensor observation")
        dev = torch.device(device) if device is not None else torch.device("cuda")
        info, out_shape = self._infer(spec.shape)
        self._info = info
        self._output_spec_shape = out_shape
        offs, total = [], 0
        for ks, bs, _, _ in info:
            k_off = total
            total = _align4(total + int(np.prod(ks)))
            b_off = total
            total = _align4(total + int(np.prod(bs)))
            offs.append((k_off, b_off))
        self._offsets = offs
        self._shapes = [(ks, bs) for ks, bs, _, _ in info]
        rng = np.random.default_rng(self._seed)
        host = np.zeros((max(total, 4),), np.float32)
        for l, (ks, bs, _, _), (k_off, b_off) in zip(self._param_layers, info, offs):
            fan_in = int(np.prod(ks[:-1]))
            fan_out = int(ks[-1]) * (int(np.prod(ks[:-2])) if len(ks) == 4 else 1)
            host[k_off:k_off + int(np.prod(ks))] = l.kernel_initializer(
                ks, rng, fan_in, fan_out).reshape(-1)
            host[b_off:b_off + int(np.prod(bs))] = l.bias_initializer(
                bs, rng, fan_in, fan_out).reshape(-1)
        self.flat_params = torch.from_numpy(host).to(dev)
        self.flat_grads = torch.zeros_like(self.flat_params)
        self._make_views()
        self._built = True
        return out_shape

    def _make_views(self):
        def views(flat):
            kv, bv = [], []
            for (ks, bs), (k_off, b_off) in zip(self._shapes, self._offsets):
                kv.append(flat[k_off:k_off + int(np.prod(ks))].view(ks))
                bv.append(flat[b_off:b_off + int(np.prod(bs))].view(bs))
            return kv, bv
        self._kviews, self._bviews = views(self.flat_params)
        self._gkviews, self._gbviews = views(self.flat_grads)

    def rebind(self, flat_params, flat_grads):
        """Moves the parameters into caller-provided storage (two fp32 views of identical length
        `flat_params.numel()` == this network's flat size).  Lets an agent keep several networks
        in ONE flat parameter / gradient buffer so that the optimizer, the global-norm clip and
        the gradient all-reduce are single passes (PPO: actor + value)."""
        self._require_built()
        n = self.flat_params.numel()
        if flat_params.numel() != n or flat_grads.numel() != n:
            raise ValueError(f"rebind needs views of {n} elements")
        if flat_params.dtype != torch.float32 or not flat_params.is_contiguous():
            raise ValueError("rebind needs contiguous float32 storage")
        flat_params.copy_(self.flat_params)
        flat_grads.zero_()
        self.flat_params = flat_params
        self.flat_grads = flat_grads
        self._make_views()

    @property
    def flat_size(self):
        self._require_built()
        return self.flat_params.numel()

    @property
    def kernels(self):
        """Kernel (weight-matrix) views, for L2 regularisation over the weights only."""
        self._require_built()
        return list(self._kviews)

    @property
    def kernel_grads(self):
        self._require_built()
        return list(self._gkviews)

    @property
    def variables(self):
        self._require_built()
        out = []
        for k, b in zip(self._kviews, self._bviews):
            out += [k, b]
        return out

    @property
    def gradients(self):
        self._require_built()
        out = []
        for k, b in zip(self._gkviews, self._gbviews):
            out += [k, b]
        return out

    @property
    def num_params(self):
        return sum(int(np.prod(ks)) + int(np.prod(bs)) for ks, bs in self._shapes)

    def segment_offsets(self):
        """[start, end) ranges of every variable inside the flat buffers (for per-tensor clips)."""
        segs = []
        for (ks, bs), (k_off, b_off) in zip(self._shapes, self._offsets):
            segs.append((k_off, k_off + int(np.prod(ks))))
            segs.append((b_off, b_off + int(np.prod(bs))))
        return segs

    def _require_built(self):
        if not self._built:
            raise RuntimeError("network variables do not exist yet; call create_variables(spec)")

    def copy(self, **kwargs):
        new = Sequential(self._layers, input_spec=self._input_tensor_spec,
                         name=kwargs.get("name", self._name), seed=self._seed)
        if self._built:
            new.create_variables(self._input_tensor_spec, device=self.flat_params.device)
            new.flat_params.copy_(self.flat_params)
        return new

    def set_weights(self, arrays):
        """Assign variables from a list of numpy arrays in `variables` order (tests / restore)."""
        self._require_built()
        for v, a in zip(self.variables, arrays):
            v.copy_(torch.as_tensor(np.asarray(a, np.float32)).reshape(v.shape))

    def get_weights(self):
        return [v.detach().cpu().numpy() for v in self.variables]

    # ---- regularisation (keras kernel_regularizer=l2) ---------------------------------------------
    @property
    def has_regularization(self):
        return any(getattr(l, "l2", 0.0) > 0 for l in self._param_layers)

    def regularization_loss(self):
        """Device scalar sum_l l2_l * sum(W_l^2) (keras regularizers.l2)."""
        lib = _lib.load()
        dev = self.flat_params.device
        total = None
        for l, k in zip(self._param_layers, self._kviews):
            l2 = getattr(l, "l2", 0.0)
            if l2 <= 0:
                continue
            s = torch.empty((1,), dtype=torch.float32, device=dev)
            _lib.check(lib.aa_sumsq_f32(k.data_ptr(), k.numel(), s.data_ptr(), _lib.stream_ptr()),
                       "aa_sumsq_f32")
            s = s * l2
            total = s if total is None else total + s
        return total

    def add_regularization_grads(self, scale=1.0):
        lib = _lib.load()
        for l, k, g in zip(self._param_layers, self._kviews, self._gkviews):
            l2 = getattr(l, "l2", 0.0)
            if l2 > 0:
                _lib.check(lib.aa_add_l2_grad(g.data_ptr(), k.data_ptr(), k.numel(),
                                              2.0 * l2 * scale, _lib.stream_ptr()),
                           "aa_add_l2_grad")

    @property
    def losses(self):
        r = self.regularization_loss() if self._built and self.has_regularization else None
        return [] if r is None else [r]

    # ---- execution -----------------------------------------------------------------------------
    def _slot(self, slot, B, need_grad):
        key = (slot, B)
        s = self._slots.get(key)
        dev = self.flat_params.device
        if s is None:
            s = _Slot()
            for ks, bs, in_shape, out_shape in self._info:
                s.ys.append(torch.empty((B,) + tuple(out_shape), dtype=torch.float32, device=dev))
            s.xs = [None] * len(self._info)
            s.dxs = [None] * len(self._info)
            self._slots[key] = s
        if need_grad and s.dz_top is None:
            dcol = 0
            for i, (ks, bs, in_shape, out_shape) in enumerate(self._info):
                if i == 0:
                    continue
                s.dxs[i] = torch.empty((B,) + tuple(in_shape), dtype=torch.float32, device=dev)
                if len(ks) == 4:
                    dcol = max(dcol, B * out_shape[0] * out_shape[1] * ks[0] * ks[1] * ks[2])
            s.dcol = torch.empty((max(dcol, 1),), dtype=torch.float32, device=dev)
            s.dz_top = torch.empty((B,) + tuple(self._info[-1][3]), dtype=torch.float32,
                                   device=dev)
        return s

    def forward(self, x, slot=0, need_grad=False, out=None):
        """Runs the stack on x [B, *input_shape]; returns the last layer's output buffer
        (owned by the network, overwritten by the next forward on the same slot and batch).
        `out` (inference only; contiguous float32 [B, units]): receives the output instead -- the
        small-MLP launch writes it directly, the other paths copy."""
        if out is not None:
            if need_grad:
                raise ValueError("forward(out=...) is for inference (the backward pass reads the "
                                 "network's own output buffer)")
            B_ = x.shape[0]
            if self._built and self._fused_small_ok() and x.dtype == torch.float32 and \
                    out.is_contiguous() and out.dtype == torch.float32 and \
                    tuple(out.shape) == (B_, self._shapes[-1][0][1]):
                _lib.require_cuda(x)
                return self._forward_fused(x, self._slot(slot, B_, False), B_, out=out)
            out.copy_(self.forward(x, slot=slot))
            return out
        self._require_built()
        _lib.require_cuda(x)
        spec = self._input_tensor_spec
        if tuple(x.shape[1:]) != tuple(spec.shape):
            raise ValueError(f"network input has shape {tuple(x.shape)}, expected [B]+"
                             f"{tuple(spec.shape)}")
        B = x.shape[0]
        s = self._slot(slot, B, need_grad)
        if self._fused_small_ok() and x.dtype == torch.float32:
            return self._forward_fused(x, s, B)
        if self.wide_ok(B) and x.dtype == torch.float32:
            return forward_wide([self], [x], slot=slot, need_grad=need_grad)[0]
        s.wide_in = None
        cur = x
        div = None
        pi = 0
        skip = 0
        pw_pair = self._pw["pair"] if (self._prepared_ok() and self._pw["pair"]) else None
        prep_pending, s.prep_issued = s.prep_issued, None
        if prep_pending is None and pw_pair is None:
            prep_pending = self._hoist_pair_prep(s, B)
        for li, l in enumerate(self._layers):
            if skip:        # second conv of a fused pair / the head of a dense tail: computed
                skip -= 1
                cur = s.ys[pi]
                pi += 1
                continue
            if isinstance(l, L.Rescale):
                div = l.divisor
            elif isinstance(l, L.Flatten):
                cur = cur.reshape(B, -1)
            elif isinstance(l, L.Conv2D):
                if cur.dtype == torch.uint8:
                    a_div = div if div is not None else 1.0
                elif div is not None:
                    raise NotImplementedError("Rescale is fused only for uint8 inputs")
                else:
                    a_div = 1.0
                div = None
                s.xs[pi] = cur
                nxt = self._layers[li + 1] if li + 1 < len(self._layers) else None
                if (FUSE_CONV_PAIRS and cur.dtype == torch.float32 and isinstance(nxt, L.Conv2D)
                        and cur.data_ptr() % 16 == 0 and cur.stride(0) % 4 == 0
                        and ops.conv_pair_supported(cur.shape, self._kviews[pi], l.stride,
                                                    self._kviews[pi + 1], nxt.stride)):
                    # two convs over frames that fit LDS: one launch, one workgroup per frame
                    s.xs[pi + 1] = s.ys[pi]
                    prepared = s.pair_prep.get(pi) if prep_pending is not None else None
                    if pw_pair is not None and pi in pw_pair:
                        prepared = pw_pair[pi]      # split by whoever wrote the weights
                    elif prepared is not None and prep_pending:
                        torch.cuda.current_stream(cur.device).wait_stream(self._prep_stream)
                        prep_pending.clear()
                    # (the middle activation is stored only on slots a backward pass reads)
                    keep = need_grad or s.dz_top is not None or \
                        ops.conv_pair_prepare_bytes(tuple(cur.shape), self._kviews[pi], l.stride,
                                                    self._kviews[pi + 1], nxt.stride) <= 0
                    ops.conv_pair_forward(cur, self._kviews[pi], self._bviews[pi], l.stride,
                                          l.activation, s.ys[pi] if keep else None,
                                          self._kviews[pi + 1],
                                          self._bviews[pi + 1], nxt.stride, nxt.activation,
                                          s.ys[pi + 1], prepared=prepared)
                    skip = 1
                else:
                    ops.conv_forward(cur, self._kviews[pi], self._bviews[pi], l.stride,
                                     l.activation, s.ys[pi], a_div=a_div)
                cur = s.ys[pi]
                pi += 1
            elif isinstance(l, L.Dense):
                if cur.dtype != torch.float32 or div is not None:
                    raise NotImplementedError(
                        "Dense needs float32 inputs (uint8 / Rescale inputs are fused into a "
                        "leading Conv2D only)")
                cur2 = cur.reshape(B, -1)
                s.xs[pi] = cur2
                nxt = self._layers[li + 1] if li + 1 < len(self._layers) else None
                if (isinstance(nxt, L.Dense) and pi + 1 < len(self._kviews)
                        and ops.dense_tail_supported(cur2, self._kviews[pi], self._kviews[pi + 1])):
                    # hidden layer + small head: the head sums the hidden layer's split-K slabs
                    s.xs[pi + 1] = s.ys[pi]
                    ops.dense_tail_forward(
                        cur2, self._kviews[pi], self._bviews[pi], l.activation, s.ys[pi],
                        self._kviews[pi + 1], self._bviews[pi + 1], nxt.activation, s.ys[pi + 1])
                    skip = 1
                else:
                    ops.dense_forward(cur2, self._kviews[pi], self._bviews[pi], l.activation,
                                      s.ys[pi])
                cur = s.ys[pi]
                pi += 1
        if prep_pending:   # a planned pair was not reached (cannot happen; keeps captures joined)
            torch.cuda.current_stream(cur.device).wait_stream(self._prep_stream)
        return cur

    # ---- weights-only pre-passes of the bf16x6 convolutions, off the critical chain -----------
    def _prep(self, device):
        st = getattr(self, "_prep_stream", None)
        if st is None:
            st = self._prep_stream = ops.new_side_stream(device)
        return st

    def _conv_param_pairs(self):
        """Param indices pi such that parametrised layers pi, pi+1 are adjacent Conv2D layers."""
        out = []
        pi = 0
        for li, l in enumerate(self._layers):
            if not l.has_params:
                continue
            nxt = self._layers[li + 1] if li + 1 < len(self._layers) else None
            if isinstance(l, L.Conv2D) and isinstance(nxt, L.Conv2D):
                out.append(pi)
            pi += 1
        return out

    def enable_prepared_weights(self):
        """Opts this network into prepared weights (see PREPARED_WEIGHTS above).  The caller takes
        over the duty of calling `refresh_prepared()` after every write to the parameters that does
        not go through a torch in-place op (optimizer / soft-update kernels)."""
        import weakref
        self._require_built()
        if self._pw is not None or self._fused_small_ok():
            return self._pw is not None
        dev = self.flat_params.device
        pair, dx = {}, {}
        if FUSE_CONV_PAIRS and "pair" in _PW_KINDS:
            for pi in self._conv_param_pairs():
                if pi in pair or (pi - 1) in pair:
                    continue
                l, nxt = self._param_layers[pi], self._param_layers[pi + 1]
                shape = (1,) + tuple(self._info[pi][2])
                w1, w2 = self._kviews[pi], self._kviews[pi + 1]
                if pi > 0 and int(np.prod(shape[1:])) % 4 == 0 and \
                        ops.conv_pair_supported(shape, w1, l.stride, w2, nxt.stride):
                    n = ops.conv_pair_prepare_bytes(shape, w1, l.stride, w2, nxt.stride)
                    if n > 0:
                        pair[pi] = torch.empty((n,), dtype=torch.uint8, device=dev)
        for i, l in enumerate(self._param_layers):
            if i == 0 or not isinstance(l, L.Conv2D) or "dx" not in _PW_KINDS:
                continue
            n = ops.conv_dx_prepare_bytes((1,) + tuple(self._info[i][2]), self._kviews[i], l.stride)
            if n > 0:
                dx[i] = torch.empty((n,), dtype=torch.uint8, device=dev)
        if not pair and not dx:
            return False
        self._pw = {"pair": pair, "dx": dx, "torch_version": -1, "scatter": None}
        _PREPARED_NETS.append(weakref.ref(self))
        self.refresh_prepared()
        self._pw["scatter"] = self._build_plane_scatter()
        return True

    def _build_plane_scatter(self):
        """Where the optimizer has to put the bf16 pieces of every filter value so that the
        prepared planes stay current without pre-pass launches (csrc/optim.hip).  The layouts are
        read off the pre-pass kernels themselves: the parameters are set to 1, 2, 3, ... (exact in
        three bf16 pieces below 2^24), the pre-passes run, and hi + mid + lo at every plane position
        names the parameter that lives there.  Returns a _lib.PlaneScatter (tables kept alive in
        self._pw), or None when the layout cannot be expressed (then the agent falls back to the
        pre-pass launches)."""
        pw = self._pw
        n = self.flat_params.numel()
        targets = []
        for kind in ("pair", "dx"):
            for pi, ws in pw.get(kind, {}).items():
                nw = int(np.prod(self._shapes[pi][0]))
                if kind == "pair":
                    nw += int(np.prod(self._shapes[pi + 1][0]))
                targets.append((ws, nw))
        if not targets or len(targets) > 4 or n >= (1 << 24):
            return None
        saved = self.flat_params.clone()
        try:
            self.flat_params.copy_(torch.arange(1, n + 1, dtype=torch.float32,
                                                device=self.flat_params.device))
            self.refresh_prepared()
            desc = _lib.PlaneScatter()
            tables = []
            for t, (ws, nw) in enumerate(targets):
                # fragment layout of both pre-passes: [tile][plane][lane][8] bf16 -> 512 per plane
                if (nw * 6) % (3 * 512 * 2) != 0 or ws.numel() < nw * 6:
                    return None
                frag = ws[:nw * 6].view(torch.bfloat16).view(-1, 3, 512).to(torch.float32)
                src = (frag[:, 0] + frag[:, 1] + frag[:, 2]).round().to(torch.int64) - 1
                tiles = src.shape[0]
                pos = (torch.arange(tiles, device=src.device)[:, None] * 1536 +
                       torch.arange(512, device=src.device)[None, :])
                valid = src >= 0
                srcv, posv = src[valid], pos[valid]
                if srcv.numel() != nw or torch.unique(srcv).numel() != nw:
                    return None            # not one plane position per weight
                lo, hi = int(srcv.min()), int(srcv.max()) + 1
                table = torch.full((hi - lo,), -1, dtype=torch.int32, device=src.device)
                table[srcv - lo] = posv.to(torch.int32)
                tables.append(table)
                desc.stride[t] = 512
                desc.lo[t], desc.hi[t] = lo, hi
                desc.pos[t] = table.data_ptr()
                desc.planes[t] = ws.data_ptr()
            desc.n = len(targets)
            pw["scatter_tables"] = tables
            return desc
        finally:
            self.flat_params.copy_(saved)
            self.refresh_prepared()

    def plane_scatter(self):
        """The descriptor an optimizer's apply_flat(..., planes=) takes to keep the prepared planes
        current, or None (no prepared weights / they are stale: somebody wrote the parameters with
        a torch op and the next refresh_prepared() has not run yet)."""
        pw = self._pw
        if pw is None or pw.get("scatter") is None or not self._prepared_ok():
            return None
        return pw["scatter"]

    def refresh_prepared(self):
        """Runs the pre-passes for the CURRENT weights on the caller's stream (capturable)."""
        pw = self._pw
        if pw is None:
            return
        for pi, ws in pw["pair"].items():
            l, nxt = self._param_layers[pi], self._param_layers[pi + 1]
            ops.conv_pair_prepare((1,) + tuple(self._info[pi][2]), self._kviews[pi], l.stride,
                                  self._kviews[pi + 1], nxt.stride, ws)
        for i, ws in pw["dx"].items():
            ops.conv_dx_prepare((1,) + tuple(self._info[i][2]), self._kviews[i],
                                self._param_layers[i].stride, ws)
        pw["torch_version"] = self.flat_params._version

    def _prepared_ok(self):
        pw = self._pw
        return pw is not None and pw["torch_version"] == self.flat_params._version

    def prepare_forward(self, B, slot=0, need_grad=False):
        """Issues, from the CALLER's stream, the weights-only pre-passes of the next
        `forward(x[B], slot)`: for a forward that will itself run on a side line (the DQN target
        network), where forking a second time is not possible under graph capture."""
        self._require_built()
        s = self._slot(slot, B, need_grad)
        if s.prep_issued is None and not self._fused_small_ok() and not self._prepared_ok():
            s.prep_issued = self._hoist_pair_prep(s, B)

    def _hoist_pair_prep(self, s, B):
        """Issues the filter split of every fused conv pair of this forward on the prep stream and
        returns the list of pending pairs (None: nothing hoisted, the pair call splits itself)."""
        if not (HOIST_PREP and _HOIST_FWD and FUSE_CONV_PAIRS):
            return None
        if ops._LINE is not None:
            # Already on a side line: that line may have joined a graph capture through an event
            # recorded before the capture had any node, and forking again from such a point makes
            # ROCm 7's hipStreamEndCapture crash.  Callers that run a forward on a side line issue
            # the pre-pass from the origin stream first (prepare_forward).
            return None
        dev = self.flat_params.device
        if s.pair_prep is None:
            if torch.cuda.is_current_stream_capturing():
                return None     # planned (and its scratch allocated) by an eager call only
            plan = {}
            for pi in self._conv_param_pairs():
                if pi == 0 or pi in plan or (pi - 1) in plan:
                    continue    # a pair fed by the caller's tensor has nothing to overlap with
                l, nxt = self._param_layers[pi], self._param_layers[pi + 1]
                shape = (B,) + tuple(self._info[pi][2])
                w1, w2 = self._kviews[pi], self._kviews[pi + 1]
                if int(np.prod(shape[1:])) % 4 == 0 and \
                        ops.conv_pair_supported(shape, w1, l.stride, w2, nxt.stride):
                    nbytes = ops.conv_pair_prepare_bytes(shape, w1, l.stride, w2, nxt.stride)
                    if nbytes > 0:
                        plan[pi] = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
            s.pair_prep = plan
        if not s.pair_prep:
            return None
        prep = self._prep(dev)
        prep.wait_stream(torch.cuda.current_stream(dev))
        with ops.side_line(prep):
            for pi, ws in s.pair_prep.items():
                l, nxt = self._param_layers[pi], self._param_layers[pi + 1]
                ops.conv_pair_prepare((B,) + tuple(self._info[pi][2]), self._kviews[pi], l.stride,
                                      self._kviews[pi + 1], nxt.stride, ws)
        return list(s.pair_prep)

    def _hoist_dx_prep(self, s, B, hi, lo, fork_event=None):
        """Same for the conv input gradients of layers lo..hi of this backward pass: True when
        something was issued (the caller joins the prep stream before the first of them).
        `fork_event`: the prep stream forks from that (earlier) point of the caller's stream
        instead of from its current one."""
        if not (HOIST_PREP and _HOIST_BWD):
            return False
        dev = self.flat_params.device
        if s.dx_prep is None:
            if torch.cuda.is_current_stream_capturing():
                return False
            plan = {}
            for i, l in enumerate(self._param_layers):
                if i == 0 or not isinstance(l, L.Conv2D):
                    continue
                shape = (B,) + tuple(self._info[i][2])
                nbytes = ops.conv_dx_prepare_bytes(shape, self._kviews[i], l.stride)
                if nbytes > 0:
                    plan[i] = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
            s.dx_prep = plan
        todo = [i for i in s.dx_prep if lo <= i <= hi]
        if not todo:
            return False
        prep = self._prep(dev)
        if fork_event is not None:
            prep.wait_event(fork_event)
        else:
            prep.wait_stream(torch.cuda.current_stream(dev))
        with ops.side_line(prep):
            for i in todo:
                ops.conv_dx_prepare((B,) + tuple(self._info[i][2]), self._kviews[i],
                                    self._param_layers[i].stride, s.dx_prep[i])
        return True

    # ---- fused small-MLP path (csrc/mlp_small.hip) -----------------------------------------------
    def _fused_small_ok(self):
        """Every parametrised layer is Dense, at most 4 of them, every width (and the input) <= 64:
        the whole stack runs as ONE forward launch and ONE backward launch (+ slab reduce)."""
        ok = getattr(self, "_fused_ok", None)
        if ok is None:
            ok = False
            if FUSED_SMALL_MLP and self._built and 1 <= len(self._param_layers) <= 4 and \
                    all(isinstance(l, (L.Dense, L.Flatten)) for l in self._layers):
                n0 = int(np.prod(self._input_tensor_spec.shape))
                widths = [n0] + [ks[1] for ks, _ in self._shapes]
                ok = all(1 <= w_ <= 64 for w_ in widths) and \
                    all(l.activation in (None, "relu", "tanh") for l in self._param_layers)
                if ok:
                    n = len(self._param_layers)
                    self._f_dims = (ctypes.c_int32 * (n + 1))(*widths)
                    self._f_acts = (ctypes.c_int32 * n)(*[ops.ACT[l.activation]
                                                          for l in self._param_layers])
                    self._f_koff = (ctypes.c_int64 * n)(*[o[0] for o in self._offsets])
                    self._f_boff = (ctypes.c_int64 * n)(*[o[1] for o in self._offsets])
                    self._f_ws = {}
            self._fused_ok = ok
        return ok and FUSED_SMALL_MLP

    def wide_layout(self):
        """The aa_mlp_layout of this stack when csrc/mlp_wide.hip can run it (every parametrised
        layer Dense, at most 4, every width <= 256, input <= 1024, activations None / relu / tanh,
        and not already a <=64-wide stack), else None."""
        lay = getattr(self, "_wide_layout", False)
        if lay is False:
            lay = None
            if self._built and 1 <= len(self._param_layers) <= 4 and \
                    all(isinstance(l, (L.Dense, L.Flatten)) for l in self._layers) and \
                    not self._fused_small_ok():
                n0 = int(np.prod(self._input_tensor_spec.shape))
                widths = [n0] + [ks[1] for ks, _ in self._shapes]
                if 1 <= n0 <= 1024 and all(1 <= w_ <= 256 for w_ in widths[1:]) and \
                        all(l.activation in (None, "relu", "tanh") for l in self._param_layers):
                    n = len(self._param_layers)
                    lay = _lib.MlpLayout()
                    lay.n_layers = n
                    for i, w_ in enumerate(widths):
                        lay.dims[i] = w_
                    for i, l in enumerate(self._param_layers):
                        lay.acts[i] = ops.ACT[l.activation]
                        lay.k_off[i] = self._offsets[i][0]
                        lay.b_off[i] = self._offsets[i][1]
            self._wide_layout = lay
        return lay

    def wide_ok(self, B):
        return FUSED_WIDE_MLP and 1 <= B <= 1024 and self.wide_layout() is not None

    def wide_key(self):
        """Two networks can share a wide-MLP launch when this is equal."""
        lay = self.wide_layout()
        if lay is None:
            return None
        n = lay.n_layers
        return (n, tuple(lay.dims[:n + 1]), tuple(lay.acts[:n]), tuple(lay.k_off[:n]),
                tuple(lay.b_off[:n]))

    def _fused_ptrs(self, s, out=None):
        n = len(self._param_layers)
        ptrs = [y.data_ptr() for y in s.ys]
        if out is not None:
            ptrs[-1] = out.data_ptr()
        return (ctypes.c_void_p * n)(*ptrs)

    def _forward_fused(self, x, s, B, out=None):
        x2 = x.reshape(B, -1)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        n = len(self._param_layers)
        s.xs[0] = x2
        for i in range(1, n):
            s.xs[i] = s.ys[i - 1]
        _lib.check(_lib.load().aa_mlp_small_forward(
            x2.data_ptr(), x2.shape[1], self.flat_params.data_ptr(), n, self._f_dims,
            self._f_acts, self._f_koff, self._f_boff, B, self._fused_ptrs(s, out),
            _lib.stream_ptr()), "aa_mlp_small_forward")
        return s.ys[-1] if out is None else out

    def _backward_fused(self, dout, s, B, input_grad):
        lib = _lib.load()
        n = len(self._param_layers)
        total = self.flat_grads.numel()
        need = int(lib.aa_mlp_small_workspace_bytes(B, total))
        ws = self._f_ws.get(B)
        if ws is None or ws.numel() < need:
            ws = torch.empty((need,), dtype=torch.uint8, device=self.flat_grads.device)
            self._f_ws[B] = ws
        x2 = s.xs[0]
        _lib.check(lib.aa_mlp_small_backward(
            x2.data_ptr(), x2.shape[1], self.flat_params.data_ptr(), n, self._f_dims,
            self._f_acts, self._f_koff, self._f_boff, B, self._fused_ptrs(s),
            dout.contiguous().data_ptr(), self.flat_grads.data_ptr(), total,
            None if input_grad is None else input_grad.data_ptr(), ws.data_ptr(), ws.numel(),
            _lib.stream_ptr()), "aa_mlp_small_backward")

    def fusable_head(self, B, slot):
        """The operands of the LAST layer's backward pass when it is a small Dense head without
        activation behind another parametrised layer (the Q head): dict(x, w, dx, dw, mask_src,
        mask_act, bias_grad) for a loss kernel that runs the head's backward in its own launch
        (ops.dqn_td_loss(head=...)), else None.  Needs the slot's gradient buffers (a forward with
        need_grad=True, or any earlier backward on this slot)."""
        s = self._slots.get((slot, B))
        n = len(self._param_layers)
        if s is None or s.dz_top is None or n < 2 or self._fused_small_ok():
            return None
        top = self._param_layers[-1]
        if not isinstance(top, L.Dense) or top.activation is not None or s.dxs[n - 1] is None:
            return None
        x = s.ys[n - 2].view(B, -1)
        prev_act = self._param_layers[n - 2].activation
        if not (SMALL_HEAD_ON_MAIN and FUSE_HEAD_BACKWARD and
                ops.dense_loss_head_ok(B, self._shapes[n - 1][0][1], x if prev_act else None)):
            return None
        return dict(x=x, w=self._kviews[n - 1], dx=s.dxs[n - 1].view(B, -1),
                    dw=self._gkviews[n - 1], mask_src=x if prev_act else None, mask_act=prev_act,
                    bias_grad=self._gbviews[n - 1])

    def backward(self, dout, slot=0, side_stream=None, param_grads=True, input_grad=None,
                 stop_layer=0, head_done=False, input_grad_cols=None, keep_dw_slabs=False):
        """Given d loss / d output [B, out], fills flat_grads (overwrites).

        `param_grads=False` skips every weight/bias gradient (only the input-gradient chain runs:
        SAC's actor loss differentiates THROUGH the critics without updating them);
        `input_grad` ([B, in] float32 buffer, Dense first layer only) also receives
        d loss / d network input (`input_grad_cols=(lo, hi)`: only those columns are needed -- a
        critic's action gradient -- the others may be left unwritten).  `head_done=True`: the last layer's backward (its dX into this
        slot's buffer, dW, db) has already been produced by the caller's loss launch
        (`fusable_head`); the walk starts at the layer below.
        `keep_dw_slabs=True`: conv weight gradients that leave split-K slabs are NOT summed into
        flat_grads; `take_grad_slabs()` hands the slabs to an optimizer that sums them itself
        (`optimizers.RMSprop.apply_flat(grad_slabs=...)`) -- for a caller with nothing between
        backward and the optimizer step (no clipping, no all-reduce).
        `stop_layer=k > 0` stops after parametrised layer k (its input gradient is computed); `backward_resume` continues with layers k-1 .. 0 -- the Learner
        starts the all-reduce of the tail's gradients in between (`grad_buckets`).

        The input-gradient chain (dX_n -> dX_{n-1} -> ...) is the critical path; every
        weight/bias-gradient GEMM only needs its layer's dZ, so with `side_stream` they are
        enqueued there (fork after the kernel that produced dZ, join at the end) and overlap with
        the chain.  Works the same eagerly and under HIP-graph capture."""
        B = dout.shape[0]
        s = self._slots.get((slot, B))
        if s is None or s.dz_top is None or s.xs[0] is None:
            raise RuntimeError("backward() needs a preceding forward(..., need_grad=True)")
        if self._fused_small_ok() and param_grads and stop_layer == 0 and \
                s.xs[0].dtype == torch.float32 and s.xs[0].dim() == 2:
            return self._backward_fused(dout, s, B, input_grad)
        if s.wide_in is not None and stop_layer == 0 and not head_done:
            return backward_wide([self], [dout], slot=slot, param_grads=param_grads,
                                 input_grads=None if input_grad is None else [input_grad],
                                 input_grad_cols=input_grad_cols)
        lib = _lib.load()
        n = len(self._param_layers)
        top = self._param_layers[-1]
        self._grad_slabs = None
        # with stop_layer > 0 the kept slabs of this call's layers wait in `_pending_keep` for
        # `backward_resume`, which adds the remaining layers' and builds the aa_grad_slabs
        self._keep_dw_slabs = bool(keep_dw_slabs) and param_grads
        self._keep_resume = self._keep_dw_slabs and stop_layer > 0
        self._pending_keep = None
        if top.activation is not None:
            _lib.check(lib.aa_act_backward(dout.data_ptr(), s.ys[-1].data_ptr(),
                                           ops.ACT[top.activation], dout.numel(),
                                           s.dz_top.data_ptr(), _lib.stream_ptr()),
                       "aa_act_backward")
            dz = s.dz_top
        else:
            dz = dout.contiguous()
        if head_done:
            if n < 2 or stop_layer > n - 1 or not param_grads:
                raise ValueError("head_done needs a head on top of another parametrised layer")
            if stop_layer <= n - 2:
                self._backward_range(s, B, s.dxs[n - 1], n - 2, stop_layer, side_stream,
                                     param_grads, input_grad)
            return
        self._backward_range(s, B, dz, n - 1, stop_layer, side_stream, param_grads, input_grad)

    def take_grad_slabs(self):
        """The unsummed weight-gradient slabs of the last `backward` as an aa_grad_slabs (None
        unless it ran with keep_dw_slabs=True and layers left slabs): the caller's optimizer step
        must consume them.  Valid until the next backward; a HIP-graph replay of a backward pass
        does not run this Python, so a replayer restores what `backward` left at capture time
        (`set_grad_slabs`; utils/graph.py: GraphedTrain)."""
        return getattr(self, "_grad_slabs", None)

    def set_grad_slabs(self, g):
        self._grad_slabs = g

    def backward_resume(self, B, slot=0, side_stream=None, from_layer=1):
        """Continues a `backward(..., stop_layer=from_layer)`: layers from_layer-1 .. 0."""
        s = self._slots[(slot, B)]
        self._keep_dw_slabs = getattr(self, "_pending_keep", None) is not None
        self._keep_resume = False
        self._backward_range(s, B, s.dxs[from_layer], from_layer - 1, 0, side_stream, True, None)

    def grad_buckets(self, split_layer):
        """(tail, head) views of flat_grads: gradients of layers >= split_layer (complete after
        `backward(stop_layer=split_layer)`) and of the layers below."""
        off = self._offsets[split_layer][0]
        return self.flat_grads[off:], self.flat_grads[:off]

    def dense_tail_start(self):
        """Index of the first layer of the trailing run of Dense layers (0 if all are Dense)."""
        k = len(self._param_layers)
        while k > 0 and isinstance(self._param_layers[k - 1], L.Dense):
            k -= 1
        return k

    def _backward_range(self, s, B, dz, hi, lo, side_stream, param_grads, input_grad):
        main = torch.cuda.current_stream(dz.device)
        if side_stream is None:
            side_stream = main

        pw_dx = self._pw["dx"] if (self._prepared_ok() and self._pw["dx"]) else None
        # the pre-passes of the conv input gradients below: issued now, or (DX_PREP_LATE) forked
        # from here but recorded behind the first input-gradient launch of the chain
        prep_fork = None
        late_prep = (pw_dx is None and DX_PREP_LATE and HOIST_PREP and _HOIST_BWD and hi > lo and
                     isinstance(self._param_layers[hi], L.Dense) and hi > 0)
        if late_prep:
            prep_fork = torch.cuda.Event()
            prep_fork.record(main)
            dx_prep_pending = False
        else:
            dx_prep_pending = False if pw_dx is not None else self._hoist_dx_prep(s, B, hi, lo)

        def issue_late_prep():
            nonlocal prep_fork, dx_prep_pending
            if prep_fork is not None:
                dx_prep_pending = self._hoist_dx_prep(s, B, hi, lo, fork_event=prep_fork)
                prep_fork = None

        side_used = False

        def on_side(fn, fork=True):
            nonlocal side_used
            if side_stream is main:
                return fn()
            side_used = True
            if fork:
                side_stream.wait_stream(main)  # dZ of this layer is ready once main gets here
            with ops.side_line(side_stream):
                return fn()

        # conv weight gradients on the side stream leave their slabs unsummed; ONE launch sums them
        # for all layers after the last of them has been enqueued (ops.PendingDwReduce)
        keep = getattr(self, "_keep_dw_slabs", False)
        self._keep_dw_slabs = False
        if keep and getattr(self, "_kept_ws", None) is None:
            self._kept_ws = ops.new_kept_workspaces()     # this network's own (advisor, round 4)
        pending_dw = getattr(self, "_pending_keep", None) if keep else None
        self._pending_keep = None
        if pending_dw is None:
            pending_dw = ops.PendingDwReduce(keep=keep, kept_ws=self._kept_ws if keep else None) \
                if (keep or side_stream is not main) else None

        for i in range(hi, lo - 1, -1):
            l = self._param_layers[i]
            ks = self._shapes[i][0]
            x = s.xs[i]
            prev_act = self._param_layers[i - 1].activation if i > 0 else None
            # The input gradient (critical chain) is enqueued BEFORE the layer's weight gradient:
            # both only read this layer's dZ, and the order in which the two branches are recorded
            # decides which one the HIP-graph executor keeps on the chain's queue.  The side stream
            # forks from main right BEFORE dX(i) is enqueued (main then holds dX(i+1), the producer
            # of this layer's dZ); layer 0 has no dX, so its dW forks on its own.
            if isinstance(l, L.Dense):
                dz2 = dz.view(B, -1)
                dz_next = None
                if i == 0 and input_grad is not None:
                    ops.dense_dx(dz2, self._kviews[0], input_grad.view(B, -1))
                fused_head = (i > 0 and param_grads and SMALL_HEAD_ON_MAIN and FUSE_HEAD_BACKWARD
                              and ops.dense_small_backward_ok(x, dz2, x if prev_act else None))
                if fused_head:
                    # the head's input and weight gradients in one launch (both a few us)
                    if DX_FIRST and side_stream is not main:
                        side_stream.wait_stream(main)
                    ops.dense_small_backward(x, dz2, self._kviews[i], s.dxs[i].view(B, -1),
                                             self._gkviews[i], mask_src=x if prev_act else None,
                                             mask_act=prev_act, bias_grad=self._gbviews[i])
                    issue_late_prep()
                    dz = s.dxs[i]
                    continue
                if i > 0:
                    dx = s.dxs[i].view(B, -1)
                    if DX_FIRST:
                        side_stream.wait_stream(main) if side_stream is not main else None
                    ops.dense_dx(dz2, self._kviews[i], dx, mask_src=x if prev_act else None,
                                 mask_act=prev_act)
                    issue_late_prep()
                    dz_next = s.dxs[i]
                if param_grads:
                    if dz2.shape[1] <= ops.SMALL_N and SMALL_HEAD_ON_MAIN:
                        # a few-microsecond head kernel: a cross-queue fork costs more than it hides
                        ops.dense_dw(x, dz2, self._gkviews[i], bias_grad=self._gbviews[i])
                    else:
                        on_side(lambda: ops.dense_dw(x, dz2, self._gkviews[i],
                                                     bias_grad=self._gbviews[i]),
                                fork=not (DX_FIRST and i > 0))
                if dz_next is not None:
                    dz = dz_next
            else:
                F = ks[3]
                dz2 = dz.view(-1, F)
                if i == 0 and input_grad is not None:
                    raise NotImplementedError("input_grad is implemented for a Dense first layer")
                dz_next = None
                if i > 0:
                    issue_late_prep()
                    if DX_FIRST:
                        side_stream.wait_stream(main) if side_stream is not main else None
                    prepared = s.dx_prep.get(i) if (s.dx_prep and HOIST_PREP) else None
                    if pw_dx is not None:
                        prepared = pw_dx.get(i)
                    elif prepared is not None and dx_prep_pending:
                        main.wait_stream(self._prep_stream)
                        dx_prep_pending = False
                    ops.conv_dx(dz2, self._kviews[i], tuple(x.shape), l.stride, s.dcol, s.dxs[i],
                                mask_src=x if prev_act else None, mask_act=prev_act,
                                prepared=prepared)
                    dz_next = s.dxs[i]
                if param_grads:
                    on_main = i == 0 and LAST_DW_ON_MAIN
                    dw = lambda: ops.conv_dw(x, dz2, ks, l.stride, self._gkviews[i],
                                             a_div=self._first_div() if i == 0 else 1.0,
                                             bias_grad=self._gbviews[i],
                                             defer=pending_dw if (keep or not on_main) else None)
                    if i == 0 and LAST_DW_ON_MAIN:
                        # layer 0 has no input gradient: main has nothing left to do, while the
                        # side stream is still finishing layer 1's weight gradient (timeline:
                        # the launch waited ~15 us behind it for no dependency)
                        dw()
                    else:
                        on_side(dw, fork=not (DX_FIRST and i > 0))
                if dz_next is not None:
                    dz = dz_next
        issue_late_prep()
        if dx_prep_pending:
            main.wait_stream(self._prep_stream)
        if pending_dw is not None and pending_dw.items:
            on_side(lambda: ops.conv_dw_flush(pending_dw), fork=False)
        if keep and getattr(self, "_keep_resume", False):
            self._pending_keep = pending_dw          # backward_resume continues this list
        elif keep:
            self._grad_slabs = ops.grad_slabs(pending_dw, self.flat_grads)
        if side_stream is not main and (side_used or lo > 0 or hi > lo):
            # (a resumed range of layer 0 alone with its weight gradient on main has not touched
            # the side stream: nothing to join -- and under capture nothing that could be joined)
            main.wait_stream(side_stream)

    def _first_div(self):
        for l in self._layers:
            if isinstance(l, L.Rescale):
                return l.divisor
            if l.has_params:
                break
        return 1.0

    def call(self, inputs, step_type=None, network_state=(), training=False, **kwargs):
        out = self.forward(inputs, slot="call")
        return out.clone(), network_state



def small_mlp_layout(net):
    """(flat params pointer, n_layers, dims, acts, k_off, b_off) of a <= 64-wide Dense stack as the
    small-MLP entry points take them (ctypes arrays owned by the network), or None when the stack
    does not qualify (`Sequential._fused_small_ok`)."""
    if not (FUSED_SMALL_MLP and net._built and net._fused_small_ok()):
        return None
    return (net.flat_params.data_ptr(), len(net._param_layers), net._f_dims, net._f_acts,
            net._f_koff, net._f_boff)


# ---- wide MLPs: several networks of one layout per launch (csrc/mlp_wide.hip) ---------------------
# Measurement aid (tools/bench_sac.py): set to a list and every forward_wide call appends
# (networks, batch, layer widths) -- the per-launch work a roofline of the kernel is computed from.
WIDE_FWD_LOG = None


def _wide_group(nets, B):
    if not 1 <= len(nets) <= 4:
        raise ValueError("a wide-MLP launch takes 1..4 networks")
    key = nets[0].wide_key()
    if key is None or any(n.wide_key() != key for n in nets[1:]) or not nets[0].wide_ok(B):
        raise ValueError("forward_wide / backward_wide: the networks do not share a layout "
                         "csrc/mlp_wide.hip supports at this batch size")
    return nets[0].wide_layout()


def forward_wide(nets, xs, slot=0, need_grad=False, x2s=None, slots=None, need_grads=None,
                 sample_tail=None):
    """`net.forward(x, slot, need_grad)` of up to four Sequential networks of the same layout in
    ONE launch.  xs[g]: float32 [B, d] with unit column stride (any row stride); with x2s the
    network input is [xs[g] | x2s[g]] -- a critic's (observation, action) -- read from the two
    tensors in place.  `slots` / `need_grads` (optional lists) give every network its own slot and
    flag -- SAC's critic update evaluates the twin TARGET critics on (next observation, next
    action) and the twin critics on (observation, action) in one launch of four networks.
    `sample_tail` (a filled `_lib.SacSampleTail`, or a pair of them for two networks of the launch
    -- the same actor listed twice with two inputs and slots): network `sample_tail.net` is a SAC actor and the
    launch also draws its tanh-squashed actions and log-probabilities
    (aa_mlp_wide_forward_sample).
    Returns the networks' output buffers (owned by their slots)."""
    B = int(xs[0].shape[0])
    lay = _wide_group(nets, B)
    d = _lib.MlpWideFwd()
    d.layout = lay
    d.n_nets = len(nets)
    d.B = B
    n = lay.n_layers
    outs = []
    for g, net in enumerate(nets):
        x = xs[g] if xs[g].dim() == 2 else xs[g].reshape(B, -1)
        x2 = None
        if x2s is not None:
            x2 = x2s[g] if x2s[g].dim() == 2 else x2s[g].reshape(B, -1)
        for t in (x, x2):
            if t is not None and (t.dtype != torch.float32 or t.stride(1) != 1 or
                                  int(t.shape[0]) != B or not t.is_cuda):
                raise ValueError("wide-MLP inputs are float32 [B, d] device tensors with unit "
                                 "column stride")
        width = int(x.shape[1]) + (int(x2.shape[1]) if x2 is not None else 0)
        if width != lay.dims[0]:
            raise ValueError(f"network input has {width} columns, expected {lay.dims[0]}")
        if g == 0:
            d.x_split = int(x.shape[1])
        elif d.x_split != int(x.shape[1]):
            raise ValueError("every network of a launch splits its input at the same column")
        s = net._slot(slot if slots is None else slots[g], B,
                      need_grad if need_grads is None else need_grads[g])
        s.wide_in = (x, x2)
        s.xs[0] = x
        for i in range(1, n):
            s.xs[i] = s.ys[i - 1]
        d.params[g] = net.flat_params.data_ptr()
        d.x[g] = x.data_ptr()
        d.ldx[g] = x.stride(0)
        if x2 is not None:
            d.x2[g] = x2.data_ptr()
            d.ldx2[g] = x2.stride(0)
        for i in range(n):
            d.y[g][i] = s.ys[i].data_ptr()
        outs.append(s.ys[-1])
    if WIDE_FWD_LOG is not None:
        WIDE_FWD_LOG.append((len(nets), B, [int(lay.dims[i]) for i in range(n + 1)]))
    with torch.cuda.device(xs[0].device):
        if isinstance(sample_tail, (tuple, list)):       # two networks draw (forward_sample2)
            _lib.check(_lib.load().aa_mlp_wide_forward_sample2(
                ctypes.byref(d), ctypes.byref(sample_tail[0]), ctypes.byref(sample_tail[1]),
                _lib.stream_ptr()), "aa_mlp_wide_forward_sample2")
        elif sample_tail is not None:
            _lib.check(_lib.load().aa_mlp_wide_forward_sample(
                ctypes.byref(d), ctypes.byref(sample_tail), _lib.stream_ptr()),
                "aa_mlp_wide_forward_sample")
        else:
            _lib.check(_lib.load().aa_mlp_wide_forward(ctypes.byref(d), _lib.stream_ptr()),
                       "aa_mlp_wide_forward")
    return outs


def backward_wide(nets, douts, slot=0, param_grads=True, input_grads=None, input_grad_cols=None,
                  gen=None, batch=None, adam=None, dw_only=False):
    """`net.backward(dout, slot, param_grads, input_grad)` of the networks of a `forward_wide`
    group: the gradient chain in one launch, every weight / bias gradient of every layer and
    network in a second one (skipped with param_grads=False).  input_grads[g]: [B, d] float32
    buffers receiving d loss / d input columns `input_grad_cols` (default: all).
    `gen` (a filled `_lib.SacDoutGen`; then douts = None and batch = B): the chain launch computes
    d loss / d output itself (aa_mlp_wide_backward_gen: SAC's critic / actor loss, the actor head's
    backward)."""
    B = int(batch) if (gen is not None or dw_only) else int(douts[0].shape[0])
    lay = _wide_group(nets, B)
    d = _lib.MlpWideBwd()
    d.layout = lay
    d.n_nets = len(nets)
    d.B = B
    n = lay.n_layers
    d.dx_lo, d.dx_hi = input_grad_cols if input_grad_cols is not None else (0, lay.dims[0])
    for g, net in enumerate(nets):
        s = net._slots.get((slot, B))
        if s is None or s.wide_in is None:
            raise RuntimeError("backward_wide() needs a preceding forward_wide on this slot")
        if s.dzs is None:
            s.dzs = [torch.empty_like(y) for y in s.ys]
        x, x2 = s.wide_in
        if g == 0:
            d.x_split = int(x.shape[1])
        if gen is None and not dw_only:
            dout = douts[g] if douts[g].dim() == 2 else douts[g].reshape(B, -1)
            if dout.dtype != torch.float32 or dout.stride(1) != 1 or \
                    int(dout.shape[1]) != lay.dims[n]:
                raise ValueError("d loss / d output must be float32 [B, out] with unit column "
                                 "stride")
            d.dout[g] = dout.data_ptr()
            d.ld_dout[g] = dout.stride(0)
        d.params[g] = net.flat_params.data_ptr()
        d.x[g] = x.data_ptr()
        d.ldx[g] = x.stride(0)
        if x2 is not None:
            d.x2[g] = x2.data_ptr()
            d.ldx2[g] = x2.stride(0)
        for i in range(n):
            d.y[g][i] = s.ys[i].data_ptr()
            d.dz[g][i] = s.dzs[i].data_ptr()
        if input_grads is not None:
            ig = input_grads[g]
            if ig.dtype != torch.float32 or ig.stride(1) != 1 or int(ig.shape[1]) < d.dx_hi:
                raise ValueError("input_grad must be a float32 [B, in] buffer")
            d.dx[g] = ig.data_ptr()
            d.ld_dx[g] = ig.stride(0)
        if param_grads:
            d.grads[g] = net.flat_grads.data_ptr()
    if adam is not None:
        # `adam` (Adam.fused_step_desc over the flat buffer the networks' parameters are views
        # of): the weight-gradient launch steps the optimizer itself; every network's slice of
        # m / v / target sits at the offset of its parameters
        if not param_grads:
            raise ValueError("backward_wide(adam=...) needs the weight gradients")
        base = adam.p[0]
        m0, v0, t0 = adam.m[0], adam.v[0], adam.target[0]
        for g, net in enumerate(nets):
            off = net.flat_params.data_ptr() - base
            if off < 0:
                raise ValueError("backward_wide(adam=...): a network outside the flat buffer")
            adam.p[g], adam.m[g], adam.v[g] = base + off, m0 + off, v0 + off
            adam.target[g] = (t0 + off) if t0 else None
    with torch.cuda.device(nets[0].flat_params.device):
        if dw_only:
            # (the gradient chain ran in an earlier call with param_grads=False; batch = B)
            _lib.check(_lib.load().aa_mlp_wide_dw_adam(
                ctypes.byref(d), None if adam is None else ctypes.byref(adam),
                _lib.stream_ptr()), "aa_mlp_wide_dw_adam")
        elif adam is not None:
            _lib.check(_lib.load().aa_mlp_wide_backward_gen_adam(
                ctypes.byref(d), None if gen is None else ctypes.byref(gen), ctypes.byref(adam),
                _lib.stream_ptr()), "aa_mlp_wide_backward_gen_adam")
        elif gen is not None:
            _lib.check(_lib.load().aa_mlp_wide_backward_gen(ctypes.byref(d), ctypes.byref(gen),
                                                            _lib.stream_ptr()),
                       "aa_mlp_wide_backward_gen")
        else:
            _lib.check(_lib.load().aa_mlp_wide_backward(ctypes.byref(d), _lib.stream_ptr()),
                       "aa_mlp_wide_backward")
