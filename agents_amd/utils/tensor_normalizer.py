"""Tensor statistics and normalisation on HIP kernels (csrc/normalizer.hip).

Same classes, arguments and behaviour as tf_agents/utils/tensor_normalizer.py:
  TensorNormalizer            :45-206   update(tensor, outer_dims), normalize(tensor, clip_value,
                                        center_mean, variance_epsilon), variables
  EMATensorNormalizer         :209-285  moving mean / variance, update rate 0.001
  StreamingTensorNormalizer   :288-395  count / avg / m2 / m2_carry over the full history
                                        (Chan's parallel variance + Kahan carry, :397-474), reset()
State lives in one fp32 device tensor per tensor-spec leaf ([4, n] or [2, n]); an update is two
launches, a normalisation one -- all capturable in HIP graphs (persistent scratch).  Variables are
fp32 (the reference maps float64 / int64 specs to float64 variables, :36-42; float64 observations
are outside the hot path and raise).
"""

import torch

from agents_amd import _lib
from agents_amd.utils import nest_utils

_EPS = 1e-8


class TensorNormalizer:
    _ROWS = 0

    def __init__(self, tensor_spec, scope="normalize_tensor", device=None):
        self._scope = scope
        self._tensor_spec = tensor_spec
        self._flat_specs = nest_utils.flatten(tensor_spec)
        for s in self._flat_specs:
            if s.dtype in (torch.float64, torch.int64):
                raise NotImplementedError(
                    "float64 normaliser variables (float64 / int64 specs) are not supported")
        self._device = torch.device(device) if device is not None else torch.device(
            "cuda", torch.cuda.current_device())
        self._n = [max(int(s.num_elements), 1) for s in self._flat_specs]
        self._state = None
        self._scratch = None
        self._create_variables()

    # ---- variables ---------------------------------------------------------------------------
    def _initial_rows(self):
        raise NotImplementedError

    def _create_variables(self):
        lib = _lib.load()
        init = self._initial_rows()
        self._state = []
        for n in self._n:
            st = torch.empty((self._ROWS, n), dtype=torch.float32, device=self._device)
            for r, v in enumerate(init):
                st[r].fill_(v)
            self._state.append(st)
        scratch = max((int(lib.aa_norm_scratch_floats(n)) for n in self._n), default=1)
        self._scratch = torch.empty((scratch,), dtype=torch.float32, device=self._device)

    def _row_nest(self, r):
        return nest_utils.pack_sequence_as(
            self._tensor_spec, [st[r].view(tuple(s.shape)) for st, s in
                                zip(self._state, self._flat_specs)])

    @property
    def variables(self):
        return tuple(self._row_nest(r) for r in range(self._ROWS))

    def state_dict(self):
        return {"state": [st.clone() for st in self._state]}

    def load_state_dict(self, sd):
        for st, src in zip(self._state, sd["state"]):
            st.copy_(src)

    # ---- helpers -----------------------------------------------------------------------------
    def _flat_inputs(self, tensor, what):
        """Per leaf: (fp32 contiguous [n_outer, n] view, outer shape).  Dtype / inner-shape checks
        of nest_utils.assert_matching_dtypes_and_inner_shapes (:119-125, :150-156)."""
        nest_utils.assert_same_structure(tensor, self._tensor_spec)
        out = []
        for t, s, n in zip(nest_utils.flatten(tensor), self._flat_specs, self._n):
            if not isinstance(t, torch.Tensor):
                t = torch.as_tensor(t, dtype=s.dtype, device=self._device)
            _lib.require_cuda(t)
            if t.dtype != s.dtype:
                raise ValueError(f"{what} has dtype {t.dtype}, the tensor_spec says {s.dtype}")
            rank = len(s.shape)
            if rank and tuple(t.shape[t.dim() - rank:]) != tuple(s.shape):
                raise ValueError(f"{what} has inner shape {tuple(t.shape[t.dim() - rank:])}, "
                                 f"the tensor_spec says {tuple(s.shape)}")
            outer = tuple(t.shape[:t.dim() - rank])
            x = t.to(torch.float32).contiguous().view(-1, n)
            out.append((x, outer))
        return out

    def _mean_var_ptrs(self, st):
        """(mean ptr, var numerator ptr, var denominator ptr or None) into the leaf state."""
        raise NotImplementedError

    def _update_leaf(self, lib, x, st, stream):
        raise NotImplementedError

    # ---- API ---------------------------------------------------------------------------------
    _CHECK_OUTER_DIMS = False

    def update(self, tensor, outer_dims=(0,)):
        """Updates the statistics with a batch.  Both normalisers reduce over ALL dims in front of
        the spec's shape: the streaming normaliser ignores `outer_dims` exactly like the reference
        (:325-333); the reference's EMA normaliser reduces over `outer_dims` only
        (tf.reduce_mean(axis=outer_dims), :236-281), so for it anything other than "all leading
        dims" -- e.g. a [B, T, ...] input with outer_dims=(0,) -- is rejected here instead of
        silently computing a different statistic."""
        lib = _lib.load()
        with torch.cuda.device(self._device):
            stream = _lib.stream_ptr()
            for (x, outer), st in zip(self._flat_inputs(tensor, "tensor"), self._state):
                if self._CHECK_OUTER_DIMS and \
                        sorted(int(d) for d in outer_dims) != list(range(len(outer))):
                    raise NotImplementedError(
                        f"EMATensorNormalizer.update reduces over all {len(outer)} leading "
                        f"dim(s) of the input; outer_dims={tuple(outer_dims)} is not supported")
                if x.shape[0] == 0:
                    continue
                self._update_leaf(lib, x, st, stream)

    def normalize(self, tensor, clip_value=5.0, center_mean=True, variance_epsilon=1e-3):
        lib = _lib.load()
        outs = []
        with torch.cuda.device(self._device):
            stream = _lib.stream_ptr()
            for (x, outer), st, s in zip(self._flat_inputs(tensor, "tensors"), self._state,
                                         self._flat_specs):
                y = torch.empty_like(x)
                mean, num, den = self._mean_var_ptrs(st)
                _lib.check(lib.aa_norm_apply(
                    x.data_ptr(), x.shape[0], x.shape[1], mean if center_mean else None, num, den,
                    float(variance_epsilon), float(clip_value), y.data_ptr(), stream),
                    "aa_norm_apply")
                y = y.view(outer + tuple(s.shape))
                outs.append(y if s.dtype == torch.float32 else y.to(s.dtype))
        return nest_utils.pack_sequence_as(self._tensor_spec, outs)


class EMATensorNormalizer(TensorNormalizer):
    """Exponential moving average of mean and variance (:209-285)."""
    _ROWS = 2
    _CHECK_OUTER_DIMS = True

    def __init__(self, tensor_spec, scope="normalize_tensor", norm_update_rate=0.001,
                 device=None):
        self._norm_update_rate = float(norm_update_rate)
        super().__init__(tensor_spec, scope, device)

    def _initial_rows(self):
        return (0.0, 1.0)

    def _mean_var_ptrs(self, st):
        return st[0].data_ptr(), st[1].data_ptr(), None

    def _update_leaf(self, lib, x, st, stream):
        _lib.check(lib.aa_ema_norm_update(x.data_ptr(), x.shape[0], x.shape[1],
                                          self._norm_update_rate, st.data_ptr(),
                                          self._scratch.data_ptr(), stream), "aa_ema_norm_update")


class StreamingTensorNormalizer(TensorNormalizer):
    """Mean and variance over the full history of values (:288-395)."""
    _ROWS = 4

    def _initial_rows(self):
        return (_EPS, 0.0, 0.0, 0.0)

    def _mean_var_ptrs(self, st):
        return st[1].data_ptr(), st[2].data_ptr(), st[0].data_ptr()

    def _update_leaf(self, lib, x, st, stream):
        _lib.check(lib.aa_streaming_norm_update(x.data_ptr(), x.shape[0], x.shape[1],
                                                st.data_ptr(), self._scratch.data_ptr(), stream),
                   "aa_streaming_norm_update")

    def reset(self):
        """count = 1e-8, avg = m2 = m2_carry = 0 (:372-385), in place."""
        for st in self._state:
            st[0].fill_(_EPS)
            st[1:].zero_()
        return []
