"""TensorSpec / BoundedTensorSpec on torch dtypes.

Minimal counterpart of tf_agents/specs/tensor_spec.py (+ array_spec.py:170,271): shape, dtype,
bounds, name.  dtypes are torch dtypes; numpy dtypes and strings are accepted and converted.
"""
import numpy as np
import torch

_NP_TO_TORCH = {
    np.dtype("float32"): torch.float32, np.dtype("float64"): torch.float64,
    np.dtype("int32"): torch.int32, np.dtype("int64"): torch.int64,
    np.dtype("uint8"): torch.uint8, np.dtype("int8"): torch.int8, np.dtype("bool"): torch.bool,
    np.dtype("int16"): torch.int16, np.dtype("float16"): torch.float16,
}
_TORCH_TO_NP = {v: k for k, v in _NP_TO_TORCH.items()}


def as_torch_dtype(dtype):
    if isinstance(dtype, torch.dtype):
        return dtype
    return _NP_TO_TORCH[np.dtype(dtype)]


def as_numpy_dtype(dtype):
    return _TORCH_TO_NP[as_torch_dtype(dtype)]


def is_discrete_dtype(dtype):
    return as_torch_dtype(dtype) in (torch.int32, torch.int64, torch.uint8, torch.int8,
                                     torch.int16)


class TensorSpec:
    """Shape + dtype (+ name) of one tensor, without outer batch/time dims."""

    __slots__ = ("_shape", "_dtype", "_name")

    def __init__(self, shape, dtype=torch.float32, name=None):
        self._shape = tuple(int(d) for d in shape)
        self._dtype = as_torch_dtype(dtype)
        self._name = name

    @property
    def shape(self):
        return self._shape

    @property
    def dtype(self):
        return self._dtype

    @property
    def name(self):
        return self._name

    @property
    def num_elements(self):
        n = 1
        for d in self._shape:
            n *= d
        return n

    @property
    def itemsize(self):
        return torch.empty((), dtype=self._dtype).element_size()

    @property
    def row_bytes(self):
        return self.num_elements * self.itemsize

    def is_compatible_with(self, tensor):
        return tuple(tensor.shape) == self._shape and tensor.dtype == self._dtype

    def __eq__(self, other):
        return (type(self) is type(other) and self._shape == other._shape and
                self._dtype == other._dtype)

    def __hash__(self):
        return hash((self._shape, self._dtype))

    def __repr__(self):
        return f"TensorSpec(shape={self._shape}, dtype={self._dtype}, name={self._name!r})"


class BoundedTensorSpec(TensorSpec):
    """TensorSpec with inclusive [minimum, maximum] bounds (scalars or arrays)."""

    __slots__ = ("_minimum", "_maximum")

    def __init__(self, shape, dtype, minimum, maximum, name=None):
        super().__init__(shape, dtype, name)
        npd = as_numpy_dtype(self._dtype)
        self._minimum = np.asarray(minimum, dtype=npd)
        self._maximum = np.asarray(maximum, dtype=npd)
        try:
            np.broadcast_to(self._minimum, self._shape)
            np.broadcast_to(self._maximum, self._shape)
        except ValueError as e:
            raise ValueError("minimum/maximum are not broadcastable to the spec shape") from e

    @property
    def minimum(self):
        return self._minimum

    @property
    def maximum(self):
        return self._maximum

    def __eq__(self, other):
        return (TensorSpec.__eq__(self, other) and np.array_equal(self._minimum, other._minimum)
                and np.array_equal(self._maximum, other._maximum))

    def __hash__(self):
        return hash((self._shape, self._dtype, self._minimum.tobytes(), self._maximum.tobytes()))

    def __repr__(self):
        return (f"BoundedTensorSpec(shape={self._shape}, dtype={self._dtype}, "
                f"name={self._name!r}, minimum={self._minimum}, maximum={self._maximum})")


# The reference distinguishes numpy ArraySpecs from TensorSpecs; here both are the same objects.
ArraySpec = TensorSpec
BoundedArraySpec = BoundedTensorSpec


def is_bounded(spec):
    return isinstance(spec, BoundedTensorSpec)


def is_discrete(spec):
    return is_discrete_dtype(spec.dtype)


def is_continuous(spec):
    return spec.dtype in (torch.float32, torch.float64, torch.float16)


def to_array_spec(spec):
    return spec


def from_spec(spec):
    return spec


def add_outer_dims_nest(specs, outer_dims):
    from agents_amd.utils import nest_utils

    def add(s):
        if isinstance(s, BoundedTensorSpec):
            return BoundedTensorSpec(tuple(outer_dims) + s.shape, s.dtype, s.minimum, s.maximum,
                                     s.name)
        return TensorSpec(tuple(outer_dims) + s.shape, s.dtype, s.name)

    return nest_utils.map_structure(add, specs)


def zeros_like_spec(spec, outer_dims=(), device=None):
    """Nest of zero tensors matching `spec` with the given outer dims (cold-path allocation)."""
    from agents_amd.utils import nest_utils
    return nest_utils.map_structure(
        lambda s: torch.zeros(tuple(outer_dims) + s.shape, dtype=s.dtype, device=device), spec)
