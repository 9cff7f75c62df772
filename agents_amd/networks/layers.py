"""Layer descriptors for `sequential.Sequential` (the keras layers the reference's scripts pass).

They carry configuration only; parameters live in the owning network's flat fp32 buffer and all
arithmetic is done by the HIP GEMM (csrc/gemm.hip).  Names/arguments follow tf.keras.layers so the
scripts' network factories translate one to one (agents/dqn/examples/v2/train_eval.py:343-376,
examples/dqn/mnih15/dqn_train_eval_atari.py:80-112).
"""
import math

import numpy as np

_TRUNC_STD_FIX = 0.87962566103423978  # keras VarianceScaling truncated_normal correction


# ---- initialisers (host side, numpy; cold path) -------------------------------------------------
class Initializer:
    def __call__(self, shape, rng, fan_in, fan_out):
        raise NotImplementedError


class Zeros(Initializer):
    def __call__(self, shape, rng, fan_in, fan_out):
        return np.zeros(shape, np.float32)


class Constant(Initializer):
    def __init__(self, value=0.0):
        self.value = value

    def __call__(self, shape, rng, fan_in, fan_out):
        v = np.asarray(self.value, np.float32)
        if v.size == 1:
            return np.full(shape, float(v.reshape(-1)[0]), np.float32)
        return v.reshape(shape).copy()


class RandomUniform(Initializer):
    def __init__(self, minval=-0.05, maxval=0.05, seed=None):
        self.minval, self.maxval = minval, maxval

    def __call__(self, shape, rng, fan_in, fan_out):
        return rng.uniform(self.minval, self.maxval, size=shape).astype(np.float32)


class VarianceScaling(Initializer):
    """keras VarianceScaling(scale, mode, distribution); default fan_in / truncated_normal."""

    def __init__(self, scale=1.0, mode="fan_in", distribution="truncated_normal", seed=None):
        self.scale, self.mode, self.distribution = scale, mode, distribution

    def __call__(self, shape, rng, fan_in, fan_out):
        n = {"fan_in": fan_in, "fan_out": fan_out, "fan_avg": (fan_in + fan_out) / 2.0}[self.mode]
        scale = self.scale / max(1.0, n)
        if self.distribution in ("truncated_normal", "normal"):
            std = math.sqrt(scale) / _TRUNC_STD_FIX
            x = rng.standard_normal(size=shape)
            bad = np.abs(x) > 2.0
            while bad.any():
                x[bad] = rng.standard_normal(size=int(bad.sum()))
                bad = np.abs(x) > 2.0
            return (x * std).astype(np.float32)
        if self.distribution == "untruncated_normal":
            return (rng.standard_normal(size=shape) * math.sqrt(scale)).astype(np.float32)
        limit = math.sqrt(3.0 * scale)
        return rng.uniform(-limit, limit, size=shape).astype(np.float32)


class GlorotUniform(VarianceScaling):
    def __init__(self, seed=None):
        super().__init__(1.0, "fan_avg", "uniform")


class Orthogonal(Initializer):
    def __init__(self, gain=1.0, seed=None):
        self.gain = gain

    def __call__(self, shape, rng, fan_in, fan_out):
        rows = int(np.prod(shape[:-1]))
        cols = shape[-1]
        a = rng.standard_normal(size=(max(rows, cols), min(rows, cols)))
        q, r = np.linalg.qr(a)
        q = q * np.sign(np.diag(r))
        if rows < cols:
            q = q.T
        return (self.gain * q[:rows, :cols]).reshape(shape).astype(np.float32)


def get_initializer(x, default):
    if x is None:
        return default
    if isinstance(x, Initializer):
        return x
    if isinstance(x, str):
        return {"zeros": Zeros(), "glorot_uniform": GlorotUniform(),
                "orthogonal": Orthogonal()}[x]
    raise TypeError(f"unsupported initializer {x!r}")


def _act_name(a):
    if a is None:
        return None
    if isinstance(a, str):
        return {"relu": "relu", "tanh": "tanh", "linear": None}[a]
    name = getattr(a, "__name__", None)
    if name in ("relu", "tanh"):
        return name
    raise ValueError(f"unsupported activation {a!r} (HIP epilogues: relu, tanh, linear)")


# ---- layers ---------------------------------------------------------------------------------
class Layer:
    has_params = False


class Rescale(Layer):
    """x / divisor -- the `Lambda(lambda x: x / 255)` of the Atari Q-network
    (dqn_train_eval_atari.py:103); fused into the first convolution's uint8 loader."""

    def __init__(self, divisor=255.0):
        self.divisor = float(divisor)


class Flatten(Layer):
    """keras Flatten on NHWC: (h, w, c) order == the memory layout; no data movement."""


class Dense(Layer):
    has_params = True

    def __init__(self, units, activation=None, use_bias=True, kernel_initializer=None,
                 bias_initializer=None, kernel_regularizer_l2=0.0, name=None):
        self.units = int(units)
        self.activation = _act_name(activation)
        if not use_bias:
            raise NotImplementedError("Dense(use_bias=False) is not supported")
        self.kernel_initializer = get_initializer(kernel_initializer, GlorotUniform())
        self.bias_initializer = get_initializer(bias_initializer, Zeros())
        self.l2 = float(kernel_regularizer_l2)
        self.name = name


class Conv2D(Layer):
    has_params = True

    def __init__(self, filters, kernel_size, strides=1, activation=None, padding="valid",
                 kernel_initializer=None, bias_initializer=None, name=None):
        self.filters = int(filters)
        ks = kernel_size if isinstance(kernel_size, (tuple, list)) else (kernel_size, kernel_size)
        self.kernel_size = (int(ks[0]), int(ks[1]))
        st = strides if not isinstance(strides, (tuple, list)) else strides[0]
        if isinstance(strides, (tuple, list)) and strides[0] != strides[1]:
            raise NotImplementedError("anisotropic strides are not supported")
        self.stride = int(st)
        if padding.lower() != "valid":
            raise NotImplementedError("only padding='valid' is supported")
        self.activation = _act_name(activation)
        self.kernel_initializer = get_initializer(kernel_initializer, GlorotUniform())
        self.bias_initializer = get_initializer(bias_initializer, Zeros())
        self.name = name


def relu(x=None):
    return "relu"


def tanh(x=None):
    return "tanh"
