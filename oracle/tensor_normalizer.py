"""numpy restatement of the tensor normalisers (TEST INFRASTRUCTURE; see oracle/__init__.py).

Follows tf_agents/utils/tensor_normalizer.py, all arithmetic in the variable dtype (float32 unless
the spec is float64/int64, :36-42):
  TensorNormalizer.normalize            :134-206   tf.nn.batch_normalization(x, mean, var, None,
                                                   None, eps) = x * rsqrt(var + eps)
                                                   + (-mean * rsqrt(var + eps)), then clip
  EMATensorNormalizer._update_ops       :236-281
  StreamingTensorNormalizer             :288-395   count init 1e-8 (:36,:291-296)
  parallel_variance_calculation         :397-449
  kahan_summation                       :452-474
Pinned on the numeric cases of tensor_normalizer_test.py (tests/test_oracle_normalizer.py): the
three parallel-variance identities, EMA normalisation (-3.1622776601, 0, 3.1622776601 / 90, 100,
110), streaming update / normalisation vs numpy moments, and the four long-run cases at the
iteration counts the reference's own comments give as its fp32 limits (62 / 41 / 383 / 54).
Unpinned by the reference: TF's reduction order inside reduce_mean / reduce_sum (numpy's pairwise
sums are used here) and the last bit of rsqrt (1 / sqrt here).
"""
import numpy as np

_EPS = 1e-8


def kahan_summation(accumulator, carry, value):
    delta = value - carry
    new_accumulator = accumulator + delta
    carry = (new_accumulator - accumulator) - delta
    return new_accumulator, carry


def parallel_variance_calculation(n_a, avg_a, m2_a, n_b, avg_b, m2_b, m2_b_c):
    n_ab = n_a + n_b
    delta = avg_b - avg_a
    s_delta = delta * n_b / n_ab
    avg_ab = avg_a + s_delta
    m2_ab, m2_ab_c = kahan_summation(m2_b, m2_b_c, m2_a + (delta * n_a * s_delta))
    return n_ab, avg_ab, m2_ab, m2_ab_c


def batch_normalization(x, mean, var, eps):
    inv = (np.asarray(1.0, x.dtype) / np.sqrt(var + np.asarray(eps, x.dtype))).astype(x.dtype)
    return (x * inv + (-mean * inv)).astype(x.dtype)


class StreamingNormalizer:
    """One leaf of shape `shape`; nests are lists of these in the tests."""

    def __init__(self, shape, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.shape = tuple(shape)
        self.reset()

    def reset(self):
        self.count = np.full(self.shape, _EPS, self.dtype)
        self.avg = np.zeros(self.shape, self.dtype)
        self.m2 = np.zeros(self.shape, self.dtype)
        self.m2_carry = np.zeros(self.shape, self.dtype)

    def update(self, x):
        x = np.asarray(x).astype(self.dtype)
        outer_rank = x.ndim - len(self.shape)
        axes = tuple(range(outer_rank))
        n_a = self.dtype.type(np.prod(x.shape[:outer_rank], dtype=np.int64))
        avg_a = x.mean(axis=axes, dtype=self.dtype)
        m2_a = np.square(x - avg_a).sum(axis=axes, dtype=self.dtype)
        self.count, self.avg, self.m2, self.m2_carry = [
            np.asarray(v, self.dtype) for v in parallel_variance_calculation(
                n_a, avg_a, m2_a, self.count, self.avg, self.m2, self.m2_carry)]

    def mean_var(self):
        return self.avg, (self.m2 / self.count).astype(self.dtype)

    def normalize(self, x, clip_value=5.0, center_mean=True, variance_epsilon=1e-3):
        x = np.asarray(x).astype(self.dtype)
        mean, var = self.mean_var()
        if not center_mean:
            mean = np.zeros_like(mean)
        y = batch_normalization(x, mean, var, variance_epsilon)
        if clip_value > 0:
            y = np.clip(y, self.dtype.type(-clip_value), self.dtype.type(clip_value))
        return y


class EMANormalizer:
    def __init__(self, shape, norm_update_rate=0.001, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.shape = tuple(shape)
        self.rate = self.dtype.type(norm_update_rate)
        self.mean = np.zeros(self.shape, self.dtype)
        self.var = np.ones(self.shape, self.dtype)

    def update(self, x, outer_dims=(0,)):
        x = np.asarray(x).astype(self.dtype)
        axes = tuple(outer_dims)
        mean = x.mean(axis=axes, dtype=self.dtype)
        var = np.square(x - self.mean).mean(axis=axes, dtype=self.dtype)
        self.mean = (self.mean + self.rate * (mean - self.mean)).astype(self.dtype)
        self.var = (self.var + self.rate * (var - self.var)).astype(self.dtype)

    def normalize(self, x, clip_value=5.0, center_mean=True, variance_epsilon=1e-3):
        x = np.asarray(x).astype(self.dtype)
        mean = self.mean if center_mean else np.zeros_like(self.mean)
        y = batch_normalization(x, mean, self.var, variance_epsilon)
        if clip_value > 0:
            y = np.clip(y, self.dtype.type(-clip_value), self.dtype.type(clip_value))
        return y
