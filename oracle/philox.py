"""Philox4x32-10 in numpy (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as
1, 2, 3", SC'11).  Restates agents_amd/csrc/common.h::philox4x32_10 and is itself pinned to the
published Random123 known-answer vectors (tests/test_oracle_philox.py).

The reference draws replay indices with tf.random.uniform(int64)
(tf_agents/replay_buffers/tf_uniform_replay_buffer.py:265-272), an unseeded TensorFlow-internal
Philox stream that no reference test pins; this module defines the canonical stream instead.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over broadcastable integer arrays.  Returns 4 uint32 arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(*[np.asarray(c, dtype=np.uint64) & MASK
                                           for c in (c0, c1, c2, c3)])
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, lo1, n2, lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def u01(bits):
    """Top 24 bits * 2^-24, as float32 (agents_amd/csrc/common.h::aa_u01)."""
    return ((np.asarray(bits, dtype=np.uint32) >> np.uint32(8)).astype(np.float32) *
            np.float32(1.0 / 16777216.0))


def uniform_bounded(lo, hi, n_rows, seed, call):
    """RandomTFPolicy on a bounded continuous spec as csrc/ppo.hip::aa_uniform_sample_kernel draws
    it: out[i, d] = lo[d] + (hi[d] - lo[d]) * u01(word 0 of Philox(counter = (element index,
    call), key = seed)) in float32, folded back to lo when rounding reaches hi.  (The reference
    samples with an unseeded tf.random.uniform -- specs/tensor_spec.py:235-312 -- the stream is
    ours, the range is the spec's.)"""
    lo = np.asarray(lo, np.float32).reshape(-1)
    hi = np.asarray(hi, np.float32).reshape(-1)
    D = lo.size
    i = np.arange(n_rows * D, dtype=np.uint64)
    r0, _, _, _ = philox4x32_10(i & MASK, i >> np.uint64(32), int(call) & 0xFFFFFFFF,
                                (int(call) >> 32) & 0xFFFFFFFF, int(seed) & 0xFFFFFFFF,
                                (int(seed) >> 32) & 0xFFFFFFFF)
    d = (np.arange(n_rows * D) % D)
    l, h = lo[d], hi[d]
    v = (l + ((h - l).astype(np.float32) * u01(r0)).astype(np.float32)).astype(np.float32)
    return np.where(v < h, v, l).reshape(n_rows, D)


def tf_uniform_u64(seed, seed2, base_block, n):
    """The n raw 64-bit words behind one execution of tf.random.uniform(shape=[n], dtype=int64) in
    TensorFlow's stream LAYOUT (from general knowledge of tensorflow/core --
    NOT in /root/reference and NOT verified against a TensorFlow run: no TF in this image):
      GuardedPhiloxRandom::Init(seed, seed2)  -> key = (seed lo32, seed hi32), counter words 2, 3 =
                                                 (seed2 lo32, seed2 hi32), counter words 0, 1 = 0
      ReserveRandomOutputs(n, 256)            -> an execution starts at the generator's current
                                                 64-bit block index and advances it by n * 256
      UniformDistribution<PhiloxRandom, int64>-> output j = words 2 (j % 2), 2 (j % 2) + 1 (low,
                                                 high) of Philox block base + j // 2
    The caller maps with minval + x mod (maxval - minval), as the op does."""
    j = np.arange(n, dtype=np.uint64)
    blk = np.uint64(base_block) + (j >> np.uint64(1))
    x0, x1, x2, x3 = philox4x32_10(blk & MASK, blk >> np.uint64(32), int(seed2) & 0xFFFFFFFF,
                                   (int(seed2) >> 32) & 0xFFFFFFFF, int(seed) & 0xFFFFFFFF,
                                   (int(seed) >> 32) & 0xFFFFFFFF)
    lo = np.where(j & np.uint64(1), x2, x0).astype(np.uint64)
    hi = np.where(j & np.uint64(1), x3, x1).astype(np.uint64)
    return (hi << np.uint64(32)) | lo
