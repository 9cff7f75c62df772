"""numpy restatement of aa_random_permutation (TEST INFRASTRUCTURE; see oracle/__init__.py).

The reference shuffles PPO minibatches with a tf.data shuffle buffer
(tf_agents/train/ppo_learner.py:228-247) whose order no reference test pins; the package's own
permutation is defined here and in csrc/replay.hip (aa_feistel_perm_kernel):
  perm[i] = orbit of i under a 4-round balanced Feistel network on 2h bits (2^(2h) >= n, h >= 1),
            followed until it re-enters [0, n) (cycle walking);
  round function F_round(half) = low h bits of (x1 << 32 | x0) of
            Philox4x32-10(counter = (half, round, call_lo, call_hi), key = (seed_lo, seed_hi)).
"""
import numpy as np

from oracle import philox


def random_permutation(n, seed, call):
    n = int(n)
    h = 1
    while (1 << (2 * h)) < n:
        h += 1
    mask = np.uint64((1 << h) - 1)
    hh = np.uint64(h)
    x = np.arange(n, dtype=np.uint64)
    out = np.empty(n, dtype=np.int64)
    todo = np.arange(n)
    while todo.size:
        v = x[todo]
        l, r = v >> hh, v & mask
        for rnd in range(4):
            x0, x1, _, _ = philox.philox4x32_10(
                r & np.uint64(0xFFFFFFFF), np.full(r.shape, rnd, np.uint64),
                call & 0xFFFFFFFF, (call >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF,
                (seed >> 32) & 0xFFFFFFFF)
            f = ((x1.astype(np.uint64) << np.uint64(32)) | x0.astype(np.uint64)) & mask
            l, r = r, l ^ f
        v = (l << hh) | r
        x[todo] = v
        done = v < np.uint64(n)
        out[todo[done]] = v[done].astype(np.int64)
        todo = todo[~done]
    return out
