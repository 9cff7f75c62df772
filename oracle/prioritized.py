"""numpy restatement of the prioritized sampler (TEST INFRASTRUCTURE; the product path never
imports this).  There is no reference implementation to follow (see
agents_amd/replay_buffers/tf_prioritized_replay_buffer.py); this states the package's own
definition, which csrc/prio.hip must reproduce bit for bit:

  q_i      = clamp(floor((|p_i| + eps)^alpha * 65536 + 0.5), 1, 2^32 - 1)         (uint32;
             x^alpha in float32 is not correctly rounded on either side, so for alpha != 1 the stored
             value may differ by one unit between device and numpy; sampling only sees q)
  valid    = rows whose stored id lies in the window-start range of
             TFUniformReplayBuffer._valid_range_ids (tf_uniform_replay_buffer.py:610-635)
  total    = sum of q over valid rows                                            (exact, uint64)
  r_s      = (Philox4x32-10(counter = (s, call), key = seed) words x | y << 32) mod total
  row_s    = the valid row (index order) whose cumulative interval [c, c + q) contains r_s
  prob_s   = float32(q_row / total)
  rows[s,t] = (id + t) mod L + block * L with id = the stored id of row_s.
PARITY UNPINNED against tf_agents: the reference has no prioritized TFUniformReplayBuffer.
"""
import numpy as np

from oracle import philox


def quantise(p, alpha, eps):
    v = (np.abs(np.asarray(p, np.float32)) + np.float32(eps)).astype(np.float32)
    if alpha != 1.0:
        v = np.power(v, np.float32(alpha)).astype(np.float32)
    q = np.floor(v.astype(np.float64) * 65536.0 + 0.5)
    q = np.where(q >= 1.0, q, 1.0)
    return np.minimum(q, 4294967295.0).astype(np.uint64)


def valid_range(last_id, max_len, T):
    if last_id < max_len:
        return 0, max(last_id + 1 - T + 1, 0)
    return last_id + 1 - max_len, last_id + 1 - T + 1


def sample(prio_q, ids, last_id, batch, max_len, S, T, seed, call):
    lo, hi = valid_range(last_id, max_len, T)
    q = np.where((ids >= lo) & (ids < hi), prio_q.astype(np.uint64), np.uint64(0))
    cum = np.cumsum(q, dtype=np.uint64)
    total = int(cum[-1]) if len(cum) else 0
    rows = np.zeros((S, T), np.int64)
    probs = np.zeros((S,), np.float32)
    if total == 0:
        return rows, probs, True
    sidx = np.arange(S, dtype=np.uint64)
    x0, x1, _, _ = philox.philox4x32_10(sidx & philox.MASK, sidx >> np.uint64(32),
                                        call & 0xFFFFFFFF, call >> 32,
                                        seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    for s in range(S):
        r = ((int(x1[s]) << 32) | int(x0[s])) % total
        row = int(np.searchsorted(cum, np.uint64(r), side="right"))
        seg = row // max_len
        for t in range(T):
            rows[s, t] = (int(ids[row]) + t) % max_len + seg * max_len
        probs[s] = np.float32(float(int(q[row])) / float(total))
    return rows, probs, False
