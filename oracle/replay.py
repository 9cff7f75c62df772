"""numpy restatement of TFUniformReplayBuffer (TEST INFRASTRUCTURE; see oracle/__init__.py).

Follows tf_agents/replay_buffers/tf_uniform_replay_buffer.py:
  __init__/layout :132-154   _num_frames :177-180   _add_batch :182-209   _get_next :211-310
  _single_deterministic_pass_dataset :369-531   _gather_all :533-557   _clear :559-579
  _increment_last_id :582-595   _get_rows_for_id :603-607   _valid_range_ids :610-635
and tf_agents/replay_buffers/table.py:86-137 (read = gather rows, write = scatter rows).

Random draws: the reference uses two unseeded tf.random.uniform(int64) ops (:265-272); parity is
therefore defined on the mapping raw draws -> rows, with the raw draws taken from the package's
documented Philox stream (oracle/philox.py):
  (x0,x1,x2,x3) = Philox(counter=(s_lo, s_hi, call_lo, call_hi), key=(seed_lo, seed_hi))
  id = min + ((x1<<32|x0) mod (max-min));  block = (x3<<32|x2) mod batch_size
"""
import numpy as np

from oracle import philox

EMPTY_MSG = ("TFUniformReplayBuffer is empty. Make sure to add items before sampling the buffer.")


def valid_range_ids(last_id, max_length, num_steps=None):
    """[min_id, max_id) of valid start ids (:610-635)."""
    if num_steps is None:
        num_steps = 1
    last_id = int(last_id)
    if last_id < max_length:
        return 0, max(last_id + 1 - num_steps + 1, 0)
    return last_id + 1 - max_length, last_id + 1 - num_steps + 1


def raw_draws(seed, call, n):
    s = np.arange(n, dtype=np.uint64)
    x0, x1, x2, x3 = philox.philox4x32_10(s & 0xFFFFFFFF, s >> np.uint64(32),
                                          call & 0xFFFFFFFF, (call >> 32) & 0xFFFFFFFF,
                                          seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    a = (x1.astype(np.uint64) << np.uint64(32)) | x0.astype(np.uint64)
    c = (x3.astype(np.uint64) << np.uint64(32)) | x2.astype(np.uint64)
    return a, c


def raw_draws_tf_layout(seed, seed2_ids, seed2_seg, base_block, n):
    """The two draws of `_get_next` (tf_uniform_replay_buffer.py:265-272) as two TensorFlow ops
    with op seeds seed2_ids / seed2_seg under one global seed, both `base_block` Philox blocks into
    their streams (philox.tf_uniform_u64: layout from general knowledge of TF core, UNVERIFIED
    against TensorFlow -- parity unpinned)."""
    return (philox.tf_uniform_u64(seed, seed2_ids, base_block, n),
            philox.tf_uniform_u64(seed, seed2_seg, base_block, n))


def rows_from_draws(id_raw, block_raw, last_id, batch_size, max_length, num_steps):
    """rows[S,T], probabilities[S] from raw 64-bit draws (:242-292)."""
    T = 1 if num_steps is None else num_steps
    lo, hi = valid_range_ids(last_id, max_length, num_steps)
    if hi <= lo:
        raise RuntimeError(EMPTY_MSG)
    num_ids = hi - lo
    ids = (np.asarray(id_raw, dtype=np.uint64) % np.uint64(num_ids)).astype(np.int64) + lo
    blocks = (np.asarray(block_raw, dtype=np.uint64) % np.uint64(batch_size)).astype(np.int64)
    steps = np.arange(T, dtype=np.int64)[None, :]
    rows = np.mod(steps + ids[:, None], max_length) + (blocks * max_length)[:, None]
    prob = np.float32(1.0) / np.float32(num_ids * batch_size)
    return rows, np.full(ids.shape, prob, dtype=np.float32)


class OracleReplayBuffer:
    """Leaves are numpy arrays [capacity, *leaf_shape]; items are lists of leaves."""

    def __init__(self, leaf_shapes, leaf_dtypes, batch_size, max_length, seed=0):
        self.batch_size = int(batch_size)
        self.max_length = int(max_length)
        self.capacity = self.batch_size * self.max_length
        self.tables = [np.zeros((self.capacity,) + tuple(s), dtype=d)
                       for s, d in zip(leaf_shapes, leaf_dtypes)]
        self.id_table = np.zeros((self.capacity,), dtype=np.int64)
        self.last_id = -1
        self.seed = int(seed)
        self.calls = 0
        self.batch_offsets = np.arange(self.batch_size, dtype=np.int64) * self.max_length

    # -- writes ---------------------------------------------------------------------------
    def add_batch(self, leaves):
        for leaf, tab in zip(leaves, self.tables):
            if leaf.shape[0] != self.batch_size:
                raise ValueError("leading dim of every item must equal batch_size")
            assert leaf.shape[1:] == tab.shape[1:]
        self.last_id += 1
        rows = self.batch_offsets + (self.last_id % self.max_length)
        self.id_table[rows] = self.last_id
        for leaf, tab in zip(leaves, self.tables):
            tab[rows] = leaf

    # -- reads ----------------------------------------------------------------------------
    def num_frames(self):
        return min((self.last_id + 1) * self.batch_size, self.capacity)

    def sample_rows(self, sample_batch_size, num_steps):
        S = 1 if sample_batch_size is None else sample_batch_size
        a, c = raw_draws(self.seed, self.calls, S)
        self.calls += 1
        return rows_from_draws(a, c, self.last_id, self.batch_size, self.max_length, num_steps)

    def get_next(self, sample_batch_size=None, num_steps=None):
        """Time-stacked sample: leaves [S,T,...] (or [S,...]/[T,...]/[...] like the reference)."""
        rows, probs = self.sample_rows(sample_batch_size, num_steps)
        data = [t[rows] for t in self.tables]
        ids = self.id_table[rows]
        if num_steps is None:
            data = [d[:, 0] for d in data]
            ids = ids[:, 0]
        if sample_batch_size is None:
            data = [d[0] for d in data]
            ids = ids[0]
            probs = probs[0]
        return data, ids, probs

    def gather_all(self):
        lo, hi = valid_range_ids(self.last_id, self.max_length)
        ids = np.arange(lo, hi, dtype=np.int64)
        rows = np.mod(ids, self.max_length)[None, :] + self.batch_offsets[:, None]
        return [t[rows] for t in self.tables]

    def clear(self, clear_all_variables=False):
        self.last_id = -1
        if clear_all_variables:
            for t in self.tables:
                t[...] = 0
            self.id_table[...] = 0

    # -- deterministic single pass (:369-531): yields arrays of GLOBAL ids ----------------------
    def deterministic_ids(self, sample_batch_size=None, num_steps=None, drop_remainder=False,
                          window_shift=None):
        return list(deterministic_pass_ids(self.last_id, self.batch_size, self.max_length,
                                           sample_batch_size, num_steps, drop_remainder,
                                           window_shift))

    def read_ids(self, ids):
        rows = np.mod(np.asarray(ids, dtype=np.int64), self.capacity)
        return [t[rows] for t in self.tables]


def _windows(seq, size, shift, drop_remainder):
    """tf.data window(size, shift).flat_map(batch(size, drop_remainder)) over a python list."""
    shift = size if shift is None else shift
    out = []
    i = 0
    n = len(seq)
    while i < n:
        w = seq[i:i + size]
        if len(w) == size or not drop_remainder:
            out.append(w)
        i += shift
    return out


def deterministic_pass_ids(last_id, batch_size, max_length, sample_batch_size, num_steps,
                           drop_remainder, window_shift):
    """Row ids ("b*L + frame_offset") in the order the reference's fixed-order dataset emits."""
    if drop_remainder and sample_batch_size is not None and sample_batch_size > batch_size:
        raise ValueError("sample_batch_size > batch_size and dataset_drop_remainder is True: "
                         "ALL data will be dropped")
    if drop_remainder and num_steps is not None and num_steps > max_length:
        raise ValueError("num_steps > max_length and dataset_drop_remainder is True: "
                         "ALL data will be dropped")
    lo, hi = valid_range_ids(last_id, max_length, None)
    if hi <= lo:
        raise RuntimeError("TFUniformReplayBuffer is empty. Make sure to add items before asking "
                           "the buffer for data.")
    frames = list(range(lo, hi))
    if sample_batch_size is None:
        for b in range(batch_size):
            ids = [b * max_length + f for f in frames]
            if num_steps is None:
                for i in ids:
                    yield np.int64(i)
            else:
                for w in _windows(ids, num_steps, window_shift, drop_remainder):
                    yield np.asarray(w, dtype=np.int64)
        return
    env_batches = _windows(list(range(batch_size)), sample_batch_size, None, drop_remainder)
    for envs in env_batches:
        per_frame = [np.asarray([f + e * max_length for e in envs], dtype=np.int64)
                     for f in frames]
        if num_steps is None:
            for v in per_frame:
                yield v
        else:
            for w in _windows(per_frame, num_steps, window_shift, True):
                yield np.stack(w, axis=0).T  # [S, num_steps]
