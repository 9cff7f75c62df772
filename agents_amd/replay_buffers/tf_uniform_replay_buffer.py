"""TFUniformReplayBuffer on MI355X: batched adds and uniform sampling, tables resident in HBM.

Drop-in for tf_agents/replay_buffers/tf_uniform_replay_buffer.py:47-635 -- same constructor,
methods, layout (B blocks of L frames, one shared `last_id`), valid-id ranges, BufferInfo -- with
every data-path operation a HIP kernel from csrc/replay.hip:

  add_batch   -> aa_rb_scatter_rows   (all leaves + id table in one launch, then last_id += 1)
  get_next    -> aa_rb_sample_gather  (Philox4x32-10 ids/blocks -> rows, probabilities, row
                                       gather of every leaf, call counter: one launch)
                 aa_rb_gather_rows    (all leaves + ids in one launch)
  gather_all  -> aa_rb_range_rows + aa_rb_gather_rows

Ordering: every op is enqueued on the current HIP stream, which replaces the reference's
tf.CriticalSection on `last_id` (:154,582-601).  `last_id` lives in device memory (graph-capture
friendly) with a host mirror that lets the emptiness check raise without a device sync.
Differences from the reference are listed in DESIGN.md (own documented random stream; `device`
names a torch/HIP device).
"""
import collections

import numpy as np
import os

import torch

from agents_amd import _lib
from agents_amd.replay_buffers import replay_buffer, table
from agents_amd.replay_buffers.dataset import Dataset
from agents_amd.specs import tensor_spec
from agents_amd.utils import graph, nest_utils

BufferInfo = collections.namedtuple("BufferInfo", ["ids", "probabilities"])

_EMPTY_SAMPLE = ("TFUniformReplayBuffer is empty. Make sure to add items before sampling the "
                 "buffer.")
_EMPTY_DATASET = ("TFUniformReplayBuffer is empty. Make sure to add items before asking the "
                  "buffer for data.")


# AA_RB_STAMPED=0: graphed datasets replay the device-counter launch instead of stamping (A/B)
STAMPED_DRAWS = True


def _valid_range_ids(last_id, max_length, num_steps=None):
    """[min_id, max_id) of valid start ids; host mirror of the kernel's range logic (:610-635)."""
    if num_steps is None:
        num_steps = 1
    if last_id < max_length:
        return 0, max(last_id + 1 - num_steps + 1, 0)
    return last_id + 1 - max_length, last_id + 1 - num_steps + 1


def _resolve_device(device):
    if device is None or (isinstance(device, str) and device in ("", "gpu:*", "GPU:*")):
        return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() \
            else torch.device("cuda")
    if isinstance(device, str) and device.lower().startswith(("gpu:", "/gpu:", "/device:gpu:")):
        idx = device.split(":")[-1]
        return torch.device("cuda", int(idx)) if idx.isdigit() else torch.device("cuda")
    d = torch.device(device) if not isinstance(device, torch.device) else device
    if d.type == "cuda" and d.index is None and torch.cuda.is_available():
        d = torch.device("cuda", torch.cuda.current_device())
    return d


class TFUniformReplayBuffer(replay_buffer.ReplayBuffer):
    """A TFUniformReplayBuffer with batched adds and uniform sampling."""

    def __init__(self, data_spec, batch_size, max_length=1000, scope="TFUniformReplayBuffer",
                 device=None, table_fn=table.Table, dataset_drop_remainder=False,
                 dataset_window_shift=None, stateful_dataset=False, seed=0, dataset_ring=8,
                 rng="philox"):
        """`seed`, `dataset_ring`, `rng` are this package's additions (the reference's sampling is
        unseeded).  rng="philox" (default): the package's own Philox stream (counter = (sample,
        call), key = seed; oracle/philox.py).  rng="tf": the two draws of `_get_next` in
        TensorFlow's stream LAYOUT -- seed = (global seed, op seed of the id draw[, op seed of the
        env-block draw = the former + 1]), blocks advancing by 256 x outputs per call (SURVEY.md
        Appendix B; include/agents_amd.h: aa_rb_draw_tf_host) -- so that replay indices can be
        compared with a fixture from a seeded TensorFlow run the day one exists.  UNVERIFIED
        against TensorFlow (none is installed here): parity of the stream stays unpinned.  The
        draw is made on the host and the rows are uploaded, so datasets are not graphed in this
        mode: it is a checking aid, not the fast path."""
        if rng not in ("philox", "tf"):
            raise ValueError("rng must be 'philox' or 'tf'")
        self._rng = rng
        self._tf_seeds = None
        self._tf_blocks = 0              # Philox blocks both TF-layout streams have consumed
        if rng == "tf":
            sd = tuple(int(v) for v in (seed if isinstance(seed, (tuple, list)) else (seed, 0)))
            if len(sd) == 2:
                sd = sd + (sd[1] + 1,)
            if len(sd) != 3:
                raise ValueError("rng='tf': seed = (global seed, op seed[, second op seed])")
            self._tf_seeds = tuple(v & 0xFFFFFFFFFFFFFFFF for v in sd)
            seed = sd[0]
        self._batch_size = int(batch_size)
        self._max_length = int(max_length)
        capacity = self._batch_size * self._max_length
        super().__init__(data_spec, capacity, stateful_dataset)
        self._scope = scope
        self._device = _resolve_device(device)
        if self._device.type != "cuda":
            raise _lib.AgentsAmdError(
                f"TFUniformReplayBuffer tables live in GPU memory; device={device!r} is not a "
                "HIP device (there is no CPU path).")
        self._table_fn = table_fn
        self._dataset_drop_remainder = dataset_drop_remainder
        self._dataset_window_shift = dataset_window_shift
        self._dataset_ring = int(dataset_ring)   # 0 -> as_dataset elements are fresh tensors
        self._id_spec = tensor_spec.TensorSpec((), torch.int64, name="id")
        self._data_table = table_fn(self._data_spec, capacity, device=self._device)
        self._id_table = table_fn(self._id_spec, capacity, device=self._device)
        self._last_id = torch.full((1,), -1, dtype=torch.int64, device=self._device)
        # arrival counters of the scatter / sample-and-gather launches (9 x one 128-byte line each)
        self._scatter_arrival = torch.zeros((144,), dtype=torch.int64, device=self._device)
        self._sample_arrival = torch.zeros((144,), dtype=torch.int64, device=self._device)
        self._err_flag = torch.zeros((1,), dtype=torch.int32, device=self._device)
        self._last_id_host = -1          # mirror: every add_batch is +1, clear() is -> -1
        self._seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self._sample_calls = 0           # Philox call counter (one per get_next), host mirror
        # the counter the kernels read: device resident so that captured graphs advance it
        self._sample_calls_dev = torch.zeros((1,), dtype=torch.int64, device=self._device)

    def device_error(self):
        """The kernels' error word (synchronises): 0 = fine, 1 = a draw found no valid range (the
        buffer was emptier than the host mirror believed), 2 = the arrival protocol of a
        sample-and-gather launch timed out (two launches shared the arrival words concurrently);
        the call counter was not advanced for that launch and its rows must not be trusted."""
        return int(self._err_flag.item())

    def check_device_error(self):
        code = self.device_error()
        if code:
            self._err_flag.zero_()
            raise RuntimeError(f"TFUniformReplayBuffer: device-side error {code} in a sample / "
                               "gather launch (see device_error())")

    # ---- properties ------------------------------------------------------------------------
    @property
    def device(self):
        return self._device

    @property
    def table_fn(self):
        return self._table_fn

    @property
    def scope(self):
        return self._scope

    @property
    def batch_size(self):
        return self._batch_size

    @property
    def max_length(self):
        return self._max_length

    def variables(self):
        return self._data_table.variables() + self._id_table.variables() + [self._last_id]

    # ---- ReplayBuffer implementation ----------------------------------------------------------
    def _num_frames(self):
        return min((self._last_id_host + 1) * self._batch_size, self._capacity)

    def _add_batch(self, items, count=None):
        """Writes one frame per env block (:182-209).
        count = (step_type int32 [n], counter int32 [n] or None, total int64 [1], mailbox device
        pointer or None): the launch also runs DynamicStepDriver's loop counter on those step types
        (a graphed driver body passes the time step it just produced: utils/graph.py)."""
        lib = _lib.load()
        graph.join_lanes(self._device)
        flat = self._data_table.check_values(items, self._batch_size)
        p = self._data_table.pack(flat)
        with torch.cuda.device(self._device):
            if count is None:
                _lib.check(lib.aa_rb_scatter_rows(
                    p.tables, p.ios, p.row_bytes, p.n, self._id_table.variables()[0].data_ptr(),
                    self._last_id.data_ptr(), self._scatter_arrival.data_ptr(), self._batch_size,
                    self._max_length, _lib.stream_ptr()),
                    "aa_rb_scatter_rows")
            else:
                st, counter, total, mailbox = count
                _lib.check(lib.aa_rb_scatter_rows_count(
                    p.tables, p.ios, p.row_bytes, p.n, self._id_table.variables()[0].data_ptr(),
                    self._last_id.data_ptr(), self._scatter_arrival.data_ptr(), self._batch_size,
                    self._max_length, st.data_ptr(), st.numel(),
                    None if counter is None else counter.data_ptr(), total.data_ptr(), mailbox,
                    _lib.stream_ptr()), "aa_rb_scatter_rows_count")
        graph.on_replay(self._bump_last_id_host)

    def add_batch_counting(self, items, step_type, counter, total, mailbox):
        """`add_batch(items)` + the driver's step count of `step_type` in one launch (see
        `_add_batch`).  Only for buffers whose add is the plain scatter (subclasses that extend
        `_add_batch` -- the prioritized buffer -- do not offer it: `supports_counting_add`)."""
        self._add_batch(items, count=(step_type, counter, total, mailbox))

    def supports_counting_add(self):
        # neither hook of the add path is overridden: a subclass that extends the PUBLIC
        # `add_batch` (preprocessing, bookkeeping) must see every add of a graphed driver too
        return type(self)._add_batch is TFUniformReplayBuffer._add_batch and \
            type(self).add_batch is replay_buffer.ReplayBuffer.add_batch

    def _bump_last_id_host(self):
        self._last_id_host += 1

    def _bump_sample_calls(self):
        self._sample_calls += 1

    def _check_not_empty(self, num_steps):
        lo, hi = _valid_range_ids(self._last_id_host, self._max_length, num_steps)
        if hi <= lo:
            raise RuntimeError(_EMPTY_SAMPLE)

    # ---- stamped draws: the host mirrors of last_id / the call counter ride in the launch ------
    def supports_stamped_draws(self):
        """True when `draw_into` reproduces `get_next` (no subclass sampler in the way)."""
        return STAMPED_DRAWS and self._rng == "philox" and \
            type(self)._sample_rows is TFUniformReplayBuffer._sample_rows and \
            type(self)._get_next is TFUniformReplayBuffer._get_next

    def stamped_slot(self, element):
        """The launch arguments of `draw_into` for one static output element
        (data [S, T, ...], BufferInfo(ids [S, T], probabilities [S])), packed once."""
        data, info = element
        outs = nest_utils.flatten(data)
        S, T = (int(d) for d in info.ids.shape)
        for t in outs + [info.ids, info.probabilities]:
            if not t.is_contiguous() or int(t.shape[0]) != S:
                raise ValueError("stamped_slot: not a [S, T, ...] element of this buffer")
        return (self._data_table.pack(outs), info.ids.data_ptr(), info.probabilities.data_ptr(),
                S, T)

    def draw_into(self, slot):
        """`get_next(S, T)` into the buffers of `slot` with ONE eager launch that carries last_id
        and the Philox call number by value (aa_rb_sample_gather_stamped): same samples, bit for
        bit, as the device-counter launch a HIP graph replays, without its dependent counter
        read.  Not capturable (the values would freeze); the caller holds the device context."""
        p, ids_ptr, probs_ptr, S, T = slot
        self._check_not_empty(T)
        _lib.check(_lib.load().aa_rb_sample_gather_stamped(
            p.tables, p.ios, p.row_bytes, p.n, self._id_table.variables()[0].data_ptr(), ids_ptr,
            probs_ptr, self._last_id_host, self._batch_size, self._max_length, S, T, self._seed,
            self._sample_calls, self._sample_calls_dev.data_ptr(), self._err_flag.data_ptr(),
            _lib.stream_ptr()), "aa_rb_sample_gather_stamped")
        self._sample_calls += 1

    def _sample_rows_tf(self, S, T):
        """rng="tf": the draw on the host in TensorFlow's stream layout, rows uploaded."""
        import ctypes
        if graph.capturing():
            raise RuntimeError("rng='tf' draws on the host: get_next cannot be captured")
        rows_h = torch.empty((S, T), dtype=torch.int64).pin_memory()
        prob = ctypes.c_float(0.0)
        g, op_ids, op_seg = self._tf_seeds
        rc = _lib.load().aa_rb_draw_tf_host(
            self._last_id_host, self._batch_size, self._max_length, S, T, g, op_ids, op_seg,
            self._tf_blocks, rows_h.data_ptr(), ctypes.byref(prob))
        if rc == -34:       # AA_ERR_RANGE: the valid id range is empty
            raise RuntimeError(_EMPTY_SAMPLE)
        _lib.check(rc, "aa_rb_draw_tf_host")
        self._tf_blocks += S * 256          # ReserveRandomOutputs(output size, 256), both ops
        self._sample_calls += 1
        rows = rows_h.to(self._device, non_blocking=True)
        probs = torch.full((S,), prob.value, dtype=torch.float32, device=self._device)
        return rows, probs

    def _sample_rows(self, S, T):
        if self._rng == "tf":
            return self._sample_rows_tf(S, T)
        lib = _lib.load()
        rows = torch.empty((S, T), dtype=torch.int64, device=self._device)
        probs = torch.empty((S,), dtype=torch.float32, device=self._device)
        # call counter = 0 + *device counter; the launch advances the device counter by one
        _lib.check(lib.aa_rb_sample_rows(
            self._last_id.data_ptr(), self._batch_size, self._max_length, S, T, self._seed,
            0, self._sample_calls_dev.data_ptr(), rows.data_ptr(), probs.data_ptr(),
            self._err_flag.data_ptr(), _lib.stream_ptr()), "aa_rb_sample_rows")
        graph.on_replay(self._bump_sample_calls)
        return rows, probs

    def _get_next(self, sample_batch_size=None, num_steps=None, time_stacked=True):
        """Uniformly sampled items (:211-310).  Returns (data, BufferInfo(ids, probabilities))."""
        graph.join_lanes(self._device)
        if not graph.capturing():
            self._check_not_empty(num_steps)
        S = 1 if sample_batch_size is None else int(sample_batch_size)
        T = 1 if num_steps is None else int(num_steps)
        with torch.cuda.device(self._device):
            if type(self)._sample_rows is TFUniformReplayBuffer._sample_rows and \
                    self._rng == "philox":
                # draw + gather + counter advance in ONE launch (csrc/replay.hip)
                lib = _lib.load()
                ids = torch.empty((S, T), dtype=torch.int64, device=self._device)
                probs = torch.empty((S,), dtype=torch.float32, device=self._device)
                outs = self._data_table.alloc_out((S, T))
                p = self._data_table.pack(outs)
                _lib.check(lib.aa_rb_sample_gather(
                    p.tables, p.ios, p.row_bytes, p.n, self._id_table.variables()[0].data_ptr(),
                    ids.data_ptr(), probs.data_ptr(), self._last_id.data_ptr(), self._batch_size,
                    self._max_length, S, T, self._seed, 0, self._sample_calls_dev.data_ptr(),
                    self._sample_arrival.data_ptr(), self._err_flag.data_ptr(),
                    _lib.stream_ptr()), "aa_rb_sample_gather")
                graph.on_replay(self._bump_sample_calls)
                data = nest_utils.pack_sequence_as(self._data_spec, outs)
            else:       # a subclass with its own index sampler (prioritized replay)
                rows, probs = self._sample_rows(S, T)
                ids = torch.empty((S, T), dtype=torch.int64, device=self._device)
                data = self._data_table.read(rows, self._id_table.variables()[0], ids)

        def squeeze(t):
            if num_steps is None:
                t = t.squeeze(1)
            if sample_batch_size is None:
                t = t.squeeze(0)
            return t

        if num_steps is not None and not time_stacked:
            # a T-tuple of [S, ...] items instead of [S, T, ...] (:295-306)
            def unstack(t, i):
                u = t[:, i]
                return u.squeeze(0) if sample_batch_size is None else u
            data = tuple(nest_utils.map_structure(lambda t, i=i: unstack(t, i), data)
                         for i in range(T))
            ids_out = tuple(unstack(ids, i) for i in range(T))
        else:
            data = nest_utils.map_structure(squeeze, data)
            ids_out = squeeze(ids)
        if sample_batch_size is None:
            probs = probs.squeeze(0)
        return data, BufferInfo(ids=ids_out, probabilities=probs)

    def as_dataset(self, sample_batch_size=None, num_steps=None, num_parallel_calls=None,
                   single_deterministic_pass=False):
        return super().as_dataset(sample_batch_size, num_steps, num_parallel_calls,
                                  single_deterministic_pass=single_deterministic_pass)

    def _as_dataset(self, sample_batch_size=None, num_steps=None, sequence_preprocess_fn=None,
                    num_parallel_calls=None, ring=None):
        """Infinite stream of get_next (:329-367).  num_parallel_calls is accepted and ignored:
        sampling kernels are already asynchronous on the stream.  `ring` (subclasses): size of the
        graphed sampler's output ring, 0 = every element is a fresh eager get_next; None = the
        buffer's `dataset_ring`, read when the dataset is iterated."""
        if sequence_preprocess_fn is not None:
            raise NotImplementedError("sequence_preprocess_fn is not supported.")

        def gen():
            # the iterator replays a HIP graph of (sample, gather) after two eager draws; its
            # elements live in a ring of static buffers (graph.GraphedSampler)
            n_ring = self._dataset_ring if ring is None else ring
            if self._rng != "philox":
                n_ring = 0        # host-made draws: nothing to replay from a graph
            if n_ring <= 0:
                while True:
                    yield self.get_next(sample_batch_size, num_steps, time_stacked=True)
            sampler = graph.GraphedSampler(self, sample_batch_size, num_steps, ring=n_ring)
            while True:
                yield sampler.next()

        return Dataset(gen, infinite=True)

    def _single_deterministic_pass_dataset(self, sample_batch_size=None, num_steps=None,
                                           sequence_preprocess_fn=None, num_parallel_calls=None):
        """Fixed-order pass (:369-531); index order computed on the host, rows gathered on device."""
        if sequence_preprocess_fn is not None:
            raise NotImplementedError("sequence_preprocess_fn is not supported.")
        drop = self._dataset_drop_remainder
        if drop and sample_batch_size is not None and sample_batch_size > self._batch_size:
            raise ValueError(
                "sample_batch_size ({}) > self.batch_size ({}) and dataset_drop_remainder is "
                "True.  In this case, ALL data will be dropped by the deterministic dataset."
                .format(sample_batch_size, self._batch_size))
        if drop and num_steps is not None and num_steps > self._max_length:
            raise ValueError(
                "num_steps_size ({}) > self.max_length ({}) and dataset_drop_remainder is "
                "True.  In this case, ALL data will be dropped by the deterministic dataset."
                .format(num_steps, self._max_length))

        def gen():
            for ids in deterministic_pass_ids(self._last_id_host, self._batch_size,
                                              self._max_length, sample_batch_size, num_steps,
                                              drop, self._dataset_window_shift):
                ids_np = np.asarray(ids, dtype=np.int64)
                ids_t = torch.as_tensor(ids_np, device=self._device)
                rows = torch.as_tensor(np.mod(ids_np, self._capacity), device=self._device)
                with torch.cuda.device(self._device):
                    data = self._data_table.read(rows)
                yield data, BufferInfo(ids=ids_t, probabilities=())

        return Dataset(gen, infinite=False)

    def _gather_all(self):
        """All valid items, shape [batch_size, n, ...] in id order (:533-557)."""
        lib = _lib.load()
        graph.join_lanes(self._device)
        lo, hi = _valid_range_ids(self._last_id_host, self._max_length)
        n = hi - lo
        rows = torch.empty((self._batch_size, max(n, 0)), dtype=torch.int64, device=self._device)
        with torch.cuda.device(self._device):
            if n > 0:
                _lib.check(lib.aa_rb_range_rows(lo, n, self._batch_size, self._max_length,
                                                rows.data_ptr(), _lib.stream_ptr()),
                           "aa_rb_range_rows")
            return self._data_table.read(rows)

    def _clear(self, clear_all_variables=False):
        """last_id = -1; tables untouched unless clear_all_variables (:559-579)."""
        graph.join_lanes(self._device)
        self._last_id.fill_(-1)
        self._last_id_host = -1
        if clear_all_variables:
            for v in self._data_table.variables() + self._id_table.variables():
                v.zero_()

    def clear(self, clear_all_variables=False):
        return self._clear(clear_all_variables)

    # ---- helpers used by tests / checkpointing ------------------------------------------------
    def _get_last_id(self):
        return self._last_id_host

    def state_dict(self):
        graph.join_lanes(self._device)
        return {"tables": [v.clone() for v in self._data_table.variables()],
                "ids": self._id_table.variables()[0].clone(), "last_id": self._last_id_host,
                "sample_calls": self._sample_calls, "seed": self._seed,
                "tf_blocks": self._tf_blocks}

    def load_state_dict(self, sd):
        graph.join_lanes(self._device)
        for v, s in zip(self._data_table.variables(), sd["tables"]):
            v.copy_(s)
        self._id_table.variables()[0].copy_(sd["ids"])
        self._last_id_host = int(sd["last_id"])
        self._last_id.fill_(self._last_id_host)
        self._sample_calls = int(sd["sample_calls"])
        self._sample_calls_dev.fill_(self._sample_calls)
        self._seed = int(sd["seed"])
        self._tf_blocks = int(sd.get("tf_blocks", 0))


def _windows(seq, size, shift, drop_remainder):
    """tf.data `window(size, shift).flat_map(batch(size, drop_remainder))` over a python list."""
    shift = size if shift is None else shift
    out, i, n = [], 0, len(seq)
    while i < n:
        w = seq[i:i + size]
        if len(w) == size or not drop_remainder:
            out.append(w)
        i += shift
    return out


def deterministic_pass_ids(last_id, batch_size, max_length, sample_batch_size, num_steps,
                           drop_remainder, window_shift):
    """Index order of the fixed-order dataset (:433-511): env-major frames when unbatched;
    blocks of `sample_batch_size` envs, frame-major inside a block, windows transposed to
    [S, num_steps] (remainder windows always dropped) when batched."""
    lo, hi = _valid_range_ids(last_id, max_length, None)
    if hi <= lo:
        raise RuntimeError(_EMPTY_DATASET)
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
    fr = np.arange(lo, hi, dtype=np.int64)
    for envs in _windows(list(range(batch_size)), sample_batch_size, None, drop_remainder):
        # (vectorised: at 2,048 envs x 129 frames the per-frame list comprehensions cost 10 ms per
        # pass, as much as 200 PPO minibatch steps)
        e = np.asarray(envs, dtype=np.int64) * max_length
        if num_steps is None:
            for f in fr:
                yield f + e
        else:
            shift = num_steps if window_shift is None else window_shift
            i = 0
            while i + num_steps <= fr.shape[0]:       # remainder windows are always dropped
                yield e[:, None] + fr[None, i:i + num_steps]      # [S, num_steps]
                i += shift
