"""TFPrioritizedReplayBuffer: TFUniformReplayBuffer with proportional prioritized sampling.

The reference has no such class (its prioritized replay is Reverb's, e.g.
tf_agents/examples/dqn/gymnasium/d3qn_train_eval.py:162); BASELINE.json's north star asks for
"uniform/segment-tree sampling", and the reference does carry the plumbing a prioritized buffer
plugs into -- `DqnLossInfo.td_error` (agents/dqn/dqn_agent.py:50-72), `BufferInfo.ids` /
`probabilities`, and `Learner(after_train_strategy_step_fn=...)` (train/learner.py:362-376):

    rb = TFPrioritizedReplayBuffer(agent.collect_data_spec, batch_size=B, max_length=L)
    learner = Learner(root, step, agent,
                      after_train_strategy_step_fn=rb.update_priorities_from_loss)

Everything of TFUniformReplayBuffer is kept (layout, add_batch, gather_all, datasets, validity of
window starts); only `get_next` differs: start rows are drawn with P(i) = p_i / sum_j p_j
(Schaul et al., 2016), `BufferInfo.probabilities` carries P(i) so callers can form importance
weights, new rows enter with the running maximum priority, and `update_priorities` stores
(|td_error| + eps)^alpha.  Kernels: csrc/prio.hip (flat two-level scan on uint32 fixed-point
priorities -- exact sums, bit-exact indices against oracle/prioritized.py).
"""
import torch

from agents_amd import _lib
from agents_amd.replay_buffers import replay_buffer
from agents_amd.replay_buffers import tf_uniform_replay_buffer as uniform
from agents_amd.replay_buffers.dataset import Dataset
from agents_amd.utils import graph, nest_utils

BufferInfo = uniform.BufferInfo

# False = block sums, draw and counter advance as three launches (aa_prio_sample_rows; 7 % slower
# in the loop: profiles/r06_zzzz_prio_one_launch_ab.txt)
ONE_LAUNCH_DRAW = True


class TFPrioritizedReplayBuffer(uniform.TFUniformReplayBuffer):
    def __init__(self, data_spec, batch_size, max_length=1000, priority_exponent=0.6,
                 priority_epsilon=1e-6, initial_priority=1.0, **kwargs):
        super().__init__(data_spec, batch_size, max_length, **kwargs)
        self._alpha = float(priority_exponent)
        self._eps = float(priority_epsilon)
        dev = self._device
        # validity of a row is decided by its STORED id here (uniform sampling draws ids, so it
        # never sees an unwritten row): unwritten rows must not look like "id 0"
        self._id_table.variables()[0].fill_(-1)
        self._prio_q = torch.zeros((self._capacity,), dtype=torch.int32, device=dev)  # uint32 bits
        q0 = min(max(int(round(float(initial_priority) * 65536.0)), 1), 2 ** 31 - 1)
        self._max_prio_q = torch.full((1,), q0, dtype=torch.int32, device=dev)
        lib = _lib.load()
        self._prio_ws = torch.empty((max(int(lib.aa_prio_workspace_bytes(self._capacity)), 8),),
                                    dtype=torch.uint8, device=dev)
        # the one-launch draw's slots + launch sequence + arrival count: zero-filled ONCE
        n_draw = int(lib.aa_prio_draw_workspace_bytes(self._capacity))
        self._draw_ws = torch.zeros((n_draw // 8,), dtype=torch.int64, device=dev) \
            if ONE_LAUNCH_DRAW and n_draw > 0 else None
        self._start_rows = None

    @property
    def priority_exponent(self):
        return self._alpha

    def priorities(self):
        """Float32 view of the stored priorities ([capacity], 0 for never-written rows)."""
        return (self._prio_q.to(torch.int64) & 0xFFFFFFFF).to(torch.float32) / 65536.0

    # ---- writes ---------------------------------------------------------------------------------
    def _add_batch(self, items, count=None):
        super()._add_batch(items, count)
        with torch.cuda.device(self._device):
            _lib.check(_lib.load().aa_prio_on_add(
                self._last_id.data_ptr(), self._batch_size, self._max_length,
                self._max_prio_q.data_ptr(), self._prio_q.data_ptr(), _lib.stream_ptr()),
                "aa_prio_on_add")

    def supports_counting_add(self):
        # the add is the plain scatter (which can carry a graphed driver's step count) followed by
        # the priority write of this class: offered unless a further subclass changes either hook
        return type(self)._add_batch is TFPrioritizedReplayBuffer._add_batch and \
            type(self).add_batch is replay_buffer.ReplayBuffer.add_batch

    def _clear(self, clear_all_variables=False):
        super()._clear(clear_all_variables)
        self._id_table.variables()[0].fill_(-1)
        self._prio_q.zero_()

    def update_priorities(self, ids, priorities):
        """p[row] = (|priority| + eps)^alpha.  `ids=None`: the window-start ROWS of the most recent
        `get_next` (`self.last_sampled_rows`); otherwise explicit row indices (a global frame id
        alone does not name the env block, so BufferInfo.ids cannot address a row)."""
        rows = (self._start_rows if self._start_rows is not None else self.last_sampled_rows) \
            if ids is None else ids
        if rows is None:
            raise RuntimeError("update_priorities: nothing has been sampled yet")
        rows = rows.reshape(-1).to(torch.int64).contiguous()
        pr = priorities.reshape(-1).to(torch.float32).contiguous()
        if rows.numel() != pr.numel():
            raise ValueError("update_priorities needs one priority per sampled item")
        graph.join_lanes(self._device)
        with torch.cuda.device(self._device):
            _lib.check(_lib.load().aa_prio_set(
                rows.data_ptr(), pr.data_ptr(), rows.numel(), self._alpha, self._eps,
                self._capacity, self._prio_q.data_ptr(), self._max_prio_q.data_ptr(),
                _lib.stream_ptr()), "aa_prio_set")

    def update_priorities_from_loss(self, experience_and_info, loss_info):
        """Learner.after_train_strategy_step_fn adapter: priorities from DqnLossInfo.td_error of
        the batch that was just trained on (the most recently sampled one)."""
        self.update_priorities(None, loss_info.extra.td_error)

    # ---- sampling -------------------------------------------------------------------------------
    def _sample_rows(self, S, T, out=None):
        """S window starts drawn with P(row) = p_row / sum; `out` = (rows [S, T], start_rows [S],
        probabilities [S]) to draw into (a dataset ring slot), else fresh tensors."""
        lib = _lib.load()
        dev = self._device
        if out is None:
            out = (torch.empty((S, T), dtype=torch.int64, device=dev),
                   torch.empty((S,), dtype=torch.int64, device=dev),
                   torch.empty((S,), dtype=torch.float32, device=dev))
        rows, start, probs = out
        if self._draw_ws is not None:
            _lib.check(lib.aa_prio_draw_rows(
                self._prio_q.data_ptr(), self._id_table.variables()[0].data_ptr(),
                self._last_id.data_ptr(), self._batch_size, self._max_length, S, T, self._seed,
                self._sample_calls_dev.data_ptr(), self._draw_ws.data_ptr(),
                self._draw_ws.numel() * 8, rows.data_ptr(), start.data_ptr(), probs.data_ptr(),
                self._err_flag.data_ptr(), _lib.stream_ptr()), "aa_prio_draw_rows")
            self._start_rows = start
        else:
            _lib.check(lib.aa_prio_sample_rows(
                self._prio_q.data_ptr(), self._id_table.variables()[0].data_ptr(),
                self._last_id.data_ptr(), self._batch_size, self._max_length, S, T, self._seed,
                self._sample_calls_dev.data_ptr(), self._prio_ws.data_ptr(),
                self._prio_ws.numel(), rows.data_ptr(), probs.data_ptr(),
                self._err_flag.data_ptr(), _lib.stream_ptr()), "aa_prio_sample_rows")
            self._start_rows = None
        graph.on_replay(self._bump_sample_calls)
        self.last_sampled_rows = rows[:, 0]
        return rows, probs

    last_sampled_rows = None

    def _as_dataset(self, sample_batch_size=None, num_steps=None, sequence_preprocess_fn=None,
                    num_parallel_calls=None):
        """Priorities change between draws and `last_sampled_rows` must name the batch being
        trained on: no graph replay of the draw (a replayed draw would leave `last_sampled_rows`
        pointing at whichever slot was captured last) and no sampling ahead.  The elements do live
        in a ring of `dataset_ring` static output slots, drawn into eagerly in turn (as the uniform
        buffer's dataset elements do): a train step that is a HIP graph of its input addresses
        then replays on them instead of copying a fresh batch in.  An element (and
        `last_sampled_rows`) is valid until `dataset_ring` further elements have been drawn;
        dataset_ring = 0: fresh tensors every time."""
        if sequence_preprocess_fn is not None:
            raise NotImplementedError("sequence_preprocess_fn is not supported.")
        n_ring = self._dataset_ring
        if n_ring <= 0 or sample_batch_size is None or num_steps is None:
            return super()._as_dataset(sample_batch_size, num_steps, sequence_preprocess_fn,
                                       num_parallel_calls, ring=0)
        S, T = int(sample_batch_size), int(num_steps)

        def gen():
            dev = self._device
            slots = []
            k = 0
            while True:
                if len(slots) < n_ring:
                    outs = self._data_table.alloc_out((S, T))
                    ids = torch.empty((S, T), dtype=torch.int64, device=dev)
                    draw = (torch.empty((S, T), dtype=torch.int64, device=dev),     # rows
                            torch.empty((S,), dtype=torch.int64, device=dev),       # start rows
                            torch.empty((S,), dtype=torch.float32, device=dev))     # P(row)
                    # the element OBJECT is the slot's too: a graphed train step recognises it
                    # by identity and skips its structure checks (utils/graph.py)
                    slots.append((self._data_table.pack(outs), ids, draw,
                                  (nest_utils.pack_sequence_as(self._data_spec, outs),
                                   BufferInfo(ids=ids, probabilities=draw[2]))))
                p, ids, draw, element = slots[k % n_ring]
                k += 1
                graph.join_lanes(dev)
                if not graph.capturing():
                    self._check_not_empty(T)
                with torch.cuda.device(dev):
                    rows, _ = self._sample_rows(S, T, out=draw)
                    _lib.check(_lib.load().aa_rb_gather_rows(
                        p.tables, p.ios, p.row_bytes, p.n,
                        self._id_table.variables()[0].data_ptr(), ids.data_ptr(),
                        rows.data_ptr(), S * T, _lib.stream_ptr()), "aa_rb_gather_rows")
                yield element

        return Dataset(gen, infinite=True)

    def state_dict(self):
        sd = super().state_dict()
        sd["prio_q"] = self._prio_q.clone()
        sd["max_prio_q"] = self._max_prio_q.clone()
        return sd

    def load_state_dict(self, sd):
        super().load_state_dict(sd)
        self._prio_q.copy_(sd["prio_q"])
        self._max_prio_q.copy_(sd["max_prio_q"])
