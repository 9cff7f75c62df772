"""Step metrics of the train_eval scripts (tf_agents/metrics/tf_metrics.py): observers fed by the
drivers, device resident and free of host synchronisation, so that they can ride inside a captured
collect graph; `result()` is the only place that reads the device.

  NumberOfEpisodes            += traj.is_last()                       (tf_metrics.py:166-199)
  EnvironmentSteps            += ~traj.is_boundary()                  (tf_metrics.py:130-163)
  AverageReturnMetric         per-env return accumulators, reset on is_first, pushed into a ring
                              of `buffer_size` finished episodes on is_last; result = their mean
                              (tf_metrics.py:202-262, TFDeque :41-95)
  AverageEpisodeLengthMetric  the same for the number of non-boundary steps  (tf_metrics.py:265-325)
Bookkeeping tensors are a few words per environment: plain torch element-wise ops on the collect
stream (measurement plumbing beside the hot path, not part of it).
"""
import torch


class _Counter:
    def __init__(self, name, prefix="Metrics", dtype=torch.int64):
        self.name, self.prefix = name, prefix
        self._count = None
        self._restore = None      # a count restored before the first call (applied on allocation)

    def _hit(self, traj):
        raise NotImplementedError

    def __call__(self, traj):
        hit = self._hit(traj)
        if self._count is None:
            # lazily allocated on the first trajectory's device; a checkpoint restored before that
            # (train_eval scripts: `train_checkpointer.initialize_or_restore()` precedes the first
            # driver run) left its count in `_restore`
            self._count = torch.full((), int(self._restore or 0), dtype=torch.int64,
                                     device=hit.device)
            self._restore = None
        self._count += hit.sum()
        return traj

    call = __call__

    def result(self):
        if self._count is None:
            return int(self._restore or 0)
        return int(self._count.item())

    def reset(self):
        self._restore = None
        if self._count is not None:
            self._count.zero_()

    def state_dict(self):
        return {"count": self.result()}

    def load_state_dict(self, sd):
        n = int(sd.get("count", 0))
        if self._count is not None:
            self._count.fill_(n)
        else:
            self._restore = n

    def tf_summaries(self, train_step=None, step_metrics=()):
        """Summary writers are outside the hot-path scope."""


class NumberOfEpisodes(_Counter):
    def __init__(self, name="NumberOfEpisodes", prefix="Metrics", dtype=torch.int64):
        super().__init__(name, prefix, dtype)

    def _hit(self, traj):
        return traj.is_last()


class EnvironmentSteps(_Counter):
    def __init__(self, name="EnvironmentSteps", prefix="Metrics", dtype=torch.int64):
        super().__init__(name, prefix, dtype)

    def _hit(self, traj):
        return ~traj.is_boundary()


class _EpisodeAverage:
    """Ring of the last `buffer_size` finished episodes' totals (TFDeque semantics: the mean is over
    the entries filled so far, 0 when none)."""

    def __init__(self, name, prefix, dtype, batch_size, buffer_size):
        self.name, self.prefix = name, prefix
        self._batch_size, self._buffer_size = int(batch_size), int(buffer_size)
        self._acc = self._ring = self._pushed = None
        self._restore = None      # a state restored before the first call (applied on allocation)

    def _alloc(self, dev):
        self._acc = torch.zeros((self._batch_size,), dtype=torch.float32, device=dev)
        # one spare slot at the end swallows the writes of environments that did not finish
        self._ring = torch.zeros((self._buffer_size + 1,), dtype=torch.float32, device=dev)
        self._pushed = torch.zeros((), dtype=torch.int64, device=dev)
        if self._restore is not None:
            sd, self._restore = self._restore, None
            self._apply(sd)

    def _apply(self, sd):
        self._acc.copy_(torch.as_tensor(sd["acc"]).to(self._acc.device))
        self._ring.copy_(torch.as_tensor(sd["ring"]).to(self._ring.device))
        self._pushed.fill_(int(sd["pushed"]))

    def _increment(self, traj):
        raise NotImplementedError

    def __call__(self, traj):
        first, last = traj.is_first().reshape(-1), traj.is_last().reshape(-1)
        if self._acc is None:
            self._alloc(first.device)
        # reset on is_first by selection, not by a multiplication (0 * inf = nan would stick; the
        # reference resets with tf.where, tf_metrics.py:236-240)
        self._acc.copy_(torch.where(first, torch.zeros_like(self._acc), self._acc))
        self._acc += self._increment(traj)
        # finished environments take consecutive ring positions in environment order; when more
        # finish in one step than the ring holds only the LAST `buffer_size` of them are kept (the
        # reference's TFDeque adds them one after the other), so no two writes share a position
        rank = torch.cumsum(last.to(torch.int64), 0) - 1
        n_last = last.sum()
        keep = last & (rank >= n_last - self._buffer_size)
        pos = torch.where(keep, (self._pushed + rank) % self._buffer_size,
                          torch.full_like(rank, self._buffer_size))
        self._ring.scatter_(0, pos, self._acc)
        self._pushed += n_last
        return traj

    call = __call__

    def result(self):
        if self._acc is None:
            if self._restore is None:
                return 0.0
            n = min(int(self._restore["pushed"]), self._buffer_size)
            ring = torch.as_tensor(self._restore["ring"])
            return float(ring[:n].mean().item()) if n else 0.0
        n = min(int(self._pushed.item()), self._buffer_size)
        return float(self._ring[:n].mean().item()) if n else 0.0

    def reset(self):
        self._restore = None
        if self._acc is not None:
            self._acc.zero_()
            self._ring.zero_()
            self._pushed.zero_()

    def state_dict(self):
        if self._acc is None:
            return dict(self._restore) if self._restore is not None else {}
        # copies (Tensor.cpu() of a host tensor is the tensor itself)
        return {"acc": self._acc.detach().cpu().clone(), "ring": self._ring.detach().cpu().clone(),
                "pushed": int(self._pushed)}

    def load_state_dict(self, sd):
        if not sd:
            return
        if self._acc is not None:
            self._apply(sd)
        else:
            self._restore = dict(sd)      # applied when the first trajectory allocates the state

    def tf_summaries(self, train_step=None, step_metrics=()):
        """Summary writers are outside the hot-path scope."""


class AverageReturnMetric(_EpisodeAverage):
    def __init__(self, name="AverageReturn", prefix="Metrics", dtype=torch.float32, batch_size=1,
                 buffer_size=10):
        super().__init__(name, prefix, dtype, batch_size, buffer_size)

    def _increment(self, traj):
        return traj.reward.reshape(-1).to(torch.float32)


class AverageEpisodeLengthMetric(_EpisodeAverage):
    def __init__(self, name="AverageEpisodeLength", prefix="Metrics", dtype=torch.float32,
                 batch_size=1, buffer_size=10):
        super().__init__(name, prefix, dtype, batch_size, buffer_size)

    def _increment(self, traj):
        return (~traj.is_boundary()).reshape(-1).to(torch.float32)
