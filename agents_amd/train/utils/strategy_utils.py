"""Data-parallel "strategy" over torch.distributed (RCCL on MI355X, gloo in CPU tests).

Stands in for tf.distribute.MirroredStrategy as used by tf_agents/train/learner.py:170-177,
356-360 and tf_agents/train/utils/strategy_utils.py:27-61: one PROCESS per GPU (torchrun), each
with a full replica of the parameters and its own replay shard; the only exchange per train step is
ONE sum all-reduce of the flat fp32 gradient buffer (the reference's implicit gradient all-reduce
inside optimizer.apply_gradients under strategy.run), plus a sum of the LossInfo scalars
(learner.py:322-337).
"""
import os

import torch
import torch.distributed as dist


class Strategy:
    """Single-replica default (tf.distribute.get_strategy())."""

    num_replicas_in_sync = 1
    rank = 0

    def all_reduce_sum_(self, tensor):
        return tensor

    def reduce_sum(self, tensor):
        return tensor

    def broadcast_(self, tensors, src=0):
        return tensors

    def all_gather_batch(self, tensor):
        return tensor

    def barrier(self):
        pass


class DataParallelStrategy(Strategy):
    """All ranks of the default torch.distributed process group."""

    def __init__(self, process_group=None):
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised; launch with "
                               "`python -m torch.distributed.run` and call init_process_group")
        self._pg = process_group
        self.num_replicas_in_sync = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        # the launcher's view must agree with the communicator's: a rank that silently fell out
        # of the group (or a second group of size 1) would train on its own
        ws = os.environ.get("WORLD_SIZE")
        if process_group is None and ws is not None and int(ws) != self.num_replicas_in_sync:
            raise RuntimeError(
                f"DataParallelStrategy: the process group has {self.num_replicas_in_sync} ranks "
                f"but the launcher started WORLD_SIZE={ws}")
        self.backend = dist.get_backend(process_group)
        # what the data path exchanged so far (bench.py reports it per step, so that the first
        # multi-GPU scaling run can be read: bytes and calls per step, and -- with `profile` on --
        # how long the calling stream sat in / waited for the collectives)
        self.stats = {"calls": 0, "bytes": 0}
        self.profile = False
        self._timed = []          # (begin event, end event) pairs while `profile` is on

    def reset_stats(self):
        self.stats = {"calls": 0, "bytes": 0}
        self._timed = []

    def _count(self, tensor):
        self.stats["calls"] += 1
        self.stats["bytes"] += tensor.numel() * tensor.element_size()

    def _stamp(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def exposed_ms(self):
        """Stream time between the stamps taken around the synchronous collectives and around the
        waits for the asynchronous ones (call after a device synchronisation)."""
        return sum(a.elapsed_time(b) for a, b in self._timed)

    def describe(self):
        return (f"{self.num_replicas_in_sync} replicas over {self.backend} "
                f"(rank {self.rank})")

    def all_reduce_sum_(self, tensor):
        """In-place SUM all-reduce (ncclAllReduce over xGMI on GPUs)."""
        self._count(tensor)
        if self.profile and tensor.is_cuda:
            a = self._stamp()
            dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self._pg)
            self._timed.append((a, self._stamp()))
            return tensor
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self._pg)
        return tensor

    def all_reduce_sum_async_(self, tensor):
        """Starts an in-place SUM all-reduce and returns its Work handle; `handle.wait()` makes the
        CURRENT stream wait for it (no host block).  RCCL runs it on its own stream after the work
        already enqueued on the current stream, so kernels enqueued afterwards overlap with it."""
        self._count(tensor)
        work = dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self._pg, async_op=True)
        if self.profile and tensor.is_cuda:
            return _TimedWork(work, self)
        return work

    def reduce_sum(self, tensor):
        out = tensor.clone()
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self._pg)
        return out

    def broadcast_(self, tensors, src=0):
        """In-place broadcast of every tensor from rank `src`: what MirroredStrategy does when it
        creates mirrored variables (every replica starts from replica 0's values)."""
        for t in tensors:
            dist.broadcast(t, src=src, group=self._pg)
        return tensors

    def all_gather_batch(self, tensor):
        """[B, ...] on every rank -> [world * B, ...] (rank order) on every rank: the global batch
        a mirrored variable's update sees in cross-replica context (the PPO tensor normalisers,
        tf_agents/agents/ppo/ppo_agent.py:1078-1086 under MirroredStrategy).  Every rank must pass
        the same shape."""
        world = self.num_replicas_in_sync
        x = tensor.contiguous()
        if dist.get_backend(self._pg) == "nccl":
            out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype,
                              device=x.device)
            dist.all_gather_into_tensor(out, x, group=self._pg)
            return out
        # gloo (CPU tests, two ranks sharing one GPU): gathers host tensors
        host = x.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=self._pg)
        return torch.cat(parts, dim=0).to(x.device)

    def barrier(self):
        dist.barrier(group=self._pg)


class _TimedWork:
    """A collective's Work handle whose wait() is bracketed by timing events on the waiting stream
    (DataParallelStrategy.profile): the stream time between them is the EXPOSED part of the
    collective, the rest ran under the kernels enqueued in between."""

    def __init__(self, work, strategy):
        self._work, self._strategy = work, strategy

    def wait(self):
        a = self._strategy._stamp()
        out = self._work.wait()
        self._strategy._timed.append((a, self._strategy._stamp()))
        return out

    def __getattr__(self, name):
        return getattr(self._work, name)


def get_strategy(tpu=None, use_gpu=True):
    """strategy_utils.get_strategy: data-parallel if a process group exists, else single."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return DataParallelStrategy()
    return Strategy()
