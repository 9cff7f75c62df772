"""Learner: runs `agent.train` on batches from an experience dataset, data-parallel across GPUs.

Mirrors tf_agents/train/learner.py:42-467:
  __init__   :58-243   dataset via experience_dataset_fn, agent.initialize(), triggers
  run        :265-305  `iterations` train steps, then triggers; returns the reduced LossInfo
  _train / single_train_step :309-378   sample = next(iterator); strategy.run(agent.train)
  loss       :380-467
Data parallelism: each rank owns a replay shard and samples it independently; the
per-replica loss is divided by B_local * num_replicas (tf.nn.compute_average_loss,
utils/common.py:1462-1467) and the flat gradient buffer is SUM all-reduced once per step through
`agent.gradient_hook`; every rank then applies the identical optimizer step, so replicas stay in
lock-step.  They START in lock-step because the constructor broadcasts rank 0's replicated state
(parameters, target networks, optimizer slots, normaliser statistics: `agent.replicated_state()`)
to every rank after the checkpoint restore -- MirroredStrategy's mirrored variables; networks
default to an unseeded initialiser, so without this each process would train its own weights.
LossInfo fields are SUM-reduced over replicas and over all axes like strategy.reduce(SUM)
(:322-337); the returned LossInfo owns its storage (one small copy per `run`), unlike the LossInfo
`agent.train` returns, whose fields are views of work buffers the next train step overwrites.
"""
import os

import torch

from agents_amd.agents import tf_agent
from agents_amd.train.utils import strategy_utils
from agents_amd.utils import nest_utils

TRAIN_DIR = "train"
POLICY_SAVED_MODEL_DIR = "policies"


# False: always reduce LossInfo fields with generic reductions (A/B measurements)
_FIELD_SUMS = True


class Learner:
    def __init__(self, root_dir, train_step, agent, experience_dataset_fn=None,
                 after_train_strategy_step_fn=None, triggers=None, checkpoint_interval=100000,
                 summary_interval=1000, max_checkpoints_to_keep=3, use_kwargs_in_agent_train=False,
                 strategy=None, run_optimizer_variable_init=True, use_reverb_v2=False,
                 direct_sampling=False, experience_dataset_options=None,
                 strategy_run_options=None, summary_root_dir=None, use_graph=True):
        if checkpoint_interval < 0:
            raise ValueError("checkpoint_interval must be non-negative")
        self._root_dir = root_dir
        self._train_dir = os.path.join(root_dir, TRAIN_DIR) if root_dir else None
        self.train_step = train_step
        self._agent = agent
        self.use_kwargs_in_agent_train = use_kwargs_in_agent_train
        self.strategy = strategy or strategy_utils.get_strategy()
        self._after_train_strategy_step_fn = after_train_strategy_step_fn
        self.triggers = triggers or []
        self._checkpoint_interval = checkpoint_interval
        self._max_checkpoints_to_keep = max_checkpoints_to_keep
        self._experience_dataset_fn = experience_dataset_fn
        self._experience_iterator = None
        if experience_dataset_fn is not None:
            self._experience_iterator = iter(experience_dataset_fn())
        # data-parallel wiring (the agent divides its loss by B_local * num_replicas)
        self._agent.num_replicas = self.strategy.num_replicas_in_sync
        if self.strategy.num_replicas_in_sync > 1:
            self._agent.gradient_hook = self.strategy.all_reduce_sum_
            # lets the graphed train step overlap the reduce of the dense layers' gradients with
            # the conv layers' backward (two buckets)
            self._agent.gradient_hook_async = getattr(self.strategy, "all_reduce_sum_async_", None)
            # replicated statistics updated from data (PPO's tensor normalisers) see the GLOBAL
            # batch, as mirrored variables updated in cross-replica context do
            self._agent.batch_gather_hook = self.strategy.all_gather_batch
        self._agent.initialize()
        # learner.py:309-337 runs the step inside tf.function; here: HIP-graph replay
        self._train_fn = self._agent.train
        if use_graph and (hasattr(self._agent, "_train_phase_grads") or
                          hasattr(self._agent, "_graph_train_whole")):
            from agents_amd.utils import graph
            self._train_fn = graph.graphed_train(self._agent)
        # learner.py:206-243: a Checkpointer over (agent, train_step) under root_dir/train, restored
        # at construction, saved every `checkpoint_interval` train steps by an IntervalTrigger
        self._checkpointer = None
        if self._train_dir is not None and hasattr(self._agent, "state_dict"):
            from agents_amd.train import triggers as triggers_lib
            from agents_amd.utils import common
            self._checkpointer = common.Checkpointer(
                self._train_dir, max_to_keep=max_checkpoints_to_keep, agent=self._agent,
                train_step=self.train_step)
            if self.train_step is not None and self._checkpointer.checkpoint_exists:
                self._agent.train_step_counter.assign(int(self.train_step))
            self.triggers = list(self.triggers) + [triggers_lib.CheckpointTrigger(
                self._checkpointer, self._agent.train_step_counter, checkpoint_interval)]
        self.sync_replicas()

    def sync_replicas(self):
        """Every replica takes rank 0's replicated state (no-op on one replica).  Call it again
        after restoring or editing the agent's weights on a subset of the ranks."""
        if self.strategy.num_replicas_in_sync <= 1:
            return
        state = getattr(self._agent, "replicated_state", None)
        if state is None:
            return
        tensors = [t for t in state() if t is not None]
        dev = next((t.device for t in tensors if t.is_cuda), None)
        if dev is not None:
            from agents_amd.utils import graph
            graph.join_lanes(dev)
        # the ranks must agree on WHAT is replicated before the collectives are issued (optimizer
        # slots exist only after the first step or a restore): sum(n) and sum(n^2) over the
        # replicas pin every rank's n to the same value
        n, world = len(tensors), self.strategy.num_replicas_in_sync
        chk = self.strategy.reduce_sum(torch.tensor(
            [float(n), float(n * n)], dtype=torch.float64, device=dev or "cpu"))
        if int(chk[0]) != n * world or int(chk[1]) != n * n * world:
            raise RuntimeError(
                "Learner.sync_replicas: the replicas hold different numbers of replicated tensors "
                f"(this rank: {n}); restore the same checkpoint on every rank")
        self.strategy.broadcast_(tensors, src=0)
        step = torch.tensor([int(self._agent.train_step_counter)], dtype=torch.int64,
                            device=dev or "cpu")
        self.strategy.broadcast_([step], src=0)
        self._agent.train_step_counter.assign(int(step.item()))
        hook = getattr(self._agent, "post_replicated_state_update", None)
        if hook is not None:      # state derived from the parameters (prepared filter planes)
            hook()

    @property
    def train_step_numpy(self):
        return int(self.train_step)

    def _reduce_loss(self, loss_info):
        """SUM over replicas and over all axes of every LossInfo field (learner.py:322-337).
        All fields travel in ONE all-reduce (a [n_fields] vector) instead of one per field."""
        hook = getattr(self._agent, "reduce_loss_info", None) if _FIELD_SUMS else None
        if hook is not None:      # an agent whose loss kernel already holds the per-field sums
            pre = hook(loss_info)
            if pre is not None:
                loss_info = pre
                if self.strategy.num_replicas_in_sync == 1 and \
                        getattr(self._agent, "reduced_owns_storage", False):
                    return loss_info       # already sums in storage of their own (one replica)
        flat = nest_utils.flatten(loss_info)
        idx = [i for i, t in enumerate(flat) if isinstance(t, torch.Tensor)]
        if not idx:
            return loss_info
        if len(idx) <= 8 and self.strategy.num_replicas_in_sync == 1 and \
                all(flat[i].numel() == 1 and flat[i].dtype == torch.float32 and flat[i].is_cuda
                    for i in idx):
            # every field is already one float32 device scalar (the agent's loss launch made the
            # sums): the packed copy with its launch arguments cached per set of source addresses
            # (the generic path below spends ~20 tiny tensor ops per call on views of views)
            return self._pack_scalars(loss_info, flat, idx)
        sums = [flat[i].sum().reshape(1) if flat[i].dim() > 0 else flat[i].reshape(1)
                for i in idx]
        # one packed copy: the result owns its storage (the inputs may be views of buffers the
        # next train step overwrites) and travels in one all-reduce.  Up to 8 float32 device
        # scalars are packed by ONE aa_copy_segments launch (no torch kernel on the timed path)
        if len(sums) <= 8 and all(s.dtype == torch.float32 and s.is_cuda for s in sums):
            from agents_amd import ops
            vec = torch.empty((len(sums),), dtype=torch.float32, device=sums[0].device)
            with torch.cuda.device(vec.device):
                ops.copy_segments([(s.view(1, 1), vec[j:j + 1].view(1, 1))
                                   for j, s in enumerate(sums)])
        else:
            vec = torch.cat([s.to(torch.float32) for s in sums])
        if self.strategy.num_replicas_in_sync > 1:
            vec = self.strategy.all_reduce_sum_(vec)
        sums = [vec[j:j + 1] for j in range(len(idx))]
        for j, i in enumerate(idx):
            flat[i] = sums[j].reshape(())
        return nest_utils.pack_sequence_as(loss_info, flat)

    def _pack_scalars(self, loss_info, flat, idx):
        import ctypes
        from agents_amd import _lib
        n = len(idx)
        key = tuple(flat[i].data_ptr() for i in idx)
        cache = self.__dict__.setdefault("_pack_cache", {})
        args = cache.get(key)
        if args is None:
            if len(cache) > 16:
                cache.clear()
            src = (ctypes.c_void_p * n)(*key)
            one64 = (ctypes.c_int64 * n)(*([1] * n))
            one32 = (ctypes.c_int32 * n)(*([1] * n))
            args = cache[key] = (src, (ctypes.c_void_p * n)(), one64, one32)
        src, dst, one64, one32 = args
        dev = flat[idx[0]].device
        vec = torch.empty((n,), dtype=torch.float32, device=dev)
        base = vec.data_ptr()
        for j in range(n):
            dst[j] = base + 4 * j
        if torch.cuda.current_device() == dev.index:
            _lib.check(_lib.load().aa_copy_segments(src, dst, one64, one64, one32, n, 1,
                                                    _lib.stream_ptr()), "aa_copy_segments")
        else:
            with torch.cuda.device(dev):
                _lib.check(_lib.load().aa_copy_segments(src, dst, one64, one64, one32, n, 1,
                                                        _lib.stream_ptr()), "aa_copy_segments")
        outs = vec.unbind(0)          # 0-dim views: the result owns its storage (vec)
        for j, i in enumerate(idx):
            flat[i] = outs[j]
        return nest_utils.pack_sequence_as(loss_info, flat)

    def single_train_step(self, iterator):
        sample = next(iterator)
        if isinstance(sample, tuple) and len(sample) == 2 and not hasattr(sample, "_fields"):
            experience, sample_info = sample
        else:
            experience, sample_info = sample, None
        if self.use_kwargs_in_agent_train:
            loss_info = self._train_fn(**experience)
        else:
            loss_info = self._train_fn(experience)
        if self._after_train_strategy_step_fn:
            self._after_train_strategy_step_fn((experience, sample_info), loss_info)
        return loss_info

    def run(self, iterations=1, iterator=None, parallel_iterations=10):
        """`iterations` train steps; returns the replica-summed LossInfo of the last one."""
        if iterations < 1:
            raise AssertionError("Iterations must be greater or equal to 1, was %d" % iterations)
        iterator = iterator or self._experience_iterator
        if iterator is None:
            raise ValueError("Learner.run needs an iterator or an experience_dataset_fn")
        loss_info = None
        for _ in range(iterations):
            loss_info = self.single_train_step(iterator)
        return self.finish_run(loss_info)

    def finish_run(self, loss_info):
        """The end of `run` (learner.py:293-305): train_step follows the agent's counter, the last
        LossInfo is reduced over replicas and axes, triggers fire.  Also called by PPOLearner when
        the agent ran a whole epoch's minibatch steps from one host call."""
        if self.train_step is not None and self.train_step is not self._agent.train_step_counter:
            self.train_step.assign(int(self._agent.train_step_counter))
        reduced = self._reduce_loss(loss_info)
        step = int(self._agent.train_step_counter)
        for trigger in self.triggers:
            trigger(step)
        return reduced

    def loss(self, experience_and_sample_info=None, reduce_op="sum"):
        """agent.loss on one batch (no update), replica-reduced (learner.py:380-467)."""
        if experience_and_sample_info is None:
            experience_and_sample_info = next(self._experience_iterator)
        if isinstance(experience_and_sample_info, tuple) and \
                not hasattr(experience_and_sample_info, "_fields"):
            experience = experience_and_sample_info[0]
        else:
            experience = experience_and_sample_info
        loss_info = self._agent.loss(experience)
        if not isinstance(loss_info, tf_agent.LossInfo):
            raise TypeError("agent.loss must return a LossInfo")
        return self._reduce_loss(loss_info)
