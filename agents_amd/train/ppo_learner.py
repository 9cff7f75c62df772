"""PPOLearner: epochs x shuffled minibatches over a fixed set of collected sequences.

Mirrors tf_agents/train/ppo_learner.py:40-345:
  __init__ validation   :141-167  (minibatching needs shuffle_buffer_size, a feed-forward
                                   network, compute_value_and_advantage_in_train=False and
                                   update_normalizers_in_train=False)
  _create_datasets      :198-262  take(num_samples) -> cache -> repeat(num_epochs) -> flatten
                                   [B,T] -> shuffle -> batch(1) -> batch(minibatch_size,
                                   drop_remainder)
  run                   :264-305  update normalizers / count frames, then
                                   iterations = int(frames / minibatch_size) * num_epochs train steps
The reference builds this with tf.data; here the `num_samples` elements are materialised once as
device tensors ([F, ...] after flattening), every epoch draws one permutation of the F frames and
gathers `minibatch_size` rows per step, each handed to `agent.train` as a [minibatch, 1, ...]
trajectory -- what batch(1).batch(minibatch_size) produces.  Differences, both in the data
pipeline (whose random order no reference test pins): the permutation is per epoch (tf.data's
shuffle buffer is filled from the repeated stream, so with a large buffer it can mix consecutive
epochs) and comes from the package's own documented generator (aa_random_permutation: a Feistel
network keyed by Philox4x32-10(seed, epoch counter), restated in oracle/perm.py), and in multi-GPU runs every rank runs all of ITS minibatches (one process per GPU with
its own replay shard) instead of dividing one dataset's batches by the replica count.
"""
import ctypes
import os

import torch

from agents_amd import _lib
from agents_amd.train import learner
from agents_amd.utils import nest_utils


# AA_PPO_FUSED_EPOCHS=0: always one `agent.train` call per gathered minibatch (A/B, tests)
FUSED_EPOCHS = True


class PPOLearner:
    def __init__(self, root_dir, train_step, agent, experience_dataset_fn,
                 normalization_dataset_fn, num_samples, num_epochs=1, minibatch_size=None,
                 shuffle_buffer_size=None, after_train_strategy_step_fn=None, triggers=None,
                 checkpoint_interval=100000, summary_interval=1000,
                 use_kwargs_in_agent_train=False, strategy=None, seed=0):
        if minibatch_size and shuffle_buffer_size is None:
            raise ValueError("shuffle_buffer_size must be provided if minibatch_size is not None.")
        if minibatch_size and (getattr(agent._actor_net, "state_spec", ()) or
                               getattr(agent._value_net, "state_spec", ())):
            raise ValueError("minibatch_size must be set to None for RNN networks.")
        if minibatch_size and agent._compute_value_and_advantage_in_train:
            raise ValueError("agent.compute_value_and_advantage_in_train should be set to False "
                             "when mini batching is used.")
        if agent.update_normalizers_in_train:
            raise ValueError("agent.update_normalizers_in_train should be set to False when "
                             "PPOLearner is used.")
        self._agent = agent
        self._minibatch_size = minibatch_size
        self._shuffle_buffer_size = shuffle_buffer_size
        self._num_epochs = num_epochs
        self._experience_dataset_fn = experience_dataset_fn
        self._normalization_dataset_fn = normalization_dataset_fn
        self._num_samples = num_samples
        self._generic_learner = learner.Learner(
            root_dir, train_step, agent, experience_dataset_fn=None,
            after_train_strategy_step_fn=after_train_strategy_step_fn, triggers=triggers,
            checkpoint_interval=checkpoint_interval, summary_interval=summary_interval,
            use_kwargs_in_agent_train=use_kwargs_in_agent_train, strategy=strategy)
        self.strategy = self._generic_learner.strategy
        self.num_replicas = self.strategy.num_replicas_in_sync
        self.num_frames_for_training = 0
        self._perm = None
        self._perm_calls = 0      # Philox call counter of the shuffle: one permutation per epoch
        self._seed = seed
        self._train_iter = None
        self._norm_iter = None
        self._mb_buffers = None
        self._mb_key = None

    # ---- data ------------------------------------------------------------------------------------
    def _take_samples(self):
        if self._train_iter is None:
            self._train_iter = iter(self._experience_dataset_fn())
        out = []
        for _ in range(self._num_samples):
            out.append(next(self._train_iter))
        return out

    def _update_normalizers(self):
        """ppo_learner.py:310-335: the agent's observation / reward normalisers are updated with
        `num_samples` elements of the normalisation dataset (the agent itself must not update
        them, :185-189), and the frames of those elements are counted."""
        if self._norm_iter is None:
            self._norm_iter = iter(self._normalization_dataset_fn())
        frames = 0
        for _ in range(self._num_samples):
            traj, _ = next(self._norm_iter)
            self._agent.update_observation_normalizer(traj.observation)
            self._agent.update_reward_normalizer(traj.reward)
            n = 1
            for d in traj.reward.shape[:2]:
                n *= int(d)
            frames += n
        return frames

    def _frames(self, samples):
        """The `num_samples` elements flattened to [F, ...] leaves + the permutation buffer."""
        trajs = [s[0] for s in samples]
        flat = []
        for tr in trajs:
            flat.append(nest_utils.map_structure(
                lambda t: t.reshape((-1,) + tuple(t.shape[2:])), tr))
        if len(flat) == 1:
            frames = flat[0]
        else:
            frames = nest_utils.map_structure(lambda *ts_: torch.cat(ts_, dim=0), *flat)
        F = int(frames.discount.shape[0])
        dev = frames.discount.device
        if self._perm is None or self._perm.numel() != F:
            self._perm = torch.empty((F,), dtype=torch.int64, device=dev)
        return frames, F, dev

    def _shuffle(self, F, dev):
        """One pseudo-random permutation of the F frames, computed index by index on the device
        (csrc/replay.hip: Feistel network keyed by Philox; oracle/perm.py)."""
        with torch.cuda.device(dev):
            _lib.check(_lib.load().aa_random_permutation(
                F, self._seed & 0xFFFFFFFFFFFFFFFF, self._perm_calls, self._perm.data_ptr(),
                _lib.stream_ptr()), "aa_random_permutation")
        self._perm_calls += 1
        return self._perm

    def _run_fused(self, samples, num_total_batches):
        """Every minibatch step of every epoch through PPOAgent.train_minibatches (one host call
        per epoch), or None when the agent / configuration does not take that path."""
        agent = self._agent
        gl = self._generic_learner
        if not (FUSED_EPOCHS and hasattr(agent, "train_minibatches") and
                gl._after_train_strategy_step_fn is None and self.num_replicas == 1 and
                not gl.use_kwargs_in_agent_train):
            return None
        frames, F, dev = self._frames(samples)
        mb = self._minibatch_size
        if not agent.fused_minibatches_ok(frames) or \
                (F // mb) * self._num_epochs != num_total_batches:
            return None
        loss_info = None
        for _ in range(self._num_epochs):
            perm = self._shuffle(F, dev)
            loss_info = agent.train_minibatches(frames, perm, mb, F // mb)
        return loss_info

    def _minibatches(self, samples):
        mb = self._minibatch_size
        frames, F, dev = self._frames(samples)
        # minibatches are gathered into ONE persistent buffer set: the train step's HIP graph is
        # bound to these addresses and replays without input copies (utils/graph.py)
        key = (mb, tuple((tuple(t.shape[1:]), t.dtype) for t in nest_utils.flatten(frames)))
        if self._mb_buffers is None or self._mb_key != key:
            self._mb_buffers = nest_utils.map_structure(
                lambda t: torch.empty((mb,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev),
                frames)
            self._mb_key = key
        bufs = self._mb_buffers
        views = nest_utils.map_structure(lambda t: t.unsqueeze(1), bufs)
        # every leaf of the minibatch in ONE row-gather launch (the replay table's gather kernel)
        srcs = [t.contiguous() for t in nest_utils.flatten(frames)]
        dsts = nest_utils.flatten(bufs)
        n = len(srcs)
        c_src = (ctypes.c_void_p * n)(*[t.data_ptr() for t in srcs])
        c_dst = (ctypes.c_void_p * n)(*[t.data_ptr() for t in dsts])
        c_rb = (ctypes.c_int64 * n)(*[t.element_size() * int(t[0].numel()) for t in srcs])
        lib = _lib.load()
        perm = self._perm
        for _ in range(self._num_epochs):
            # one pseudo-random permutation of the F frames per epoch -- no sort, no torch
            # arithmetic
            perm = self._shuffle(F, dev)
            for i in range(F // mb):
                idx = perm[i * mb:(i + 1) * mb]
                with torch.cuda.device(dev):
                    _lib.check(lib.aa_rb_gather_rows(c_src, c_dst, c_rb, n, None, None,
                                                     idx.data_ptr(), mb, _lib.stream_ptr()),
                               "aa_rb_gather_rows")
                yield views, None

    def _full_batches(self, samples):
        for _ in range(self._num_epochs):
            for s in samples:
                yield s

    # ---- run -------------------------------------------------------------------------------------
    def run(self, parallel_iterations=10):
        num_frames = self._update_normalizers()
        self.num_frames_for_training = num_frames
        samples = self._take_samples()
        if self._minibatch_size:
            # the reference sizes the run from the NORMALISATION dataset's frame count (:281-291)
            # and lets tf.data raise OutOfRange if the train dataset is shorter; here a train
            # dataset that cannot fill the run is reported up front, any other mismatch is
            # accepted with a warning like the reference accepts it
            train_frames = 0
            for traj, _ in samples:
                n = 1
                for d in traj.discount.shape[:2]:
                    n *= int(d)
                train_frames += n
            needed = int(num_frames / self._minibatch_size) * self._minibatch_size
            if train_frames < needed:
                raise ValueError(
                    "PPOLearner: the normalization dataset yielded {} frames but the experience "
                    "dataset only {} for num_samples={}: not enough to fill the {} frames of "
                    "minibatches per epoch the run is sized for.".format(
                        num_frames, train_frames, self._num_samples, needed))
            if train_frames != num_frames:
                import warnings
                warnings.warn(
                    "PPOLearner: the normalization dataset yielded {} frames, the experience "
                    "dataset {}; the run is sized from the former (ppo_learner.py:281-291)".format(
                        num_frames, train_frames))
            num_total_batches = int(num_frames / self._minibatch_size) * self._num_epochs
            if num_total_batches > 0:
                loss_info = self._run_fused(samples, num_total_batches)
                if loss_info is not None:
                    return self._generic_learner.finish_run(loss_info)
            it = self._minibatches(samples)
        else:
            num_total_batches = self._num_samples * self._num_epochs
            it = self._full_batches(samples)
        if num_total_batches == 0:
            raise ValueError(
                "Cannot distribute {} batches across {} replicas. Please increase "
                "PPOLearner.num_samples. See PPOLeaner.num_samples documentation for more "
                "details.".format(num_total_batches, self.num_replicas))
        return self._generic_learner.run(num_total_batches, it,
                                         parallel_iterations=parallel_iterations)

    @property
    def train_step_numpy(self):
        return self._generic_learner.train_step_numpy
