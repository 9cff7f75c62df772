"""DqnAgent / DdqnAgent on HIP kernels.

Same constructor and behaviour as tf_agents/agents/dqn/dqn_agent.py:82-753 for feed-forward
Q-networks:
  _initialize         :380-383   hard copy theta -> theta_target
  _train              :412-449   loss -> grads -> (per-tensor clip) -> optimizer -> counter -> target
  _loss               :462-579   n-step transition, q / next-q, TD target, Huber|squared, ~is_last
                                 mask, aggregate_losses (mean over the FULL batch, /replicas)
  _compute_q_values / _compute_next_q_values :581-645 (DQN), :659-700 (Double DQN)
  _get_target_updater :385-409   Periodically(period) -> soft_variables_update(tau)

One train step is: online forward (activations kept), target forward (+ online forward on the
next observation for DDQN) -> aa_dqn_td_loss (loss, td_error, dL/dq in one kernel) -> backward
GEMMs into the flat gradient buffer -> [gradient hook: the Learner's RCCL all-reduce] -> fused
optimizer over the flat parameter buffer -> target update.  Nothing syncs the device; LossInfo
holds device tensors.

Lifetime of the returned LossInfo: `train` returns VIEWS of the agent's persistent work buffers
(loss, td_loss[B], td_error[B]) -- also through the HIP-graph replay -- which the next train step
of the same batch size overwrites.  Consume them (or `.clone()`) before the next `train`; a
prioritized replay's `update_priorities(td_error)` enqueued right after `train` is ordered before
the next step on the stream and is safe.  `loss()` and `Learner.run` return tensors that own their
storage.
"""
import collections
import os

import numpy as np
import torch

from agents_amd import _lib, ops
from agents_amd.agents import tf_agent
from agents_amd.policies import q_policy
from agents_amd.utils import common, graph, nest_utils


# A/B knob: AA_TRAIN_SINGLE_STREAM=1 keeps the whole train step on the caller's stream (no target /
# weight-gradient side stream): fewer branches for the HIP-graph launch, no kernel overlap
_SINGLE_STREAM = os.environ.get("AA_TRAIN_SINGLE_STREAM", "0") == "1"
# AA_OPT_SUMS_SLABS=0: the conv weight gradients' slabs are summed by reduce launches into
# flat_grads even when nothing but the optimizer reads them (A/B measurements; bit-identical)
OPT_SUMS_SLABS = True
# AA_PACK_IN_OPTIMIZER=0: Learner.run's LossInfo scalars are packed by a launch of their own behind
# the optimizer step (A/B measurements; same values either way)
PACK_IN_OPTIMIZER = True


class DqnLossInfo(collections.namedtuple("DqnLossInfo", ("td_loss", "td_error"))):
    """td_loss / td_error per example, masked by ~is_last (dqn_agent.py:50-72)."""


def compute_td_targets(next_q_values, rewards, discounts):
    """rewards + discounts * next_q (dqn_agent.py:75-78); API parity helper."""
    return (rewards + discounts * next_q_values).detach()


class _Work:
    """Static per-batch-size outputs of the loss kernel (graph-capture friendly)."""

    def __init__(self, B, A, device):
        f = dict(dtype=torch.float32, device=device)
        self.loss = torch.zeros((1,), **f)
        self.td_loss = torch.zeros((B,), **f)
        self.td_error = torch.zeros((B,), **f)
        self.dq = torch.zeros((B, A), **f)
        self.head_done = False      # the last loss launch also ran the Q head's backward pass
        self.field_sums = torch.zeros((2,), **f)   # sum(td_loss), sum(td_error): Learner's reduce


class DqnAgent(tf_agent.TFAgent):
    _double_q = False

    def __init__(self, time_step_spec, action_spec, q_network, optimizer,
                 observation_and_action_constraint_splitter=None, epsilon_greedy=0.1,
                 n_step_update=1, boltzmann_temperature=None, emit_log_probability=False,
                 target_q_network=None, target_update_tau=1.0, target_update_period=1,
                 td_errors_loss_fn=None, gamma=1.0, reward_scale_factor=1.0,
                 gradient_clipping=None, debug_summaries=False, summarize_grads_and_vars=False,
                 train_step_counter=None, training_data_spec=None, name=None, seed=0):
        self._check_action_spec(action_spec)
        if epsilon_greedy is not None and boltzmann_temperature is not None:
            raise ValueError(
                "Configured both epsilon_greedy value {} and temperature {}, however only one of "
                "them can be used for exploration.".format(epsilon_greedy, boltzmann_temperature))
        self._boltzmann_temperature = boltzmann_temperature
        self._observation_and_action_constraint_splitter = \
            observation_and_action_constraint_splitter
        self._q_network = q_network
        net_observation_spec = time_step_spec.observation
        if observation_and_action_constraint_splitter:
            net_observation_spec, _ = observation_and_action_constraint_splitter(
                net_observation_spec)
        q_network.create_variables(net_observation_spec)
        if target_q_network is not None:
            target_q_network.create_variables(net_observation_spec)
            if target_q_network is q_network or \
                    target_q_network.flat_params.data_ptr() == q_network.flat_params.data_ptr():
                raise ValueError("target_q_network shares weights with the original network")
            self._target_q_network = target_q_network
        else:
            self._target_q_network = q_network.copy(name="TargetQNetwork")
        self._check_network_output(self._q_network, "q_network")
        self._check_network_output(self._target_q_network, "target_q_network")
        self._epsilon_greedy = epsilon_greedy
        self._n_step_update = n_step_update
        self._optimizer = optimizer
        self._td_errors_loss_fn = td_errors_loss_fn or common.element_wise_huber_loss
        self._gamma = gamma
        self._reward_scale_factor = reward_scale_factor
        self._gradient_clipping = gradient_clipping
        self._target_update_tau = target_update_tau
        # Early target forward (utils/graph.py: GraphedTrain): what a target forward computed ahead
        # of its train step depends on besides the batch -- writes of theta_target by this agent,
        # and launches outside the graphs that overwrite the target network's activation slot
        self._target_writes = 0
        self._target_fwd_epoch = 0
        self._update_target = self._get_target_updater(target_update_tau, target_update_period)
        self._seed = seed
        policy, collect_policy = self._setup_policy(time_step_spec, action_spec,
                                                    emit_log_probability)
        if getattr(q_network, "state_spec", ()) and n_step_update != 1:
            raise NotImplementedError(
                "DqnAgent does not currently support n-step updates with stateful networks "
                "(i.e., RNNs), but n_step_update = {}".format(n_step_update))
        super().__init__(time_step_spec, action_spec, policy, collect_policy,
                         train_sequence_length=n_step_update + 1,
                         debug_summaries=debug_summaries,
                         summarize_grads_and_vars=summarize_grads_and_vars,
                         train_step_counter=train_step_counter,
                         training_data_spec=training_data_spec)
        self._work = {}
        self._side_streams = {}
        self._seg_offsets = None
        self._seg_sumsq = None
        # Data-parallel hooks installed by train.Learner: number of replicas the loss is averaged
        # over (tf.nn.compute_average_loss, utils/common.py:1462-1467) and the gradient all-reduce.
        self.num_replicas = 1
        self.gradient_hook = None
        self.gradient_hook_async = None   # Work-returning variant (bucketed overlap)
        self.check_numerics = False  # the reference's check_numerics needs a device sync

    # ---- construction helpers -----------------------------------------------------------------
    def _check_action_spec(self, action_spec):
        flat = nest_utils.flatten(action_spec)
        if len(flat) > 1 or len(flat[0].shape) > 0:
            raise ValueError("Only scalar actions are supported now, but action spec is: {}"
                             .format(action_spec))
        spec = flat[0]
        if int(np.asarray(spec.minimum)) != 0:
            raise ValueError("Action specs should have minimum of 0, but saw: {0}".format(spec))
        self._num_actions = int(np.asarray(spec.maximum) - np.asarray(spec.minimum) + 1)

    def _check_network_output(self, net, label):
        out_shape = net.create_variables()
        if tuple(out_shape) != (self._num_actions,):
            raise ValueError(f"Expected {label} to emit a floating point tensor with inner dims "
                             f"({self._num_actions},); but saw network output spec: {out_shape}")

    def _setup_policy(self, time_step_spec, action_spec, emit_log_probability):
        splitter = self._observation_and_action_constraint_splitter
        policy = q_policy.QPolicy(time_step_spec, action_spec, q_network=self._q_network,
                                  emit_log_probability=emit_log_probability,
                                  observation_and_action_constraint_splitter=splitter,
                                  seed=self._seed)
        if self._boltzmann_temperature is not None:       # dqn_agent.py:357-360
            from agents_amd.policies import boltzmann_policy
            collect_policy = boltzmann_policy.BoltzmannPolicy(
                policy, temperature=self._boltzmann_temperature)
        else:
            collect_policy = q_policy.EpsilonGreedyPolicy(policy, epsilon=self._epsilon_greedy)
        greedy = q_policy.GreedyPolicy(policy)
        target_policy = q_policy.QPolicy(time_step_spec, action_spec,
                                         q_network=self._target_q_network,
                                         observation_and_action_constraint_splitter=splitter)
        self._target_greedy_policy = q_policy.GreedyPolicy(target_policy)
        return greedy, collect_policy

    def _get_target_updater(self, tau=1.0, period=1):
        import weakref
        me = weakref.ref(self)      # (the agent owns the updater: no cycle through the closure)

        def update():
            self = me()
            self._target_writes += 1
            out = common.soft_variables_update(self._q_network.flat_params,
                                               self._target_q_network.flat_params, tau)
            self._refresh_prepared(self._target_q_network)
            return out
        return common.Periodically(update, period, "periodic_update_targets")

    # Prepared weights (networks/sequential.py): this agent is the only writer of its networks'
    # parameters through raw-pointer kernels (optimizer step, soft target update), so it re-runs the
    # bf16x6 filter pre-passes right there -- once per write instead of once per forward / backward
    # that reads the weights (policy forward, online forward, target forward, two input gradients).
    def _enable_prepared(self):
        from agents_amd.networks import sequential
        if sequential.PREPARED_WEIGHTS:
            for net in (self._q_network, self._target_q_network):
                if hasattr(net, "enable_prepared_weights") and net.flat_params is not None:
                    net.enable_prepared_weights()

    @staticmethod
    def _refresh_prepared(net):
        if getattr(net, "_pw", None) is not None:
            net.refresh_prepared()

    def post_replicated_state_update(self):
        """Learner.sync_replicas calls this after broadcasting rank 0's parameters."""
        self._target_writes += 1
        self._refresh_prepared(self._q_network)
        self._refresh_prepared(self._target_q_network)

    # ---- TFAgent implementation ----------------------------------------------------------------
    def _initialize(self):
        common.soft_variables_update(self._q_network.flat_params,
                                     self._target_q_network.flat_params, tau=1.0)
        self._enable_prepared()
        self.post_replicated_state_update()

    def _side_stream(self, device):
        if _SINGLE_STREAM:
            return None
        key = (device.type, device.index)
        st = self._side_streams.get(key)
        if st is None:
            st = self._side_streams[key] = ops.new_side_stream(device)
        return st

    def _get_work(self, B, device):
        w = self._work.get(B)
        if w is None:
            w = _Work(B, self._num_actions, device)
            self._work[B] = w
        return w

    def _loss_kind(self, fn):
        """AA_LOSS_HUBER / AA_LOSS_SQUARED for the two `common.*` functions (the fused loss kernel
        implements them), None for any other callable (`_custom_loss`)."""
        return getattr(fn, "aa_loss_kind", None)

    def _custom_loss(self, fn, w, experience, weights, B, need_grad):
        """An arbitrary `td_errors_loss_fn(td_targets, q_values)` (agents/dqn/dqn_agent.py:114,
        250-251, 458): the kernel has left td_targets in w.td_loss and q_values in w.td_error
        (AA_LOSS_TARGETS); the callable is evaluated on those two [B] tensors with torch, its
        derivative with respect to q_values comes from torch.autograd, and mask / weights /
        aggregation follow `_loss` (:514-538) and `common.aggregate_losses` (common.py:1400-1476).
        This path is NOT captured into HIP graphs (`graph_train_ok` is False for such an agent: the
        train step runs eagerly), and it is the one place of the DQN train step where torch
        evaluates arithmetic -- the user's own."""
        td_targets = w.td_loss.clone()
        q = w.td_error.clone().requires_grad_(need_grad)
        with torch.enable_grad() if need_grad else torch.no_grad():
            elem = fn(td_targets, q)
            if elem.shape != q.shape:
                raise ValueError("td_errors_loss_fn must return one loss per sample: got shape "
                                 f"{tuple(elem.shape)} for {tuple(q.shape)} q-values")
            valid = (experience.step_type[:, 0] != 2).to(torch.float32)
            td_loss = valid * elem
            weighted = td_loss
            if weights is not None:
                # tf.math.multiply_no_nan(losses, weights)
                weighted = torch.where(weights == 0, torch.zeros_like(td_loss), td_loss * weights)
            total = weighted.sum() / float(B * self.num_replicas)
            if need_grad:
                (gq,) = torch.autograd.grad(total, q)
                acts = experience.action[:, 0] if experience.action.dim() == 2 \
                    else experience.action
                w.dq.zero_()
                w.dq.scatter_(1, acts.to(torch.int64).reshape(B, 1), gq.reshape(B, 1))
        w.td_error.copy_(valid * (td_targets - q.detach()))
        w.td_loss.copy_(td_loss.detach())
        w.loss.copy_(total.detach().reshape(1))
        w.field_sums[0] = w.td_loss.sum()
        w.field_sums[1] = w.td_error.sum()

    def _early_target_key(self):
        """Changes whenever a target forward computed earlier (GraphedTrain's early target
        forward) may have gone stale: theta_target written by this agent or by a torch op, or the
        target network's `train` activation slot overwritten by an un-captured launch."""
        return (self._target_writes, self._target_fwd_epoch,
                self._target_q_network.flat_params._version)

    def _train_phase_target(self, experience):
        """The target network's forward on the batch's LAST observation, alone: depends on the
        batch and on theta_target only, not on the optimizer step in front of the train step that
        consumes it (`_train_phase_grads(..., q_next_target=...)`)."""
        obs = experience.observation
        if self._observation_and_action_constraint_splitter is not None:
            obs, _ = self._observation_and_action_constraint_splitter(obs)
        return self._target_q_network.forward(obs[:, -1], slot="train")

    def _forward_and_loss(self, experience, td_errors_loss_fn, gamma, reward_scale_factor, weights,
                          need_grad, q_next_target=None):
        obs = experience.observation
        mask = None
        if self._observation_and_action_constraint_splitter is not None:
            obs, mask = self._observation_and_action_constraint_splitter(obs)
        B = experience.discount.shape[0]
        dev = experience.discount.device
        graph.join_lanes(dev)
        w = self._get_work(B, dev)
        obs_t = obs[:, 0]
        obs_next = obs[:, -1]
        # The online and the target forward are independent chains of small kernels: run the
        # target one on a side stream so the two overlap (fork / join, also under graph capture).
        main = torch.cuda.current_stream(dev)
        side = self._side_stream(dev)
        forked = False
        if q_next_target is not None:
            pass    # computed ahead of this step (GraphedTrain's early target forward)
        elif side is None:
            if not graph.capturing():
                self._target_fwd_epoch += 1
            q_next_target = self._target_q_network.forward(obs_next, slot="train")
        else:
            if not graph.capturing():
                self._target_fwd_epoch += 1
            forked = True
            if hasattr(self._target_q_network, "prepare_forward"):
                self._target_q_network.prepare_forward(obs_next.shape[0], slot="train")
            side.wait_stream(main)
            with ops.side_line(side):
                q_next_target = self._target_q_network.forward(obs_next, slot="train")
        q_online = self._q_network.forward(obs_t, slot="train", need_grad=need_grad)
        q_next_select = None
        if self._double_q:
            q_next_select = self._q_network.forward(obs_next, slot="next")
        if forked:
            main.wait_stream(side)
        next_mask = None
        if mask is not None:
            next_mask = mask[:, -1].to(torch.int32).contiguous()
        if weights is not None:
            if not isinstance(weights, torch.Tensor):
                weights = torch.full((B,), float(weights), dtype=torch.float32, device=dev)
            elif weights.dim() == 0:
                weights = weights.to(torch.float32).expand(B).contiguous()
            else:
                weights = weights.to(torch.float32).contiguous()
        st = experience.step_type
        if st.dtype != torch.int32:
            st = st.to(torch.int32)
        # training: the Q head's backward pass rides in the loss launch (dL/dq of a sample only
        # needs that sample: csrc/dqn.hip), one launch less on the critical chain
        head = None
        if need_grad and hasattr(self._q_network, "fusable_head"):
            head = self._q_network.fusable_head(B, "train")
        fn = td_errors_loss_fn or self._td_errors_loss_fn
        kind = self._loss_kind(fn)
        if kind is None:
            head = None       # the head's backward needs dL/dq, which torch produces below
        w.head_done = head is not None
        ops.dqn_td_loss(q_online, q_next_target, q_next_select, next_mask,
                        experience.action, experience.reward.contiguous(),
                        experience.discount.contiguous(), st.contiguous(), weights,
                        self._gamma, reward_scale_factor,
                        _lib.AA_LOSS_TARGETS if kind is None else kind,
                        float(B * self.num_replicas), w.loss, w.td_loss, w.td_error, w.dq,
                        gamma_loss=gamma, field_sums_out=w.field_sums, head=head)
        if kind is None:
            self._custom_loss(fn, w, experience if st is experience.step_type else
                              experience._replace(step_type=st), weights, B, need_grad)
        return w

    def reduce_loss_info(self, loss_info):
        """Learner.run sums every LossInfo field over all axes (train/learner.py:322-337).  For the
        LossInfo of the last train step the loss kernel has already produced those sums: returns a
        LossInfo of scalars (views of the work buffer), or None for any other LossInfo."""
        extra = getattr(loss_info, "extra", None)
        td = getattr(extra, "td_loss", None)
        self.reduced_owns_storage = False
        packed = getattr(self, "_packed", None)
        if packed is not None and td is packed[0].td_loss and extra.td_error is packed[0].td_error \
                and packed[2] + 1 == int(self._train_step_counter) \
                and not self._q_network.has_regularization:
            # (the step tag: only the eager optimizer phase refreshes `_packed`; a step whose
            # optimizer phase REPLAYED from its graph must not be answered with an older step's
            # sums -- the work-buffer branch below serves it)
            # the optimizer launch of this step already copied the three sums into storage of
            # their own (`_train_phase_apply`): nothing left to launch
            v = packed[1].unbind(0)
            self.reduced_owns_storage = True
            return tf_agent.LossInfo(v[0], DqnLossInfo(td_loss=v[1], td_error=v[2]))
        for w in self._work.values():
            if td is w.td_loss and extra.td_error is w.td_error:
                return tf_agent.LossInfo(loss_info.loss, DqnLossInfo(
                    td_loss=w.field_sums[0], td_error=w.field_sums[1]))
        return None

    def _loss(self, experience, td_errors_loss_fn=None, gamma=1.0, reward_scale_factor=1.0,
              weights=None, training=False):
        """Scalar loss + DqnLossInfo.  NOTE: like the reference, `gamma` here defaults to 1.0 and
        only scales the bootstrap discount; `_train` passes the agent's gamma (dqn_agent.py:
        412-421,462-470)."""
        with torch.cuda.device(experience.discount.device):
            w = self._forward_and_loss(experience, td_errors_loss_fn, gamma, reward_scale_factor,
                                       weights, need_grad=False)
            total = w.loss.clone()
            if self._q_network.has_regularization:
                total = total + self._q_network.regularization_loss() / self.num_replicas
        return tf_agent.LossInfo(total.reshape(()),
                                 DqnLossInfo(td_loss=w.td_loss.clone(),
                                             td_error=w.td_error.clone()))

    def _clip_gradients(self, net):
        """Per-tensor tf.clip_by_norm (eager_utils.clip_gradient_norms, eager_utils.py:227-246)."""
        lib = _lib.load()
        dev = net.flat_grads.device
        if self._seg_offsets is None:
            # consecutive variables: [start_i, start_{i+1}); alignment padding holds zeros
            starts = [s0 for s0, _ in net.segment_offsets()] + [net.flat_grads.numel()]
            self._seg_offsets = torch.tensor(starts, dtype=torch.int64, device=dev)
            self._seg_sumsq = torch.zeros((len(starts) - 1,), dtype=torch.float32, device=dev)
        st = _lib.stream_ptr()
        n_seg = self._seg_sumsq.numel()
        _lib.check(lib.aa_segment_sumsq(net.flat_grads.data_ptr(), self._seg_offsets.data_ptr(),
                                        n_seg, self._seg_sumsq.data_ptr(), st),
                   "aa_segment_sumsq")
        _lib.check(lib.aa_clip_by_norm(net.flat_grads.data_ptr(), self._seg_offsets.data_ptr(),
                                       n_seg, self._seg_sumsq.data_ptr(),
                                       float(self._gradient_clipping), 1, st), "aa_clip_by_norm")

    # The train step is split in three so that it can be replayed from HIP graphs
    # (agents_amd/utils/graph.py): two capturable device phases around the gradient hook (the
    # Learner's RCCL all-reduce) and a host phase (counters, periodic target update).
    def _train_phase_grads(self, experience, weights, q_next_target=None):
        """forward(s) + loss + backward (+ regulariser, clipping): fills flat_grads.
        `q_next_target`: the output of `_train_phase_target` on this batch, already computed."""
        net = self._q_network
        w = self._forward_and_loss(experience, self._td_errors_loss_fn, self._gamma,
                                   self._reward_scale_factor, weights, need_grad=True,
                                   q_next_target=q_next_target)
        self._last_work = w
        # head_done only exists in networks that exposed `fusable_head` (Sequential): a q_network
        # with the plain backward(dout, slot, side_stream, stop_layer) contract never sees it
        extra = {"head_done": True} if w.head_done else {}
        if self._optimizer_sums_slabs(net):
            extra["keep_dw_slabs"] = True
        net.backward(w.dq, slot="train", side_stream=self._side_stream(w.dq.device), **extra)
        total = w.loss
        if net.has_regularization:
            net.add_regularization_grads(1.0 / self.num_replicas)
            total = w.loss + net.regularization_loss() / self.num_replicas
        if self._gradient_clipping is not None:
            self._clip_gradients(net)
        return tf_agent.LossInfo(total.reshape(()),
                                 DqnLossInfo(td_loss=w.td_loss, td_error=w.td_error))

    # Two-bucket variant for data-parallel runs: the dense tail (fc1 + fc2 = 95 % of the Atari
    # net's parameters) finishes its gradients first, so the Learner can start their all-reduce
    # while the conv layers are still in backward (utils/graph.py: GraphedTrain, bucket mode).
    def _bucket_split(self):
        """Layer index where the gradient buffer is split, or None when one bucket is required
        (per-replica clipping / regularisation run over the whole buffer before the reduce)."""
        net = self._q_network
        if net.has_regularization or self._gradient_clipping is not None:
            return None
        k = net.dense_tail_start()
        return k if 0 < k < len(net._param_layers) else None

    def _train_phase_grads_a(self, experience, weights, q_next_target=None):
        """First half of the gradient phase: everything above the data-parallel bucket split."""
        net = self._q_network
        w = self._forward_and_loss(experience, self._td_errors_loss_fn, self._gamma,
                                   self._reward_scale_factor, weights, need_grad=True,
                                   q_next_target=q_next_target)
        self._last_work = w
        self._bucket_B = w.dq.shape[0]
        self._split_at = self._bucket_split()
        extra = {"head_done": True} if w.head_done else {}
        net.backward(w.dq, slot="train", side_stream=self._side_stream(w.dq.device),
                     stop_layer=self._split_at, **extra)
        return tf_agent.LossInfo(w.loss.reshape(()),
                                 DqnLossInfo(td_loss=w.td_loss, td_error=w.td_error))

    def _train_phase_grads_b(self):
        net = self._q_network
        net.backward_resume(self._bucket_B, slot="train",
                            side_stream=self._side_stream(net.flat_grads.device),
                            from_layer=self._split_at)

    def _optimizer_sums_slabs(self, net):
        """Nothing reads flat_grads between backward and the optimizer step (no clipping, no
        regulariser, no gradient hook = no all-reduce) and the optimizer can take the conv weight
        gradients as unsummed slabs: the reduce launches of the backward pass are dropped."""
        return OPT_SUMS_SLABS and self.gradient_hook is None and \
            self._gradient_clipping is None and not net.has_regularization and \
            getattr(self._optimizer, "supports_grad_slabs", False) and \
            hasattr(net, "take_grad_slabs")

    @property
    def graph_train_ok(self):
        """False when the train step cannot be recorded into HIP graphs: a td_errors_loss_fn other
        than the two fused ones is evaluated by torch (with autograd) inside the step."""
        return self._loss_kind(self._td_errors_loss_fn) is not None

    def _graph_capture_key(self):
        """What `_train_phase_grads` decides on the host while it is recorded (GraphedTrain replays
        an entry only under the state it was captured in): with a gradient hook installed the
        backward pass must sum its weight gradients into flat_grads for the all-reduce."""
        return (self.gradient_hook is None, self.gradient_hook_async is None)

    # GraphedTrain: the optimizer phase of an entry replays behind ITS gradient graph.  State =
    # (the unsummed weight-gradient slabs the backward pass left, the work buffers its loss wrote)
    def _apply_state(self):
        net = self._q_network
        slabs = net.take_grad_slabs() if hasattr(net, "take_grad_slabs") else None
        return None if slabs is None else (slabs, getattr(self, "_last_work", None))

    def _set_apply_state(self, state):
        slabs, work = state if state is not None else (None, None)
        if hasattr(self._q_network, "set_grad_slabs"):
            self._q_network.set_grad_slabs(slabs)
        if work is not None:
            self._last_work = work

    def _train_phase_apply(self):
        net = self._q_network
        planes = net.plane_scatter() if hasattr(net, "plane_scatter") else None
        slabs = net.take_grad_slabs() if hasattr(net, "take_grad_slabs") else None
        self._packed = None
        if slabs is not None:
            pack = None
            w = getattr(self, "_last_work", None)
            if PACK_IN_OPTIMIZER and w is not None and not graph.capturing() and \
                    getattr(self._optimizer, "supports_pack", False):
                # the three sums Learner.run returns (reduce_loss_info) leave with this launch, in
                # storage of their own: a fresh 3-float tensor per step, like the copy it replaces
                import ctypes
                vec = torch.empty((3,), dtype=torch.float32, device=net.flat_params.device)
                src = w.__dict__.get("_pack_src")
                if src is None:
                    src = w._pack_src = (ctypes.c_void_p * 3)(
                        w.loss.data_ptr(), w.field_sums.data_ptr(), w.field_sums.data_ptr() + 4)
                pack = (src, vec)
                self._packed = (w, vec, int(self._train_step_counter))
            self._optimizer.apply_flat(net.flat_params, net.flat_grads, planes=planes,
                                       grad_slabs=slabs, **({"pack": pack} if pack else {}))
            if planes is None:
                self._refresh_prepared(net)
        elif planes is not None and getattr(self._optimizer, "supports_planes", False):
            # the optimizer kernel writes the new filters' bf16 pieces into the prepared planes
            self._optimizer.apply_flat(net.flat_params, net.flat_grads, planes=planes)
        else:
            self._optimizer.apply_flat(net.flat_params, net.flat_grads)
            self._refresh_prepared(net)

    def _train_phase_host(self):
        self._train_step_counter.assign_add(1)
        self._update_target()

    def _train(self, experience, weights):
        net = self._q_network
        with torch.cuda.device(experience.discount.device):
            loss_info = self._train_phase_grads(experience, weights)
            if self.check_numerics and not bool(torch.isfinite(loss_info.loss).all()):
                raise FloatingPointError("Loss is inf or nan")
            if self.gradient_hook is not None:
                self.gradient_hook(net.flat_grads)
            self._train_phase_apply()
            self._train_phase_host()
        return loss_info

    def replicated_state(self):
        """Tensors every data-parallel replica must hold identically (train.Learner broadcasts
        rank 0's at construction): online and target parameters, existing optimizer slots."""
        return [self._q_network.flat_params, self._target_q_network.flat_params] + \
            (self._optimizer.variables() if self._optimizer is not None else [])

    # ---- checkpointing ---------------------------------------------------------------------------
    def state_dict(self):
        graph.join_lanes(self._q_network.flat_params.device)
        return {"q": self._q_network.flat_params.clone(),
                "target": self._target_q_network.flat_params.clone(),
                "train_step": int(self._train_step_counter),
                "target_update_calls": self._update_target._counter,
                "optimizer": self._optimizer.state_dict() if self._optimizer else None,
                "collect_policy": self._collect_policy.state_dict()}

    def load_state_dict(self, sd):
        """Restores IN PLACE: captured HIP graphs keep pointing at the same buffers."""
        graph.join_lanes(self._q_network.flat_params.device)
        self._q_network.flat_params.copy_(sd["q"])
        self._target_q_network.flat_params.copy_(sd["target"])
        self._target_writes += 1
        self._train_step_counter.assign(sd["train_step"])
        self._update_target._counter = int(sd.get("target_update_calls", 0))
        if sd.get("optimizer") is not None and self._optimizer is not None:
            self._optimizer.load_state_dict(sd["optimizer"])
        if sd.get("collect_policy") is not None:
            self._collect_policy.load_state_dict(sd["collect_policy"])
        self._initialized = True
        self._enable_prepared()
        self.post_replicated_state_update()


class DdqnAgent(DqnAgent):
    """Double DQN: the online network selects the next action, the target network evaluates it
    (dqn_agent.py:659-700)."""
    _double_q = True
