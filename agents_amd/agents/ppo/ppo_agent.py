"""PPOAgent on HIP kernels.

Same constructor, defaults and training flow as tf_agents/agents/ppo/ppo_agent.py:113-1690 for
feed-forward actor / value networks with a diagonal-Normal action distribution:
  compute_advantages              :440-479   GAE via aa_gae (bootstrap quirk reproduced)
  get_loss                        :481-615   aa_ppo_loss_dist (pg, value, entropy, KL penalty)
  compute_return_and_advantage    :617-719   aa_ppo_discounts + aa_discounted_return (+ aa_gae)
  _preprocess                     :721-807   value predictions, returns/advantages padded by a 0
  _train                          :834-1076  mask, old log-probs, advantage normalisation,
                                             num_epochs x (loss, backward, global-norm clip, Adam)
  l2_regularization_loss          :1088-1157 aa_sumsq_f32 / aa_add_l2_grad over kernel weights
  kl_penalty_loss & friends       :1514-1690 fused in aa_ppo_loss_dist; aa_ppo_update_kl_beta
Actor, std bias and value parameters live in ONE flat fp32 buffer (gradients likewise), so the
global-norm clip, the optimizer step and the Learner's RCCL all-reduce are single passes.
Reward / observation normalisers (:347-366, 650-652, 991-993, 1078-1086) are
utils/tensor_normalizer.StreamingTensorNormalizer (csrc/normalizer.hip).
Not implemented (raise NotImplementedError): RNN networks, discrete action distributions.
"""
import collections
import os

import numpy as np
import torch

from agents_amd import _lib, ops
from agents_amd.agents import tf_agent
from agents_amd.agents.ppo import ppo_policy
from agents_amd.networks import network
from agents_amd.specs import tensor_spec
from agents_amd.utils import common, graph, nest_utils, tensor_normalizer

# AA_PPO_FUSED=0: the layer-by-layer train step (A/B measurements; also the path for shapes the
# fused step does not take)
FUSED_STEP = os.environ.get("AA_PPO_FUSED", "1") != "0"

PPOLossInfo = collections.namedtuple(
    "PPOLossInfo", ("policy_gradient_loss", "value_estimation_loss", "l2_regularization_loss",
                    "entropy_regularization_loss", "kl_penalty_loss", "clip_fraction"))


class PPOAgent(tf_agent.TFAgent):
    def __init__(self, time_step_spec, action_spec, optimizer=None, actor_net=None, value_net=None,
                 greedy_eval=True, importance_ratio_clipping=0.0, lambda_value=0.95,
                 discount_factor=0.99, entropy_regularization=0.0, policy_l2_reg=0.0,
                 value_function_l2_reg=0.0, shared_vars_l2_reg=0.0, value_pred_loss_coef=0.5,
                 num_epochs=25, use_gae=False, use_td_lambda_return=False, normalize_rewards=True,
                 reward_norm_clipping=10.0, normalize_observations=True, log_prob_clipping=0.0,
                 kl_cutoff_factor=2.0, kl_cutoff_coef=1000.0, initial_adaptive_kl_beta=1.0,
                 adaptive_kl_target=0.01, adaptive_kl_tolerance=0.3, gradient_clipping=None,
                 value_clipping=None, check_numerics=False,
                 compute_value_and_advantage_in_train=True, update_normalizers_in_train=True,
                 aggregate_losses_across_replicas=True, debug_summaries=False,
                 summarize_grads_and_vars=False, train_step_counter=None, name=None, seed=0):
        if not isinstance(actor_net, network.Network):
            raise TypeError("actor_net must be an instance of a network.Network.")
        if not isinstance(value_net, network.Network):
            raise TypeError("value_net must be an instance of a network.Network.")
        # shared_vars_l2_reg (ppo_agent.py:1130-1146) penalises the variables the actor and the
        # value network SHARE; the networks here own disjoint segments of one flat buffer, so
        # the set is empty and the term is identically zero: accepted, nothing to compute.
        self._shared_vars_l2_reg = shared_vars_l2_reg
        # aggregate_losses_across_replicas=False (ppo_agent.py:1170-1181, 1281-1292, 1403-1408):
        # policy-gradient, value and entropy terms are tf.reduce_mean over the LOCAL batch instead
        # of common.aggregate_losses' sum / (batch x replicas); see `_loss_denominator`
        self._aggregate_losses_across_replicas = bool(aggregate_losses_across_replicas)
        actor_net.create_variables(time_step_spec.observation)
        value_net.create_variables(time_step_spec.observation)
        self._optimizer = optimizer
        self._actor_net = actor_net
        self._value_net = value_net
        self._importance_ratio_clipping = float(importance_ratio_clipping)
        self._lambda = float(lambda_value)
        self._discount_factor = float(discount_factor)
        self._entropy_regularization = float(entropy_regularization)
        self._policy_l2_reg = float(policy_l2_reg)
        self._value_function_l2_reg = float(value_function_l2_reg)
        self._value_pred_loss_coef = float(value_pred_loss_coef)
        self._num_epochs = int(num_epochs)
        self._use_gae = use_gae
        self._use_td_lambda_return = use_td_lambda_return
        self._log_prob_clipping = float(log_prob_clipping)
        self._kl_cutoff_factor = float(kl_cutoff_factor)
        self._kl_cutoff_coef = float(kl_cutoff_coef)
        self._adaptive_kl_target = float(adaptive_kl_target)
        self._adaptive_kl_tolerance = float(adaptive_kl_tolerance)
        self._gradient_clipping = float(gradient_clipping or 0.0)
        self._value_clipping = float(value_clipping or 0.0)
        self._check_numerics = check_numerics
        self._compute_value_and_advantage_in_train = compute_value_and_advantage_in_train
        self.update_normalizers_in_train = update_normalizers_in_train
        self._initial_adaptive_kl_beta = float(initial_adaptive_kl_beta)
        self._seed = seed

        # ---- one flat parameter / gradient buffer: [actor | value] ------------------------------
        dev = actor_net.body.flat_params.device
        na, nv = actor_net.flat_size, value_net.flat_size
        self.flat_params = torch.zeros((na + nv,), dtype=torch.float32, device=dev)
        self.flat_grads = torch.zeros_like(self.flat_params)
        actor_net.rebind(self.flat_params[:na], self.flat_grads[:na])
        value_net.rebind(self.flat_params[na:], self.flat_grads[na:])
        self._adaptive_kl_beta = None
        if initial_adaptive_kl_beta > 0.0:
            self._adaptive_kl_beta = torch.full((1,), float(initial_adaptive_kl_beta),
                                                dtype=torch.float32, device=dev)
        self._device = dev
        # ppo_agent.py:347-360
        self._reward_norm_clipping = float(reward_norm_clipping)
        self._reward_normalizer = None
        if normalize_rewards:
            self._reward_normalizer = tensor_normalizer.StreamingTensorNormalizer(
                tensor_spec.TensorSpec((), torch.float32), scope="normalize_reward", device=dev)
        self._observation_normalizer = None
        if normalize_observations:
            self._observation_normalizer = tensor_normalizer.StreamingTensorNormalizer(
                time_step_spec.observation, scope="normalize_observations", device=dev)

        policy = ppo_policy.PPOPolicy(time_step_spec, action_spec, actor_net, value_net,
                                      observation_normalizer=self._observation_normalizer,
                                      clip=False, collect=False, greedy=greedy_eval, seed=seed)
        collect_policy = ppo_policy.PPOPolicy(
            time_step_spec, action_spec, actor_net, value_net,
            observation_normalizer=self._observation_normalizer, clip=False, collect=True,
            compute_value_and_advantage_in_train=compute_value_and_advantage_in_train,
            seed=seed + 1)
        if compute_value_and_advantage_in_train:
            training_data_spec = None
        else:
            info = dict(collect_policy.trajectory_spec.policy_info)
            info["return"] = tensor_spec.TensorSpec((), torch.float32)
            info["advantage"] = tensor_spec.TensorSpec((), torch.float32)
            training_data_spec = collect_policy.trajectory_spec.replace(policy_info=info)
        super().__init__(time_step_spec, action_spec, policy, collect_policy,
                         train_sequence_length=None, training_data_spec=training_data_spec,
                         debug_summaries=debug_summaries,
                         summarize_grads_and_vars=summarize_grads_and_vars,
                         train_step_counter=train_step_counter)
        spec = nest_utils.flatten(action_spec)[0]
        self._D = int(np.prod(spec.shape)) if len(spec.shape) else 1
        self._obs_rank = len(time_step_spec.observation.shape)
        self._work = {}
        # [0, total] segment table of the global-norm clip + its accumulator: created here, not
        # lazily inside `_train` -- an H2D copy is not allowed while a HIP graph is being captured
        self._norm_seg = torch.tensor([0, self.flat_grads.numel()], dtype=torch.int64,
                                      device=self.flat_grads.device)
        self._norm_sumsq = torch.zeros((1,), dtype=torch.float32, device=self.flat_grads.device)
        self.num_replicas = 1       # installed by train.Learner
        self.gradient_hook = None
        self._clip_fraction = 0.0
        self._grad_norm = None
        self._zero = None

    # ---- accessors ------------------------------------------------------------------------------
    @property
    def actor_net(self):
        return self._actor_net

    @property
    def adaptive_kl_beta(self):
        return self._adaptive_kl_beta

    def _initialize(self):
        pass

    def replicated_state(self):
        """Tensors every data-parallel replica must hold identically (train.Learner broadcasts
        rank 0's at construction)."""
        out = [self.flat_params, self._adaptive_kl_beta]
        out += self._optimizer.variables() if self._optimizer is not None else []
        for n in (self._reward_normalizer, self._observation_normalizer):
            if n is not None:
                out += list(n._state)
        return out

    # ---- checkpointing ---------------------------------------------------------------------------
    def state_dict(self):
        return {"params": self.flat_params.clone(),
                "train_step": int(self._train_step_counter),
                "optimizer": self._optimizer.state_dict() if self._optimizer else None,
                "adaptive_kl_beta": None if self._adaptive_kl_beta is None
                else self._adaptive_kl_beta.clone(),
                "policies": [self._policy.state_dict(), self._collect_policy.state_dict()],
                "normalizers": [None if n is None else n.state_dict() for n in
                                (self._reward_normalizer, self._observation_normalizer)]}

    def load_state_dict(self, sd):
        self.flat_params.copy_(sd["params"])
        self._train_step_counter.assign(sd["train_step"])
        if sd.get("optimizer") is not None and self._optimizer is not None:
            self._optimizer.load_state_dict(sd["optimizer"])
        if sd.get("adaptive_kl_beta") is not None and self._adaptive_kl_beta is not None:
            self._adaptive_kl_beta.copy_(sd["adaptive_kl_beta"])
        self._policy.load_state_dict(sd["policies"][0])
        self._collect_policy.load_state_dict(sd["policies"][1])
        for n, nsd in zip((self._reward_normalizer, self._observation_normalizer),
                          sd.get("normalizers", (None, None))):
            if n is not None and nsd is not None:
                n.load_state_dict(nsd)
        self._initialized = True

    def _st(self):
        return _lib.stream_ptr()

    def _w(self, N):
        w = self._work.get(N)
        if w is None:
            f = dict(dtype=torch.float32, device=self._device)
            w = {k: torch.zeros((N,), **f) for k in
                 ("old_logp", "adv_norm", "mask", "dv")}
            w["dloc"] = torch.zeros((N, self._D), **f)
            w["dscale"] = torch.zeros((N, self._D), **f)
            w["stats"] = torch.zeros((_lib.AA_PPO_DIST_STATS,), **f)
            w["norm_stats"] = torch.zeros((2 + 256,), **f)
            w["reg"] = torch.zeros((1,), **f)
            self._work[N] = w
        return w

    # ---- return / advantage --------------------------------------------------------------------
    def compute_advantages(self, rewards, returns, discounts, value_preds):
        """[B, T] advantages; value_preds is [B, T+1] (ppo_agent.py:440-479)."""
        lib = _lib.load()
        vp = value_preds[:, :-1]  # NOTE: GAE then bootstraps from vp[:, -1] = V(s_{T-1})
        B, T = vp.shape
        if not self._use_gae:
            return returns - vp
        out = torch.empty((B, T), dtype=torch.float32, device=vp.device)
        vpc = vp.contiguous()  # aa_gae reads every operand with one (stride_b, stride_t)
        final = vpc[:, -1].contiguous()
        r = rewards.contiguous()
        d = discounts.contiguous()
        _lib.check(lib.aa_gae(vpc.data_ptr(), final.data_ptr(), d.data_ptr(), r.data_ptr(),
                              self._lambda, B, T, T, 1, out.data_ptr(), self._st()), "aa_gae")
        return out

    def compute_return_and_advantage(self, next_time_steps, value_preds):
        """next_time_steps fields are [B, T]; value_preds [B, T+1] (ppo_agent.py:617-719)."""
        lib = _lib.load()
        B, T = next_time_steps.discount.shape
        dev = value_preds.device
        rewards = next_time_steps.reward.contiguous()
        if self._reward_normalizer is not None:     # ppo_agent.py:650-654
            rewards = self._reward_normalizer.normalize(
                rewards, center_mean=False, clip_value=self._reward_norm_clipping)
        # discount * gamma * episode_mask on a [B, T] view: pad the kernel's [B, T+1] contract
        disc_in = torch.empty((B, T + 1), dtype=torch.float32, device=dev)
        disc_in[:, :T].copy_(next_time_steps.discount)
        st_in = torch.zeros((B, T + 1), dtype=torch.int32, device=dev)
        st_in[:, :T].copy_(next_time_steps.step_type)
        discounts = torch.empty((B, T), dtype=torch.float32, device=dev)
        _lib.check(lib.aa_ppo_discounts(disc_in.data_ptr(), st_in.data_ptr(),
                                        self._discount_factor, B, T + 1, discounts.data_ptr(),
                                        self._st()), "aa_ppo_discounts")
        final = value_preds[:, -1].contiguous()
        returns = torch.empty((B, T), dtype=torch.float32, device=dev)
        _lib.check(lib.aa_discounted_return(rewards.data_ptr(), discounts.data_ptr(),
                                            final.data_ptr(), B, T, T, 1, returns.data_ptr(),
                                            self._st()), "aa_discounted_return")
        advantages = self.compute_advantages(rewards, returns, discounts, value_preds)
        if self._use_td_lambda_return and self._use_gae:
            returns = advantages + value_preds[:, :-1]
        return returns, advantages

    def _preprocess(self, experience):
        from agents_amd.trajectories import time_step as ts
        disc = experience.discount
        if disc.dim() == 1:
            experience = nest_utils.map_structure(lambda t: t.unsqueeze(0), experience)
            unbatch = True
        else:
            unbatch = False
        B, T1 = experience.discount.shape
        if T1 <= 1:
            raise ValueError("Experience used for advantage calculation must have >1 num_steps.")
        with torch.cuda.device(experience.discount.device):
            if self._compute_value_and_advantage_in_train:
                value_preds, _ = self._collect_policy.apply_value_network(
                    experience.observation, experience.step_type, training=False)
                value_preds = value_preds.clone()
            else:
                value_preds = experience.policy_info["value_prediction"]
            next_ts = ts.TimeStep(step_type=experience.next_step_type[:, :-1],
                                  reward=experience.reward[:, :-1],
                                  discount=experience.discount[:, :-1],
                                  observation=None)
            returns, advantages = self.compute_return_and_advantage(next_ts, value_preds)
            pad = torch.zeros((B, 1), dtype=torch.float32, device=returns.device)
            info = {"dist_params": experience.policy_info["dist_params"],
                    "value_prediction": value_preds,
                    "return": torch.cat([returns, pad], dim=1),
                    "advantage": torch.cat([advantages, pad], dim=1)}
        out = experience.replace(policy_info=info)
        if unbatch:
            out = nest_utils.map_structure(lambda t: t.squeeze(0), out)
        return out

    def _preprocess_sequence(self, experience):
        if self._compute_value_and_advantage_in_train:
            return experience
        return self._preprocess(experience)

    # ---- loss ------------------------------------------------------------------------------------
    def _loss_denominator(self, N):
        """What the per-sample loss terms are divided by: N x replicas
        (tf.nn.compute_average_loss inside common.aggregate_losses, utils/common.py:1462-1467) or,
        with aggregate_losses_across_replicas=False, N (tf.reduce_mean on the replica)."""
        return float(N * (self.num_replicas if self._aggregate_losses_across_replicas else 1))

    def l2_regularization_loss(self, debug_summaries=False):
        """policy_l2_reg * sum(actor kernels^2) + value_function_l2_reg * sum(value kernels^2),
        divided by the replica count (ppo_agent.py:1088-1157)."""
        lib = _lib.load()
        total = torch.zeros((), dtype=torch.float32, device=self._device)
        for net, coef in ((self._actor_net, self._policy_l2_reg),
                          (self._value_net, self._value_function_l2_reg)):
            if coef <= 0:
                continue
            for k in net.kernels:
                s = torch.empty((1,), dtype=torch.float32, device=self._device)
                _lib.check(lib.aa_sumsq_f32(k.data_ptr(), k.numel(), s.data_ptr(), self._st()),
                           "aa_sumsq_f32")
                total = total + s[0] * coef
        return total / self.num_replicas

    def _add_l2_grads(self):
        lib = _lib.load()
        for net, coef in ((self._actor_net, self._policy_l2_reg),
                          (self._value_net, self._value_function_l2_reg)):
            if coef <= 0:
                continue
            for k, g in zip(net.kernels, net.kernel_grads):
                _lib.check(lib.aa_add_l2_grad(g.data_ptr(), k.data_ptr(), k.numel(),
                                              2.0 * coef / self.num_replicas, self._st()),
                           "aa_add_l2_grad")

    def _flat_obs(self, obs):
        return obs.reshape((-1,) + tuple(obs.shape[obs.dim() - self._obs_rank:]))

    def _normalized_obs(self, obs_flat):
        """What PPOPolicy._apply_actor_network / apply_value_network feed the networks
        (ppo_policy.py:231-241).  The statistics do not change during the update epochs (they are
        updated after them, ppo_agent.py:991-993), so `_train` normalises once."""
        if self._observation_normalizer is None:
            return obs_flat
        return self._observation_normalizer.normalize(obs_flat)

    def _global_batch(self, nest):
        """Data-parallel runs: the normalisers are replicated state (MirroredStrategy variables
        updated in cross-replica context), so every replica updates them with the batches of ALL
        replicas, gathered in rank order -- the statistics stay identical everywhere.  The hook is
        installed by train.Learner (strategy.all_gather_batch); one small all-gather per
        collected batch, off the minibatch loop."""
        hook = getattr(self, "batch_gather_hook", None)
        if hook is None or self.num_replicas <= 1:
            return nest
        return nest_utils.map_structure(hook, nest)

    def update_observation_normalizer(self, batched_observations):   # ppo_agent.py:1078-1082
        if self._observation_normalizer is not None:
            self._observation_normalizer.update(self._global_batch(batched_observations),
                                                outer_dims=[0, 1])

    def update_reward_normalizer(self, batched_rewards):             # ppo_agent.py:1084-1086
        if self._reward_normalizer is not None:
            self._reward_normalizer.update(self._global_batch(batched_rewards),
                                           outer_dims=[0, 1])

    def _loss_forward_backward(self, obs_flat, actions, old_logp, returns, adv, old_loc, old_scale,
                               weights, old_vpred, training, slot):
        """One evaluation of get_loss on N flattened samples; with training=True also fills
        flat_grads.  Returns the device stats vector."""
        lib = _lib.load()
        N = actions.shape[0]
        w = self._w(N)
        loc, scale = self._actor_net.forward(obs_flat, slot=slot, need_grad=training)
        vpred = self._value_net.forward(obs_flat, slot=slot, need_grad=training)
        use_kl = not (self._initial_adaptive_kl_beta == 0 and self._kl_cutoff_factor == 0)
        kl_cut_coef = self._kl_cutoff_coef if self._kl_cutoff_factor > 0 else 0.0
        _lib.check(lib.aa_ppo_loss_dist(
            loc.data_ptr(), scale.data_ptr(),
            old_loc.data_ptr() if use_kl else None, old_scale.data_ptr() if use_kl else None,
            actions.data_ptr(), old_logp.data_ptr(), adv.data_ptr(), returns.data_ptr(),
            vpred.data_ptr(), _lib.ptr(old_vpred) if self._value_clipping > 0 else None,
            weights.data_ptr(), N, self._D, self._importance_ratio_clipping,
            self._value_clipping, self._value_pred_loss_coef, self._entropy_regularization,
            self._loss_denominator(N), self._log_prob_clipping,
            _lib.ptr(self._adaptive_kl_beta) if use_kl else None, kl_cut_coef,
            self._kl_cutoff_factor * self._adaptive_kl_target,
            w["dloc"].data_ptr() if training else None,
            w["dscale"].data_ptr() if training else None,
            w["dv"].data_ptr() if training else None, w["stats"].data_ptr(), self._st()),
            "aa_ppo_loss_dist")
        if training:
            self._actor_net.backward(w["dloc"], w["dscale"], slot=slot)
            self._value_net.backward(w["dv"], slot=slot)
            self._add_l2_grads()
        return w["stats"]

    def _loss_info_from_stats(self, stats, l2):
        """LossInfo whose fields are views of ONE private copy of the stats vector (the work
        buffer is overwritten by the next evaluation); the copy and `total = stats[6] + l2` are one
        launch (aa_pack_small_f32)."""
        s = torch.empty((9,), dtype=torch.float32, device=stats.device)
        _lib.check(_lib.load().aa_pack_small_f32(
            stats.data_ptr(), 8, l2.data_ptr() if isinstance(l2, torch.Tensor) else None, 6,
            s.data_ptr(), self._st()), "aa_pack_small_f32")
        return tf_agent.LossInfo(s[6], PPOLossInfo(
            policy_gradient_loss=s[0], value_estimation_loss=s[1], l2_regularization_loss=s[8],
            entropy_regularization_loss=s[2], kl_penalty_loss=s[5], clip_fraction=s[3]))

    def get_loss(self, time_steps, actions, act_log_probs, returns, normalized_advantages,
                 action_distribution_parameters, weights, train_step=None, debug_summaries=False,
                 old_value_predictions=None, training=False):
        """LossInfo for a batch with ONE or TWO outer dims (ppo_agent.py:481-615)."""
        dev = time_steps.discount.device
        with torch.cuda.device(dev):
            obs = self._normalized_obs(self._flat_obs(time_steps.observation))
            N = obs.shape[0]
            f = lambda t: t.reshape(N, -1).to(torch.float32).contiguous()
            v = lambda t: t.reshape(N).to(torch.float32).contiguous()
            stats = self._loss_forward_backward(
                obs, f(actions), v(act_log_probs), v(returns), v(normalized_advantages),
                f(action_distribution_parameters["loc"]),
                f(action_distribution_parameters["scale"]), v(weights),
                None if old_value_predictions is None else v(old_value_predictions), training,
                slot=("loss", N))
            l2 = self.l2_regularization_loss() if (
                self._policy_l2_reg > 0 or self._value_function_l2_reg > 0) else \
                torch.zeros((), dtype=torch.float32, device=dev)
            self._clip_fraction = stats[3]
            return self._loss_info_from_stats(stats, l2)

    # ---- train -----------------------------------------------------------------------------------
    def _global_norm_clip(self):
        lib = _lib.load()
        if self._norm_seg is None:
            self._norm_seg = torch.tensor([0, self.flat_grads.numel()], dtype=torch.int64,
                                          device=self._device)
            self._norm_sumsq = torch.zeros((1,), dtype=torch.float32, device=self._device)
        _lib.check(lib.aa_segment_sumsq(self.flat_grads.data_ptr(), self._norm_seg.data_ptr(), 1,
                                        self._norm_sumsq.data_ptr(), self._st()),
                   "aa_segment_sumsq")
        if self._gradient_clipping > 0:
            _lib.check(lib.aa_clip_by_norm(self.flat_grads.data_ptr(), self._norm_seg.data_ptr(),
                                           1, self._norm_sumsq.data_ptr(),
                                           self._gradient_clipping, 0, self._st()),
                       "aa_clip_by_norm")
        self._grad_norm = self._norm_sumsq

    def _bump_train_step(self):
        self._train_step_counter.assign_add(1)

    @property
    def graph_train_whole_ok(self):
        """`common.function(agent.train)` replays the train step as ONE HIP graph only for the
        static configuration (minibatches of a fixed shape prepared by PPOLearner: no in-train
        preprocessing, no in-train normaliser update, one epoch per call).  The reference
        script's `tf_agent.train(gather_all())` -- a different [B, T] every iteration, GAE and
        `num_epochs` epochs inside the call -- stays on eager launches."""
        return (not self._compute_value_and_advantage_in_train and
                not self.update_normalizers_in_train and self._num_epochs == 1)

    def _graph_train_whole(self, experience, weights):
        """The whole train step is device work plus host counters registered with
        graph.on_replay: capturable as one HIP graph (utils/graph.py: GraphedTrain)."""
        return self._train(experience, weights)

    def _train(self, experience, weights):
        if self._optimizer is None:
            raise ValueError("Optimizer is undefined.")
        lib = _lib.load()
        dev = experience.discount.device
        with torch.cuda.device(dev):
            if self._compute_value_and_advantage_in_train:
                processed = self._preprocess(experience)
            else:
                processed = experience
            B, T1 = processed.discount.shape
            N = B * T1
            w = self._w(N)
            info = processed.policy_info
            returns = info["return"].reshape(N).contiguous()
            advantages = info["advantage"].reshape(N).contiguous()
            step_type = processed.step_type.to(torch.int32).reshape(N).contiguous()
            wts = None
            if weights is not None:
                wts = torch.as_tensor(weights, dtype=torch.float32, device=dev)
                wts = wts.expand(B, T1).reshape(N).contiguous()
            if self._fused_step_ok() and processed.observation.dtype == torch.float32:
                # three launches per epoch instead of ~28 (csrc/ppo_fused.hip)
                return self._train_fused(processed, returns, advantages, step_type, wts, N)
            _lib.check(lib.aa_ppo_trajectory_mask(step_type.data_ptr(), returns.data_ptr(),
                                                  advantages.data_ptr(), _lib.ptr(wts), N,
                                                  w["mask"].data_ptr(), self._st()),
                       "aa_ppo_trajectory_mask")
            old_loc = info["dist_params"]["loc"].reshape(N, self._D).to(torch.float32).contiguous()
            old_scale = info["dist_params"]["scale"].reshape(N, self._D).to(
                torch.float32).contiguous()
            actions = processed.action.reshape(N, self._D).to(torch.float32).contiguous()
            _lib.check(lib.aa_normal_log_prob(old_loc.data_ptr(), old_scale.data_ptr(),
                                              actions.data_ptr(), N, self._D,
                                              w["old_logp"].data_ptr(), self._st()),
                       "aa_normal_log_prob")
            _lib.check(lib.aa_normalize_moments(advantages.data_ptr(), N, 1e-8,
                                                w["adv_norm"].data_ptr(),
                                                w["norm_stats"].data_ptr(), self._st()),
                       "aa_normalize_moments")
            old_vpred = info["value_prediction"].reshape(N).contiguous()
            obs = self._normalized_obs(self._flat_obs(processed.observation))
            if self._zero is None:
                self._zero = torch.zeros((), dtype=torch.float32, device=dev)
            l2 = self._zero
            stats = None
            for _ in range(self._num_epochs):
                stats = self._loss_forward_backward(
                    obs, actions, w["old_logp"], returns, w["adv_norm"], old_loc, old_scale,
                    w["mask"], old_vpred, True, slot=("train", N))
                if self._policy_l2_reg > 0 or self._value_function_l2_reg > 0:
                    l2 = self.l2_regularization_loss()
                self._global_norm_clip()
                if self.gradient_hook is not None:
                    self.gradient_hook(self.flat_grads)
                self._optimizer.apply_flat(self.flat_params, self.flat_grads)
                graph.on_replay(self._bump_train_step)
            loss_info = self._loss_info_from_stats(stats, l2)
            self._clip_fraction = loss_info.extra.clip_fraction
            if self._initial_adaptive_kl_beta > 0:
                # mean KL(old || current) * mask after the update epochs -> beta update
                s2 = self._loss_forward_backward(
                    obs, actions, w["old_logp"], returns, w["adv_norm"], old_loc, old_scale,
                    w["mask"], old_vpred, False, slot=("train", N))
                _lib.check(lib.aa_ppo_update_kl_beta(s2[7:].data_ptr(), self._adaptive_kl_target,
                                                     self._adaptive_kl_tolerance,
                                                     self._adaptive_kl_beta.data_ptr(),
                                                     self._st()), "aa_ppo_update_kl_beta")
            if self.update_normalizers_in_train:      # ppo_agent.py:991-993
                self.update_observation_normalizer(processed.observation)
                self.update_reward_normalizer(processed.reward)
        return loss_info

    # ---- the fused minibatch step (csrc/ppo_fused.hip) ---------------------------------------------
    def _fused_step_ok(self):
        """The train step can run as aa_ppo_fused_step: tanh-Normal actor + value MLP with every
        layer <= 64 wide and <= 3 layers, no KL / L2 terms, Adam, one replica."""
        ok = getattr(self, "_fused_ok", None)
        if ok is None:
            from agents_amd import optimizers
            from agents_amd.agents.ppo import ppo_actor_network as pan
            a, v = self._actor_net, self._value_net
            ok = bool(
                FUSED_STEP and isinstance(a, pan.TanhNormalActorNet) and isinstance(v, pan.ValueNet)
                and a.body._fused_small_ok() and v.body._fused_small_ok()
                and len(a.body._param_layers) <= 3 and len(v.body._param_layers) <= 3
                and all(ks[1] % 4 == 0 or ks[0] * ks[1] <= 1024
                        for ks, _ in a.body._shapes + v.body._shapes)
                and a.body._param_layers[-1].activation is None
                and v.body._param_layers[-1].activation is None
                and self._D <= 16 and self._obs_rank == 1
                and self._initial_adaptive_kl_beta == 0 and self._kl_cutoff_factor == 0
                and self._policy_l2_reg == 0 and self._value_function_l2_reg == 0
                and type(self._optimizer) in (optimizers.Adam, optimizers.AdamOptimizer))
            self._fused_ok = ok
        return ok and self.gradient_hook is None and self.num_replicas == 1

    @staticmethod
    def _mlp_layout(body, base):
        from agents_amd import ops
        lay = _lib.MlpLayout()
        n = len(body._param_layers)
        lay.n_layers = n
        lay.dims[0] = int(np.prod(body._input_tensor_spec.shape))
        for i, ((ks, _), (k_off, b_off), l) in enumerate(zip(body._shapes, body._offsets,
                                                             body._param_layers)):
            lay.dims[i + 1] = int(ks[1])
            lay.acts[i] = ops.ACT[l.activation]
            lay.k_off[i] = base + k_off
            lay.b_off[i] = base + b_off
        return lay

    def _fused_desc(self, obs, actions, old_loc, old_scale, returns, advantages, old_vpred,
                    step_type, wts, N):
        """aa_ppo_fused_desc over the given sample arrays (rows = the minibatch, or all frames when
        the minibatch is addressed through a row index) + the cached workspace of size N."""
        lib = _lib.load()
        dev = self._device
        D = self._D
        w = self._w(N)
        fw = w.get("fused")
        total = self.flat_params.numel()
        if fw is None:
            nbytes = int(lib.aa_ppo_fused_workspace_bytes(N, total))
            fw = {"ws": torch.zeros((nbytes,), dtype=torch.uint8, device=dev),   # zeroed ONCE
                  "stats": torch.zeros((9,), dtype=torch.float32, device=dev)}
            w["fused"] = fw
        if self._norm_seg is None:
            self._norm_seg = torch.tensor([0, total], dtype=torch.int64, device=dev)
            self._norm_sumsq = torch.zeros((1,), dtype=torch.float32, device=dev)
        a, v = self._actor_net, self._value_net
        d = _lib.PpoFusedDesc()
        d.obs, d.ld_obs, d.obs_dim, d.D = obs.data_ptr(), obs.stride(0), obs.shape[1], D
        d.actions, d.old_loc, d.old_scale = actions.data_ptr(), old_loc.data_ptr(), \
            old_scale.data_ptr()
        d.returns, d.adv, d.old_vpred = returns.data_ptr(), advantages.data_ptr(), \
            old_vpred.data_ptr()
        d.step_type, d.weights, d.N = step_type.data_ptr(), _lib.ptr(wts), N
        nrm = self._observation_normalizer
        if nrm is not None:
            st = nrm._state[0]
            d.nrm_count, d.nrm_avg, d.nrm_m2 = st[0].data_ptr(), st[1].data_ptr(), \
                st[2].data_ptr()
            d.nrm_eps, d.nrm_clip = 1e-3, 5.0        # TensorNormalizer.normalize defaults
        d.params, d.total, d.head_off = self.flat_params.data_ptr(), total, a.body.flat_size
        d.actor = self._mlp_layout(a.body, 0)
        d.value = self._mlp_layout(v.body, a.flat_size)
        d.act_mean, d.act_mag = _lib.ptr(a._mean), _lib.ptr(a._mag)
        d.clip_eps, d.value_clip = self._importance_ratio_clipping, self._value_clipping
        d.c_v, d.c_e = self._value_pred_loss_coef, self._entropy_regularization
        d.denom, d.logp_clip, d.adv_eps = self._loss_denominator(N), \
            self._log_prob_clipping, 1e-8
        return d, fw

    def _fused_loss_info(self, fw):
        """LossInfo over the stats vector of the last fused step: VIEWS of a work buffer that the
        next train step overwrites (consume or clone them before; Learner.run returns copies)."""
        s = fw["stats"]
        loss_info = tf_agent.LossInfo(s[6], PPOLossInfo(
            policy_gradient_loss=s[0], value_estimation_loss=s[1], l2_regularization_loss=s[8],
            entropy_regularization_loss=s[2], kl_penalty_loss=s[5], clip_fraction=s[3]))
        self._clip_fraction = loss_info.extra.clip_fraction
        self._grad_norm = self._norm_sumsq
        return loss_info

    def _train_fused(self, processed, returns, advantages, step_type, wts, N):
        import ctypes
        lib = _lib.load()
        D = self._D
        info = processed.policy_info
        f2 = lambda t: t.reshape(N, D).to(torch.float32).contiguous()
        obs = self._flat_obs(processed.observation)
        if obs.stride(-1) != 1:
            obs = obs.contiguous()
        d, fw = self._fused_desc(
            obs, f2(processed.action), f2(info["dist_params"]["loc"]),
            f2(info["dist_params"]["scale"]), returns, advantages,
            info["value_prediction"].reshape(N).contiguous(), step_type, wts, N)
        opt = self._optimizer
        slot = opt._slot(self.flat_params, ("m", "v"))
        for _ in range(self._num_epochs):
            _lib.check(lib.aa_ppo_fused_step(
                ctypes.byref(d), self.flat_grads.data_ptr(), slot["m"].data_ptr(),
                slot["v"].data_ptr(), slot["step"].data_ptr(), opt.learning_rate, opt.beta_1,
                opt.beta_2, opt.epsilon, self._gradient_clipping, fw["stats"].data_ptr(),
                self._norm_sumsq.data_ptr(), fw["ws"].data_ptr(), fw["ws"].numel(), self._st()),
                "aa_ppo_fused_step")
            graph.on_replay(opt._bump_iterations)
            graph.on_replay(self._bump_train_step)
        loss_info = self._fused_loss_info(fw)
        if self.update_normalizers_in_train:      # ppo_agent.py:991-993
            self.update_observation_normalizer(processed.observation)
            self.update_reward_normalizer(processed.reward)
        return loss_info

    # ---- a whole epoch's minibatch steps from one host call ------------------------------------------
    def fused_minibatches_ok(self, frames):
        """PPOLearner may hand this agent (frames, permutation) instead of one gathered minibatch
        per call: the fused step reads its rows through the permutation, so an epoch is ONE host
        call issuing three launches per minibatch (no gather launch, no per-step Python)."""
        return (self._fused_step_ok() and self._num_epochs == 1 and
                not self._compute_value_and_advantage_in_train and
                not self.update_normalizers_in_train and self._initialized and
                frames.observation.dtype == torch.float32 and frames.observation.dim() == 2 and
                frames.step_type.dtype == torch.int32 and self._fused_leaves_are_f32(frames))

    @staticmethod
    def _fused_leaves_are_f32(frames):
        """aa_ppo_fused_epoch reads these leaves as raw float32 pointers (no cast on the way, unlike
        `_train` / `_train_fused`): anything else -- a float64 action spec, a float64 policy_info
        leaf -- must take the per-minibatch path, which casts."""
        info = frames.policy_info
        try:
            leaves = (frames.action, info["dist_params"]["loc"], info["dist_params"]["scale"],
                      info["return"], info["advantage"], info["value_prediction"])
        except (KeyError, TypeError, IndexError):
            return False
        return all(torch.is_tensor(t) and t.dtype == torch.float32 for t in leaves)

    def train_minibatches(self, frames, perm, minibatch_size, n_steps):
        """`n_steps` train steps, step s on rows perm[s * mb : (s + 1) * mb] of the flattened
        frames ([F, ...] leaves of a preprocessed trajectory) -- exactly what `train` does when it
        is called once per gathered minibatch [mb, 1, ...] (tf_agents/train/ppo_learner.py:220-248
        + ppo_agent.py:834-1076), bit for bit.  Returns the LossInfo of the last step."""
        import ctypes
        lib = _lib.load()
        N, D = int(minibatch_size), self._D
        if perm.dtype != torch.int64 or perm.numel() < N * n_steps or not perm.is_contiguous():
            raise ValueError("perm must be a contiguous int64 vector of n_steps * minibatch rows")
        info = frames.policy_info
        F = frames.discount.shape[0]
        c = lambda t, *shape: t.reshape((F,) + shape).contiguous()
        with torch.cuda.device(self._device):
            d, fw = self._fused_desc(
                c(frames.observation, frames.observation.shape[1]), c(frames.action, D),
                c(info["dist_params"]["loc"], D), c(info["dist_params"]["scale"], D),
                c(info["return"]), c(info["advantage"]), c(info["value_prediction"]),
                c(frames.step_type), None, N)
            opt = self._optimizer
            slot = opt._slot(self.flat_params, ("m", "v"))
            _lib.check(lib.aa_ppo_fused_epoch(
                ctypes.byref(d), perm.data_ptr(), int(n_steps), self.flat_grads.data_ptr(),
                slot["m"].data_ptr(), slot["v"].data_ptr(), slot["step"].data_ptr(),
                opt.learning_rate, opt.beta_1, opt.beta_2, opt.epsilon, self._gradient_clipping,
                fw["stats"].data_ptr(), self._norm_sumsq.data_ptr(), fw["ws"].data_ptr(),
                fw["ws"].numel(), self._st()), "aa_ppo_fused_epoch")
        opt.iterations += int(n_steps)
        self._train_step_counter.assign_add(int(n_steps))
        return self._fused_loss_info(fw)

    def kl_cutoff_loss(self, kl_divergence, debug_summaries=False):
        """Host-side helper on an explicit KL tensor (API parity, ppo_agent.py:1514-1560)."""
        if self._kl_cutoff_factor <= 0.0:
            return torch.zeros((), dtype=torch.float32)
        kl = torch.as_tensor(kl_divergence, dtype=torch.float32)
        over = torch.clamp(kl.mean() - self._kl_cutoff_factor * self._adaptive_kl_target, min=0.0)
        return self._kl_cutoff_coef * over * over

    def update_adaptive_kl_beta(self, kl_divergence):
        """ppo_agent.py:1642-1690 on an explicit KL tensor; returns the beta tensor."""
        if self._adaptive_kl_beta is None:
            return None
        lib = _lib.load()
        kl = torch.as_tensor(kl_divergence, dtype=torch.float32, device=self._device)
        m = kl.reshape(-1).mean().reshape(1).contiguous()
        with torch.cuda.device(self._device):
            _lib.check(lib.aa_ppo_update_kl_beta(m.data_ptr(), self._adaptive_kl_target,
                                                 self._adaptive_kl_tolerance,
                                                 self._adaptive_kl_beta.data_ptr(), self._st()),
                       "aa_ppo_update_kl_beta")
        return self._adaptive_kl_beta
