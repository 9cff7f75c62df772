"""SacAgent on MI355X: twin-Q critics, tanh-squashed Normal actor, learned entropy temperature.

Drop-in for tf_agents/agents/sac/sac_agent.py:61-827 (constructor arguments, `train` / `loss` /
`critic_loss` / `actor_loss` / `alpha_loss`, `SacLossInfo`, `std_clip_transform`), with every
data-path step a HIP kernel:
  _actions_and_log_probs (:533-558)  actor MLP (fp32 MFMA GEMMs) + aa_sac_sample
  critic_loss (:559-644)             target/online critic MLPs + aa_sac_critic_loss (+ backward)
  actor_loss  (:646-694)             aa_sac_actor_loss -> critic input gradients (no critic weight
                                     gradients) -> aa_sac_head_backward -> actor backward
  alpha_loss  (:696-740)             aa_sac_alpha_loss
  _apply_gradients (:463-484)        aa_clip_by_norm (per variable, optional) + aa_adam_step x3
  _get_target_updater (:486-531)     aa_soft_update over the flat critic buffer
_train (:314-410) keeps the reference's order: critic update, then actor update through the
UPDATED critics, then alpha with a fresh sample from the UPDATED actor, then the target update.
Both critics (and both targets) live in one flat parameter buffer, so the critic optimizer, the
clip and the soft update are single launches.

Networks: `networks.actor_distribution_network.ActorDistributionNetwork` (tanh-Normal projection)
and `networks.critic_network.CriticNetwork`.  The reference lets callers pass `actor_policy_ctor`;
here the policy is always `SacPolicy` (an ActorPolicy over the same kernels).
"""
import collections
import os

import numpy as np
import torch

from agents_amd import _lib, ops
from agents_amd.agents import tf_agent
from agents_amd.networks import actor_distribution_network as adn
from agents_amd.networks import critic_network
from agents_amd.policies import tf_policy
from agents_amd.trajectories import policy_step
from agents_amd.utils import common, graph, nest_utils

SacLossInfo = collections.namedtuple("SacLossInfo", ("critic_loss", "actor_loss", "alpha_loss"))

# (Tried in round 3: the critic phase's four independent forwards and the twin critics' backward on
# parallel graph branches -- 0.768 vs 0.556 ms per iteration: branch edges of a HIP graph cost more
# than the overlap of latency-bound kernels gains.  Removed; the twin critics now share launches.)

std_clip_transform = adn.std_clip_transform


# A/B knob: AA_SAC_QUAD_FORWARD=0 evaluates the target pair and the critic pair in two launches
_QUAD_FORWARD = True
# A/B knob: AA_SAC_FUSE_TARGET_UPDATE=0 keeps the soft target update a launch of its own
_FUSE_TARGET_UPDATE = True
# AA_SAC_FUSE_SAMPLE=0: the actor's tanh-normal sample stays a launch of its own behind the
# network's forward (A/B measurements; bit-identical either way)
_FUSE_SAMPLE = os.environ.get("AA_SAC_FUSE_SAMPLE", "1") != "0"
# AA_SAC_FUSE_LOSSES=0: critic loss, actor loss and the actor head's backward stay launches of their
# own in front of the gradient-chain launches that consume them (A/B; bit-identical either way)
_FUSE_LOSSES = os.environ.get("AA_SAC_FUSE_LOSSES", "1") != "0"
# Switches of the round-6 launch fusions (module variables, not environment knobs: each was
# A/B'd through one -- profiles/r06_zzzz_sac_part_split_ab.txt -- and the bit-identity tests flip
# them): False = the actor's forward + sample on the observations is a launch of its own instead
# of sharing the one on the next observations (aa_mlp_wide_forward_sample2) ...
_PAIR_SAMPLE = True
# ... / the critics' and the actor's Adam steps (+ soft target update) are launches of their own
# instead of the epilogue of their weight-gradient launches (aa_mlp_wide_backward_gen_adam)
_FUSE_DW_ADAM = True


def _spec_means_and_magnitudes(spec):
    """common.spec_means_and_magnitudes (utils/common.py:548-577)."""
    lo = np.broadcast_to(np.asarray(spec.minimum, np.float32), spec.shape or (1,)).reshape(-1)
    hi = np.broadcast_to(np.asarray(spec.maximum, np.float32), spec.shape or (1,)).reshape(-1)
    return ((hi + lo) / 2.0).astype(np.float32), ((hi - lo) / 2.0).astype(np.float32)


class SacPolicy(tf_policy.TFPolicy):
    """ActorPolicy (tf_agents/policies/actor_policy.py) for the tanh-Normal actor: `action` draws
    one reparameterised sample per call (Philox stream of its own seed / call counter)."""

    def __init__(self, time_step_spec, action_spec, actor_network, training=False, seed=0,
                 name=None):
        super().__init__(time_step_spec, action_spec, name=name)
        self._actor_network = actor_network
        self._training = training
        self._seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self._spec = nest_utils.flatten(action_spec)[0]
        self._A = int(np.prod(self._spec.shape)) or 1
        self._mean_h, self._mag_h = _spec_means_and_magnitudes(self._spec)
        self._dev_consts = None
        self._call_counter = None
        self._bufs = {}

    def _variables(self):
        return self._actor_network.variables

    def _consts(self, dev):
        if self._dev_consts is None:
            self._dev_consts = (torch.from_numpy(self._mean_h.copy()).to(dev),
                                torch.from_numpy(self._mag_h.copy()).to(dev))
            self._call_counter = torch.zeros((1,), dtype=torch.int64, device=dev)
            self._arrival = torch.zeros((16,), dtype=torch.int64, device=dev)
        return self._dev_consts

    def _tail(self, b, mean, mag, eps, save, out, net_index=0):
        t = _lib.SacSampleTail()
        t.net, t.A, t.std_kind = net_index, self._A, self._actor_network.projection.std_kind
        t.act_mean, t.act_mag = mean.data_ptr(), mag.data_ptr()
        t.eps_in = _lib.ptr(eps)
        t.seed = self._seed
        t.call_counter_dev = self._call_counter.data_ptr()
        t.arrival_dev = self._arrival.data_ptr()
        t.action = (b["action"] if out is None else out).data_ptr()
        t.logp = b["logp"].data_ptr()
        if save:
            t.save_tanh, t.save_sigma, t.save_eps = (_lib.ptr(save["tanh"]),
                                                     _lib.ptr(save["sigma"]),
                                                     _lib.ptr(save["eps"]))
        return t

    def _sample_bufs(self, slot, B, dev):
        key = (slot, B)
        b = self._bufs.get(key)
        if b is None:
            b = {"action": torch.empty((B, self._A), dtype=torch.float32, device=dev),
                 "logp": torch.empty((B,), dtype=torch.float32, device=dev)}
            self._bufs[key] = b
        return b

    @staticmethod
    def sample_pair(pol_a, obs_a, slot_a, eps_a, save_a, pol_b, obs_b, slot_b, eps_b, save_b):
        """`pol_a.sample(obs_a, slot_a)` and `pol_b.sample(obs_b, slot_b, need_grad=True)` of two
        policies over ONE actor network as one launch (two inputs, two slots, each policy's own
        Philox counter): what the two calls return, or None when the launch does not apply."""
        net = pol_a._actor_network
        if not _FUSE_SAMPLE or pol_b._actor_network is not net or pol_a is pol_b or \
                getattr(net, "forward_sample2", None) is None or \
                obs_a.dtype != torch.float32 or obs_b.dtype != torch.float32 or \
                obs_a.shape != obs_b.shape or not net.forward_sample_ok(obs_a) or \
                not net.forward_sample_ok(obs_b):
            return None
        dev = obs_a.device
        B = int(obs_a.shape[0])
        mean_a, mag_a = pol_a._consts(dev)
        mean_b, mag_b = pol_b._consts(dev)
        ba, bb = pol_a._sample_bufs(slot_a, B, dev), pol_b._sample_bufs(slot_b, B, dev)
        ta = pol_a._tail(ba, mean_a, mag_a, eps_a, save_a, None, net_index=0)
        tb = pol_b._tail(bb, mean_b, mag_b, eps_b, save_b, None, net_index=1)
        za, zb = net.forward_sample2(obs_a, ta, slot_a, False, obs_b, tb, slot_b, True)
        return (ba["action"], ba["logp"], za), (bb["action"], bb["logp"], zb)

    def sample(self, observation, slot, need_grad=False, eps=None, save=None, out=None):
        """(action [B,A], log_pi [B], z) for a batch of observations; `save` = dict of [B,A]
        buffers (tanh, sigma, eps) kept for the backward pass; `out` = a contiguous float32 [B,A]
        tensor of the caller's that receives the action instead of the slot's own buffer."""
        lib = _lib.load()
        dev = observation.device
        mean, mag = self._consts(dev)
        if observation.dtype != torch.float32:     # float64 / integer observation specs
            observation = observation.to(torch.float32)
        B = int(observation.shape[0])
        key = (slot, B)
        b = self._bufs.get(key)
        if b is None:
            b = {"action": torch.empty((B, self._A), dtype=torch.float32, device=dev),
                 "logp": torch.empty((B,), dtype=torch.float32, device=dev)}
            self._bufs[key] = b
        net = self._actor_network
        if _FUSE_SAMPLE and getattr(net, "forward_sample_ok", None) is not None and \
                net.forward_sample_ok(observation):
            # the actor's forward launch also draws the sample of its own head output (the
            # workgroup that produced a row's [mean | raw_std] has it in LDS): aa_sac_sample's
            # arithmetic and Philox counters, one launch less on the train step's chain
            t = self._tail(b, mean, mag, eps, save, out)
            z = net.forward_sample(observation, t, slot=slot, need_grad=need_grad)
            return (b["action"] if out is None else out), b["logp"], z
        z = net.forward(observation, slot=slot, need_grad=need_grad)
        st = _lib.stream_ptr()
        _lib.check(lib.aa_sac_sample(
            z.data_ptr(), B, self._A, mean.data_ptr(), mag.data_ptr(),
            self._actor_network.projection.std_kind, _lib.ptr(eps), self._seed,
            self._call_counter.data_ptr(), self._arrival.data_ptr(),
            (b["action"] if out is None else out).data_ptr(), b["logp"].data_ptr(),
            _lib.ptr(save["tanh"]) if save else None, _lib.ptr(save["sigma"]) if save else None,
            _lib.ptr(save["eps"]) if save else None, st), "aa_sac_sample")
        # (the launch itself advances the call counter when it drew the noise)
        return (b["action"] if out is None else out), b["logp"], z

    def state_dict(self):
        return {"call_counter": None if self._call_counter is None
                else int(self._call_counter.item())}

    def load_state_dict(self, sd):
        if sd.get("call_counter") is not None:
            self._consts(self._actor_network.flat_params.device)
            self._call_counter.fill_(int(sd["call_counter"]))

    def _action(self, time_step, policy_state, seed):
        obs = time_step.observation
        batched = time_step.step_type.dim() > 0
        if not batched:
            obs = obs.unsqueeze(0)
        graph.join_lanes(obs.device)
        with torch.cuda.device(obs.device):
            # the sample kernel writes the caller's tensor (a fresh one per call; inside a captured
            # driver body a static one of the graph's pool): no copy of the slot's buffer
            action = torch.empty((obs.shape[0], self._A), dtype=torch.float32, device=obs.device)
            self.sample(obs, slot="policy", out=action)
            action = action.reshape((obs.shape[0],) + tuple(self._spec.shape))
        if not batched:
            action = action.squeeze(0)
        return policy_step.PolicyStep(action, policy_state, ())


class SacAgent(tf_agent.TFAgent):
    def __init__(self, time_step_spec, action_spec, critic_network, actor_network,
                 actor_optimizer, critic_optimizer, alpha_optimizer, actor_loss_weight=1.0,
                 critic_loss_weight=0.5, alpha_loss_weight=1.0, actor_policy_ctor=None,
                 critic_network_2=None, target_critic_network=None, target_critic_network_2=None,
                 target_update_tau=1.0, target_update_period=1,
                 td_errors_loss_fn=common.element_wise_squared_loss, gamma=1.0,
                 reward_scale_factor=1.0, initial_log_alpha=0.0, use_log_alpha_in_alpha_loss=True,
                 target_entropy=None, gradient_clipping=None, debug_summaries=False,
                 summarize_grads_and_vars=False, train_step_counter=None, name=None, seed=0):
        flat_spec = nest_utils.flatten(action_spec)
        for spec in flat_spec:
            if spec.dtype in (torch.int32, torch.int64):
                raise NotImplementedError(
                    "SacAgent does not currently support discrete actions. "
                    "Action spec: {}".format(action_spec))
        if actor_policy_ctor is not None:
            raise NotImplementedError("actor_policy_ctor: the policy is SacPolicy")
        obs_spec = time_step_spec.observation
        critic_network.create_variables((obs_spec, action_spec))
        critic_network_2 = critic_network_2 or critic_network.copy(name="CriticNetwork2")
        critic_network_2.create_variables((obs_spec, action_spec))
        target_critic_network = target_critic_network or critic_network.copy(
            name="TargetCriticNetwork1")
        target_critic_network.create_variables((obs_spec, action_spec))
        target_critic_network_2 = target_critic_network_2 or critic_network.copy(
            name="TargetCriticNetwork2")
        target_critic_network_2.create_variables((obs_spec, action_spec))
        actor_network.create_variables(obs_spec)
        self._critic_network_1, self._critic_network_2 = critic_network, critic_network_2
        self._target_critic_network_1 = target_critic_network
        self._target_critic_network_2 = target_critic_network_2
        self._actor_network = actor_network
        dev = critic_network.flat_params.device
        # both critics in one flat buffer (critic optimizer / clip / soft update = one launch each)
        n1, n2 = critic_network.flat_size, critic_network_2.flat_size
        self._critic_params = torch.empty((n1 + n2,), dtype=torch.float32, device=dev)
        self._critic_grads = torch.zeros_like(self._critic_params)
        critic_network.rebind(self._critic_params[:n1], self._critic_grads[:n1])
        critic_network_2.rebind(self._critic_params[n1:], self._critic_grads[n1:])
        self._target_params = torch.empty_like(self._critic_params)
        tg = torch.zeros_like(self._critic_params)
        target_critic_network.rebind(self._target_params[:n1], tg[:n1])
        target_critic_network_2.rebind(self._target_params[n1:], tg[n1:])
        # log_alpha: one trainable scalar, padded to a 16-byte vector for the fused optimizer
        self._log_alpha_buf = torch.zeros((4,), dtype=torch.float32, device=dev)
        self._log_alpha_buf[0] = float(initial_log_alpha)
        self._log_alpha_grad = torch.zeros((4,), dtype=torch.float32, device=dev)
        self._spec = flat_spec[0]
        self._A = int(np.prod(self._spec.shape)) or 1
        if target_entropy is None:
            target_entropy = -float(sum(int(np.prod(s.shape)) or 1 for s in flat_spec)) / 2.0
        self._target_entropy = float(target_entropy)
        self._use_log_alpha_in_alpha_loss = bool(use_log_alpha_in_alpha_loss)
        self._target_update_tau = float(target_update_tau)
        self._target_update_period = int(target_update_period)
        self._actor_optimizer = actor_optimizer
        self._critic_optimizer = critic_optimizer
        self._alpha_optimizer = alpha_optimizer
        self._actor_loss_weight = float(actor_loss_weight)
        self._critic_loss_weight = float(critic_loss_weight)
        self._alpha_loss_weight = float(alpha_loss_weight)
        self._td_errors_loss_fn = td_errors_loss_fn
        self._gamma = float(gamma)
        self._reward_scale_factor = float(reward_scale_factor)
        self._gradient_clipping = gradient_clipping
        policy = SacPolicy(time_step_spec, action_spec, actor_network, training=False, seed=seed)
        self._train_policy = SacPolicy(time_step_spec, action_spec, actor_network, training=True,
                                       seed=seed + 1)
        # The losses sample the actor on noise streams of their OWN (next-state actions of the
        # critic loss, the alpha loss): with the collect policy's stream they would interleave with
        # the collect step's draws in launch order -- which a graphed loop changes when it runs the
        # critic update beside the collect step -- and two concurrent launches would share one
        # device-resident call counter.
        self._loss_policy = SacPolicy(time_step_spec, action_spec, actor_network, training=False,
                                      seed=seed + 2)
        self._update_target = common.Periodically(common.weak_method(self._soft_update_targets),
                                                  self._target_update_period, "update_targets")
        super().__init__(time_step_spec, action_spec, policy=policy, collect_policy=policy,
                         train_sequence_length=2, debug_summaries=debug_summaries,
                         summarize_grads_and_vars=summarize_grads_and_vars,
                         train_step_counter=train_step_counter)
        self.num_replicas = 1
        self.gradient_hook = None
        self._work = {}
        self._trans = {}      # B -> buffers of the unpacked transition + the critics' inputs
        self._xcat = None     # those of the batch being trained on (None: generic path)
        obs_spec = nest_utils.flatten(time_step_spec.observation)
        self._O = int(np.prod(obs_spec[0].shape)) if len(obs_spec) == 1 else -1
        self._clip_state = {}
        # Test hook: keep the N(0,1) draws of the "next action" and "alpha" samples as well (the
        # actor phase always keeps its own for the backward pass), so that a CPU oracle can be fed
        # the very noise a (graphed) train step used.  Set before the first train call.
        self.record_noise = False

    # ---- accessors ------------------------------------------------------------------------------
    @property
    def log_alpha(self):
        return self._log_alpha_buf[0]

    @property
    def actor_network(self):
        return self._actor_network

    @property
    def critic_networks(self):
        return self._critic_network_1, self._critic_network_2

    @property
    def target_critic_networks(self):
        return self._target_critic_network_1, self._target_critic_network_2

    def _initialize(self):
        common.soft_variables_update(self._critic_params, self._target_params, tau=1.0)

    def _soft_update_targets(self):
        common.soft_variables_update(self._critic_params, self._target_params,
                                     tau=self._target_update_tau)

    def _loss_kind(self):
        kind = getattr(self._td_errors_loss_fn, "aa_loss_kind", None)
        if kind is None:
            raise NotImplementedError(
                "td_errors_loss_fn must be common.element_wise_squared_loss "
                "(tf.math.squared_difference) or common.element_wise_huber_loss")
        return kind

    def _w(self, B, dev):
        w = self._work.get(B)
        if w is None:
            f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
            w = {"closs": f(1), "aloss": f(1), "lloss": f(1), "td": f(B), "dq1": f(B), "dq2": f(B),
                 "dlogp": f(B), "dz": f(B, 2 * self._A), "da": f(B, self._A),
                 "save": {"tanh": f(B, self._A), "sigma": f(B, self._A), "eps": f(B, self._A)}}
            if self.record_noise:
                for k in ("save_next", "save_alpha"):
                    w[k] = {"tanh": f(B, self._A), "sigma": f(B, self._A), "eps": f(B, self._A)}
            self._work[B] = w
        return w

    def _weights(self, weights, B, dev):
        if weights is None:
            return None
        if not isinstance(weights, torch.Tensor):
            return torch.full((B,), float(weights), dtype=torch.float32, device=dev)
        if weights.dim() == 0:
            return weights.to(torch.float32).expand(B).contiguous()
        return weights.to(torch.float32).reshape(B).contiguous()

    def _as_transition(self, experience):
        """AsTransition(squeeze_time_dim=True) on a [B, 2] trajectory (data_converter.py:300-380):
        time_steps = frame 0; next_time_steps = frame 1 with reward/discount of frame 0.
        ONE launch (ops.copy_segments) writes the five slices into buffers of their own and, at
        the same time, the observation halves of the three [observation | action] critic inputs
        of the train step (self._xcat)."""
        obs = experience.observation
        B = experience.discount.shape[0]
        dev = obs.device
        act_src = experience.action[:, 0].reshape(B, -1)
        O = int(np.prod(obs.shape[2:]))
        ok = (obs.dtype == torch.float32 and act_src.dtype == torch.float32 and
              experience.reward.dtype == torch.float32 and
              experience.discount.dtype == torch.float32 and obs.is_cuda and obs[:, 0].reshape(B, -1).stride(-1) == 1
              and O == self._O and act_src.shape[1] == self._A)
        if not ok:
            self._xcat = None
            # generic path: the networks compute in float32 (the reference casts observations and
            # actions, critic_network.py:150-170): float64 / integer specs are converted here
            f32 = lambda t: t.to(torch.float32).contiguous()
            return (f32(obs[:, 0]), f32(act_src), f32(obs[:, 1]),
                    f32(experience.reward[:, 0]), f32(experience.discount[:, 0]))
        t = self._trans.get(B)
        if t is None:
            f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
            t = {"obs0": f(B, O), "obs1": f(B, O), "act": f(B, self._A), "rew": f(B), "disc": f(B),
                 "x_sa": f(B, O + self._A), "x_next": f(B, O + self._A), "x_pi": f(B, O + self._A)}
            self._trans[B] = t
        o0, o1 = obs[:, 0].reshape(B, -1), obs[:, 1].reshape(B, -1)
        with torch.cuda.device(dev):
            ops.copy_segments([
                (o0, t["obs0"]), (o0, t["x_sa"][:, :O]), (o0, t["x_pi"][:, :O]),
                (act_src, t["x_sa"][:, O:]), (o1, t["obs1"]), (o1, t["x_next"][:, :O]),
                (experience.reward[:, :1], t["rew"].view(B, 1)),
                (experience.discount[:, :1], t["disc"].view(B, 1))])
        self._xcat = t
        return (t["obs0"].view((B,) + tuple(obs.shape[2:])), t["x_sa"][:, O:],
                t["obs1"].view((B,) + tuple(obs.shape[2:])), t["rew"], t["disc"])

    # ---- the three losses (forward + gradients) --------------------------------------------------
    def _critic_phase(self, obs, actions, next_obs, reward, discount, weights, need_grad,
                      eps_next=None, actor_obs=None, eps_actor=None, fuse_apply=False):
        """`actor_obs` (train steps): the actor update that follows evaluates the SAME actor
        weights on these observations -- its forward + sample then shares the launch of this
        phase's forward + sample on the next observations (`SacPolicy.sample_pair`) and
        `_actor_phase` picks the result up."""
        lib = _lib.load()
        B = obs.shape[0]
        dev = obs.device
        w = self._w(B, dev)
        xc = self._xcat
        fast = xc is not None and xc["obs1"].data_ptr() == next_obs.data_ptr() and \
            xc["obs0"].data_ptr() == obs.data_ptr()
        # twin critics per launch, [observation | action] read in place (csrc/mlp_wide.hip)
        pair = critic_network.pair_ok(self._critic_network_1, self._critic_network_2, obs, actions) \
            and critic_network.pair_ok(self._target_critic_network_1,
                                       self._target_critic_network_2, next_obs, actions)
        self._pre_actor = None
        pre = None
        if actor_obs is not None and need_grad and _PAIR_SAMPLE:
            pre = SacPolicy.sample_pair(self._loss_policy, next_obs, "next", eps_next,
                                        w.get("save_next"), self._train_policy, actor_obs,
                                        "actor", eps_actor, w["save"])
        if pre is not None:
            (na, nlogp, _), self._pre_actor = pre
        else:
            na, nlogp, _ = self._loss_policy.sample(next_obs, slot="next", eps=eps_next,
                                                    save=w.get("save_next"))
        targets = (self._target_critic_network_1, self._target_critic_network_2)
        critics = (self._critic_network_1, self._critic_network_2)
        if pair and _QUAD_FORWARD and critic_network.two_pairs_ok(targets, critics):
            # target pair + critic pair: one launch of four networks (256 workgroups)
            (tq1, tq2), (q1, q2) = critic_network.forward_two_pairs(
                targets, next_obs, na, "target", critics, obs, actions, "critic",
                need_grad_b=need_grad)
        elif pair:
            tq1, tq2 = critic_network.forward_pair(
                self._target_critic_network_1, self._target_critic_network_2, next_obs, na,
                slot="target")
            q1, q2 = critic_network.forward_pair(
                self._critic_network_1, self._critic_network_2, obs, actions, slot="critic",
                need_grad=need_grad)
        else:
            x_next = x_sa = None
            if fast:
                # [observation | action] once for both twin critics; the observation halves are
                # there already (_as_transition)
                x_next, x_sa = xc["x_next"], xc["x_sa"]
                ops.copy_segments([(na.reshape(B, -1), x_next[:, self._O:])])
            tq1 = self._target_critic_network_1.forward(next_obs, na, slot="target", x_cat=x_next)
            tq2 = self._target_critic_network_2.forward(next_obs, na, slot="target", x_cat=x_next)
            q1 = self._critic_network_1.forward(obs, actions, slot="critic", need_grad=need_grad,
                                                x_cat=x_sa)
            q2 = self._critic_network_2.forward(obs, actions, slot="critic", need_grad=need_grad,
                                                x_cat=x_sa)
        if need_grad and pair and _FUSE_LOSSES:
            # the critics' gradient-chain launch computes the loss and d loss / d q itself
            # (csrc/mlp_wide.hip: aa_mlp_wide_backward_gen; aa_sac_critic_loss's arithmetic)
            gen = _lib.SacDoutGen()
            gen.kind = _lib.AA_SAC_GEN_CRITIC
            gen.q1, gen.q2, gen.tq1, gen.tq2 = (q1.data_ptr(), q2.data_ptr(), tq1.data_ptr(),
                                                tq2.data_ptr())
            gen.next_logp, gen.reward, gen.discount = (nlogp.data_ptr(), reward.data_ptr(),
                                                       discount.data_ptr())
            gen.weights = _lib.ptr(weights)
            gen.log_alpha = self._log_alpha_buf.data_ptr()
            gen.gamma, gen.reward_scale = self._gamma, self._reward_scale_factor
            gen.loss_kind, gen.loss_weight = self._loss_kind(), self._critic_loss_weight
            gen.global_batch = float(B * self.num_replicas)
            gen.loss_out, gen.td_target_out = w["closs"].data_ptr(), w["td"].data_ptr()
            adam = None
            if fuse_apply and self._critic_apply_fusable():
                # the critics' weight-gradient launch steps their optimizer (and the soft update
                # of the targets) itself: aa_mlp_wide_backward_gen_adam
                adam = self._critic_optimizer.fused_step_desc(
                    self._critic_params,
                    soft_target=(self._target_params, self._target_update_tau)
                    if self._fuse_target_update() else None)
                self._critic_applied = True
            critic_network.backward_pair(self._critic_network_1, self._critic_network_2, None,
                                         None, slot="critic", gen=gen, batch=B, adam=adam)
            return w["closs"]
        _lib.check(lib.aa_sac_critic_loss(
            q1.data_ptr(), q2.data_ptr(), tq1.data_ptr(), tq2.data_ptr(), nlogp.data_ptr(),
            reward.data_ptr(), discount.data_ptr(), _lib.ptr(weights),
            self._log_alpha_buf.data_ptr(), self._gamma, self._reward_scale_factor,
            self._loss_kind(), self._critic_loss_weight, B, float(B * self.num_replicas),
            w["closs"].data_ptr(), w["td"].data_ptr(),
            w["dq1"].data_ptr() if need_grad else None, w["dq2"].data_ptr() if need_grad else None,
            _lib.stream_ptr()), "aa_sac_critic_loss")
        if need_grad:
            if pair:
                critic_network.backward_pair(self._critic_network_1, self._critic_network_2,
                                             w["dq1"], w["dq2"], slot="critic")
            else:
                self._critic_network_1.backward(w["dq1"], slot="critic")
                self._critic_network_2.backward(w["dq2"], slot="critic")
        return w["closs"]

    def _actor_apply_fusable(self):
        from agents_amd import optimizers as _opt
        return _FUSE_DW_ADAM and self._gradient_clipping is None and \
            self.gradient_hook is None and \
            type(self._actor_optimizer) in (_opt.Adam, _opt.AdamOptimizer)

    def _actor_phase(self, obs, weights, need_grad, eps=None, defer_dw=False):
        lib = _lib.load()
        B = obs.shape[0]
        w = self._w(B, obs.device)
        pol = self._train_policy if need_grad else self._loss_policy
        pre, self._pre_actor = getattr(self, "_pre_actor", None), None
        if pre is not None and need_grad:
            a, logp, z = pre       # drawn by the critic phase's launch (same actor weights)
        else:
            a, logp, z = pol.sample(obs, slot="actor", need_grad=need_grad, eps=eps,
                                    save=w["save"] if need_grad else None)
        x_pi = None
        xc = self._xcat
        pair = critic_network.pair_ok(self._critic_network_1, self._critic_network_2, obs, a)
        if pair:
            q1, q2 = critic_network.forward_pair(self._critic_network_1, self._critic_network_2,
                                                 obs, a, slot="actor_q", need_grad=need_grad)
        else:
            if xc is not None and xc["obs0"].data_ptr() == obs.data_ptr():
                x_pi = xc["x_pi"]
                ops.copy_segments([(a.reshape(B, -1), x_pi[:, self._O:])])
            q1 = self._critic_network_1.forward(obs, a, slot="actor_q", need_grad=need_grad,
                                                x_cat=x_pi)
            q2 = self._critic_network_2.forward(obs, a, slot="actor_q", need_grad=need_grad,
                                                x_cat=x_pi)
        if need_grad and pair and _FUSE_LOSSES and self._actor_network.body.wide_ok(B) and \
                not self._actor_network.body._fused_small_ok():
            # actor loss inside the critics' chain launch (d loss / d q, d loss / d log pi, the
            # loss value), the head's backward inside the actor's (aa_mlp_wide_backward_gen)
            gen = _lib.SacDoutGen()
            gen.kind = _lib.AA_SAC_GEN_ACTOR
            gen.q1, gen.q2, gen.logp = q1.data_ptr(), q2.data_ptr(), logp.data_ptr()
            gen.weights = _lib.ptr(weights)
            gen.log_alpha = self._log_alpha_buf.data_ptr()
            gen.loss_weight = self._actor_loss_weight
            gen.global_batch = float(B * self.num_replicas)
            gen.loss_out, gen.dlogp_out = w["aloss"].data_ptr(), w["dlogp"].data_ptr()
            da1, da2 = critic_network.backward_pair(
                self._critic_network_1, self._critic_network_2, None, None, slot="actor_q",
                param_grads=False, want_action_grad=True, gen=gen, batch=B)
            mag = self._train_policy._consts(obs.device)[1]
            hg = _lib.SacDoutGen()
            hg.kind = _lib.AA_SAC_GEN_HEAD
            hg.z, hg.A = z.data_ptr(), self._A
            hg.std_kind = self._actor_network.projection.std_kind
            hg.act_mag = mag.data_ptr()
            hg.save_tanh, hg.save_sigma, hg.save_eps = (w["save"]["tanh"].data_ptr(),
                                                        w["save"]["sigma"].data_ptr(),
                                                        w["save"]["eps"].data_ptr())
            hg.daction, hg.ld_daction = da1.data_ptr(), da1.stride(0)
            hg.daction2, hg.ld_daction2 = da2.data_ptr(), da2.stride(0)
            hg.dlogp = w["dlogp"].data_ptr()
            from agents_amd.networks import sequential
            defer = defer_dw and self._actor_apply_fusable()
            # (defer: the weight gradients are computed by the launch that also applies them, in
            # part (b) behind the collect step -- `_train_part_b`)
            sequential.backward_wide([self._actor_network.body], None, slot="actor", gen=hg,
                                     batch=B, param_grads=not defer)
            self._actor_dw_pending = B if defer else None
            return w["aloss"]
        _lib.check(lib.aa_sac_actor_loss(
            q1.data_ptr(), q2.data_ptr(), logp.data_ptr(), _lib.ptr(weights),
            self._log_alpha_buf.data_ptr(), self._actor_loss_weight, B,
            float(B * self.num_replicas), w["aloss"].data_ptr(),
            w["dq1"].data_ptr() if need_grad else None, w["dq2"].data_ptr() if need_grad else None,
            w["dlogp"].data_ptr() if need_grad else None, _lib.stream_ptr()),
            "aa_sac_actor_loss")
        if need_grad:
            # d loss / d action through BOTH critics (their weights are not touched here)
            if pair:
                da1, da2 = critic_network.backward_pair(
                    self._critic_network_1, self._critic_network_2, w["dq1"], w["dq2"],
                    slot="actor_q", param_grads=False, want_action_grad=True)
            else:
                da1 = self._critic_network_1.backward(w["dq1"], slot="actor_q",
                                                      param_grads=False, want_action_grad=True)
                da2 = self._critic_network_2.backward(w["dq2"], slot="actor_q",
                                                      param_grads=False, want_action_grad=True)
            # d loss / d action = da1 + da2, both column slices of the critics' input-gradient
            # buffers: added inside the head's backward launch
            mag = self._train_policy._consts(obs.device)[1]
            _lib.check(lib.aa_sac_head_backward(
                z.data_ptr(), B, self._A, mag.data_ptr(),
                self._actor_network.projection.std_kind, w["save"]["tanh"].data_ptr(),
                w["save"]["sigma"].data_ptr(), w["save"]["eps"].data_ptr(), da1.data_ptr(),
                da1.stride(0), da2.data_ptr(), da2.stride(0),
                w["dlogp"].data_ptr(), w["dz"].data_ptr(), _lib.stream_ptr()),
                "aa_sac_head_backward")
            self._actor_network.backward(w["dz"], slot="actor")
        return w["aloss"]

    def _alpha_phase(self, obs, weights, need_grad, eps=None, tail=None):
        """`tail` = (critic_loss, actor_loss, packed4): loss, the Adam step on log_alpha and the
        LossInfo pack run as ONE launch (aa_sac_alpha_step; bit-identical to the three)."""
        lib = _lib.load()
        B = obs.shape[0]
        w = self._w(B, obs.device)
        _, logp, _ = self._loss_policy.sample(obs, slot="alpha", eps=eps,
                                              save=w.get("save_alpha"))
        if tail is not None:
            opt = self._alpha_optimizer
            slot = opt._slot(self._log_alpha_buf, ("m", "v"))
            _lib.check(lib.aa_sac_alpha_step(
                logp.data_ptr(), _lib.ptr(weights), self._log_alpha_buf.data_ptr(),
                self._target_entropy, 1 if self._use_log_alpha_in_alpha_loss else 0,
                self._alpha_loss_weight, B, float(B * self.num_replicas), w["lloss"].data_ptr(),
                self._log_alpha_grad.data_ptr(), slot["m"].data_ptr(), slot["v"].data_ptr(),
                slot["step"].data_ptr(), opt.learning_rate, opt.beta_1, opt.beta_2, opt.epsilon,
                tail[0].data_ptr(), tail[1].data_ptr(), tail[2].data_ptr(), _lib.stream_ptr()),
                "aa_sac_alpha_step")
            graph.on_replay(opt._bump_iterations)
            return w["lloss"]
        _lib.check(lib.aa_sac_alpha_loss(
            logp.data_ptr(), _lib.ptr(weights), self._log_alpha_buf.data_ptr(),
            self._target_entropy, 1 if self._use_log_alpha_in_alpha_loss else 0,
            self._alpha_loss_weight, B, float(B * self.num_replicas), w["lloss"].data_ptr(),
            self._log_alpha_grad.data_ptr() if need_grad else None, _lib.stream_ptr()),
            "aa_sac_alpha_loss")
        return w["lloss"]

    # ---- public loss API (sac_agent.py:559-740; values only) ------------------------------------
    def critic_loss(self, time_steps, actions, next_time_steps, td_errors_loss_fn=None, gamma=None,
                    reward_scale_factor=None, weights=None, training=False):
        B = time_steps.observation.shape[0]
        dev = time_steps.observation.device
        saved = (self._td_errors_loss_fn, self._gamma, self._reward_scale_factor)
        if td_errors_loss_fn is not None:
            self._td_errors_loss_fn = td_errors_loss_fn
        if gamma is not None:
            self._gamma = float(gamma)
        if reward_scale_factor is not None:
            self._reward_scale_factor = float(reward_scale_factor)
        weight = self._critic_loss_weight
        self._critic_loss_weight = 1.0
        try:
            with torch.cuda.device(dev):
                out = self._critic_phase(
                    time_steps.observation, actions.reshape(B, -1).to(torch.float32).contiguous(),
                    next_time_steps.observation, next_time_steps.reward.contiguous(),
                    next_time_steps.discount.contiguous(), self._weights(weights, B, dev),
                    False).clone().reshape(())
        finally:
            self._td_errors_loss_fn, self._gamma, self._reward_scale_factor = saved
            self._critic_loss_weight = weight
        return out

    def actor_loss(self, time_steps, weights=None, training=True):
        B = time_steps.observation.shape[0]
        dev = time_steps.observation.device
        weight = self._actor_loss_weight
        self._actor_loss_weight = 1.0
        try:
            with torch.cuda.device(dev):
                return self._actor_phase(time_steps.observation, self._weights(weights, B, dev),
                                         False).clone().reshape(())
        finally:
            self._actor_loss_weight = weight

    def alpha_loss(self, time_steps, weights=None, training=False):
        B = time_steps.observation.shape[0]
        dev = time_steps.observation.device
        weight = self._alpha_loss_weight
        self._alpha_loss_weight = 1.0
        try:
            with torch.cuda.device(dev):
                return self._alpha_phase(time_steps.observation, self._weights(weights, B, dev),
                                         False).clone().reshape(())
        finally:
            self._alpha_loss_weight = weight

    # ---- train ----------------------------------------------------------------------------------
    def _critic_bodies(self):
        """The Sequentials behind `_critic_params`, in its order (per-variable clipping)."""
        out = []
        for c in (self._critic_network_1, self._critic_network_2):
            out += c.bodies if hasattr(c, "bodies") else [c.body]
        return out

    def _apply(self, optimizer, params, grads, net_for_clip=None, soft_target=None):
        if self._gradient_clipping is not None:
            self._clip(params, grads, net_for_clip)
        if self.gradient_hook is not None:
            self.gradient_hook(grads)
        if soft_target is not None:
            optimizer.apply_flat(params, grads, soft_target=soft_target)
        else:
            optimizer.apply_flat(params, grads)

    def _critic_apply_fusable(self):
        """Nothing stands between the critics' weight gradients and their Adam step (no clipping,
        no cross-replica reduction), and both critics' parameters are views of the one flat buffer
        the optimizer's slots mirror."""
        from agents_amd import optimizers as _opt
        if not _FUSE_DW_ADAM or self._gradient_clipping is not None or \
                self.gradient_hook is not None or \
                type(self._critic_optimizer) not in (_opt.Adam, _opt.AdamOptimizer):
            return False
        lo = self._critic_params.data_ptr()
        hi = lo + 4 * self._critic_params.numel()
        return all(lo <= b.flat_params.data_ptr() and
                   b.flat_params.data_ptr() + 4 * b.flat_params.numel() <= hi
                   for b in (self._critic_network_1.body, self._critic_network_2.body)) and \
            not self._critic_network_1.has_towers and not self._critic_network_2.has_towers

    def _fuse_target_update(self):
        """The soft update of the target critics rides in the critic optimizer's launch when it
        happens every step (sac_agent.py:385-410 with target_update_period == 1): the critics do not
        change between their Adam step and the end of the train step, so updating the targets right
        behind it is the same computation, one launch earlier and one launch less."""
        return (_FUSE_TARGET_UPDATE and self._target_update_period == 1 and
                getattr(self._critic_optimizer, "supports_soft_target", False))

    def _clip(self, params, grads, nets):
        """Per-variable tf.clip_by_norm (eager_utils.clip_gradient_norms, eager_utils.py:227-246)."""
        lib = _lib.load()
        key = grads.data_ptr()
        st = self._clip_state.get(key)
        if st is None:
            starts, base = [], 0
            for net in nets or []:
                starts += [base + s0 for s0, _ in net.segment_offsets()]
                base += net.flat_size
            if not nets:
                starts = [0]
                base = grads.numel()
            offs = torch.tensor(starts + [base], dtype=torch.int64, device=grads.device)
            st = (offs, torch.zeros((len(starts),), dtype=torch.float32, device=grads.device))
            self._clip_state[key] = st
        offs, sumsq = st
        s = _lib.stream_ptr()
        _lib.check(lib.aa_segment_sumsq(grads.data_ptr(), offs.data_ptr(), sumsq.numel(),
                                        sumsq.data_ptr(), s), "aa_segment_sumsq")
        _lib.check(lib.aa_clip_by_norm(grads.data_ptr(), offs.data_ptr(), sumsq.numel(),
                                       sumsq.data_ptr(), float(self._gradient_clipping), 1, s),
                   "aa_clip_by_norm")

    def _train(self, experience, weights, eps=None):
        """`eps` (tests): dict of externally supplied N(0,1) noise {"next", "actor", "alpha"}."""
        self._train_part_a(experience, weights, eps)
        return self._train_part_b()

    def _train_part_a(self, experience, weights, eps=None):
        """The critic update (sac_agent.py:286-330, first third) and the actor's loss + gradient
        (second third, without its optimizer step): reads the actor, writes the critics (their
        Adam step rides in their weight-gradient launch) and the actor's gradient CHAIN (its weight
        gradients are computed in part (b), by the launch that also applies them, unless clipping
        or a gradient hook stands in between) -- nothing the collect policy uses is modified, so a graphed loop
        runs it beside the collect step (utils/graph.py: GraphedTrain, whole mode in two parts).
        (Until round 6 the actor phase opened part (b): the GPU timeline of tools/bench_sac.py had
        part (b) start 16 us after part (a) had ended, waiting for the collect step to release
        weights that only the actor's optimizer launch -- 70 us further down -- overwrites.)"""
        eps = eps or {}
        obs, actions, next_obs, reward, discount = self._as_transition(experience)
        B = obs.shape[0]
        dev = obs.device
        graph.join_lanes(dev)
        with torch.cuda.device(dev):
            wts = self._weights(weights, B, dev)
            self._critic_applied = False
            closs = self._critic_phase(obs, actions, next_obs, reward, discount, wts, True,
                                       eps_next=eps.get("next"), actor_obs=obs,
                                       eps_actor=eps.get("actor"), fuse_apply=True)
            if not self._critic_applied:
                self._apply(self._critic_optimizer, self._critic_params, self._critic_grads,
                            self._critic_bodies(),
                            soft_target=(self._target_params, self._target_update_tau)
                            if self._fuse_target_update() else None)
            self._actor_dw_pending = None
            aloss = self._actor_phase(obs, wts, True, eps=eps.get("actor"), defer_dw=True)
        # (the deferred weight-gradient launch travels with the hand-over: an eagerly issued part
        # (b) behind a REPLAYED part (a) sees the state of that graph's capture)
        self._part_a = (obs, wts, closs, aloss, eps, self._actor_dw_pending)

    def _train_part_b(self):
        """The actor's optimizer step, the alpha update, LossInfo, counters and the soft target
        update."""
        obs, wts, closs, aloss, eps, pending = self._part_a
        dev = obs.device
        with torch.cuda.device(dev):
            if pending is not None:
                # the actor's weight gradients and its Adam step in one launch (the gradient
                # chain ran in part (a)): aa_mlp_wide_dw_adam
                from agents_amd.networks import sequential
                sequential.backward_wide(
                    [self._actor_network.body], None, slot="actor", batch=pending, dw_only=True,
                    adam=self._actor_optimizer.fused_step_desc(self._actor_network.flat_params))
            else:
                self._apply(self._actor_optimizer, self._actor_network.flat_params,
                            self._actor_network.flat_grads, [self._actor_network.body])
            # total + storage of its own for the three terms
            packed = torch.empty((4,), dtype=torch.float32, device=dev)
            from agents_amd import optimizers as _opt
            if type(self._alpha_optimizer) in (_opt.Adam, _opt.AdamOptimizer) and \
                    self._gradient_clipping is None and self.gradient_hook is None:
                # alpha loss + its Adam step + the pack: one launch
                self._alpha_phase(obs, wts, True, eps=eps.get("alpha"),
                                  tail=(closs, aloss, packed))
            else:
                lloss = self._alpha_phase(obs, wts, True, eps=eps.get("alpha"))
                self._apply(self._alpha_optimizer, self._log_alpha_buf, self._log_alpha_grad,
                            None)
                _lib.check(_lib.load().aa_pack_sum3_f32(closs.data_ptr(), aloss.data_ptr(),
                                                        lloss.data_ptr(), packed.data_ptr(),
                                                        _lib.stream_ptr()), "aa_pack_sum3_f32")
            info = tf_agent.LossInfo(packed[0], SacLossInfo(critic_loss=packed[1],
                                                            actor_loss=packed[2],
                                                            alpha_loss=packed[3]))
            # counter + (periodic) soft target update: device work of the update is enqueued here
            graph.on_replay(self._bump_counter)
            if not self._fuse_target_update():
                self._update_target()
        # issued directly (not recorded into a graph), `packed` is storage of this step's own
        self._own_info = None if graph.capturing() else info
        return info

    reduced_owns_storage = False

    def reduce_loss_info(self, loss_info):
        """Learner.run sums every LossInfo field over all axes and hands out storage of its own
        (train/learner.py:322-337).  The LossInfo of a train step whose last part was issued
        directly already is four scalars in a tensor allocated for that step: nothing to launch."""
        self.reduced_owns_storage = loss_info is getattr(self, "_own_info", None)
        return loss_info if self.reduced_owns_storage else None

    def _bump_counter(self):
        self._train_step_counter.assign_add(1)

    @property
    def graph_train_whole_ok(self):
        # the periodic target update is a host decision unless it happens every step
        return self._target_update_period == 1 and self._gradient_clipping is None or \
            (self._target_update_period == 1 and bool(self._clip_state))

    @property
    def graph_train_whole_b_eager_ok(self):
        # with the actor phase in part (a), part (b) is three or four launches and no host decision
        return True

    def _graph_train_whole(self, experience, weights):
        """The train step is device work plus host counters registered with graph.on_replay:
        one HIP graph per input signature (utils/graph.py: GraphedTrain, whole mode)."""
        return self._train(experience, weights)

    def _graph_train_whole_a(self, experience, weights):
        self._train_part_a(experience, weights)

    def _graph_train_whole_b(self):
        return self._train_part_b()

    # ---- checkpointing ---------------------------------------------------------------------------
    def replicated_state(self):
        """Tensors every data-parallel replica must hold identically (train.Learner broadcasts
        rank 0's at construction)."""
        out = [self._actor_network.flat_params, self._critic_params, self._target_params,
               self._log_alpha_buf]
        for o in (self._actor_optimizer, self._critic_optimizer, self._alpha_optimizer):
            if o is not None:
                out += o.variables()
        return out

    def state_dict(self):
        return {"actor": self._actor_network.flat_params.clone(),
                "critics": self._critic_params.clone(), "targets": self._target_params.clone(),
                "log_alpha": self._log_alpha_buf.clone(),
                "train_step": int(self._train_step_counter),
                "target_update_calls": self._update_target._counter,
                "optimizers": [o.state_dict() for o in (self._actor_optimizer,
                                                        self._critic_optimizer,
                                                        self._alpha_optimizer)],
                "policies": [self._policy.state_dict(), self._train_policy.state_dict(),
                             self._loss_policy.state_dict()]}

    def load_state_dict(self, sd):
        self._actor_network.flat_params.copy_(sd["actor"])
        self._critic_params.copy_(sd["critics"])
        self._target_params.copy_(sd["targets"])
        self._log_alpha_buf.copy_(sd["log_alpha"])
        self._train_step_counter.assign(sd["train_step"])
        self._update_target._counter = int(sd["target_update_calls"])
        for o, osd in zip((self._actor_optimizer, self._critic_optimizer, self._alpha_optimizer),
                          sd["optimizers"]):
            o.load_state_dict(osd)
        self._policy.load_state_dict(sd["policies"][0])
        self._train_policy.load_state_dict(sd["policies"][1])
        if len(sd["policies"]) > 2:
            self._loss_policy.load_state_dict(sd["policies"][2])
        self._initialized = True

    def _loss(self, experience, weights=None, training=False):
        obs, actions, next_obs, reward, discount = self._as_transition(experience)
        B = obs.shape[0]
        dev = obs.device
        with torch.cuda.device(dev):
            wts = self._weights(weights, B, dev)
            closs = self._critic_phase(obs, actions, next_obs, reward, discount, wts, False).clone()
            aloss = self._actor_phase(obs, wts, False).clone()
            lloss = self._alpha_phase(obs, wts, False).clone()
            total = (closs + aloss + lloss).reshape(())
        return tf_agent.LossInfo(total, SacLossInfo(closs.reshape(()), aloss.reshape(()),
                                                    lloss.reshape(())))
