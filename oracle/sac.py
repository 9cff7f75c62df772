"""torch-CPU fp32 restatement of the SAC losses and train step (TEST INFRASTRUCTURE: only tests/,
__graft_entry__.smoke() and bench baselines may import this; the product path never does).

Follows
  SacAgent._train                   tf_agents/agents/sac/sac_agent.py:314-410
  SacAgent._actions_and_log_probs   :533-558
  SacAgent.critic_loss              :559-644
  SacAgent.actor_loss               :646-694
  SacAgent.alpha_loss               :696-740
  SacAgent._get_target_updater      :486-531  (soft update of both target critics)
  TanhNormalProjectionNetwork.call  tf_agents/agents/sac/tanh_normal_projection_network.py:108-143
  std_clip_transform                tf_agents/agents/sac/sac_agent.py:48-57
  SquashToSpecNormal                tf_agents/distributions/utils.py:40-160
  common.aggregate_losses           tf_agents/utils/common.py:1400-1476
Pinned on the reference's own known answers (sac_agent_test.py:269-396: critic loss with the
DummyCriticNet / DummyActorPolicy mocks, the l2 regularisation case, actor loss 6.0, alpha loss -52)
in tests/test_oracle_sac.py.  PARITY UNPINNED (third-party arithmetic, no reference test fixes it):
TFP's MultivariateNormalDiag.log_prob / tanh bijector rounding, the Normal sampler's random stream,
Adam's update arithmetic (restated from the documented formulas, oracle/optim.py).
"""
import math

import torch

from oracle import nets, optim

HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


def std_transform(raw, kind):
    if kind == "clip_exp":
        raw = torch.clamp(raw, -20.0, 2.0)
    return torch.exp(raw)


def tanh_normal(z, eps, act_mean, act_mag, kind="exp"):
    """z [B,2A] = [mean | raw_std], eps [B,A] ~ N(0,1) -> (action [B,A], log_pi [B]).

    Reparameterised x = mean + sigma*eps; action = act_mean + act_mag*tanh(x); log_pi is the
    MultivariateNormalDiag log-density at x minus the log-det-Jacobian of Shift(Scale) o Tanh with
    the numerically stable tanh form 2(log 2 - x - softplus(-2x))."""
    A = z.shape[1] // 2
    mu, raw = z[:, :A], z[:, A:]
    sigma = std_transform(raw, kind)
    x = mu + sigma * eps
    action = act_mean + act_mag * torch.tanh(x)
    e = (x - mu) / sigma
    fldj = 2.0 * (math.log(2.0) - x - torch.nn.functional.softplus(-2.0 * x))
    lp = (-0.5 * e * e - torch.log(sigma) - HALF_LOG_2PI - torch.log(torch.abs(act_mag))
          - fldj).sum(-1)
    return action, lp


def aggregate(per_example, weights, global_batch, regularization=0.0):
    """common.aggregate_losses for a [B] loss: weighted sum / global batch + regularisation."""
    if weights is not None:
        w = torch.as_tensor(weights, dtype=torch.float32)
        per_example = torch.where(w == 0, torch.zeros_like(per_example), per_example * w)
    return per_example.sum() / global_batch + regularization


def squared_difference(td, q):
    return (td - q) ** 2


def huber(td, q):
    e = (q - td).abs()
    quad = torch.clamp(e, max=1.0)
    return 0.5 * quad * quad + (e - quad)


def critic_loss(q1_fn, q2_fn, tq1_fn, tq2_fn, next_actions_and_logp, log_alpha, obs, actions,
                next_obs, reward, discount, td_errors_loss_fn=squared_difference, gamma=1.0,
                reward_scale_factor=1.0, weights=None, regularization=0.0, num_replicas=1):
    next_actions, next_logp = next_actions_and_logp
    tq = torch.minimum(tq1_fn(next_obs, next_actions), tq2_fn(next_obs, next_actions)) \
        - torch.exp(log_alpha) * next_logp
    td = (reward_scale_factor * reward + gamma * discount * tq).detach()
    l = td_errors_loss_fn(td, q1_fn(obs, actions)) + td_errors_loss_fn(td, q2_fn(obs, actions))
    return aggregate(l, weights, float(l.shape[0] * num_replicas), regularization), td


def actor_loss(q1_fn, q2_fn, actions_and_logp, log_alpha, obs, weights=None, num_replicas=1):
    actions, logp = actions_and_logp
    q = torch.minimum(q1_fn(obs, actions), q2_fn(obs, actions))
    l = torch.exp(log_alpha) * logp - q
    return aggregate(l, weights, float(l.shape[0] * num_replicas))


def alpha_loss(logp, log_alpha, target_entropy, use_log_alpha=True, weights=None, num_replicas=1):
    diff = (-logp - target_entropy).detach()
    l = (log_alpha if use_log_alpha else torch.exp(log_alpha)) * diff
    return aggregate(l, weights, float(l.shape[0] * num_replicas))


class OracleSacAgent:
    """MLP actor (projection Dense emitting 2A) + twin MLP critics on concat(obs, action)."""

    def __init__(self, obs_dim, act_dim, actor_fc, critic_fc, act_mean, act_mag, actor_params,
                 critic1_params, critic2_params, actor_lr=3e-4, critic_lr=3e-4, alpha_lr=3e-4,
                 adam_eps=1e-7, gamma=0.99, reward_scale_factor=1.0, tau=0.005,
                 target_update_period=1, initial_log_alpha=0.0, target_entropy=None,
                 std_kind="exp", critic_loss_weight=0.5, actor_loss_weight=1.0,
                 alpha_loss_weight=1.0, td_errors_loss_fn=squared_difference,
                 use_log_alpha_in_alpha_loss=True, dtype=torch.float32, critic_obs_fc=(),
                 critic_act_fc=()):
        # dtype=torch.float64: the same training in double precision (parameters, forwards, losses,
        # Adam arithmetic with the same fp32-rounded hyper-parameters) -- the yardstick of the
        # free-running envelope test (tests/test_gpu_free_running.py)
        self.dtype = dtype
        cast = lambda ps: [p.to(dtype) for p in ps]
        actor_params, critic1_params, critic2_params = cast(actor_params), cast(critic1_params), \
            cast(critic2_params)
        self.A = act_dim
        self.actor_layers = nets.mlp_q_layers(actor_fc, 2 * act_dim, "relu")
        self.critic_layers = nets.mlp_q_layers(critic_fc, 1, "relu")
        # CriticNetwork's optional per-input Dense(relu) towers (agents/ddpg/critic_network.py:126-145,
        # :163-185); their parameters come first in a critic's list: observation, action, joint
        tower = lambda fc: [{"kind": "dense", "units": int(u), "act": "relu"} for u in fc]
        self.critic_obs_layers, self.critic_act_layers = tower(critic_obs_fc), tower(critic_act_fc)
        self.actor = [p.clone().requires_grad_(True) for p in actor_params]
        self.c1 = [p.clone().requires_grad_(True) for p in critic1_params]
        self.c2 = [p.clone().requires_grad_(True) for p in critic2_params]
        self.t1 = [p.detach().clone() for p in critic1_params]
        self.t2 = [p.detach().clone() for p in critic2_params]
        self.log_alpha = torch.tensor(float(initial_log_alpha), dtype=dtype, requires_grad=True)
        self.act_mean = torch.as_tensor(act_mean, dtype=torch.float32).to(dtype)
        self.act_mag = torch.as_tensor(act_mag, dtype=torch.float32).to(dtype)
        self.opt_actor = optim.Adam(actor_lr, eps=adam_eps)
        self.opt_critic = optim.Adam(critic_lr, eps=adam_eps)
        self.opt_alpha = optim.Adam(alpha_lr, eps=adam_eps)
        self.gamma, self.scale, self.tau = gamma, reward_scale_factor, tau
        self.period = target_update_period
        self.target_entropy = -act_dim / 2.0 if target_entropy is None else target_entropy
        self.kind = std_kind
        self.wc, self.wa, self.wl = critic_loss_weight, actor_loss_weight, alpha_loss_weight
        self.loss_fn = td_errors_loss_fn
        self.use_log_alpha = use_log_alpha_in_alpha_loss
        self.steps = 0

    def _mlp(self, layers, params, x, masks=None, tag=""):
        """nets.forward, or -- with `masks` (another implementation's ReLU activation pattern, one
        0/1 tensor per hidden layer, None for the head) -- the same stack on THAT linear branch
        (nets.forward_branch); units where the pattern differs from this evaluation's own are
        logged in self.flips as (tag, count, largest |z| / max|z|): legitimate only if the
        pre-activation is numerically zero (oracle/arbiter.py explains why this matters)."""
        if masks is None:
            return nets.forward(layers, params, x, dtype=self.dtype)
        out, pre = nets.forward_branch(layers, params, x, masks=masks, dtype=self.dtype)
        for z, m in zip(pre, masks):
            if m is None:
                continue
            z = z.detach()
            diff = (z > 0) != m.to(torch.bool)
            if bool(diff.any()):
                self.flips.append((tag, int(diff.sum()),
                                   float(z[diff].abs().max() / z.abs().max())))
        return out

    def q(self, params, obs, act, masks=None, tag=""):
        no, na = 2 * len(self.critic_obs_layers), 2 * len(self.critic_act_layers)
        if no or na:
            if masks is not None:
                raise NotImplementedError("activation masks with critic towers")
            o, a = obs.to(self.dtype), act.to(self.dtype)
            if no:
                o = nets.forward(self.critic_obs_layers, params[:no], o, dtype=self.dtype)
            if na:
                a = nets.forward(self.critic_act_layers, params[no:no + na], a, dtype=self.dtype)
            return nets.forward(self.critic_layers, params[no + na:], torch.cat([o, a], -1),
                                dtype=self.dtype).reshape(-1)
        return self._mlp(self.critic_layers, params,
                         torch.cat([obs.to(self.dtype), act.to(self.dtype)], -1), masks,
                         tag).reshape(-1)

    def pi(self, obs, eps, masks=None, tag=""):
        z = self._mlp(self.actor_layers, self.actor, obs.to(self.dtype), masks, tag)
        return tanh_normal(z, eps.to(self.dtype), self.act_mean, self.act_mag, self.kind)

    def train(self, obs, actions, next_obs, reward, discount, eps_next, eps_actor, eps_alpha,
              weights=None, grads_override=None, masks=None):
        """One SacAgent._train step with the three noise draws supplied by the caller.
        `grads_override` = dict(critic=[...], actor=[...], alpha=float): the optimizer steps are
        taken with THESE gradients (another implementation's) while this agent's own are still
        returned -- the two then enter every later phase from (rounding-)identical parameters, so
        each phase's gradients and each optimizer step can be compared at a tight tolerance
        without the chaotic drift of two free-running trainings.
        `masks` = {"c1.critic", "c2.critic", "c1.actor_q", "c2.actor_q", "actor": [...]}: ReLU
        activation patterns of the other implementation for the five DIFFERENTIATED forwards (see
        `_mlp`; the forwards that only produce values are continuous in a boundary flip)."""
        masks = masks or {}
        self.flips = []
        with torch.no_grad():
            na = self.pi(next_obs, eps_next)
        closs, _ = critic_loss(
            lambda o, a: self.q(self.c1, o, a, masks.get("c1.critic"), "c1.critic"),
            lambda o, a: self.q(self.c2, o, a, masks.get("c2.critic"), "c2.critic"),
            lambda o, a: self.q(self.t1, o, a), lambda o, a: self.q(self.t2, o, a),
            na, self.log_alpha.detach(), obs, actions, next_obs, reward,
            discount, self.loss_fn, self.gamma, self.scale, weights)
        closs = self.wc * closs
        cparams = self.c1 + self.c2
        cgrads = torch.autograd.grad(closs, cparams)
        self.opt_critic.step(cparams, grads_override["critic"] if grads_override else cgrads)

        aloss = self.wa * actor_loss(
            lambda o, a: self.q(self.c1, o, a, masks.get("c1.actor_q"), "c1.actor_q"),
            lambda o, a: self.q(self.c2, o, a, masks.get("c2.actor_q"), "c2.actor_q"),
            self.pi(obs, eps_actor, masks.get("actor"), "actor"),
            self.log_alpha.detach(), obs, weights)
        agrads = torch.autograd.grad(aloss, self.actor)
        self.opt_actor.step(self.actor, grads_override["actor"] if grads_override else agrads)

        with torch.no_grad():
            _, logp = self.pi(obs, eps_alpha)
        lloss = self.wl * alpha_loss(logp, self.log_alpha, self.target_entropy,
                                     self.use_log_alpha, weights)
        lgrad = torch.autograd.grad(lloss, [self.log_alpha])
        self.opt_alpha.step([self.log_alpha],
                            [torch.tensor(float(grads_override["alpha"]), dtype=self.dtype)]
                            if grads_override else lgrad)

        self.steps += 1
        if self.steps % self.period == 0:
            optim.soft_update(self.t1, self.c1, self.tau)
            optim.soft_update(self.t2, self.c2, self.tau)
        total = closs + aloss + lloss
        return dict(loss=float(total), critic_loss=float(closs), actor_loss=float(aloss),
                    alpha_loss=float(lloss), critic_grads=[g.clone() for g in cgrads],
                    actor_grads=[g.clone() for g in agrads], alpha_grad=float(lgrad[0]))
