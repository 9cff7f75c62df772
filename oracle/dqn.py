"""CPU restatement of the DqnAgent loss / train step (TEST INFRASTRUCTURE; oracle/__init__.py).

numpy part (elementwise, float32, reference op order):
  to_n_step_transition            tf_agents/trajectories/trajectory.py:716-850
  discounted_return (foldr)       tf_agents/utils/value_ops.py:21-99
  index_with_actions              tf_agents/utils/common.py:367-411
  compute_td_targets              tf_agents/agents/dqn/dqn_agent.py:75-78
  _td_loss / _loss                tf_agents/agents/dqn/dqn_agent.py:451-460, 462-579
  Ddqn _compute_next_q_values     tf_agents/agents/dqn/dqn_agent.py:659-700
  huber (tf.compat.v1.losses.huber_loss, delta=1) / squared   tf_agents/utils/common.py:1199-1208
  aggregate_losses                tf_agents/utils/common.py:1400-1476
torch-CPU part: network forward/backward (autograd) + optimizer + target update for the
end-to-end train-step parity tests (dqn_agent.py:412-449, 385-409).
"""
import numpy as np
import torch

from oracle import nets, optim

f32 = np.float32
LAST = 2


def n_step_return(reward, discount, gamma):
    """reward/discount [B,T]; returns (n-step reward [B], final discount [B]) over frames [:-1]."""
    reward = np.asarray(reward, f32)[:, :-1]
    discount = np.asarray(discount, f32)[:, :-1]
    n = reward.shape[1]
    acc = np.zeros(reward.shape[0], f32)
    g = f32(gamma)
    for t in range(n - 1, -1, -1):
        acc = ((acc * (g * discount[:, t]).astype(f32)).astype(f32) + reward[:, t]).astype(f32)
    dprod = np.ones(reward.shape[0], f32)
    for t in range(n):
        dprod = (dprod * discount[:, t]).astype(f32)
    gpow = f32(float(gamma) ** (n - 1))
    return acc, (gpow * dprod).astype(f32)


def huber(td_targets, q):
    err = (q - td_targets).astype(f32)
    abs_err = np.abs(err)
    quad = np.minimum(abs_err, f32(1.0))
    lin = (abs_err - quad).astype(f32)
    return ((f32(0.5) * quad).astype(f32) * quad + f32(1.0) * lin).astype(f32)


def squared(td_targets, q):
    d = (td_targets - q).astype(f32)
    return (d * d).astype(f32)


def greedy_action(q, mask=None):
    q = np.asarray(q, f32).copy()
    if mask is not None:
        q[np.asarray(mask) == 0] = -np.finfo(np.float32).max
    return np.argmax(q, axis=1)  # first arg-max, like Categorical(logits).mode()


def td_loss_from_q(q_online, q_next_target, actions, reward, discount, step_type, gamma=1.0,
                   reward_scale=1.0, weights=None, loss="huber", q_next_select=None,
                   next_mask=None, global_batch=None, gamma_loss=None):
    """Everything after the network forwards.  Returns dict(loss, td_loss, td_error, dq).

    `gamma` is the agent's gamma used by AsNStepTransition (data_converter.py:613-655);
    `gamma_loss` is the `gamma` argument of DqnAgent._loss (dqn_agent.py:462-470, default 1.0
    when `loss()` is called directly, self._gamma inside `_train`); None -> same as gamma."""
    if gamma_loss is None:
        gamma_loss = gamma
    q_online = np.asarray(q_online, f32)
    q_next_target = np.asarray(q_next_target, f32)
    B, A = q_online.shape
    ret, final_disc = n_step_return(reward, discount, gamma)
    sel = q_next_target if q_next_select is None else np.asarray(q_next_select, f32)
    best = greedy_action(sel, next_mask)
    next_q = q_next_target[np.arange(B), best]
    acts = np.asarray(actions).reshape(B, -1)[:, 0].astype(np.int64)
    q = q_online[np.arange(B), acts]
    rewards = (f32(reward_scale) * ret).astype(f32)
    discounts = (f32(gamma_loss) * final_disc).astype(f32)
    td_targets = (rewards + (discounts * next_q).astype(f32)).astype(f32)
    td_error = (td_targets - q).astype(f32)
    if callable(loss):
        # a td_errors_loss_fn of the caller's own (dqn_agent.py:114, 250-251, 458): the callable
        # returns (element-wise loss, its derivative with respect to q) as float32 arrays
        td_loss, dl_dq = loss(td_targets, q)
        td_loss, dl_dq = np.asarray(td_loss, f32), np.asarray(dl_dq, f32)
    elif loss == "huber":
        td_loss = huber(td_targets, q)
        err = (q - td_targets).astype(f32)
        dl_dq = np.clip(err, -1.0, 1.0).astype(f32)
    else:
        td_loss = squared(td_targets, q)
        dl_dq = (f32(-2.0) * (td_targets - q)).astype(f32)
    st0 = np.asarray(step_type).reshape(B, -1)[:, 0]
    valid = (st0 != LAST).astype(f32)
    td_error = (valid * td_error).astype(f32)
    td_loss = (valid * td_loss).astype(f32)
    w = np.ones(B, f32) if weights is None else np.broadcast_to(np.asarray(weights, f32), (B,))
    weighted = np.where(w == 0, f32(0), td_loss * w).astype(f32)
    gb = f32(B if global_batch is None else global_batch)
    total = f32(weighted.sum(dtype=f32) / gb)
    dq = np.zeros((B, A), f32)
    dq[np.arange(B), acts] = ((valid * dl_dq).astype(f32) * w / gb).astype(f32)
    return dict(loss=total, td_loss=td_loss, td_error=td_error, dq=dq, td_targets=td_targets)


class OracleDqnAgent:
    """torch-CPU DQN/DDQN agent on a layer list (oracle/nets.py), batch-major [B,T] experience."""

    def __init__(self, layers, input_shape, num_actions, params, optimizer=None, gamma=1.0,
                 reward_scale=1.0, loss="huber", double_q=False, target_update_tau=1.0,
                 target_update_period=1, n_step=1, l2=0.0):
        self.layers, self.input_shape, self.A = layers, tuple(input_shape), num_actions
        self.params = [p.clone().requires_grad_(True) for p in params]
        self.target = [p.detach().clone() for p in params]
        self.opt = optimizer
        self.gamma, self.reward_scale, self.loss_kind = gamma, reward_scale, loss
        self.double_q, self.tau, self.period, self.n = double_q, target_update_tau, \
            target_update_period, n_step
        self.l2 = l2
        self.update_calls = 0
        self.train_steps = 0

    def q_values(self, obs, target=False):
        return nets.forward(self.layers, self.target if target else self.params, obs)

    def loss(self, obs, actions, reward, discount, step_type, weights=None, global_batch=None):
        """obs [B,T,...] torch; other fields numpy/torch [B,T].  Returns (loss tensor, extras)."""
        B = obs.shape[0]
        q_all = self.q_values(obs[:, 0])
        with torch.no_grad():
            qt = self.q_values(obs[:, -1], target=True)
            qsel = self.q_values(obs[:, -1]) if self.double_q else None
        aux = td_loss_from_q(q_all.detach().numpy(), qt.numpy(), np.asarray(actions),
                             np.asarray(reward), np.asarray(discount), np.asarray(step_type),
                             self.gamma, self.reward_scale, weights, self.loss_kind,
                             None if qsel is None else qsel.numpy(), None, global_batch)
        # scalar loss as a differentiable function of q_all: the analytic dq reproduces the
        # gradient of sum(valid*loss*w)/global_batch, so use it as the upstream gradient.
        surrogate = (q_all * torch.from_numpy(aux["dq"])).sum()
        total = torch.tensor(float(aux["loss"]))
        if self.l2 > 0:
            reg = sum((p ** 2).sum() for p in self.params[0::2]) * self.l2
            surrogate = surrogate + reg
            total = total + reg.detach()
        return total, surrogate, aux

    def train(self, obs, actions, reward, discount, step_type, weights=None, global_batch=None,
              grad_hook=None):
        total, surrogate, aux = self.loss(obs, actions, reward, discount, step_type, weights,
                                          global_batch)
        grads = torch.autograd.grad(surrogate, self.params)
        if grad_hook is not None:
            grads = grad_hook(grads)
        self.opt.step(self.params, [g.detach() for g in grads])
        self.train_steps += 1
        self.update_calls += 1
        if self.update_calls % self.period == 0:
            optim.soft_update(self.target, [p.detach() for p in self.params], self.tau)
        return total, aux, grads
