"""Float64 twins for FREE-RUNNING parity (TEST INFRASTRUCTURE; oracle/__init__.py).

The bench-configuration parity tests compare the HIP path with the fp32 oracle link by link (the
oracle is re-synchronised every step: two fp32 trainings of a ReLU network drift apart
chaotically).  That leaves one question open: do K FREE steps of the HIP path stay as close to
the exact trajectory as an independent fp32 implementation does?  The yardstick is the same
training in float64 -- same data, same fp32-rounded hyper-parameters, double-precision
parameters / forwards / losses / optimizer arithmetic.  `envelope` states the acceptance rule.

  F64DqnAgent     DqnAgent._train in float64 (tf_agents/agents/dqn/dqn_agent.py:412-449, 462-579)
                  on a layer list of oracle/nets.py, batch-major [B, 2] experience (n_step = 1)
  OracleSacAgent(dtype=torch.float64)   lives in oracle/sac.py
"""
import numpy as np
import torch

from oracle import nets, optim

LAST = 2


class F64DqnAgent:
    def __init__(self, layers, params, optimizer, gamma=0.99, reward_scale=1.0, loss="huber",
                 target_update_tau=1.0, target_update_period=1):
        self.layers = layers
        self.params = [p.detach().double().clone().requires_grad_(True) for p in params]
        self.target = [p.detach().double().clone() for p in params]
        self.opt, self.gamma, self.reward_scale, self.loss_kind = optimizer, gamma, reward_scale, \
            loss
        self.tau, self.period, self.calls = target_update_tau, target_update_period, 0

    def train(self, obs, actions, reward, discount, step_type):
        """obs [B, 2, ...] (uint8 / float), the other fields [B, 2].  Returns the scalar loss."""
        f64 = torch.float64
        as64 = lambda a: torch.as_tensor(np.asarray(a)).to(f64)
        q_all = nets.forward(self.layers, self.params, obs[:, 0], dtype=f64)
        with torch.no_grad():
            qt = nets.forward(self.layers, self.target, obs[:, 1], dtype=f64)
        next_q = qt.max(dim=1).values
        acts = torch.as_tensor(np.asarray(actions)).reshape(obs.shape[0], -1)[:, :1].long()
        q = q_all.gather(1, acts)[:, 0]
        # n_step = 1: the n-step reward is reward[:, 0], the discount gamma * discount[:, 0]
        # (trajectory.py:716-850 with one transition; data_converter.py:613-655)
        td_target = (self.reward_scale * as64(reward)[:, 0] +
                     self.gamma * as64(discount)[:, 0] * next_q).detach()
        err = q - td_target
        if self.loss_kind == "huber":
            a = err.abs()
            quad = torch.clamp(a, max=1.0)
            el = 0.5 * quad * quad + (a - quad)
        else:
            el = err * err
        valid = (torch.as_tensor(np.asarray(step_type)).reshape(obs.shape[0], -1)[:, 0] != LAST)
        loss = (el * valid.to(f64)).sum() / obs.shape[0]
        grads = torch.autograd.grad(loss, self.params)
        self.opt.step(self.params, [g.detach() for g in grads])
        self.calls += 1
        if self.calls % self.period == 0:
            optim.soft_update(self.target, [p.detach() for p in self.params], self.tau)
        return float(loss)


def envelope(hip, fp32, f64, factor=3.0, floor=1e-6):
    """Acceptance rule for a free-running loss trajectory.  With e_x[k] = |x[k] - f64[k]|:
        e_hip[k] <= factor * max_{j <= k} e_fp32[j] + floor * |f64[k]|      for every step k
    i.e. the HIP path may be at most `factor` times as far from the exact trajectory as the
    independent fp32 implementation has been so far (the running maximum, because the drift of
    a chaotic pair is not monotone), plus an fp32-rounding floor.  Returns (ok, rows) with
    rows[k] = (k, f64, e_hip, e_fp32, bound)."""
    rows, ok, run = [], True, 0.0
    for k, (h, s, d) in enumerate(zip(hip, fp32, f64)):
        e_h, e_s = abs(float(h) - float(d)), abs(float(s) - float(d))
        run = max(run, e_s)
        bound = factor * run + floor * abs(float(d))
        ok = ok and e_h <= bound
        rows.append((k, float(d), e_h, e_s, bound))
    return ok, rows


def format_rows(rows):
    return "\n".join(f"  step {k}: f64 loss {d:.9g}  |HIP - f64| {eh:.2e}  |fp32 - f64| {es:.2e}  "
                     f"bound {b:.2e}" for k, d, eh, es, b in rows)
