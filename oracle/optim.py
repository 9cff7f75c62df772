"""torch-CPU restatement of the optimizer arithmetic (TEST INFRASTRUCTURE).

The reference delegates to tf / tf-keras kernels that are not vendored (optimizer
arithmetic is "parity unpinned" -- reference tests only check that variables change).  Restated
from the documented update rules:
  Adam (TF ApplyAdam):  alpha = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)(1-b1); v += (g^2-v)(1-b2);
                        p -= m*alpha/(sqrt(v)+eps)          (agents/dqn/examples/v2/train_eval.py:180)
  RMSprop (keras):      ms = rho*ms+(1-rho)g^2; mg = rho*mg+(1-rho)g; denom = ms-mg^2+eps;
                        mom = momentum*mom + lr*g*rsqrt(denom); p -= mom
                                                 (examples/dqn/mnih15/dqn_train_eval_atari.py:176-182)
"""
import numpy as np
import torch

f32 = np.float32


class Adam:
    def __init__(self, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7):
        self.lr, self.b1, self.b2, self.eps = f32(lr), f32(beta1), f32(beta2), f32(eps)
        self.t = 0
        self.m = self.v = None

    def step(self, params, grads):
        if self.m is None:
            self.m = [torch.zeros_like(p) for p in params]
            self.v = [torch.zeros_like(p) for p in params]
        self.t += 1
        t = f32(self.t)
        alpha = f32(self.lr * np.sqrt(f32(1) - np.power(self.b2, t, dtype=f32), dtype=f32) /
                    (f32(1) - np.power(self.b1, t, dtype=f32)))
        with torch.no_grad():
            for p, g, m, v in zip(params, grads, self.m, self.v):
                m.add_((g - m) * float(f32(1) - self.b1))
                v.add_((g * g - v) * float(f32(1) - self.b2))
                p.sub_((m * float(alpha)) / (v.sqrt() + float(self.eps)))


class RMSprop:
    def __init__(self, lr=1e-3, rho=0.9, momentum=0.0, eps=1e-7, centered=False):
        self.lr, self.rho, self.mom, self.eps = f32(lr), f32(rho), f32(momentum), f32(eps)
        self.centered = centered
        self.ms = self.mg = self.mo = None

    def step(self, params, grads):
        if self.ms is None:
            self.ms = [torch.zeros_like(p) for p in params]
            self.mg = [torch.zeros_like(p) for p in params]
            self.mo = [torch.zeros_like(p) for p in params]
        rho, omr = float(self.rho), float(f32(1) - self.rho)
        with torch.no_grad():
            for p, g, ms, mg, mo in zip(params, grads, self.ms, self.mg, self.mo):
                ms.copy_(rho * ms + omr * (g * g))
                if self.centered:
                    mg.copy_(rho * mg + omr * g)
                    denom = ms - mg * mg + float(self.eps)
                else:
                    denom = ms + float(self.eps)
                inc = float(self.lr) * g * (1.0 / denom.sqrt())
                if self.mom > 0:
                    mo.copy_(float(self.mom) * mo + inc)
                    p.sub_(mo)
                else:
                    p.sub_(inc)


def soft_update(target, source, tau):
    """w_t = (1-tau)*w_t + tau*w_s  (tf_agents/utils/common.py:314-346)."""
    with torch.no_grad():
        for t, s in zip(target, source):
            t.copy_((1.0 - tau) * t + tau * s)
