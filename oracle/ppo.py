"""CPU restatement of the PPOAgent arithmetic (TEST INFRASTRUCTURE; see oracle/__init__.py).

torch-CPU float32; gradients come from autograd, i.e. independently of the analytic gradients of
csrc/ppo.hip.  Follows tf_agents/agents/ppo/ppo_agent.py:
  _normalize_advantages            :100-110   (tf.nn.moments + batch_normalization)
  compute_advantages               :440-479   (incl. the value_preds[:, :-1] / final_value quirk)
  compute_return_and_advantage     :617-719
  _preprocess (padding)            :721-807
  get_loss                         :481-615
  entropy_regularization_loss      :1159-1201
  value_estimation_loss            :1203-1327
  policy_gradient_loss             :1329-1512
  kl_cutoff_loss / adaptive_kl_loss / kl_penalty_loss / update_adaptive_kl_beta   :1514-1690
  l2_regularization_loss           :1088-1157
and agents/ppo/ppo_utils.py:35-59 (make_trajectory_mask), utils/common.py:883-895
(get_episode_mask), utils/common.py:1400-1476 (aggregate_losses: sum(loss*w)/(N*replicas), entries
with w == 0 contribute exactly 0), utils/value_ops.py (oracle/value_ops.py).
TFP closed forms restated (third party, not vendored): Normal.log_prob, Normal.entropy,
kl_normal_normal; MultivariateNormalDiag = sum over the event dimension.
Pinned on the reference's known answers (tests/test_oracle_ppo.py): 123.205, -0.0164646133,
-3.70111 * 0.1, the epoch-loss composition, the GAE vector, kl cutoff 30 * 0.24^2, beta 1 -> 1.5
-> 1.  Unpinned by the reference: the combination value_clipping / log_prob_clipping / td-lambda
returns (no numeric test) -- restated from the code.
"""
import math

import numpy as np
import torch

from oracle import value_ops

HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)
LAST = 2


def normal_log_prob(loc, scale, x):
    """sum_d log N(x; loc, scale): -0.5*squared_difference(x/s, loc/s) - (0.5 log 2pi + log s)."""
    z = x / scale - loc / scale
    return (-0.5 * z * z - (HALF_LOG_2PI + torch.log(scale))).sum(-1)


def normal_entropy(scale):
    return (0.5 + HALF_LOG_2PI + torch.log(scale)).sum(-1)


def normal_kl(loc_a, scale_a, loc_b, scale_b):
    """kl(a || b) summed over the event dim (tfp kl_normal_normal)."""
    dl = torch.log(scale_a) - torch.log(scale_b)
    dm = loc_a / scale_b - loc_b / scale_b
    return (0.5 * dm * dm + 0.5 * torch.expm1(2.0 * dl) - dl).sum(-1)


def aggregate(per_example, weights, replicas=1):
    """common.aggregate_losses(...).total_loss for a [N] loss."""
    w = torch.as_tensor(weights, dtype=torch.float32)
    prod = torch.where(w == 0, torch.zeros_like(per_example), per_example * w)
    return prod.sum() / (per_example.numel() * replicas)


def losses(loc, scale, actions, old_logp, adv, returns, vpred, weights, *, clip_eps=0.0,
           value_clip=0.0, c_v=0.5, c_e=0.0, logp_clip=0.0, old_loc=None, old_scale=None,
           old_vpred=None, kl_beta=0.0, kl_cutoff_coef=0.0, kl_cutoff=0.0, replicas=1):
    """All terms of PPOAgent.get_loss for flattened [N] samples.  Inputs are torch tensors (loc /
    scale / vpred may require grad).  Returns a dict of scalar tensors."""
    w = torch.as_tensor(weights, dtype=torch.float32)
    lp = normal_log_prob(loc, scale, actions)
    if logp_clip > 0:
        lp = torch.clamp(lp, -logp_clip, logp_clip)
    ratio = torch.exp(lp - old_logp)
    ratio_c = torch.clamp(ratio, 1 - clip_eps, 1 + clip_eps)
    obj, obj_c = ratio * adv, ratio_c * adv
    pg_el = -torch.minimum(obj, obj_c) if clip_eps > 0 else -obj
    pg = aggregate(pg_el, w, replicas)
    clip_fraction = ((ratio - 1.0).abs() > clip_eps).float().mean() if clip_eps > 0 \
        else torch.zeros(())
    verr = (returns - vpred) ** 2
    if value_clip > 0:
        vc = old_vpred + torch.clamp(vpred - old_vpred, -value_clip, value_clip)
        verr = torch.maximum(verr, (returns - vc) ** 2)
    ve = aggregate(verr, w, replicas) * c_v
    ent = normal_entropy(scale)
    ent_loss = aggregate(-ent, w, replicas) * c_e if c_e > 0 else torch.zeros(())
    out = dict(policy_gradient_loss=pg, value_estimation_loss=ve,
               entropy_regularization_loss=ent_loss, clip_fraction=clip_fraction,
               entropy_mean=(ent * w).mean())
    if old_loc is not None:
        kl = normal_kl(old_loc, old_scale, loc, scale) * w
        mean_kl = kl.mean()
        adaptive = kl_beta * mean_kl
        cutoff = kl_cutoff_coef * torch.clamp(mean_kl - kl_cutoff, min=0.0) ** 2 \
            if (kl_cutoff_coef > 0 and kl_cutoff > 0) else torch.zeros(())
        out.update(mean_kl=mean_kl, adaptive_kl_loss=adaptive, kl_cutoff_loss=cutoff,
                   kl_penalty_loss=adaptive + cutoff)
    else:
        out.update(mean_kl=torch.zeros(()), adaptive_kl_loss=torch.zeros(()),
                   kl_cutoff_loss=torch.zeros(()), kl_penalty_loss=torch.zeros(()))
    out["total"] = pg + ve + ent_loss + out["kl_penalty_loss"]
    return out


def kl_cutoff_loss(kl_divergence, kl_cutoff_factor, adaptive_kl_target, kl_cutoff_coef):
    """ppo_agent.py:1514-1560 on an explicit kl tensor (for the reference's unit test)."""
    kl = np.asarray(kl_divergence, np.float32)
    over = max(float(kl.mean()) - kl_cutoff_factor * adaptive_kl_target, 0.0)
    return kl_cutoff_coef * over * over


def update_adaptive_kl_beta(beta, mean_kl, target, tolerance):
    """ppo_agent.py:1642-1690."""
    f = 1.0
    if mean_kl < target * (1.0 - tolerance):
        f = 1.0 / 1.5
    elif mean_kl > target * (1.0 + tolerance):
        f = 1.5
    return float(np.clip(np.float32(beta) * np.float32(f), 10e-16, 10e16))


def episode_discounts(discount, next_step_type, gamma):
    """discount * gamma * (next_step_type != LAST) over the first T of T+1 columns."""
    d = np.asarray(discount, np.float32)[:, :-1] * np.float32(gamma)
    m = (np.asarray(next_step_type)[:, :-1] != LAST).astype(np.float32)
    return (d * m).astype(np.float32)


def compute_advantages(rewards, returns, discounts, value_preds, use_gae, lam):
    """ppo_agent.py:440-479.  value_preds is [B, T+1]; NOTE the reference first drops the last
    column and THEN bootstraps GAE from the new last column (V(s_{T-1}), not V(s_T))."""
    vp = np.asarray(value_preds, np.float32)[:, :-1]
    if use_gae:
        return value_ops.generalized_advantage_estimation(
            vp, vp[:, -1], discounts, rewards, lam, time_major=False)
    return (np.asarray(returns, np.float32) - vp).astype(np.float32)


def compute_return_and_advantage(reward, discount, next_step_type, value_preds, gamma=0.99,
                                 lam=0.95, use_gae=False, use_td_lambda_return=False):
    """ppo_agent.py:617-719 without reward normalisation.  reward/discount/next_step_type are the
    [B, T+1] trajectory fields; value_preds [B, T+1].  Returns (returns, advantages), both [B, T]."""
    rewards = np.asarray(reward, np.float32)[:, :-1]
    discounts = episode_discounts(discount, next_step_type, gamma)
    vp = np.asarray(value_preds, np.float32)
    returns = value_ops.discounted_return(rewards, discounts, vp[:, -1], time_major=False)
    adv = compute_advantages(rewards, returns, discounts, vp, use_gae, lam)
    if use_td_lambda_return and use_gae:
        returns = (adv + vp[:, :-1]).astype(np.float32)
    return returns, adv


def pad_last(x):
    """_preprocess: returns / advantages padded with one zero column (ppo_agent.py:790-799)."""
    x = np.asarray(x, np.float32)
    return np.concatenate([x, np.zeros((x.shape[0], 1), np.float32)], axis=1)


def trajectory_mask(step_type, returns, advantages):
    """ppo_utils.make_trajectory_mask on [B, T+1] (padded) fields."""
    st = np.asarray(step_type)
    valid = ~((np.asarray(returns) == 0) & (np.asarray(advantages) == 0))
    return ((st != LAST) & valid).astype(np.float32)


def normalize_advantages(adv, eps=1e-8):
    return value_ops.normalize_advantages(adv, eps)[0]
