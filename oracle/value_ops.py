"""numpy restatement of tf_agents/utils/value_ops.py (TEST INFRASTRUCTURE).

discounted_return :21-99, generalized_advantage_estimation :102-164, and the advantage normaliser
tf_agents/agents/ppo/ppo_agent.py:100-110.  All arithmetic in float32, in the op order of the
reference's tf.scan bodies (no FMA), so the HIP scans can be compared bit-for-bit.
"""
import numpy as np

f32 = np.float32


def discounted_return(rewards, discounts, final_value=None, time_major=True,
                      provide_all_returns=True):
    rewards = np.asarray(rewards, dtype=f32)
    discounts = np.asarray(discounts, dtype=f32)
    if not time_major:
        rewards, discounts = rewards.T, discounts.T
    acc = np.zeros_like(rewards[-1]) if final_value is None else np.asarray(final_value, f32)
    out = np.zeros_like(rewards)
    for t in range(rewards.shape[0] - 1, -1, -1):
        acc = (acc * discounts[t]).astype(f32) + rewards[t]
        acc = acc.astype(f32)
        out[t] = acc
    if not provide_all_returns:
        return acc
    return out if time_major else out.T


def generalized_advantage_estimation(values, final_value, discounts, rewards, td_lambda=1.0,
                                     time_major=True):
    values = np.asarray(values, f32)
    discounts = np.asarray(discounts, f32)
    rewards = np.asarray(rewards, f32)
    final_value = np.asarray(final_value, f32)
    if not time_major:
        values, discounts, rewards = values.T, discounts.T, rewards.T
    next_values = np.concatenate([values[1:], final_value[None]], axis=0)
    delta = ((rewards + (discounts * next_values).astype(f32)).astype(f32) - values).astype(f32)
    weighted = (discounts * f32(td_lambda)).astype(f32)
    acc = np.zeros_like(final_value)
    out = np.zeros_like(values)
    for t in range(values.shape[0] - 1, -1, -1):
        acc = (delta[t] + (weighted[t] * acc).astype(f32)).astype(f32)
        out[t] = acc
    return out if time_major else out.T


def normalize_advantages(advantages, variance_epsilon=1e-8):
    """(A - mean) * rsqrt(var + eps) over all axes (ppo_agent.py:100-110)."""
    a = np.asarray(advantages, f32)
    mean = a.mean(dtype=np.float64)
    var = ((a.astype(np.float64) - mean) ** 2).mean()
    inv = 1.0 / np.sqrt(var + variance_epsilon)
    return (a * f32(inv) + f32(-mean * inv)).astype(f32), f32(mean), f32(var)
