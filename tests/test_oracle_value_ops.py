"""Pins oracle/value_ops.py to the reference's precomputed vectors and naive ground truths
(tf_agents/utils/value_ops_test.py:28-84,179-201,239-278).  CPU only."""
import numpy as np
import pytest

from oracle import value_ops


def _naive_gae(discounts, rewards, values, final_value, td_lambda):
    """Definition-based GAE: sum_l (prod discounts * lambda^l) * delta_{t+l} (test :28-61)."""
    T, B = rewards.shape
    next_values = np.concatenate([values[1:], final_value[None]], 0)
    delta = rewards + discounts * next_values - values
    out = np.zeros_like(rewards, dtype=np.float64)
    for t in range(T):
        w = np.ones(B)
        for l in range(T - t):
            out[t] += w * delta[t + l]
            w = w * discounts[t + l] * td_lambda
    return out


def _numpy_discounted_return(rewards, discounts, final_value):
    T = rewards.shape[0]
    out = np.zeros_like(rewards, dtype=np.float64)
    acc = final_value.astype(np.float64)
    for t in range(T - 1, -1, -1):
        acc = rewards[t] + discounts[t] * acc
        out[t] = acc
    return out


def test_discounted_return_precomputed():  # :179-201
    got = value_ops.discounted_return(np.ones(9, np.float32),
                                      np.array([1, 1, 1, 1, 0, .9, .9, .9, .9], np.float32),
                                      final_value=np.float32(8))
    expected = [5, 4, 3, 2, 1, 8 * 0.9**4 + 3.439, 8 * 0.9**3 + 2.71, 8 * 0.9**2 + 1.9,
                8 * 0.9 + 1]
    np.testing.assert_allclose(got, expected, rtol=1e-6)


@pytest.mark.parametrize("B,T", [(1, 1), (7, 9)])
def test_discounted_return_random(B, T):
    rng = np.random.RandomState(0)
    r = rng.rand(T, B).astype(np.float32)
    d = rng.rand(T, B).astype(np.float32)
    fv = rng.rand(B).astype(np.float32)
    np.testing.assert_allclose(value_ops.discounted_return(r, d, fv),
                               _numpy_discounted_return(r, d, fv), rtol=1e-5)
    np.testing.assert_allclose(value_ops.discounted_return(r.T, d.T, fv, time_major=False),
                               _numpy_discounted_return(r, d, fv).T, rtol=1e-5)
    np.testing.assert_allclose(
        value_ops.discounted_return(r, d, fv, provide_all_returns=False),
        _numpy_discounted_return(r, d, fv)[0], rtol=1e-5)


@pytest.mark.parametrize("B,T,lam", [(1, 1, 0.7), (7, 9, 0.7), (7, 9, 0.0), (7, 9, 1.0)])
def test_gae_random(B, T, lam):  # :208-237
    rng = np.random.RandomState(1)
    r, d, v = (rng.rand(T, B).astype(np.float32) for _ in range(3))
    fv = rng.rand(B).astype(np.float32)
    got = value_ops.generalized_advantage_estimation(v, fv, d, r, lam)
    np.testing.assert_allclose(got, _naive_gae(d, r, v, fv, lam), rtol=1e-5, atol=1e-6)


def test_gae_precomputed():  # :239-278
    d = np.array([[1, 1, 1, 1, 0, .9, .9, .9, 0]] * 2, np.float32)
    got = value_ops.generalized_advantage_estimation(
        values=np.full((2, 9), 3.0, np.float32), final_value=np.full(2, 3.0, np.float32),
        discounts=d, rewards=np.ones((2, 9), np.float32), td_lambda=0.95, time_major=False)
    truth = [2.0808625, 1.13775, 0.145, -0.9, -2.0, 0.56016475, -0.16355, -1.01, -2.0]
    np.testing.assert_allclose(got, [truth, truth], rtol=1e-5)


def test_normalize_advantages():  # ppo_agent.py:100-110
    a = np.array([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]], np.float32)
    out, mean, var = value_ops.normalize_advantages(a, 1e-8)
    np.testing.assert_allclose(mean, 3.5)
    np.testing.assert_allclose(var, 35.0 / 12.0, rtol=1e-6)
    np.testing.assert_allclose(out, (a - 3.5) / np.sqrt(35.0 / 12.0 + 1e-8), rtol=1e-5)
