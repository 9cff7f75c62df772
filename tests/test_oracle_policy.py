"""CPU: oracle/policy.py (Boltzmann action selection) against the reference's own numbers
(tf_agents/policies/boltzmann_policy_test.py:87-123) and against the softmax it samples."""
import numpy as np

from oracle import policy as opolicy


def test_reference_logits_and_mode():
    # DummyNet: Dense(2) with kernel [[1, 1.5], [1, 1.5]], bias 1 on observation (1, 2) -> Q = 4, 5.5
    q = np.array([[1 * 1 + 2 * 1 + 1, 1 * 1.5 + 2 * 1.5 + 1]], np.float32)
    assert q.tolist() == [[4.0, 5.5]]                                     # testLogits
    assert opolicy.boltzmann_logits(q, 0.5).tolist() == [[8.0, 11.0]]     # testLogits
    assert int(np.argmax(opolicy.boltzmann_logits(q, 0.9))) == 1          # testDistribution


def test_masked_logits_are_float32_min():
    q = np.array([[1.0, 2.0, 3.0]], np.float32)
    lg = opolicy.boltzmann_logits(q, 2.0, mask=[[1, 0, 1]])
    assert lg[0, 1] == -np.finfo(np.float32).max and lg[0, 0] == 0.5 and lg[0, 2] == 1.5


def test_samples_follow_the_softmax_and_the_mask():
    rng = np.random.default_rng(0)
    q = np.tile(rng.normal(size=(1, 5)).astype(np.float32), (20000, 1))
    T = 0.7
    a = opolicy.boltzmann_actions(q, T, seed=3, call=0)
    p = np.exp(q[0] / T - (q[0] / T).max())
    p /= p.sum()
    freq = np.bincount(a, minlength=5) / len(a)
    assert np.abs(freq - p).max() < 0.012
    a2 = opolicy.boltzmann_actions(q, T, seed=3, call=1)
    assert not np.array_equal(a, a2)             # the call number moves the stream
    mask = np.tile(np.array([[0, 1, 0, 1, 0]]), (q.shape[0], 1))
    am = opolicy.boltzmann_actions(q, T, seed=3, call=0, mask=mask, action_min=10)
    assert set(np.unique(am)) == {11, 13}
    # temperature -> 0: the arg-max (boltzmann_policy.py:50-53)
    cold = opolicy.boltzmann_actions(q[:64], 1e-3, seed=3, call=0)
    assert (cold == int(np.argmax(q[0]))).all()
