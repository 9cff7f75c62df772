"""Pins oracle/ppo.py on the known answers of tf_agents/agents/ppo/ppo_agent_test.py (CPU only).

The reference's DummyActorNet is Dense(2 -> 2, kernel [[2,1],[1,1]], bias [5,5]) split into
(loc, scale); DummyValueNet is Dense(2 -> 1, kernel [2,1], bias 5)  (ppo_agent_test.py:45-170).
On observations [[1,2],[3,4]]: loc = [9, 15], scale = [8, 12], value = [9, 15]."""
import numpy as np
import torch

from oracle import ppo as oppo

OBS = np.array([[1, 2], [3, 4]], np.float32)
ACTOR_K = np.array([[2.0, 1.0], [1.0, 1.0]], np.float32)
ACTOR_B = np.array([5.0, 5.0], np.float32)
VALUE_K = np.array([[2.0], [1.0]], np.float32)
VALUE_B = np.array([5.0], np.float32)


def dummy_nets(obs):
    out = obs @ ACTOR_K + ACTOR_B
    loc, scale = out[:, :1], out[:, 1:]
    val = (obs @ VALUE_K + VALUE_B)[:, 0]
    return torch.tensor(loc), torch.tensor(scale), torch.tensor(val)


def test_value_estimation_loss_known_answer():  # ppo_agent_test.py:912-937 -> 123.205
    loc, scale, val = dummy_nets(OBS)
    out = oppo.losses(loc, scale, torch.zeros(2, 1), torch.zeros(2), torch.zeros(2),
                      torch.tensor([1.9, 1.0]), val, torch.ones(2), c_v=1.0)
    np.testing.assert_allclose(float(out["value_estimation_loss"]), 123.205, rtol=1e-6)


def test_policy_gradient_loss_known_answer():  # ppo_agent_test.py:939-981 -> -0.0164646133
    loc, scale, val = dummy_nets(OBS)
    out = oppo.losses(loc, scale, torch.tensor([[0.0], [1.0]]), torch.tensor([0.9, 0.3]),
                      torch.tensor([1.9, 1.0]), torch.zeros(2), val, torch.ones(2),
                      clip_eps=10.0)
    np.testing.assert_allclose(float(out["policy_gradient_loss"]), -0.0164646133, rtol=1e-5)
    assert float(out["clip_fraction"]) == 0.0


def test_entropy_regularization_known_answer():  # ppo_agent_test.py:864-910 -> -3.70111 * 0.1
    loc, scale, val = dummy_nets(OBS)
    out = oppo.losses(loc, scale, torch.tensor([[0.0], [1.0]]), torch.tensor([0.9, 0.3]),
                      torch.tensor([1.9, 1.0]), torch.zeros(2), val, torch.ones(2), c_e=0.1)
    np.testing.assert_allclose(float(out["entropy_regularization_loss"]), -0.370111, rtol=1e-5)
    out0 = oppo.losses(loc, scale, torch.tensor([[0.0], [1.0]]), torch.tensor([0.9, 0.3]),
                       torch.tensor([1.9, 1.0]), torch.zeros(2), val, torch.ones(2), c_e=0.0)
    assert float(out0["entropy_regularization_loss"]) == 0.0


def test_epoch_loss_composition():  # ppo_agent_test.py:644-727 (masked half of the batch)
    obs = np.concatenate([OBS, OBS])
    loc, scale, val = dummy_nets(obs)
    w = torch.tensor([1.0, 1.0, 0.0, 0.0])
    out = oppo.losses(loc, scale, torch.tensor([[0.0], [1.0], [0.0], [1.0]]),
                      torch.tensor([0.9, 0.3, 0.9, 0.3]), torch.tensor([1.9, 1.0, 1.9, 1.0]),
                      torch.tensor([1.9, 1.0, 1.9, 1.0]), val, w, clip_eps=10.0, c_v=1.0, c_e=0.1,
                      old_loc=torch.tensor([[9.0], [15.0], [9.0], [15.0]]),
                      old_scale=torch.tensor([[8.0], [12.0], [8.0], [12.0]]), kl_beta=1.0,
                      kl_cutoff_coef=1000.0, kl_cutoff=2.0 * 0.01)
    np.testing.assert_allclose(float(out["policy_gradient_loss"]), -0.0164646133 * 2 / 4,
                               rtol=1e-5)
    np.testing.assert_allclose(float(out["value_estimation_loss"]), 123.205 * 2 / 4, rtol=1e-6)
    np.testing.assert_allclose(float(out["entropy_regularization_loss"]), -0.370111 * 2 / 4,
                               rtol=1e-5)
    np.testing.assert_allclose(float(out["kl_penalty_loss"]), 0.0, atol=1e-7)
    # l2: 1e-4 * ((2^2+1) + (2^2+1+1+1)) = 1e-4 * 12 (unmasked: not part of aggregate_losses)
    l2 = 1e-4 * (float((VALUE_K ** 2).sum()) + float((ACTOR_K ** 2).sum()))
    np.testing.assert_allclose(l2, 1e-4 * 12, rtol=1e-6)


def test_compute_advantages_no_gae_and_gae():  # ppo_agent_test.py:255-347
    rewards = np.ones((2, 9), np.float32)
    discounts = np.array([[1, 1, 1, 1, 0, .9, .9, .9, 0]] * 2, np.float32)
    returns = np.array([[5.0, 4.0, 3.0, 2.0, 1.0, 3.439, 2.71, 1.9, 1.0],
                        [3.0, 4.0, 7.0, 2.0, -1.0, 5.439, 2.71, -2.9, 1.0]], np.float32)
    vp = np.full((2, 10), 3.0, np.float32)
    adv = oppo.compute_advantages(rewards, returns, discounts, vp, False, 0.95)
    np.testing.assert_allclose(adv, returns - 3.0, rtol=1e-6)
    gae = oppo.compute_advantages(rewards, returns, discounts, vp, True, 0.95)
    truth = [2.0808625, 1.13775, 0.145, -0.9, -2.0, 0.56016475, -0.16355, -1.01, -2.0]
    np.testing.assert_allclose(gae, [truth, truth], rtol=1e-5)


def test_gae_bootstrap_quirk_is_reproduced():
    """compute_advantages bootstraps from V(s_{T-1}) (ppo_agent.py:465-469), so changing only the
    LAST value prediction must not change the GAE advantages."""
    rng = np.random.default_rng(0)
    r, d = rng.normal(size=(3, 6)).astype(np.float32), np.full((3, 6), 0.9, np.float32)
    vp = rng.normal(size=(3, 7)).astype(np.float32)
    a1 = oppo.compute_advantages(r, None, d, vp, True, 0.95)
    vp2 = vp.copy()
    vp2[:, -1] += 100.0
    a2 = oppo.compute_advantages(r, None, d, vp2, True, 0.95)
    np.testing.assert_array_equal(a1, a2)


def test_kl_cutoff_and_adaptive_beta():  # ppo_agent_test.py:1037-1075, 1126-1164
    kl = [[1.5, -0.5, 6.5, -1.5, -2.3]]
    np.testing.assert_allclose(oppo.kl_cutoff_loss(kl, 5.0, 0.1, 30.0), 30.0 * 0.24 ** 2,
                               rtol=1e-5)
    assert oppo.kl_cutoff_loss(kl, 5.0, 0.1, 0.0) == 0.0
    b0 = oppo.update_adaptive_kl_beta(1.0, 10.0, 10.0, 0.5)
    b1 = oppo.update_adaptive_kl_beta(b0, 100.0, 10.0, 0.5)
    b2 = oppo.update_adaptive_kl_beta(b1, 1.0, 10.0, 0.5)
    assert (b0, b1) == (1.0, 1.5)
    np.testing.assert_allclose(b2, 1.0, rtol=1e-6)


def test_trajectory_mask_and_padding():
    st = np.array([[0, 1, 2, 0, 1]], np.int32)
    ret = oppo.pad_last(np.array([[1.0, 0.0, 0.5, 2.0]], np.float32))
    adv = oppo.pad_last(np.array([[1.0, 0.0, 0.0, 1.0]], np.float32))
    # col 1: return == advantage == 0 -> invalid; col 2: boundary; col 4: padding
    np.testing.assert_array_equal(oppo.trajectory_mask(st, ret, adv), [[1, 0, 0, 1, 0]])


def test_return_and_advantage_episode_mask():
    """discounts are zeroed where next_step_type is LAST (ppo_agent.py:664-676)."""
    reward = np.ones((1, 4), np.float32)
    discount = np.ones((1, 4), np.float32)
    nst = np.array([[1, 2, 0, 1]], np.int32)
    vp = np.zeros((1, 4), np.float32)
    ret, adv = oppo.compute_return_and_advantage(reward, discount, nst, vp, gamma=1.0)
    # t=2: 1 ; t=1: next is LAST -> 1 ; t=0: 1 + 1 = 2
    np.testing.assert_allclose(ret, [[2.0, 1.0, 1.0]])
