"""Pins oracle/dqn.py to the reference's DQN known-answer tests
(tf_agents/agents/dqn/dqn_agent_test.py:74-81,178-218,220-267,269-299,301-355,416-481,483-561).
DummyNet = Dense(2) with kernel [[2,1],[1,1]], bias [1,1] (:38-69).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import dqn, nets

FIRST, MID, LAST = 0, 1, 2
LAYERS = [{"kind": "dense", "units": 2, "act": None}]


def params():
    return [torch.tensor([[2.0, 1.0], [1.0, 1.0]]), torch.tensor([1.0, 1.0])]


def agent(double_q=False, n_step=1, l2=0.0):
    return dqn.OracleDqnAgent(LAYERS, (2,), 2, params(), gamma=1.0, loss="huber",
                              double_q=double_q, n_step=n_step, l2=l2)


def stack(frames):
    """frames: list over time of dict(step_type, obs, action, next_step_type, reward, discount)."""
    f = lambda k, dt: np.stack([np.asarray(fr[k], dt) for fr in frames], axis=1)
    return (torch.tensor(f("obs", np.float32)), f("action", np.int32), f("reward", np.float32),
            f("discount", np.float32), f("step_type", np.int32))


def two_frame(next_obs):
    f0 = dict(step_type=[FIRST, FIRST], obs=[[1, 2], [3, 4]], action=[0, 1],
              reward=[10, 20], discount=[0.9, 0.9])
    f1 = dict(step_type=[MID, MID], obs=next_obs, action=[0, 1], reward=[10, 20],
              discount=[0.9, 0.9])
    return stack([f0, f1])


def test_compute_td_targets():  # :74-81
    out = dqn.td_loss_from_q(np.zeros((2, 1), np.float32), np.array([[10.], [20.]], np.float32),
                             np.zeros((2, 2), np.int32), np.array([[10, 0], [20, 0]], np.float32),
                             np.array([[.9, 1], [.9, 1]], np.float32),
                             np.zeros((2, 2), np.int32))
    np.testing.assert_allclose(out["td_targets"], [19.0, 38.0], rtol=1e-6)


@pytest.mark.parametrize("double_q", [False, True])
def test_loss(double_q):  # :178-218 -> 26.0
    total, _, aux = agent(double_q).loss(*two_frame([[5, 6], [7, 8]]))
    np.testing.assert_allclose(float(total), 26.0, rtol=1e-6)
    np.testing.assert_allclose(aux["td_loss"], [19.8, 32.2], rtol=1e-6)
    np.testing.assert_allclose(aux["td_error"], [20.3, 32.7], rtol=1e-6)


@pytest.mark.parametrize("double_q", [False, True])
def test_loss_changed_optimal_actions(double_q):  # :220-267 -> 9.8
    total, _, _ = agent(double_q).loss(*two_frame([[-5, 6], [-7, 8]]))
    np.testing.assert_allclose(float(total), 9.8, rtol=1e-6)


def test_loss_l2():  # :269-299 -> 33.0
    total, _, _ = agent(l2=1.0).loss(*two_frame([[5, 6], [7, 8]]))
    np.testing.assert_allclose(float(total), 33.0, rtol=1e-6)


def test_loss_n_step():  # :301-355 -> 47.42
    f0 = dict(step_type=[FIRST] * 2, obs=[[1, 2], [3, 4]], action=[0, 1], reward=[10, 20],
              discount=[.9, .9])
    f1 = dict(step_type=[MID] * 2, obs=[[5, 6], [7, 8]], action=[0, 1], reward=[10, 20],
              discount=[.9, .9])
    f2 = dict(step_type=[MID] * 2, obs=[[9, 10], [11, 12]], action=[0, 1], reward=[10, 20],
              discount=[.9, .9])
    total, _, _ = agent(n_step=2).loss(*stack([f0, f1, f2]))
    np.testing.assert_allclose(float(total), 47.42, rtol=1e-6)


def test_loss_n_step_mid_mid_last_first():  # :416-481 -> 21.5
    f0 = dict(step_type=[MID] * 2, obs=[[1, 2], [3, 4]], action=[0, 1], reward=[10, 20],
              discount=[.9, .9])
    f1 = dict(step_type=[MID] * 2, obs=[[5, 6], [7, 8]], action=[0, 1], reward=[10, 20],
              discount=[0.0, 0.0])           # next step is LAST: termination -> discount 0
    f2 = dict(step_type=[LAST] * 2, obs=[[9, 10], [11, 12]], action=[0, 1], reward=[0, 0],
              discount=[1.0, 1.0])           # boundary row -> restart: reward 0, discount 1
    f3 = dict(step_type=[FIRST] * 2, obs=[[13, 14], [15, 16]], action=[0, 1], reward=[0, 0],
              discount=[1.0, 1.0])
    total, _, _ = agent(n_step=3).loss(*stack([f0, f1, f2, f3]))
    np.testing.assert_allclose(float(total), 21.5, rtol=1e-6)


def test_loss_masked_actions():  # :483-561 -> 23.75 (DQN)
    obs, act, rew, disc, st = two_frame([[5, 6], [7, 8]])
    a = agent()
    q = a.q_values(obs[:, 0]).detach().numpy()
    qt = a.q_values(obs[:, -1], target=True).detach().numpy()
    out = dqn.td_loss_from_q(q, qt, act, rew, disc, st,
                             next_mask=np.array([[0, 1], [1, 0]], np.int32))
    np.testing.assert_allclose(out["loss"], 23.75, rtol=1e-6)


def test_last_steps_are_masked():  # dqn_agent.py:514-517
    obs, act, rew, disc, st = two_frame([[5, 6], [7, 8]])
    st = st.copy()
    st[1, 0] = LAST
    total, _, aux = agent().loss(obs, act, rew, disc, st)
    np.testing.assert_allclose(aux["td_loss"], [19.8, 0.0], rtol=1e-6)
    np.testing.assert_allclose(float(total), 19.8 / 2, rtol=1e-6)  # mean over the FULL batch


def test_dq_matches_autograd():
    rng = np.random.RandomState(0)
    B, A = 16, 4
    q = torch.tensor(rng.randn(B, A).astype(np.float32), requires_grad=True)
    qt = rng.randn(B, A).astype(np.float32)
    act = rng.randint(0, A, size=(B, 2)).astype(np.int64)
    rew = rng.randn(B, 2).astype(np.float32)
    disc = (rng.rand(B, 2) > 0.2).astype(np.float32)
    st = rng.randint(0, 3, size=(B, 2)).astype(np.int32)
    w = rng.rand(B).astype(np.float32)
    for kind in ("huber", "squared"):
        out = dqn.td_loss_from_q(q.detach().numpy(), qt, act, rew, disc, st, gamma=0.99,
                                 weights=w, loss=kind)
        tgt = torch.tensor(out["td_targets"])
        qa = q[torch.arange(B), torch.tensor(act[:, 0])]
        if kind == "huber":
            l = torch.nn.functional.huber_loss(qa, tgt, reduction="none", delta=1.0)
        else:
            l = (tgt - qa) ** 2
        valid = torch.tensor((st[:, 0] != LAST).astype(np.float32))
        loss = (l * valid * torch.tensor(w)).sum() / B
        g, = torch.autograd.grad(loss, q)
        np.testing.assert_allclose(out["dq"], g.numpy(), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(out["loss"], float(loss), rtol=1e-5)
