"""CPU: the SAC oracle reproduces the reference's known answers
(tf_agents/agents/sac/sac_agent_test.py:269-396) with the test's mocks restated:
DummyCriticNet  q(obs, a) = obs[:, 1] + a  (value kernel [[0],[1]], action kernel [[1]], :84-135)
DummyActorPolicy  action = spec maximum (1.0), log_prob = 10.0 (:40-82)."""
import numpy as np
import torch

from oracle import sac as osac


def dummy_q(obs, act):
    return (obs[:, 1:2] + act).reshape(-1)


def dummy_policy(obs):
    B = obs.shape[0]
    return torch.ones(B, 1), torch.full((B,), 10.0)


def test_critic_loss_known_answer():
    # testCriticLoss: td_targets [7.3, 19.1], predictions [7, 10], loss = 2 * MSE
    obs = torch.tensor([[1., 2.], [3., 4.]])
    actions = torch.tensor([[5.], [6.]])
    reward = torch.tensor([10., 20.])
    discount = torch.tensor([0.9, 0.9])
    next_obs = torch.tensor([[5., 6.], [7., 8.]])
    loss, td = osac.critic_loss(dummy_q, dummy_q, dummy_q, dummy_q, dummy_policy(next_obs),
                                torch.tensor(0.0), obs, actions, next_obs, reward, discount)
    np.testing.assert_allclose(td.numpy(), [7.3, 19.1], rtol=1e-6)
    expected = 2 * np.mean((np.array([7.3, 19.1]) - np.array([7.0, 10.0])) ** 2)
    np.testing.assert_allclose(float(loss), expected, rtol=1e-6)


def test_critic_reg_loss_known_answer():
    # testCriticRegLoss: all-zero inputs, l2 weight 0.5 on kernels [[0],[1]] and [[1]] of both
    # critics -> regularisation only: 2 critics * 0.5 * (0 + 1 + 1) = 2.0
    z = torch.zeros(2, 2)
    reg = 2 * 0.5 * (0.0 ** 2 + 1.0 ** 2 + 1.0 ** 2)
    zero_q = lambda o, a: torch.zeros(o.shape[0])
    loss, _ = osac.critic_loss(zero_q, zero_q, zero_q, zero_q,
                               (torch.zeros(2, 1), torch.zeros(2)), torch.tensor(0.0), z,
                               torch.zeros(2, 1), z, torch.zeros(2), torch.zeros(2),
                               regularization=reg)
    np.testing.assert_allclose(float(loss), 2.0)


def test_actor_loss_known_answer():
    # testActorLoss: (2*10 - (2+1) - (4+1)) / 2 = 6
    obs = torch.tensor([[1., 2.], [3., 4.]])
    loss = osac.actor_loss(dummy_q, dummy_q, dummy_policy(obs), torch.tensor(0.0), obs)
    np.testing.assert_allclose(float(loss), (2 * 10 - (2 + 1) - (4 + 1)) / 2)


def test_alpha_loss_known_answer():
    # testAlphaLoss: initial_log_alpha 4, target_entropy 3, log_pi 10 -> 4 * (-10 - 3) = -52
    _, logp = dummy_policy(torch.zeros(2, 2))
    loss = osac.alpha_loss(logp, torch.tensor(4.0), 3.0)
    np.testing.assert_allclose(float(loss), 4.0 * (-10 - 3))


def test_tanh_normal_matches_torch_distributions():
    """log_pi of the squashed sample == TransformedDistribution(Normal, [Tanh, Affine]).log_prob
    evaluated with torch.distributions (an independent implementation of the same density)."""
    import torch.distributions as D
    g = torch.Generator().manual_seed(0)
    B, A = 64, 3
    z = torch.randn(B, 2 * A, generator=g) * 0.5
    eps = torch.randn(B, A, generator=g)
    mean = torch.tensor([0.5, -1.0, 0.0])
    mag = torch.tensor([1.5, 2.0, 1.0])
    action, logp = osac.tanh_normal(z, eps, mean, mag, "clip_exp")
    base = D.Independent(D.Normal(z[:, :A], torch.exp(torch.clamp(z[:, A:], -20, 2))), 1)
    td = D.TransformedDistribution(base, [D.TanhTransform(cache_size=1),
                                          D.AffineTransform(mean, mag, event_dim=1)])
    x = z[:, :A] + torch.exp(torch.clamp(z[:, A:], -20, 2)) * eps
    y = mean + mag * torch.tanh(x)
    np.testing.assert_allclose(action.numpy(), y.numpy(), rtol=1e-6, atol=1e-6)
    ref = base.log_prob(x) - (torch.log(mag) + 2 * (np.log(2.0) - x
                                                    - torch.nn.functional.softplus(-2 * x))).sum(-1)
    np.testing.assert_allclose(logp.numpy(), ref.numpy(), rtol=2e-5, atol=2e-5)
    # and through torch's own transformed distribution at interior points
    inner = (x.abs() < 2.5).all(-1)
    np.testing.assert_allclose(logp[inner].numpy(), td.log_prob(y)[inner].numpy(), rtol=2e-3,
                               atol=2e-3)


def test_oracle_train_step_runs_and_updates():
    g = torch.Generator().manual_seed(1)
    obs_dim, A, B = 5, 2, 8
    al = osac.nets.mlp_q_layers((16,), 2 * A)
    cl = osac.nets.mlp_q_layers((16,), 1)
    ap = osac.nets.init_params(al, (obs_dim,), seed=1)
    c1 = osac.nets.init_params(cl, (obs_dim + A,), seed=2)
    c2 = osac.nets.init_params(cl, (obs_dim + A,), seed=3)
    ag = osac.OracleSacAgent(obs_dim, A, (16,), (16,), [0.0, 0.0], [1.0, 2.0], ap, c1, c2)
    r = lambda *s: torch.randn(*s, generator=g)
    before = [p.detach().clone() for p in ag.actor + ag.c1 + ag.c2]
    out = ag.train(r(B, obs_dim), torch.tanh(r(B, A)), r(B, obs_dim), r(B), torch.ones(B),
                   r(B, A), r(B, A), r(B, A))
    assert np.isfinite(out["loss"])
    after = ag.actor + ag.c1 + ag.c2
    assert all(not torch.equal(a, b) for a, b in zip(before, after))
    assert float(ag.log_alpha) != 0.0
    # soft update moved the targets by tau towards the critics
    assert not torch.equal(ag.t1[0], c1[0])


def test_critic_towers_follow_the_reference_call():
    """OracleSacAgent.q with CriticNetwork's per-input towers (agents/ddpg/critic_network.py:
    126-145, call :163-185): each input through its Dense(relu) stack, concatenation, joint stack --
    against the formula written out in numpy; parameter order observation, action, joint."""
    rng = np.random.default_rng(4)
    OD, A, B = 5, 2, 6
    shapes = [(OD, 7), (7,), (A, 3), (3,), (7 + 3, 8), (8,), (8, 1), (1,)]
    params = [torch.from_numpy(rng.standard_normal(sh).astype(np.float32)) for sh in shapes]
    al = osac.nets.mlp_q_layers((4,), 2 * A)
    ap = osac.nets.init_params(al, (OD,), seed=1)
    ag = osac.OracleSacAgent(OD, A, (4,), (8,), [0.0, 0.0], [1.0, 1.0], ap, params, params,
                             critic_obs_fc=(7,), critic_act_fc=(3,))
    obs = rng.standard_normal((B, OD)).astype(np.float32)
    act = rng.standard_normal((B, A)).astype(np.float32)
    w = [p.numpy().astype(np.float64) for p in params]
    relu = lambda v: np.maximum(v, 0.0)
    o = relu(obs @ w[0] + w[1])
    a = relu(act @ w[2] + w[3])
    j = relu(np.concatenate([o, a], -1) @ w[4] + w[5])
    want = (j @ w[6] + w[7]).reshape(-1)
    got = ag.q(ag.c1, torch.from_numpy(obs), torch.from_numpy(act)).detach().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
    # the tower-less layout is untouched: concat(obs, act) straight into the joint stack
    cl = osac.nets.mlp_q_layers((8,), 1)
    c = osac.nets.init_params(cl, (OD + A,), seed=2)
    plain = osac.OracleSacAgent(OD, A, (4,), (8,), [0.0, 0.0], [1.0, 1.0], ap, c, c)
    x = np.concatenate([obs, act], -1).astype(np.float64)
    wc = [p.numpy().astype(np.float64) for p in c]
    want2 = (relu(x @ wc[0] + wc[1]) @ wc[2] + wc[3]).reshape(-1)
    got2 = plain.q(plain.c1, torch.from_numpy(obs), torch.from_numpy(act)).detach().numpy()
    np.testing.assert_allclose(got2, want2, rtol=1e-5, atol=1e-6)


def test_golden_file_lists_the_same_numbers():
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden",
                                    "reference_known_answers.json")))["sac"]
    td = np.array(g["critic_loss"]["td_targets"])
    pred = np.array(g["critic_loss"]["pred_td_targets"])
    np.testing.assert_allclose(2 * np.mean((td - pred) ** 2), g["critic_loss"]["loss_value"],
                               rtol=1e-9)
    assert g["actor_loss"]["loss_value"] == 6.0 and g["alpha_loss"]["loss_value"] == -52.0
