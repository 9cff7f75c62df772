"""GPU: SAC kernels and SacAgent against the torch-CPU oracle (oracle/sac.py, itself pinned on the
reference's known answers in tests/test_oracle_sac.py) and directly against the reference's numbers
(tf_agents/agents/sac/sac_agent_test.py:269-396).  Tolerances: 1e-5 relative on losses (north
star), 2e-5 on gradients / updated parameters (fp32 summation order of the MFMA GEMMs)."""
import ctypes

import numpy as np
import pytest
import torch

from agents_amd import _lib, optimizers
from agents_amd.agents.sac import sac_agent
from agents_amd.drivers import dynamic_step_driver
from agents_amd.environments import random_tf_environment
from agents_amd.networks import actor_distribution_network as adn
from agents_amd.networks import critic_network
from agents_amd.networks import layers as L
from agents_amd.replay_buffers import tf_uniform_replay_buffer as rb_lib
from agents_amd.specs import tensor_spec
from agents_amd.trajectories import time_step as ts
from agents_amd.trajectories import trajectory
from agents_amd.utils import common
from oracle import nets as onets
from oracle import sac as osac

pytestmark = pytest.mark.gpu


def close(got, ref, rtol=1e-5, atol=1e-6):
    got = got.detach().cpu().double().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    ref = ref.detach().cpu().double().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(got - ref).max()
    assert err <= rtol * scale + atol, f"max err {err:.3e} vs scale {scale:.3e}"


def f32(dev, x):
    return torch.as_tensor(np.asarray(x, np.float32)).to(dev)


# ---- the reference's known answers, through the loss kernels --------------------------------------
def test_critic_loss_kernel_reference_numbers(dev):
    """testCriticLoss: q = obs[:,1] + a, next action 1, log_pi 10 -> td [7.3, 19.1], loss 2*MSE."""
    lib = _lib.load()
    q = f32(dev, [2 + 5, 4 + 6])
    tq = f32(dev, [6 + 1, 8 + 1])
    nlogp = f32(dev, [10, 10])
    out, td = torch.zeros(1, device=dev), torch.zeros(2, device=dev)
    dq1, dq2 = torch.zeros(2, device=dev), torch.zeros(2, device=dev)
    la = torch.zeros(4, device=dev)
    rew, disc = f32(dev, [10, 20]), f32(dev, [0.9, 0.9])   # (kept alive across the launch)
    _lib.check(lib.aa_sac_critic_loss(q.data_ptr(), q.data_ptr(), tq.data_ptr(), tq.data_ptr(),
                                      nlogp.data_ptr(), rew.data_ptr(),
                                      disc.data_ptr(), None, la.data_ptr(), 1.0,
                                      1.0, _lib.AA_LOSS_SQUARED, 1.0, 2, 2.0, out.data_ptr(),
                                      td.data_ptr(), dq1.data_ptr(), dq2.data_ptr(),
                                      _lib.stream_ptr()), "critic")
    close(td, [7.3, 19.1])
    close(out, [2 * np.mean((np.array([7.3, 19.1]) - np.array([7.0, 10.0])) ** 2)])
    close(dq1, [-2 * (7.3 - 7.0) / 2, -2 * (19.1 - 10.0) / 2])


def test_actor_and_alpha_loss_kernel_reference_numbers(dev):
    lib = _lib.load()
    logp = f32(dev, [10, 10])
    q = f32(dev, [2 + 1, 4 + 1])
    la = torch.zeros(4, device=dev)
    out = torch.zeros(1, device=dev)
    _lib.check(lib.aa_sac_actor_loss(q.data_ptr(), q.data_ptr(), logp.data_ptr(), None,
                                     la.data_ptr(), 1.0, 2, 2.0, out.data_ptr(), None, None, None,
                                     _lib.stream_ptr()), "actor")
    close(out, [(2 * 10 - (2 + 1) - (4 + 1)) / 2])            # testActorLoss: 6.0
    la[0] = 4.0
    g = torch.zeros(4, device=dev)
    _lib.check(lib.aa_sac_alpha_loss(logp.data_ptr(), None, la.data_ptr(), 3.0, 1, 1.0, 2, 2.0,
                                     out.data_ptr(), g.data_ptr(), _lib.stream_ptr()), "alpha")
    close(out, [4.0 * (-10 - 3)])                              # testAlphaLoss: -52
    close(g[:1], [-13.0])


# ---- actor head ------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["exp", "clip_exp"])
@pytest.mark.parametrize("B,A", [(1, 1), (37, 3), (1024, 17)])
def test_sample_and_head_backward_vs_autograd(dev, kind, B, A):
    lib = _lib.load()
    g = torch.Generator().manual_seed(B + A)
    z = torch.randn(B, 2 * A, generator=g)
    z[:, A:] *= 1.5
    if kind == "clip_exp" and B > 1:
        z[0, A] = 3.0         # outside the clip range: zero gradient to the raw std
        z[1, A] = -25.0
    eps = torch.randn(B, A, generator=g)
    mean = torch.linspace(-0.5, 0.5, A)
    mag = torch.linspace(0.5, 2.0, A)
    zr = z.clone().requires_grad_(True)
    act_o, logp_o = osac.tanh_normal(zr, eps, mean, mag, kind)
    da = torch.randn(B, A, generator=g)
    dl = torch.randn(B, generator=g)
    ((act_o * da).sum() + (logp_o * dl).sum()).backward()
    zd, epsd, meand, magd, dad, dld = (t.to(dev) for t in (z, eps, mean, mag, da, dl))
    action = torch.empty(B, A, device=dev)
    logp = torch.empty(B, device=dev)
    sv = [torch.empty(B, A, device=dev) for _ in range(3)]
    k = _lib.AA_SAC_STD_EXP if kind == "exp" else _lib.AA_SAC_STD_CLIP_EXP
    _lib.check(lib.aa_sac_sample(zd.data_ptr(), B, A, meand.data_ptr(),
                                 magd.data_ptr(), k, epsd.data_ptr(), 0, None, None,
                                 action.data_ptr(), logp.data_ptr(), sv[0].data_ptr(),
                                 sv[1].data_ptr(), sv[2].data_ptr(), _lib.stream_ptr()), "sample")
    close(action, act_o, rtol=2e-6)
    close(logp, logp_o, rtol=1e-5, atol=1e-5)
    dz = torch.empty(B, 2 * A, device=dev)
    _lib.check(lib.aa_sac_head_backward(zd.data_ptr(), B, A, magd.data_ptr(), k,
                                        sv[0].data_ptr(), sv[1].data_ptr(), sv[2].data_ptr(),
                                        dad.data_ptr(), A, None, 0, dld.data_ptr(),
                                        dz.data_ptr(), _lib.stream_ptr()), "head bwd")
    close(dz, zr.grad, rtol=2e-5, atol=1e-5)
    # d loss / d action as the sum of two strided tensors (the twin critics' input gradients):
    # the same bits as adding them first
    g = torch.Generator().manual_seed(9)
    part = torch.randn(B, A + 3, generator=g).to(dev)
    rest = torch.zeros(B, A + 5, device=dev)
    rest[:, :A] = dad - part[:, :A]
    summed = (part[:, :A] + rest[:, :A]).contiguous()
    dz_a, dz_b = torch.empty(B, 2 * A, device=dev), torch.empty(B, 2 * A, device=dev)
    for da1, ld1, da2, ld2, out in ((summed, A, None, 0, dz_a), (part, A + 3, rest, A + 5, dz_b)):
        _lib.check(lib.aa_sac_head_backward(
            zd.data_ptr(), B, A, magd.data_ptr(), k, sv[0].data_ptr(), sv[1].data_ptr(),
            sv[2].data_ptr(), da1.data_ptr(), ld1, None if da2 is None else da2.data_ptr(), ld2,
            dld.data_ptr(), out.data_ptr(), _lib.stream_ptr()), "head bwd")
    assert torch.equal(dz_a, dz_b)


def test_sample_internal_noise_is_standard_normal_and_advances(dev):
    lib = _lib.load()
    B, A = 20000, 4
    z = torch.zeros(B, 2 * A, device=dev)          # mean 0, sigma 1
    mean, mag = torch.zeros(A, device=dev), torch.ones(A, device=dev)
    ctr = torch.zeros(1, dtype=torch.int64, device=dev)
    outs = []
    for call in range(2):
        a = torch.empty(B, A, device=dev)
        lp = torch.empty(B, device=dev)
        sv = [torch.empty(B, A, device=dev) for _ in range(3)]
        _lib.check(lib.aa_sac_sample(z.data_ptr(), B, A, mean.data_ptr(), mag.data_ptr(), 0, None,
                                     1234, ctr.data_ptr(), None, a.data_ptr(), lp.data_ptr(),
                                     sv[0].data_ptr(), sv[1].data_ptr(), sv[2].data_ptr(),
                                     _lib.stream_ptr()), "sample")
        ctr += 1
        outs.append(sv[2].cpu())
        assert float(a.abs().max()) < 1.0           # tanh-squashed into the spec
    e = outs[0].double()
    assert abs(float(e.mean())) < 0.02 and abs(float(e.std()) - 1.0) < 0.02
    assert not torch.equal(outs[0], outs[1])        # the call counter moves the stream


# ---- agent vs oracle -------------------------------------------------------------------------------
OBS_DIM, A = 11, 3
OBS = tensor_spec.BoundedTensorSpec((OBS_DIM,), torch.float32, -1.0, 1.0)
ACT = tensor_spec.BoundedTensorSpec((A,), torch.float32, [-1.0, -2.0, 0.0], [1.0, 2.0, 4.0])
TSS = ts.time_step_spec(OBS)


def make_pair(dev, actor_fc=(32, 32), critic_fc=(32, 32), kind="clip_exp", clip=None, **kw):
    actor = adn.ActorDistributionNetwork(
        OBS, ACT, fc_layer_params=actor_fc,
        continuous_projection_net=lambda spec: adn.TanhNormalProjectionNetwork(
            spec, std_transform=kind), seed=1)
    critic = critic_network.CriticNetwork((OBS, ACT), joint_fc_layer_params=critic_fc,
                                          kernel_initializer=L.GlorotUniform(),
                                          last_kernel_initializer=L.GlorotUniform(), seed=2)
    agent = sac_agent.SacAgent(
        TSS, ACT, critic_network=critic, actor_network=actor,
        actor_optimizer=optimizers.Adam(3e-3), critic_optimizer=optimizers.Adam(3e-3),
        alpha_optimizer=optimizers.Adam(3e-3), target_update_tau=0.05, target_update_period=1,
        td_errors_loss_fn=common.element_wise_squared_loss, gamma=0.99, reward_scale_factor=0.5,
        gradient_clipping=clip, **kw)
    agent.initialize()
    ap = [torch.from_numpy(w.copy()) for w in actor.get_weights()]
    c1 = [torch.from_numpy(w.copy()) for w in agent.critic_networks[0].get_weights()]
    c2 = [torch.from_numpy(w.copy()) for w in agent.critic_networks[1].get_weights()]
    mean, mag = sac_agent._spec_means_and_magnitudes(ACT)
    oracle = osac.OracleSacAgent(
        OBS_DIM, A, actor_fc, critic_fc, mean, mag, ap, c1, c2, actor_lr=3e-3, critic_lr=3e-3,
        alpha_lr=3e-3, gamma=0.99, reward_scale_factor=0.5, tau=0.05, std_kind=kind,
        initial_log_alpha=kw.get("initial_log_alpha", 0.0),
        target_entropy=kw.get("target_entropy"),
        critic_loss_weight=kw.get("critic_loss_weight", 0.5),
        use_log_alpha_in_alpha_loss=kw.get("use_log_alpha_in_alpha_loss", True))
    return agent, oracle


def batch(dev, B, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    obs = torch.tanh(r(B, 2, OBS_DIM))
    mean, mag = sac_agent._spec_means_and_magnitudes(ACT)
    act = torch.from_numpy(mean) + torch.from_numpy(mag) * torch.tanh(r(B, 2, A))
    reward = r(B, 2)
    discount = (torch.rand(B, 2, generator=g) > 0.1).float()
    exp = trajectory.Trajectory(
        step_type=torch.ones(B, 2, dtype=torch.int32), observation=obs, action=act,
        policy_info=(), next_step_type=torch.ones(B, 2, dtype=torch.int32), reward=reward,
        discount=discount)
    eps = {k: r(B, A) for k in ("next", "actor", "alpha")}
    from agents_amd.utils import nest_utils
    return (nest_utils.map_structure(lambda t: t.to(dev), exp), exp,
            {k: v.to(dev) for k, v in eps.items()}, eps)


@pytest.mark.parametrize("B", [8, 256])
@pytest.mark.parametrize("cfg", [dict(), dict(initial_log_alpha=0.7, target_entropy=-1.0,
                                              use_log_alpha_in_alpha_loss=False,
                                              critic_loss_weight=1.0)])
def test_train_steps_match_oracle(dev, B, cfg):
    agent, oracle = make_pair(dev, **cfg)
    for step in range(3):
        exp_d, exp_h, eps_d, eps_h = batch(dev, B, 100 + step)
        li = agent.train(exp_d, eps=eps_d)
        out = oracle.train(exp_h.observation[:, 0], exp_h.action[:, 0], exp_h.observation[:, 1],
                           exp_h.reward[:, 0], exp_h.discount[:, 0], eps_h["next"], eps_h["actor"],
                           eps_h["alpha"])
        close(li.extra.critic_loss, out["critic_loss"], rtol=1e-5)
        close(li.extra.actor_loss, out["actor_loss"], rtol=1e-5, atol=1e-5)
        close(li.extra.alpha_loss, out["alpha_loss"], rtol=1e-5, atol=1e-6)
        close(li.loss, out["loss"], rtol=1e-5, atol=1e-5)
        if step == 0:   # gradients of the first step (same weights on both sides)
            got = agent.actor_network.body.gradients
            for gd, go in zip(got, out["actor_grads"]):
                close(gd, go, rtol=5e-5, atol=1e-6)
            close(agent._log_alpha_grad[:1], [out["alpha_grad"]], rtol=1e-5)
    for v, o in zip(agent.actor_network.variables, oracle.actor):
        close(v, o, rtol=2e-4, atol=2e-5)
    for net, o in zip(agent.critic_networks, (oracle.c1, oracle.c2)):
        for v, ov in zip(net.variables, o):
            close(v, ov, rtol=2e-4, atol=2e-5)
    for net, o in zip(agent.target_critic_networks, (oracle.t1, oracle.t2)):
        for v, ov in zip(net.variables, o):
            close(v, ov, rtol=2e-4, atol=2e-5)
    close(agent.log_alpha.reshape(1), [float(oracle.log_alpha)], rtol=1e-4, atol=1e-6)
    assert int(agent.train_step_counter.numpy()) == 3


def test_loss_values_and_weights(dev):
    agent, oracle = make_pair(dev)
    exp_d, exp_h, _, _ = batch(dev, 32, 7)
    li = agent.loss(exp_d)
    assert torch.isfinite(li.loss).item()
    assert li.extra.critic_loss.shape == () and li.extra.alpha_loss.shape == ()
    # zero weights switch every term off (aggregate_losses with sample_weight = 0)
    li0 = agent.loss(exp_d, weights=torch.zeros(32, device=dev))
    assert float(li0.loss) == 0.0


def test_gradient_clipping_bounds_update(dev):
    agent, _ = make_pair(dev, clip=1e-3)
    exp_d, _, eps_d, _ = batch(dev, 64, 3)
    agent.train(exp_d, eps=eps_d)
    for net in agent.critic_networks + (agent.actor_network,):
        for gvar in net.body.gradients:
            assert float(gvar.norm()) <= 1e-3 * (1 + 1e-4)


def test_collect_train_loop(dev):
    """DynamicStepDriver + TFUniformReplayBuffer + SacAgent.train end to end (examples/sac/
    haarnoja18/sac_train_eval.py structure), actions inside the spec bounds."""
    agent, _ = make_pair(dev)
    B = 16
    env = random_tf_environment.RandomTFEnvironment(TSS, ACT, batch_size=B,
                                                    episode_end_probability=0.05, seed=1,
                                                    device=dev)
    rb = rb_lib.TFUniformReplayBuffer(agent.collect_data_spec, batch_size=B, max_length=64,
                                      device=dev)
    drv = dynamic_step_driver.DynamicStepDriver(env, agent.collect_policy,
                                                observers=[rb.add_batch], num_steps=B * 8)
    drv.run()
    data = rb.gather_all()
    lo = torch.tensor([-1.0, -2.0, 0.0], device=dev)
    hi = torch.tensor([1.0, 2.0, 4.0], device=dev)
    assert bool(((data.action >= lo) & (data.action <= hi)).all())
    it = iter(rb.as_dataset(sample_batch_size=32, num_steps=2))
    before = agent.actor_network.flat_params.clone()
    losses = []
    for _ in range(5):
        exp, _ = next(it)
        losses.append(float(agent.train(exp).loss))
    assert all(np.isfinite(losses)) and not torch.equal(before, agent.actor_network.flat_params)
    assert int(agent.train_step_counter.numpy()) == 5


def test_discrete_action_spec_is_rejected(dev):
    disc = tensor_spec.BoundedTensorSpec((), torch.int64, 0, 3)
    critic = critic_network.CriticNetwork((OBS, ACT), joint_fc_layer_params=(8,))
    with pytest.raises(NotImplementedError, match="does not currently support discrete actions"):
        sac_agent.SacAgent(TSS, disc, critic_network=critic, actor_network=None,
                           actor_optimizer=None, critic_optimizer=None, alpha_optimizer=None)


@pytest.mark.parametrize("b_eager", [True, False])
def test_sac_train_graph_matches_eager(dev, b_eager, monkeypatch):
    """SacAgent.train through the Learner (one HIP graph per sampler ring slot; part (b) -- the
    actor's optimizer launch, the alpha update -- issued directly behind part (a)'s graph, or as
    a graph of its own) == eager train: same parameters, log_alpha, targets and counters after 30
    steps on the same replay stream."""
    from agents_amd.train import learner
    from agents_amd.utils import graph
    monkeypatch.setattr(graph, "WHOLE_B_EAGER", b_eager)
    stacks = []
    for _ in range(2):
        agent, _ = make_pair(dev)
        env = random_tf_environment.RandomTFEnvironment(TSS, ACT, batch_size=8,
                                                        episode_end_probability=0.05, seed=1,
                                                        device=dev)
        rb = rb_lib.TFUniformReplayBuffer(agent.collect_data_spec, batch_size=8, max_length=64,
                                          device=dev, seed=5)
        dynamic_step_driver.DynamicStepDriver(env, agent.collect_policy, observers=[rb.add_batch],
                                              num_steps=8 * 16).run()
        stacks.append((agent, rb))
    (ag_e, rb_e), (ag_g, rb_g) = stacks
    lrn = learner.Learner(None, common.Variable(0), ag_g)
    it_g = iter(rb_g.as_dataset(sample_batch_size=32, num_steps=2))
    for i in range(30):
        exp_e, _ = rb_e.get_next(32, 2)
        li_e = ag_e.train(exp_e)
        prev = None if i == 0 else (li_g, [float(li_g.loss)] + [float(x) for x in li_g.extra])
        li_g = lrn.run(iterations=1, iterator=it_g)
        assert float(li_e.loss) == float(li_g.loss), f"step {i}"
        for x, y in zip(li_e.extra, li_g.extra):
            assert float(x) == float(y), f"step {i}"
        if prev is not None:       # what Learner.run returned owns its storage: the next step left it alone
            assert [float(prev[0].loss)] + [float(x) for x in prev[0].extra] == prev[1]
    assert graph.graphed_train(ag_g).replays > 15
    assert torch.equal(ag_e.actor_network.flat_params, ag_g.actor_network.flat_params)
    assert torch.equal(ag_e._critic_params, ag_g._critic_params)
    assert torch.equal(ag_e._target_params, ag_g._target_params)
    assert torch.equal(ag_e._log_alpha_buf, ag_g._log_alpha_buf)
    assert int(ag_e.train_step_counter.numpy()) == int(ag_g.train_step_counter.numpy()) == 30
    assert ag_e._actor_optimizer.iterations == ag_g._actor_optimizer.iterations == 30


@pytest.mark.parametrize("cfg", [dict(), dict(initial_log_alpha=0.7, target_entropy=-1.0,
                                              use_log_alpha_in_alpha_loss=False)])
def test_fused_alpha_tail_is_bit_identical(dev, cfg):
    """aa_sac_alpha_step (alpha loss + Adam step on log_alpha + LossInfo pack in one launch) against
    the three launches it replaces (an identity gradient hook sends the agent down that path):
    every parameter, optimizer slot, log_alpha and loss after five train steps, bit for bit."""
    a_fused, _ = make_pair(dev, **cfg)
    a_three, _ = make_pair(dev, **cfg)
    a_three.gradient_hook = lambda g: g
    for step in range(5):
        exp_d, _, eps_d, _ = batch(dev, 64, 300 + step)
        li_f = a_fused.train(exp_d, eps=eps_d)
        li_t = a_three.train(exp_d, eps=eps_d)
        for x, y in zip([li_f.loss] + list(li_f.extra), [li_t.loss] + list(li_t.extra)):
            assert torch.equal(x, y)
    for x, y in zip(a_fused.replicated_state(), a_three.replicated_state()):
        assert torch.equal(x, y)
    assert a_fused._alpha_optimizer.iterations == a_three._alpha_optimizer.iterations == 5


@pytest.mark.parametrize("B", [7, 256])
@pytest.mark.parametrize("kind", ["exp", "clip_exp"])
def test_fused_forward_sample_is_bit_identical(dev, B, kind, monkeypatch):
    """aa_mlp_wide_forward_sample (the actor's forward launch also draws the tanh-squashed action
    and its log-probability) against forward + aa_sac_sample: head output, actions, log pi and the
    tensors kept for the backward pass, with supplied noise and with the policy's own Philox stream
    (whose call counter must move the same way), bit for bit; and whole train steps either way."""
    def policy():
        actor = adn.ActorDistributionNetwork(
            OBS, ACT, fc_layer_params=(128, 96),
            continuous_projection_net=lambda spec: adn.TanhNormalProjectionNetwork(
                spec, std_transform=kind), seed=4)
        actor.create_variables(device=dev)
        return sac_agent.SacPolicy(TSS, ACT, actor, seed=99)
    pf, ps = policy(), policy()
    assert pf._actor_network.forward_sample_ok(torch.zeros(B, OBS_DIM, device=dev))
    g = torch.Generator().manual_seed(8)
    obs = torch.tanh(torch.randn(B, OBS_DIM, generator=g)).to(dev)
    eps = torch.randn(B, A, generator=g).to(dev)
    mk = lambda: {k: torch.empty(B, A, device=dev) for k in ("tanh", "sigma", "eps")}
    for use_eps in (True, False, False):
        sv_f, sv_s = mk(), mk()
        monkeypatch.setattr(sac_agent, "_FUSE_SAMPLE", True)
        a_f, lp_f, z_f = pf.sample(obs, slot="t", need_grad=True, eps=eps if use_eps else None,
                                   save=sv_f)
        monkeypatch.setattr(sac_agent, "_FUSE_SAMPLE", False)
        a_s, lp_s, z_s = ps.sample(obs, slot="t", need_grad=True, eps=eps if use_eps else None,
                                   save=sv_s)
        torch.cuda.synchronize()
        assert torch.equal(z_f, z_s) and torch.equal(a_f, a_s) and torch.equal(lp_f, lp_s)
        for k in sv_f:
            assert torch.equal(sv_f[k], sv_s[k]), k
        assert int(pf._call_counter.item()) == int(ps._call_counter.item())
    assert int(pf._call_counter.item()) == 2          # two draws of the policy's own noise
    # the agent's train step (critic / actor / alpha phases each sample once) either way
    monkeypatch.setattr(sac_agent, "_FUSE_SAMPLE", True)
    ag_f, _ = make_pair(dev, actor_fc=(128, 96), critic_fc=(128, 128), kind=kind)
    monkeypatch.setattr(sac_agent, "_FUSE_SAMPLE", False)
    ag_s, _ = make_pair(dev, actor_fc=(128, 96), critic_fc=(128, 128), kind=kind)
    for step in range(3):
        exp_d, _, eps_d, _ = batch(dev, 64, 500 + step)
        for fused, ag in ((True, ag_f), (False, ag_s)):
            monkeypatch.setattr(sac_agent, "_FUSE_SAMPLE", fused)
            li = ag.train(exp_d) if step == 2 else ag.train(exp_d, eps=eps_d)
            if fused:
                li_f = li
        for x, y in zip([li_f.loss] + list(li_f.extra), [li.loss] + list(li.extra)):
            assert torch.equal(x, y)
    for x, y in zip(ag_f.replicated_state(), ag_s.replicated_state()):
        assert torch.equal(x, y)


@pytest.mark.parametrize("cfg", [dict(), dict(critic_loss_weight=1.0, initial_log_alpha=0.7,
                                              td_errors_loss_fn=common.element_wise_huber_loss)])
@pytest.mark.parametrize("weighted", [False, True])
def test_losses_inside_the_gradient_chain_are_bit_identical(dev, cfg, weighted, monkeypatch):
    """aa_mlp_wide_backward_gen (critic loss / actor loss / actor-head backward computed by the
    gradient-chain launches that consume them) against the three launches it replaces: losses, every
    parameter, optimizer slot and log_alpha after four train steps, bit for bit."""
    kw = dict(cfg)
    loss_fn = kw.pop("td_errors_loss_fn", None)

    def build(fused):
        monkeypatch.setattr(sac_agent, "_FUSE_LOSSES", fused)
        ag, _ = make_pair(dev, actor_fc=(128, 96), critic_fc=(128, 128), **kw)
        if loss_fn is not None:
            ag._td_errors_loss_fn = loss_fn
        return ag
    ag_f, ag_s = build(True), build(False)
    g = torch.Generator().manual_seed(21)
    for step in range(4):
        exp_d, _, eps_d, _ = batch(dev, 64, 700 + step)
        wts = (torch.rand(64, generator=g) * (torch.rand(64, generator=g) > 0.2)).to(dev) \
            if weighted else None
        outs = []
        for fused, ag in ((True, ag_f), (False, ag_s)):
            monkeypatch.setattr(sac_agent, "_FUSE_LOSSES", fused)
            outs.append(ag.train(exp_d, weights=wts, eps=eps_d))
        for x, y in zip([outs[0].loss] + list(outs[0].extra), [outs[1].loss] + list(outs[1].extra)):
            assert torch.equal(x, y)
    for x, y in zip(ag_f.replicated_state(), ag_s.replicated_state()):
        assert torch.equal(x, y)


@pytest.mark.parametrize("kind", ["exp", "clip_exp"])
def test_paired_actor_sample_is_bit_identical(dev, kind, monkeypatch):
    """aa_mlp_wide_forward_sample2: the actor's forward + sample on the next observations (critic
    update) and on the observations (actor update) as ONE launch -- the same weights, two inputs,
    two policies' Philox counters -- against the two launches: losses of every step, every
    variable and optimizer slot, the policies' call counters, with supplied noise and with the
    policies' own; and the path must actually be taken."""
    calls = []
    real = sac_agent.SacPolicy.sample_pair

    def spy(*a, **k):
        out = real(*a, **k)
        calls.append(out is not None)
        return out
    monkeypatch.setattr(sac_agent.SacPolicy, "sample_pair", staticmethod(spy))
    monkeypatch.setattr(sac_agent, "_PAIR_SAMPLE", True)
    ag_p, _ = make_pair(dev, actor_fc=(128, 96), critic_fc=(128, 128), kind=kind)
    monkeypatch.setattr(sac_agent, "_PAIR_SAMPLE", False)
    ag_s, _ = make_pair(dev, actor_fc=(128, 96), critic_fc=(128, 128), kind=kind)
    for step in range(4):
        exp_d, _, eps_d, _ = batch(dev, 64, 700 + step)
        outs = []
        for paired, ag in ((True, ag_p), (False, ag_s)):
            monkeypatch.setattr(sac_agent, "_PAIR_SAMPLE", paired)
            outs.append(ag.train(exp_d) if step >= 2 else ag.train(exp_d, eps=eps_d))
        for x, y in zip([outs[0].loss] + list(outs[0].extra), [outs[1].loss] + list(outs[1].extra)):
            assert torch.equal(x, y), f"step {step}"
    assert calls == [True] * 4
    for x, y in zip(ag_p.replicated_state(), ag_s.replicated_state()):
        assert torch.equal(x, y)
    for pp, ps in ((ag_p._loss_policy, ag_s._loss_policy), (ag_p._train_policy, ag_s._train_policy)):
        assert int(pp._call_counter.item()) == int(ps._call_counter.item()) > 0
    for op, os_ in ((ag_p._actor_optimizer, ag_s._actor_optimizer),
                    (ag_p._critic_optimizer, ag_s._critic_optimizer)):
        for x, y in zip(op.variables(), os_.variables()):
            assert torch.equal(x, y)


@pytest.mark.parametrize("cfg", [dict(), dict(critic_loss_weight=1.0, initial_log_alpha=0.7)])
def test_adam_step_inside_the_weight_gradient_launch_is_bit_identical(dev, cfg, monkeypatch):
    """aa_mlp_wide_backward_gen_adam / aa_mlp_wide_dw_adam: the critics' weight-gradient launch
    applies their Adam step (and the soft update of the target critics) to the tile it has just
    finished; the actor's weight-gradient launch, deferred behind its gradient chain, does the
    same for the actor -- against the optimizer launches behind them: every variable, target, optimizer slot,
    step counter and loss over five steps, bit for bit; and the path must actually be taken."""
    monkeypatch.setattr(sac_agent, "_FUSE_DW_ADAM", True)
    ag_f, _ = make_pair(dev, actor_fc=(128, 96), critic_fc=(128, 128), **cfg)
    monkeypatch.setattr(sac_agent, "_FUSE_DW_ADAM", False)
    ag_s, _ = make_pair(dev, actor_fc=(128, 96), critic_fc=(128, 128), **cfg)
    took = []
    for step in range(5):
        exp_d, _, eps_d, _ = batch(dev, 64, 900 + step)
        outs = []
        for fused, ag in ((True, ag_f), (False, ag_s)):
            monkeypatch.setattr(sac_agent, "_FUSE_DW_ADAM", fused)
            outs.append(ag.train(exp_d, eps=eps_d))
            if fused:
                took.append(ag._critic_applied and ag._part_a[5] == 64)
            else:
                assert ag._part_a[5] is None
        for x, y in zip([outs[0].loss] + list(outs[0].extra), [outs[1].loss] + list(outs[1].extra)):
            assert torch.equal(x, y), f"step {step}"
    assert took == [True] * 5 and not ag_s._critic_applied
    for x, y in zip(ag_f.replicated_state(), ag_s.replicated_state()):
        assert torch.equal(x, y)
    assert torch.equal(ag_f._critic_grads, ag_s._critic_grads)
    assert torch.equal(ag_f._actor_network.flat_grads, ag_s._actor_network.flat_grads)
    for of, os_ in ((ag_f._critic_optimizer, ag_s._critic_optimizer),
                    (ag_f._actor_optimizer, ag_s._actor_optimizer)):
        for x, y in zip(of.variables(), os_.variables()):
            assert torch.equal(x, y)
        assert of.iterations == os_.iterations == 5
