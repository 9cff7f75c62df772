"""GPU parity of DqnAgent / DdqnAgent (product API) against the reference's known answers and
against oracle.OracleDqnAgent (torch-CPU fp32) over several train steps.

Tolerances (north star): losses within 1e-5 relative; parameters after K optimizer steps within
2e-5 of max|param| (fp32 accumulation-order noise through the MFMA GEMMs)."""
import numpy as np
import pytest
import torch

from agents_amd import optimizers
from agents_amd.agents.dqn import dqn_agent
from agents_amd.networks import layers as L
from agents_amd.networks import q_network, sequential
from agents_amd.specs import tensor_spec
from agents_amd.trajectories import time_step as ts
from agents_amd.trajectories import trajectory
from agents_amd.utils import common
from oracle import dqn as odqn
from oracle import nets as onets
from oracle import optim as ooptim

pytestmark = pytest.mark.gpu
FIRST, MID, LAST = 0, 1, 2


def dummy_net(l2=0.0):
    return sequential.Sequential([L.Dense(2, kernel_initializer=L.Constant([[2, 1], [1, 1]]),
                                          bias_initializer=L.Constant([1, 1]),
                                          kernel_regularizer_l2=l2)])


def specs():
    obs = tensor_spec.TensorSpec((2,), torch.float32)
    return ts.time_step_spec(obs), tensor_spec.BoundedTensorSpec((), torch.int32, 0, 1)


def experience(dev, frames):
    def col(k, dt):
        return torch.tensor(np.stack([np.asarray(f[k]) for f in frames], 1), dtype=dt, device=dev)
    n = len(frames)
    nxt = [frames[min(i + 1, n - 1)]["step_type"] for i in range(n)]
    return trajectory.Trajectory(
        step_type=col("step_type", torch.int32), observation=col("obs", torch.float32),
        action=col("action", torch.int32), policy_info=(),
        next_step_type=torch.tensor(np.stack(nxt, 1), dtype=torch.int32, device=dev),
        reward=col("reward", torch.float32), discount=col("discount", torch.float32))


def two_frame(dev, next_obs):
    f0 = dict(step_type=[FIRST] * 2, obs=[[1, 2], [3, 4]], action=[0, 1], reward=[10, 20],
              discount=[.9, .9])
    f1 = dict(step_type=[MID] * 2, obs=next_obs, action=[0, 1], reward=[10, 20],
              discount=[.9, .9])
    return experience(dev, [f0, f1])


@pytest.mark.parametrize("cls", [dqn_agent.DqnAgent, dqn_agent.DdqnAgent])
def test_known_answer_losses(dev, cls):
    tss, aspec = specs()
    with torch.cuda.device(dev):
        agent = cls(tss, aspec, q_network=dummy_net(), optimizer=None)
        loss, extra = agent._loss(two_frame(dev, [[5, 6], [7, 8]]))
        np.testing.assert_allclose(loss.item(), 26.0, rtol=1e-6)          # testLoss
        np.testing.assert_allclose(extra.td_loss.cpu().numpy(), [19.8, 32.2], rtol=1e-6)
        loss, _ = agent._loss(two_frame(dev, [[-5, 6], [-7, 8]]))
        np.testing.assert_allclose(loss.item(), 9.8, rtol=1e-6)           # changed optimal actions
        agent = cls(tss, aspec, q_network=dummy_net(l2=1.0), optimizer=None)
        loss, _ = agent._loss(two_frame(dev, [[5, 6], [7, 8]]))
        np.testing.assert_allclose(loss.item(), 33.0, rtol=1e-6)          # L2 regularisation


def test_known_answer_n_step(dev):
    tss, aspec = specs()
    base = dict(action=[0, 1], reward=[10, 20], discount=[.9, .9])
    with torch.cuda.device(dev):
        agent = dqn_agent.DqnAgent(tss, aspec, q_network=dummy_net(), optimizer=None,
                                   n_step_update=2)
        frames = [dict(step_type=[FIRST] * 2, obs=[[1, 2], [3, 4]], **base),
                  dict(step_type=[MID] * 2, obs=[[5, 6], [7, 8]], **base),
                  dict(step_type=[MID] * 2, obs=[[9, 10], [11, 12]], **base)]
        loss, _ = agent._loss(experience(dev, frames))
        np.testing.assert_allclose(loss.item(), 47.42, rtol=1e-6)         # testLossNStep
        agent = dqn_agent.DqnAgent(tss, aspec, q_network=dummy_net(), optimizer=None,
                                   n_step_update=3)
        frames = [dict(step_type=[MID] * 2, obs=[[1, 2], [3, 4]], **base),
                  dict(step_type=[MID] * 2, obs=[[5, 6], [7, 8]], action=[0, 1], reward=[10, 20],
                       discount=[0., 0.]),
                  dict(step_type=[LAST] * 2, obs=[[9, 10], [11, 12]], action=[0, 1],
                       reward=[0, 0], discount=[1., 1.]),
                  dict(step_type=[FIRST] * 2, obs=[[13, 14], [15, 16]], action=[0, 1],
                       reward=[0, 0], discount=[1., 1.])]
        loss, _ = agent._loss(experience(dev, frames))
        np.testing.assert_allclose(loss.item(), 21.5, rtol=1e-6)          # MidMidLastFirst
        with pytest.raises(ValueError, match="train_sequence_length"):
            agent.loss(two_frame(dev, [[5, 6], [7, 8]]))


def test_known_answer_masked_actions(dev):
    obs_spec = (tensor_spec.TensorSpec((2,), torch.float32),
                tensor_spec.BoundedTensorSpec((2,), torch.int32, 0, 1))
    tss = ts.time_step_spec(obs_spec)
    aspec = tensor_spec.BoundedTensorSpec((), torch.int32, 0, 1)
    with torch.cuda.device(dev):
        agent = dqn_agent.DqnAgent(tss, aspec, q_network=dummy_net(), optimizer=None,
                                   observation_and_action_constraint_splitter=lambda x: (x[0], x[1]))
        e = two_frame(dev, [[5, 6], [7, 8]])
        mask = torch.tensor([[[1, 1], [0, 1]], [[1, 1], [1, 0]]], dtype=torch.int32, device=dev)
        e = e._replace(observation=(e.observation, mask))
        loss, _ = agent._loss(e)
        np.testing.assert_allclose(loss.item(), 23.75, rtol=1e-6)
        step = agent.policy.action(ts.restart((e.observation[0][:, 1], mask[:, 1]), batch_size=2))
        assert step.action.cpu().tolist() == [1, 0]


def test_policy_and_validation(dev):
    tss, aspec = specs()
    with torch.cuda.device(dev):
        agent = dqn_agent.DqnAgent(tss, aspec, q_network=dummy_net(), optimizer=None)
        obs = torch.tensor([[1., 2.], [3., 4.]], device=dev)
        a = agent.policy.action(ts.restart(obs, batch_size=2)).action
        assert a.cpu().tolist() == [0, 0] and a.dtype == torch.int32     # testPolicy / restore
        c = agent.collect_policy.action(ts.restart(obs, batch_size=2)).action
        assert set(c.cpu().tolist()) <= {0, 1}
    with pytest.raises(ValueError, match="minimum of 0"):
        dqn_agent.DqnAgent(tss, tensor_spec.BoundedTensorSpec((), torch.int32, 1, 2),
                           q_network=dummy_net(), optimizer=None)
    with pytest.raises(ValueError, match="scalar actions"):
        dqn_agent.DqnAgent(tss, tensor_spec.BoundedTensorSpec((2,), torch.int32, 0, 1),
                           q_network=dummy_net(), optimizer=None)


# ---- multi-step train parity vs the torch-CPU oracle --------------------------------------------
def _copy_params(net, oparams):
    net.set_weights([p.detach().numpy() for p in oparams])


def _rand_experience(rng, B, T, obs_shape, obs_dtype, A, dev):
    if obs_dtype == torch.uint8:
        obs = rng.integers(0, 256, size=(B, T) + obs_shape, dtype=np.uint8)
    else:
        obs = rng.standard_normal((B, T) + obs_shape).astype(np.float32)
    act = rng.integers(0, A, size=(B, T)).astype(np.int64)
    rew = rng.choice([-1.0, 0.0, 1.0], size=(B, T)).astype(np.float32)
    disc = (rng.random((B, T)) > 0.1).astype(np.float32)
    st = rng.integers(0, 3, size=(B, T)).astype(np.int32)
    e = trajectory.Trajectory(
        step_type=torch.tensor(st, device=dev), observation=torch.tensor(obs, device=dev),
        action=torch.tensor(act, device=dev), policy_info=(),
        next_step_type=torch.tensor(np.roll(st, -1, 1), device=dev),
        reward=torch.tensor(rew, device=dev), discount=torch.tensor(disc, device=dev))
    return e, (torch.tensor(obs), act, rew, disc, st)


def _run_parity(dev, layers_gpu, olayers, obs_shape, obs_dtype, A, B, steps, make_opt, make_oopt,
                loss_fn, okind, double_q=False, period=2, tau=1.0, n_step=1, clip=None,
                gamma=0.99, tol_p=3e-5):
    rng = np.random.default_rng(B + A + steps)
    tss = ts.time_step_spec(tensor_spec.TensorSpec(obs_shape, obs_dtype))
    aspec = tensor_spec.BoundedTensorSpec((), torch.int64, 0, A - 1)
    oparams = onets.init_params(olayers, obs_shape, seed=3)
    with torch.cuda.device(dev):
        net = sequential.Sequential(layers_gpu)
        cls = dqn_agent.DdqnAgent if double_q else dqn_agent.DqnAgent
        agent = cls(tss, aspec, q_network=net, optimizer=make_opt(), td_errors_loss_fn=loss_fn,
                    gamma=gamma, target_update_period=period, target_update_tau=tau,
                    n_step_update=n_step, gradient_clipping=clip, reward_scale_factor=0.7)
        _copy_params(net, oparams)
        agent.initialize()
        oagent = odqn.OracleDqnAgent(olayers, obs_shape, A, oparams, optimizer=make_oopt(),
                                     gamma=gamma, reward_scale=0.7, loss=okind, double_q=double_q,
                                     target_update_tau=tau, target_update_period=period,
                                     n_step=n_step)
        hook = None
        if clip is not None:
            def hook(grads):
                return [g * (clip / max(float(g.norm()), clip)) for g in grads]
        for k in range(steps):
            e, (obs, act, rew, disc, st) = _rand_experience(rng, B, n_step + 1, obs_shape,
                                                            obs_dtype, A, dev)
            li = agent.train(e)
            ototal, aux, ograds = oagent.train(obs, act, rew, disc, st, grad_hook=hook)
            np.testing.assert_allclose(li.loss.item(), float(ototal), rtol=1e-5, atol=1e-7,
                                       err_msg=f"loss at step {k}")
            np.testing.assert_allclose(li.extra.td_error.cpu().numpy(), aux["td_error"],
                                       rtol=2e-4, atol=2e-5, err_msg=f"td_error at step {k}")
            if k == 0 and clip is None:
                for g, og in zip(net.gradients, ograds):
                    scale = max(float(og.abs().max()), 1e-12)
                    err = float((g.cpu() - og).abs().max())
                    assert err <= 3e-5 * scale, f"grad mismatch {err/scale:.2e}"
        assert int(agent.train_step_counter) == steps
        for v, ov in zip(net.variables, oagent.params):
            scale = max(float(ov.abs().max()), 1e-12)
            err = float((v.cpu() - ov.detach()).abs().max())
            assert err <= tol_p * scale, f"param mismatch {err/scale:.2e}"
        for v, ov in zip(agent._target_q_network.variables, oagent.target):
            scale = max(float(ov.abs().max()), 1e-12)
            assert float((v.cpu() - ov).abs().max()) <= tol_p * scale


def test_train_parity_cartpole_mlp_adam(dev):
    """BASELINE config 1 shapes: obs f32[4], QNetwork fc=(100,), Adam, squared loss, B=64."""
    layers = [L.Dense(100, "relu"), L.Dense(2)]
    _run_parity(dev, layers, onets.mlp_q_layers((100,), 2), (4,), torch.float32, 2, 64, 6,
                lambda: optimizers.AdamOptimizer(1e-3), lambda: ooptim.Adam(1e-3, eps=1e-8),
                common.element_wise_squared_loss, "squared")


def test_train_parity_mlp_ddqn_nstep_clip_soft_target(dev):
    layers = [L.Dense(64, "tanh"), L.Dense(32, "relu"), L.Dense(5)]
    olayers = [{"kind": "dense", "units": 64, "act": "tanh"},
               {"kind": "dense", "units": 32, "act": "relu"},
               {"kind": "dense", "units": 5, "act": None}]
    _run_parity(dev, layers, olayers, (17,), torch.float32, 5, 96, 5,
                lambda: optimizers.RMSprop(1e-3, 0.9, 0.0, 1e-7, False),
                lambda: ooptim.RMSprop(1e-3, 0.9, 0.0, 1e-7, False),
                common.element_wise_huber_loss, "huber", double_q=True, period=1, tau=0.05,
                n_step=3, clip=0.5)


def _atari_gpu_layers(A):
    vs = lambda: L.VarianceScaling(2.0)
    return [L.Rescale(255.0), L.Conv2D(32, (8, 8), 4, "relu", kernel_initializer=vs()),
            L.Conv2D(64, (4, 4), 2, "relu", kernel_initializer=vs()),
            L.Conv2D(64, (3, 3), 1, "relu", kernel_initializer=vs()), L.Flatten(),
            L.Dense(512, "relu", kernel_initializer=vs()),
            L.Dense(A, None, kernel_initializer=vs())]


def test_train_parity_atari_convnet_rmsprop(dev):
    """BASELINE config 2 network (Mnih-15 stack on uint8 84x84x4), centred RMSProp, Huber, at a
    reduced batch the CPU oracle finishes in seconds."""
    A = 6
    _run_parity(dev, _atari_gpu_layers(A), onets.atari_q_layers(A), (84, 84, 4), torch.uint8, A,
                32, 3, lambda: optimizers.RMSprop(2.5e-4, 0.95, 0.95, 0.01, True),
                lambda: ooptim.RMSprop(2.5e-4, 0.95, 0.95, 0.01, True),
                common.element_wise_huber_loss, "huber", period=2, tol_p=5e-5)


def test_qnetwork_builder_and_call(dev):
    with torch.cuda.device(dev):
        obs = tensor_spec.TensorSpec((84, 84, 4), torch.float32)
        aspec = tensor_spec.BoundedTensorSpec((), torch.int64, 0, 3)
        net = q_network.QNetwork(obs, aspec, conv_layer_params=((32, 8, 4), (64, 4, 2)),
                                 fc_layer_params=(64,), seed=0)
        net.create_variables()
        x = torch.randn(5, 84, 84, 4, device=dev)
        out, state = net(x)
        assert tuple(out.shape) == (5, 4) and state == ()
        w = [torch.tensor(a) for a in net.get_weights()]
        ol = [{"kind": "conv", "filters": 32, "kernel": (8, 8), "stride": 4, "act": "relu"},
              {"kind": "conv", "filters": 64, "kernel": (4, 4), "stride": 2, "act": "relu"},
              {"kind": "flatten"}, {"kind": "dense", "units": 64, "act": "relu"},
              {"kind": "dense", "units": 4, "act": None}]
        ref = onets.forward(ol, w, x.cpu())
        scale = float(ref.abs().max())
        assert float((out.cpu() - ref).abs().max()) <= 3e-5 * scale
        np.testing.assert_allclose(net.get_weights()[-1], -0.2)  # Q-layer bias init
        assert np.abs(net.get_weights()[-2]).max() <= 0.03


def test_learner_field_sums_come_from_the_loss_kernel(dev):
    """Learner.run returns every LossInfo field summed over all axes (learner.py:322-337).  The
    loss kernel produces those sums itself; they equal the sums of the per-sample fields, for the
    train step's LossInfo only (any other LossInfo goes through the generic reduction)."""
    from agents_amd.train import learner
    tss, aspec = specs()
    with torch.cuda.device(dev):
        net = dummy_net()
        agent = dqn_agent.DqnAgent(tss, aspec, q_network=net,
                                   optimizer=optimizers.SGD(learning_rate=0.0),
                                   td_errors_loss_fn=common.element_wise_squared_loss)
        agent.initialize()
        exp = two_frame(dev, [[5, 6], [7, 8]])
        li = agent.train(exp)
        pre = agent.reduce_loss_info(li)
        assert pre is not None and pre.extra.td_loss.dim() == 0
        np.testing.assert_allclose(pre.extra.td_loss.item(), li.extra.td_loss.double().sum().item(),
                                   rtol=1e-6)
        np.testing.assert_allclose(pre.extra.td_error.item(),
                                   li.extra.td_error.double().sum().item(), rtol=1e-6)
        assert agent.reduce_loss_info(agent.loss(exp)) is None      # clones: generic path
        lrn = learner.Learner(None, common.create_variable("train_step"), agent,
                              experience_dataset_fn=None, use_graph=False)
        red = lrn._reduce_loss(li)
        np.testing.assert_allclose(red.extra.td_loss.item(), li.extra.td_loss.sum().item(), rtol=1e-6)
        np.testing.assert_allclose(red.loss.item(), li.loss.item(), rtol=0)


# ---- TD loss + Q-head backward in one launch ------------------------------------------------------
@pytest.mark.parametrize("double_q,loss", [(False, "huber"), (True, "squared")])
def test_fused_loss_head_backward_is_bit_identical(dev, double_q, loss):
    """csrc/dqn.hip: aa_dqn_loss_head_backward == aa_dqn_td_loss_sums + aa_dense_small_backward
    (every workgroup of the head's backward recomputes dL/dq in LDS): loss, td fields, dq, every
    gradient and the parameters after the optimizer step, bit for bit, eager and graphed."""
    from agents_amd import ops
    from agents_amd.utils import graph
    cls = dqn_agent.DdqnAgent if double_q else dqn_agent.DqnAgent
    obs_spec = tensor_spec.TensorSpec((20, 20, 4), torch.uint8)
    aspec = tensor_spec.BoundedTensorSpec((), torch.int64, 0, 4)
    tss = ts.time_step_spec(obs_spec)

    def make():
        net = sequential.Sequential([L.Rescale(255.0), L.Conv2D(8, 4, 4, "relu"),
                                     L.Conv2D(16, 3, 1, "relu"), L.Flatten(),
                                     L.Dense(64, "relu"), L.Dense(5)], seed=11)
        fn = common.element_wise_huber_loss if loss == "huber" else \
            common.element_wise_squared_loss
        return cls(tss, aspec, q_network=net, optimizer=optimizers.RMSprop(1e-3, 0.95, 0.9, 0.01,
                                                                           True),
                   td_errors_loss_fn=fn, gamma=0.97, target_update_period=2, n_step_update=2), net

    rng = np.random.default_rng(3)
    B = 48

    def batch():
        return trajectory.Trajectory(
            step_type=torch.as_tensor(rng.integers(0, 3, (B, 3)).astype(np.int32), device=dev),
            observation=torch.as_tensor(rng.integers(0, 256, (B, 3, 20, 20, 4), dtype=np.uint8),
                                        device=dev),
            action=torch.as_tensor(rng.integers(0, 5, (B, 3)).astype(np.int64), device=dev),
            policy_info=(),
            next_step_type=torch.as_tensor(rng.integers(0, 3, (B, 3)).astype(np.int32),
                                           device=dev),
            reward=torch.as_tensor(rng.standard_normal((B, 3)).astype(np.float32), device=dev),
            discount=torch.as_tensor((rng.random((B, 3)) > 0.2).astype(np.float32), device=dev))

    with torch.cuda.device(dev):
        (a_f, n_f), (a_s, n_s) = make(), make()
        a_f.initialize()
        a_s.initialize()
        train_f = graph.graphed_train(a_f)
        for step in range(5):
            exp = batch()
            w = torch.as_tensor(rng.uniform(0.5, 1.5, B).astype(np.float32), device=dev) \
                if step % 2 else None
            ops.FUSE_LOSS_HEAD = True
            li_f = train_f(exp, weights=w) if w is None else a_f.train(exp, weights=w)
            assert a_f._get_work(B, dev).head_done
            ops.FUSE_LOSS_HEAD = False
            try:
                li_s = a_s.train(exp, weights=w)
            finally:
                ops.FUSE_LOSS_HEAD = True
            assert not a_s._get_work(B, dev).head_done
            assert torch.equal(li_f.loss, li_s.loss)
            assert torch.equal(li_f.extra.td_loss, li_s.extra.td_loss)
            assert torch.equal(li_f.extra.td_error, li_s.extra.td_error)
            assert torch.equal(a_f._get_work(B, dev).dq, a_s._get_work(B, dev).dq)
            assert torch.equal(n_f.flat_grads, n_s.flat_grads), f"step {step}"
            assert torch.equal(n_f.flat_params, n_s.flat_params)
