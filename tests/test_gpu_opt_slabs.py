"""The optimizer that reads conv weight gradients straight from their split-K slabs
(csrc/optim.hip: aa_rmsprop_step_slabs; agents/dqn/dqn_agent.py:412-449 with
gradient_clipping=None): bit-identical to reduce launches + aa_rmsprop_step_planes."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sum16(slabs):
    """The reduce launch's association (csrc/splitk_reduce.h, 16 z-lanes): lane j adds slabs j,
    j + 16, ... in order, the 16 partials are then added in lane order; fp32 throughout."""
    z = slabs.shape[0]
    part = []
    for j in range(16):
        acc = torch.zeros_like(slabs[0])
        for k in range(j, z, 16):
            acc = acc + slabs[k]
        part.append(acc)
    out = part[0]
    for j in range(1, 16):
        out = out + part[j]
    return out


@pytest.mark.parametrize("centered,momentum", [(True, 0.9), (False, 0.0), (True, 0.0)])
def test_slab_optimizer_equals_reduce_then_rmsprop(centered, momentum):
    from agents_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    # two slab segments (kernel + bias each) inside a longer flat vector, flat ranges around them
    n = 40 + (8 * 8 * 4 * 32 + 32) + 100 + (3 * 3 * 16 * 64 + 64) + 4242 + 3   # tail of 3
    segs = [(40, 8 * 8 * 4 * 32, 32, 67), (40 + 8224 + 100, 3 * 3 * 16 * 64, 64, 33)]
    p0 = torch.randn(n, generator=g)
    grads = torch.randn(n, generator=g) * 0.1
    slots0 = [torch.rand(n, generator=g) + 0.5, torch.randn(n, generator=g) * 0.1,
              torch.randn(n, generator=g) * 0.01]
    slabs, expected = [], grads.clone()
    for off, mn, nt, z in segs:
        sl = torch.randn(z * mn + z * nt, generator=g) * 0.05
        slabs.append(sl)
        expected[off:off + mn] = _sum16(sl[:z * mn].view(z, mn))
        expected[off + mn:off + mn + nt] = _sum16(sl[z * mn:].view(z, nt))
    res = []
    for use_slabs in (False, True):
        p = p0.clone().to(dev)
        ms, mg, mom = (t.clone().to(dev) for t in slots0)
        gd = (grads.clone() if use_slabs else expected.clone()).to(dev)   # segments hold garbage
        args = [p.data_ptr(), gd.data_ptr(), ms.data_ptr(), mg.data_ptr() if centered else None,
                mom.data_ptr() if momentum > 0 else None, n, 2.5e-4, 0.95, momentum, 0.01, None]
        if use_slabs:
            G = _lib.GradSlabs()
            G.n = len(segs)
            keep = [s.to(dev) for s in slabs]
            for i, ((off, mn, nt, z), sd) in enumerate(zip(segs, keep)):
                G.splits[i], G.mn[i], G.n_tail[i], G.offset[i], G.slab[i] = z, mn, nt, off, \
                    sd.data_ptr()
            _lib.check(lib.aa_rmsprop_step_slabs(*args, ctypes.byref(G), _lib.stream_ptr()),
                       "aa_rmsprop_step_slabs")
        else:
            _lib.check(lib.aa_rmsprop_step_planes(*args, _lib.stream_ptr()),
                       "aa_rmsprop_step_planes")
        torch.cuda.synchronize()
        res.append([t.cpu() for t in (p, ms, mg, mom, gd)])
    for a, b in zip(res[0][:4], res[1][:4]):
        assert torch.equal(a, b)
    assert torch.equal(res[1][4], expected)   # the sums are stored to g: the buffer is complete
    assert not torch.equal(res[0][0], p0)


def _atari_agent(dev, A=6):
    from agents_amd import optimizers
    from agents_amd.agents.dqn import dqn_agent
    from agents_amd.networks import layers as L
    from agents_amd.networks import sequential
    from agents_amd.specs import tensor_spec
    from agents_amd.trajectories import time_step as ts
    from agents_amd.utils import common
    obs_spec = tensor_spec.TensorSpec((84, 84, 4), torch.uint8)
    aspec = tensor_spec.BoundedTensorSpec((), torch.int64, 0, A - 1)
    vs = lambda: L.VarianceScaling(2.0)
    net = sequential.Sequential([
        L.Rescale(255.0), L.Conv2D(32, 8, 4, "relu", kernel_initializer=vs()),
        L.Conv2D(64, 4, 2, "relu", kernel_initializer=vs()),
        L.Conv2D(64, 3, 1, "relu", kernel_initializer=vs()), L.Flatten(),
        L.Dense(512, "relu", kernel_initializer=vs()), L.Dense(A, kernel_initializer=vs())], seed=3)
    agent = dqn_agent.DqnAgent(ts.time_step_spec(obs_spec), aspec, q_network=net,
                               optimizer=optimizers.RMSprop(2.5e-4, 0.95, 0.95, 0.01, True),
                               td_errors_loss_fn=common.element_wise_huber_loss, gamma=0.99,
                               target_update_period=2)
    agent.initialize()
    return agent, net


def test_dqn_atari_step_with_slab_optimizer_is_bit_identical(monkeypatch):
    """Bench-config DQN step (B = 256: 64 / 64 / 256 slabs for conv3 / conv2 / conv1): the three
    conv layers take the slab path, parameters and slots equal the reduce-launch path bit for bit,
    eagerly and from the HIP graph."""
    from agents_amd import ops
    from agents_amd.agents.dqn import dqn_agent
    from agents_amd.trajectories import trajectory
    from agents_amd.utils import graph
    dev = torch.device("cuda", 0)
    B = 256
    rng = np.random.default_rng(0)

    def batch():
        return trajectory.Trajectory(
            step_type=torch.as_tensor(rng.integers(0, 3, (B, 2)).astype(np.int32), device=dev),
            observation=torch.as_tensor(rng.integers(0, 256, (B, 2, 84, 84, 4), dtype=np.uint8),
                                        device=dev),
            action=torch.as_tensor(rng.integers(0, 6, (B, 2)).astype(np.int64), device=dev),
            policy_info=(),
            next_step_type=torch.as_tensor(rng.integers(0, 3, (B, 2)).astype(np.int32),
                                           device=dev),
            reward=torch.as_tensor(rng.standard_normal((B, 2)).astype(np.float32), device=dev),
            discount=torch.as_tensor((rng.random((B, 2)) > 0.1).astype(np.float32), device=dev))

    taken = []
    real = ops.grad_slabs

    def spy(pending, flat):
        g = real(pending, flat)
        taken.append(0 if g is None else int(g.n))
        return g
    monkeypatch.setattr(ops, "grad_slabs", spy)
    with torch.cuda.device(dev):
        monkeypatch.setattr(dqn_agent, "OPT_SUMS_SLABS", True)
        (a_s, n_s), (a_g, n_g) = _atari_agent(dev), _atari_agent(dev)
        monkeypatch.setattr(dqn_agent, "OPT_SUMS_SLABS", False)
        a_r, n_r = _atari_agent(dev)
        train_g = graph.graphed_train(a_g)
        for step in range(4):
            exp = batch()
            monkeypatch.setattr(dqn_agent, "OPT_SUMS_SLABS", True)
            li_s = a_s.train(exp)
            li_g = train_g(exp)
            monkeypatch.setattr(dqn_agent, "OPT_SUMS_SLABS", False)
            li_r = a_r.train(exp)
            torch.cuda.synchronize()
            assert torch.equal(li_s.loss, li_r.loss) and torch.equal(li_g.loss, li_r.loss)
            assert torch.equal(n_s.flat_params, n_r.flat_params), f"step {step}"
            assert torch.equal(n_g.flat_params, n_r.flat_params), f"step {step} (graph)"
            for vs_, vr in zip(a_s._optimizer.variables(), a_r._optimizer.variables()):
                assert torch.equal(vs_, vr)
            assert torch.equal(a_s._target_q_network.flat_params,
                               a_r._target_q_network.flat_params)
    # conv1 (GEMM slabs) + conv2 + conv3 on every slab-path backward (4 eager + the captures);
    # the reduce-launch agent never asked
    assert len(taken) >= 5 and all(t == 3 for t in taken), taken


def test_clipping_or_a_gradient_hook_keep_the_reduce_launches():
    from agents_amd.agents.dqn import dqn_agent
    dev = torch.device("cuda", 0)
    with torch.cuda.device(dev):
        agent, net = _atari_agent(dev)
        assert agent._optimizer_sums_slabs(net)
        agent.gradient_hook = lambda g: None
        assert not agent._optimizer_sums_slabs(net)
        agent.gradient_hook = None
        agent._gradient_clipping = 1.0
        assert not agent._optimizer_sums_slabs(net)
