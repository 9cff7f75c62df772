"""GPU: the bf16 filter planes of the bf16x6 convolutions stay current WITHOUT pre-pass launches --
the optimizer kernels write the three pieces of every new filter value straight into every
prepared plane set (csrc/optim.hip: aa_rmsprop_step_planes / aa_adam_step_planes;
networks/sequential.py: plane_scatter).  Bit-for-bit against the pre-pass kernels, with identical
parameter updates to the plain optimizer kernels, through the eager and the graphed DQN loop."""
import numpy as np
import pytest
import torch

import bench
from agents_amd import optimizers
from agents_amd.networks import sequential as _seq_mod
from agents_amd.networks import layers as L
from agents_amd.networks import sequential
from agents_amd.specs import tensor_spec
from agents_amd.utils import common, graph

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _both_plane_kinds(monkeypatch):
    """Exercise both plane kinds (the default prepares "pair" only)."""
    monkeypatch.setattr(_seq_mod, "_PW_KINDS", ("pair", "dx"))


def _planes(net):
    pw = net._pw
    return [ws.clone() for kind in ("pair", "dx") for ws in pw[kind].values()]


def _net(dev, seed):
    net = sequential.Sequential(bench.atari_layers(L, 6), seed=seed)
    net.create_variables(tensor_spec.TensorSpec(bench.OBS_SHAPE, torch.uint8), device=dev)
    return net


@pytest.mark.parametrize("opt_name", ["rmsprop", "rmsprop_plain", "adam"])
def test_optimizer_step_keeps_planes_bit_identical_to_the_prepass(dev, opt_name):
    mk = {"rmsprop": lambda: optimizers.RMSprop(2.5e-4, 0.95, 0.95, 0.01, True),
          "rmsprop_plain": lambda: optimizers.RMSprop(1e-3),
          "adam": lambda: optimizers.Adam(1e-3)}[opt_name]
    with torch.cuda.device(dev):
        a, b = _net(dev, 3), _net(dev, 3)
        assert a.enable_prepared_weights()
        desc = a.plane_scatter()
        assert desc is not None and desc.n == 3          # pair (conv2+conv3), conv2.dX, conv3.dX
        covered = sum(int(desc.hi[t] - desc.lo[t]) for t in range(desc.n))
        assert covered >= 2 * (4 * 4 * 32 * 64 + 3 * 3 * 64 * 64)
        oa, ob = mk(), mk()
        g = torch.Generator(device=dev).manual_seed(1)
        for step in range(4):
            grads = torch.randn(a.flat_grads.shape, generator=g, device=dev) * 0.1
            a.flat_grads.copy_(grads)
            b.flat_grads.copy_(grads)
            oa.apply_flat(a.flat_params, a.flat_grads, planes=a.plane_scatter())
            ob.apply_flat(b.flat_params, b.flat_grads)
            assert torch.equal(a.flat_params, b.flat_params)       # the update itself is unchanged
            kept = _planes(a)
            a.refresh_prepared()                                   # what the pre-passes produce
            for k, (x, y) in enumerate(zip(kept, _planes(a))):
                assert torch.equal(x, y), f"plane set {k} differs after step {step}"


def test_stale_planes_are_not_scattered_into(dev):
    """A torch write to the parameters makes the prepared planes stale: plane_scatter() then
    answers None (the agent falls back to the pre-pass launches) until they are refreshed."""
    with torch.cuda.device(dev):
        net = _net(dev, 5)
        net.enable_prepared_weights()
        assert net.plane_scatter() is not None
        net.flat_params.mul_(1.5)
        assert net.plane_scatter() is None and not net._prepared_ok()
        net.refresh_prepared()
        assert net.plane_scatter() is not None


def test_dqn_loop_has_no_prepass_launches_and_planes_stay_current(dev):
    """The benchmarked loop (graphs, three streams): after a number of iterations the planes the
    forwards / backward read equal a fresh split of the current weights, for the online network
    (maintained by the optimizer) and the target network (refreshed at target updates)."""
    with torch.cuda.device(dev):
        w = bench.build_workload(dev, 0, 1, 64, 8, 64, seed=1)
        agent, net = w["agent"], w["net"]
        agent._update_target = agent._get_target_updater(1.0, 5)       # a few target copies inside
        w["init_driver"]._num_steps = 64 * 8
        w["init_driver"].run()
        assert net._pw is not None and net.plane_scatter() is not None
        run = common.function(w["collect_driver"].run)
        it = iter(w["dataset"])
        graph.enable_overlap(dev)
        try:
            ts_ = None
            before = net.flat_params.clone()
            for _ in range(12):
                ts_, _ = run(ts_)
                w["learner"].run(iterations=1, iterator=it)
            graph.join_lanes(dev)
            torch.cuda.synchronize()
        finally:
            graph.disable_overlap()
        assert not torch.equal(before, net.flat_params)
        assert graph.graphed_train(agent).replays == 10
        for n_ in (net, agent._target_q_network):
            assert n_._prepared_ok()
            kept = _planes(n_)
            n_.refresh_prepared()
            for x, y in zip(kept, _planes(n_)):
                assert torch.equal(x, y)
