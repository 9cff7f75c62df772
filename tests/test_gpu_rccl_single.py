"""GPU: the multi-GPU wiring exercised on ONE device with a real RCCL process group (world size 1):
`torch.distributed` (backend "nccl" == RCCL on ROCm) initialised, the Learner's gradient hook issuing
`all_reduce` between the two train graphs, HIP-graph captures happening while the process group
(and its watchdog thread) is alive.  With one rank the all-reduce is the identity, so the run must
match a run without any process group bit for bit.  (World size 2 semantics are covered on CPU with
gloo in tests/test_distributed_gloo.py; 8-GPU runs belong to the driver.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from agents_amd.train import learner
from agents_amd.train.utils import strategy_utils
from agents_amd.utils import common, graph
from tests.test_gpu_graphs import _stack

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(dev, strategy, steps=30):
    env, agent, rb, drv, net = _stack(dev, 8, 64, 0.2, 1)
    run = common.function(drv.run)
    lrn = learner.Learner(None, common.Variable(0), agent, strategy=strategy)
    if strategy is not None:      # one replica: install the hook by hand to exercise the collective
        agent.gradient_hook = strategy.all_reduce_sum_
    for _ in range(4):
        run()
    it = iter(rb.as_dataset(sample_batch_size=16, num_steps=2).prefetch(3))
    li = None
    for _ in range(steps):
        run()
        li = lrn.run(iterations=1, iterator=it)
    torch.cuda.synchronize()
    return net.flat_params.clone(), float(li.loss), graph.graphed_train(agent).replays


def test_rccl_hook_between_train_graphs(dev):
    ref_params, ref_loss, _ = _run(dev, None)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        strat = strategy_utils.DataParallelStrategy()
        assert strat.num_replicas_in_sync == 1
        params, loss, replays = _run(dev, strat)
        t = torch.ones(4, device=dev)
        assert torch.equal(strat.all_reduce_sum_(t), torch.ones(4, device=dev))
    finally:
        dist.destroy_process_group()
    assert replays > 15, "train graphs did not replay with the process group alive"
    assert torch.equal(params, ref_params) and loss == ref_loss


def test_bucketed_allreduce_overlap_matches_single_bucket(dev):
    """Bucket mode of the graphed train step (dense-tail gradients all-reduced asynchronously
    while the conv layers are still in backward, then the head): same parameters as the single
    synchronous all-reduce and as no process group at all."""
    ref_params, ref_loss, _ = _run(dev, None)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        strat = strategy_utils.DataParallelStrategy()
        results = {}
        for bucketed in (True, False):
            graph.BUCKETED_ALLREDUCE = bucketed
            env, agent, rb, drv, net = _stack(dev, 8, 64, 0.2, 1)
            run = common.function(drv.run)
            lrn = learner.Learner(None, common.Variable(0), agent, strategy=strat)
            agent.gradient_hook = strat.all_reduce_sum_
            agent.gradient_hook_async = strat.all_reduce_sum_async_
            assert agent._bucket_split() is not None
            for _ in range(4):
                run()
            it = iter(rb.as_dataset(sample_batch_size=16, num_steps=2).prefetch(3))
            for _ in range(30):
                run()
                li = lrn.run(iterations=1, iterator=it)
            torch.cuda.synchronize()
            gt = graph.graphed_train(agent)
            ents = [e for b in gt._cache.values() for e in b.values()]
            assert any(e.g_grads_b is not None for e in ents) == bucketed
            results[bucketed] = (net.flat_params.clone(), float(li.loss))
    finally:
        graph.BUCKETED_ALLREDUCE = True
        dist.destroy_process_group()
    assert torch.equal(results[True][0], results[False][0])
    assert torch.equal(results[True][0], ref_params) and results[True][1] == ref_loss


class _IdentityStrategy:
    """The hooks of a one-replica DataParallelStrategy without a process group: the reference run
    takes the same code paths (PPO's train step is the unfused one whenever a hook is installed)."""
    num_replicas_in_sync = 1
    stats = {"calls": 0}

    def all_reduce_sum_(self, t):
        return t

    def all_gather_batch(self, t):
        return t

    def reduce_sum(self, t):
        return t


def _loop_other(dev, kind, strategy):
    """A few iterations of the PPO / SAC benchmark loop at toy sizes through a Learner on
    `strategy`, the gradient hook installed by hand (one replica): returns (parameters, loss)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "tools"))
    if kind == "sac":
        import bench_sac
        w = bench_sac.build(dev, envs=16, max_length=16, batch=8)
        agent, lrn = w["agent"], w["learner"]
        if strategy is not None:
            lrn.strategy = strategy
            agent.gradient_hook = strategy.all_reduce_sum_
        it = iter(w["dataset"])
        t, li = None, None
        for _ in range(8):
            t, _ = w["collect"](t)
            li = lrn.run(iterations=1, iterator=it)
        params = torch.cat([p.reshape(-1) for p in agent.replicated_state() if p is not None
                            and p.dtype == torch.float32])
    else:
        import bench_ppo
        w = bench_ppo.build(dev, envs=16, steps=8, minibatch=32, epochs=2)
        agent, lrn, rb = w["agent"], w["learner"], w["rb"]
        if strategy is not None:
            lrn.strategy = strategy
            agent.gradient_hook = strategy.all_reduce_sum_
            agent.batch_gather_hook = strategy.all_gather_batch
        t, li = None, None
        for _ in range(3):
            rb.clear()
            t, _ = w["collect_driver"].run(t)
            lrn._train_iter = lrn._norm_iter = None
            li = lrn.run()
        params = agent.flat_params.clone()
    torch.cuda.synchronize()
    return params.clone(), float(li.loss)


@pytest.mark.parametrize("kind", ["ppo", "sac"])
def test_rccl_hook_in_the_ppo_and_sac_train_steps(dev, kind):
    """PPO and SAC have no bucket mode (their gradient buffers are 44 KB / 2 MB): the hook is one
    synchronous all-reduce per optimizer, inside the train step.  With a live RCCL group of one
    rank -- collectives issued on the training stream between the kernels of the step, the
    normaliser gather through all_gather_into_tensor (PPO), HIP-graph captures of the collect
    loop with the watchdog alive -- the run equals the run without a process group bit for bit."""
    ref_params, ref_loss = _loop_other(dev, kind, _IdentityStrategy())
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        strat = strategy_utils.DataParallelStrategy()
        params, loss = _loop_other(dev, kind, strat)
        assert strat.stats["calls"] > 0, "the gradient hook never reached the communicator"
    finally:
        dist.destroy_process_group()
    assert torch.equal(params, ref_params) and loss == ref_loss
