"""CPU, world_size 2, gloo: the Learner's data-parallel wiring.

Uses a small stand-in agent (linear model, gradients by hand on CPU tensors) because the real
agents need HIP kernels; what is under test is the Learner / strategy contract that is identical
for every agent: num_replicas is pushed into the agent, the flat gradient buffer is SUM
all-reduced once per step through `gradient_hook`, per-replica losses divided by
B_local * num_replicas sum to the single-process loss on the concatenated batch, replicas stay
bit-identical, and Learner.run returns the replica-SUM of LossInfo
(cf. tf_agents/train/learner_test.py:446-562 testLossLearnerDifferentDistStrat)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from agents_amd.agents import tf_agent
from agents_amd.train import learner
from agents_amd.train.utils import strategy_utils
from agents_amd.utils import common


class LinearAgent:
    """loss = sum_b (x_b . w - y_b)^2 / (B_local * num_replicas); SGD; flat params/grads."""

    def __init__(self, dim, lr=0.1):
        self.w = torch.zeros(dim, dtype=torch.float32)
        self.flat_grads = torch.zeros(dim, dtype=torch.float32)
        self.lr = lr
        self.num_replicas = 1
        self.gradient_hook = None
        self.train_step_counter = common.Variable(0)
        self.initialized = False

    def initialize(self):
        self.initialized = True

    def replicated_state(self):
        return [self.w]

    def train(self, experience):
        x, y = experience
        denom = x.shape[0] * self.num_replicas
        err = x @ self.w - y
        loss = (err * err).sum() / denom
        self.flat_grads.copy_(2.0 * (x.T @ err) / denom)
        if self.gradient_hook is not None:
            self.gradient_hook(self.flat_grads)
        self.w -= self.lr * self.flat_grads
        self.train_step_counter.assign_add(1)
        return tf_agent.LossInfo(loss, {"per_example": err * err / denom})

    def loss(self, experience):
        x, y = experience
        err = x @ self.w - y
        return tf_agent.LossInfo((err * err).sum() / (x.shape[0] * self.num_replicas), ())


def _data(seed, n, dim):
    rng = np.random.RandomState(seed)
    return (torch.tensor(rng.randn(n, dim).astype(np.float32)),
            torch.tensor(rng.randn(n).astype(np.float32)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, steps, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x, y = _data(0, 16, 5)
        shard = (x[rank::world], y[rank::world])            # every rank owns its own shard
        agent = LinearAgent(5)
        agent.w += float(rank)     # replicas that start apart (default, unseeded initialisers) ...
        lrn = learner.Learner(None, agent.train_step_counter, agent,
                              experience_dataset_fn=lambda: iter(lambda: (shard, "info"), None))
        assert float(agent.w.abs().max()) == 0.0   # ... start from rank 0's weights (broadcast)
        assert isinstance(lrn.strategy, strategy_utils.DataParallelStrategy)
        assert agent.num_replicas == world and agent.gradient_hook is not None
        losses = []
        for _ in range(steps):
            li = lrn.run(iterations=1)
            losses.append(float(li.loss))
        extra = float(li.extra["per_example"])
        out.put((rank, agent.w.clone().numpy(), losses, extra, int(agent.train_step_counter)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_replicas_equal_single_process_on_global_batch():
    world, steps = 2, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    res.sort(key=lambda r: r[0])
    # single process on the concatenated (global) batch
    x, y = _data(0, 16, 5)
    ref = LinearAgent(5)
    lrn = learner.Learner(None, ref.train_step_counter, ref,
                          experience_dataset_fn=lambda: iter(lambda: ((x, y), "info"), None))
    assert type(lrn.strategy) is strategy_utils.Strategy and ref.gradient_hook is None
    ref_losses = [float(lrn.run(iterations=1).loss) for _ in range(steps)]
    w0, w1 = res[0][1], res[1][1]
    np.testing.assert_array_equal(w0, w1)                      # replicas stay bit-identical
    np.testing.assert_allclose(w0, ref.w.numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(res[0][2], ref_losses, rtol=1e-5)   # SUM over replicas == global
    np.testing.assert_allclose(res[1][2], ref_losses, rtol=1e-5)
    np.testing.assert_allclose(res[0][3], ref_losses[-1], rtol=1e-5)  # extra reduced over axes too
    assert res[0][4] == steps and res[1][4] == steps


def test_single_replica_strategy_is_identity():
    s = strategy_utils.get_strategy()
    assert s.num_replicas_in_sync == 1
    t = torch.ones(3)
    assert s.all_reduce_sum_(t) is t and torch.equal(s.reduce_sum(t), t)
    with pytest.raises(RuntimeError):
        strategy_utils.DataParallelStrategy()


def test_learner_uses_the_agents_precomputed_field_sums():
    """An agent whose loss kernel already holds sum-over-all-axes of its LossInfo fields hands them
    to Learner.run through `reduce_loss_info` (DqnAgent does); any other LossInfo takes the generic
    reduction.  Both give the reference's SUM semantics (learner.py:322-337)."""
    class HookAgent(LinearAgent):
        def __init__(self, dim):
            super().__init__(dim)
            self.last = None
            self.hook_calls = 0

        def train(self, experience):
            li = super().train(experience)
            self.last = li
            self.sums = li.extra["per_example"].sum()
            return li

        def reduce_loss_info(self, loss_info):
            self.hook_calls += 1
            if loss_info is self.last:
                return tf_agent.LossInfo(loss_info.loss, {"per_example": self.sums})
            return None

    x, y = _data(3, 8, 4)
    agent = HookAgent(4)
    lrn = learner.Learner(None, agent.train_step_counter, agent,
                          experience_dataset_fn=lambda: iter(lambda: ((x, y), "info"), None))
    li = lrn.run(iterations=1)
    assert agent.hook_calls == 1
    assert li.extra["per_example"].dim() == 0
    np.testing.assert_allclose(float(li.extra["per_example"]), float(li.loss), rtol=1e-6)
    other = lrn.loss(((x, y), "info"))          # not the train step's LossInfo: generic path
    assert agent.hook_calls == 2 and other.loss.dim() == 0
