"""GPU: checkpoint / resume (utils/common.Checkpointer as in
tf_agents/agents/dqn/examples/v2/train_eval.py:214-232,317-324 and train/learner.py:206-243):
a run restored from a checkpoint continues BIT-IDENTICALLY to the uninterrupted run -- parameters,
optimizer slots, replay contents, random streams, counters -- also through the HIP-graph path."""
import os

import pytest
import torch

from agents_amd.train import learner
from agents_amd.utils import common, graph
from tests.test_gpu_graphs import _stack

pytestmark = pytest.mark.gpu


def _loop(stack, run, lrn, it, n):
    out = None
    for _ in range(n):
        run()
        out = lrn.run(iterations=1, iterator=it)
    return out


def _build(dev, root=None):
    env, agent, rb, drv, net = _stack(dev, 8, 64, 0.2, 8)
    run = common.function(drv.run)
    lrn = learner.Learner(root, common.Variable(0, name="train_step"), agent,
                          checkpoint_interval=10)
    return dict(env=env, agent=agent, rb=rb, drv=drv, net=net, run=run, lrn=lrn)


def test_resume_is_bit_identical(dev, tmp_path):
    a = _build(dev)
    for _ in range(4):
        a["run"]()
    it_a = iter(a["rb"].as_dataset(sample_batch_size=16, num_steps=2))
    _loop(a, a["run"], a["lrn"], it_a, 15)
    ckpt = common.Checkpointer(str(tmp_path / "ck"), agent=a["agent"], replay_buffer=a["rb"],
                               env=a["env"], train_step=a["agent"].train_step_counter)
    ckpt.save(int(a["agent"].train_step_counter))
    li_a = _loop(a, a["run"], a["lrn"], it_a, 12)

    b = _build(dev)            # fresh process stand-in: new objects, then restore
    for _ in range(2):         # graphs / buffers may already exist when a checkpoint is loaded
        b["run"]()
    ck_b = common.Checkpointer(str(tmp_path / "ck"), agent=b["agent"], replay_buffer=b["rb"],
                               env=b["env"], train_step=b["agent"].train_step_counter)
    assert ck_b.checkpoint_exists
    assert int(b["agent"].train_step_counter) == 15
    it_b = iter(b["rb"].as_dataset(sample_batch_size=16, num_steps=2))
    li_b = _loop(b, b["run"], b["lrn"], it_b, 12)
    assert float(li_a.loss) == float(li_b.loss)
    assert torch.equal(a["net"].flat_params, b["net"].flat_params)
    assert torch.equal(a["agent"]._target_q_network.flat_params,
                       b["agent"]._target_q_network.flat_params)
    for va, vb in zip(a["rb"].variables(), b["rb"].variables()):
        assert torch.equal(va, vb)
    assert a["agent"]._optimizer.iterations == b["agent"]._optimizer.iterations == 27
    for sa, sb in zip(a["agent"]._optimizer._slots.values(),
                      b["agent"]._optimizer._slots.values()):
        for k in sa:
            assert torch.equal(sa[k], sb[k]), k


def test_learner_checkpoints_and_restores(dev, tmp_path):
    root = str(tmp_path / "root")
    a = _build(dev, root)
    for _ in range(4):
        a["run"]()
    it = iter(a["rb"].as_dataset(sample_batch_size=16, num_steps=2))
    _loop(a, a["run"], a["lrn"], it, 25)
    files = sorted(os.listdir(os.path.join(root, "train")))
    assert files == ["ckpt-10.pt", "ckpt-20.pt"]
    b = _build(dev, root)      # constructing the Learner restores the latest checkpoint
    assert int(b["agent"].train_step_counter) == 20
    assert b["lrn"].train_step_numpy == 20
    assert not torch.equal(b["net"].flat_params, _build(dev)["net"].flat_params)
