"""DynamicEpisodeDriver (tf_agents/drivers/dynamic_episode_driver.py) and PPOLearner
(tf_agents/train/ppo_learner.py) on the GPU: loop termination semantics, minibatch pipeline,
constructor validation."""
import numpy as np
import pytest
import torch

from agents_amd import optimizers
from agents_amd.agents.ppo import ppo_actor_network as pan
from agents_amd.agents.ppo import ppo_clip_agent
from agents_amd.drivers import dynamic_episode_driver, dynamic_step_driver
from agents_amd.environments import random_tf_environment
from agents_amd.replay_buffers import tf_uniform_replay_buffer as rb_lib
from agents_amd.specs import tensor_spec
from agents_amd.train import ppo_learner
from agents_amd.trajectories import time_step as ts
from agents_amd.utils import common

pytestmark = pytest.mark.gpu

OBS = tensor_spec.BoundedTensorSpec((17,), torch.float32, -1.0, 1.0)
ACT = tensor_spec.BoundedTensorSpec((6,), torch.float32, -1.0, 1.0)
TSS = ts.time_step_spec(OBS)


def make_agent(**kw):
    actor = pan.PPOActorNetwork().create_sequential_actor_net((32, 32), ACT, seed=1)
    value = pan.value_network((32, 32), "tanh", seed=2)
    kw.setdefault("normalize_observations", False)
    kw.setdefault("normalize_rewards", False)
    return ppo_clip_agent.PPOClipAgent(TSS, ACT, optimizers.Adam(3e-4, epsilon=1e-5),
                                       actor_net=actor, value_net=value,
                                       importance_ratio_clipping=0.2, use_gae=True, **kw)


@pytest.mark.parametrize("B,num_episodes", [(1, 3), (16, 40), (64, 10)])
def test_episode_driver_stops_after_num_episodes(dev, B, num_episodes):
    env = random_tf_environment.RandomTFEnvironment(TSS, ACT, batch_size=B,
                                                    episode_end_probability=0.2, seed=B, device=dev)
    agent = make_agent()
    seen = []
    drv = dynamic_episode_driver.DynamicEpisodeDriver(
        env, agent.collect_policy, observers=[lambda tr: seen.append(tr.step_type.clone())],
        num_episodes=num_episodes)
    final_ts, _ = drv.run()
    boundaries = [int((s == 2).sum().item()) for s in seen]
    total = sum(boundaries)
    # loop ran while sum(counter) < num_episodes: done after the last iteration, not before it
    assert total >= num_episodes
    assert total - boundaries[-1] < num_episodes
    assert final_ts.step_type.shape == (B,)
    # maximum_iterations caps the loop
    seen.clear()
    drv.run(num_episodes=10 ** 6, maximum_iterations=5)
    assert len(seen) == 5


def collect_into_replay(agent, dev, B, T):
    env = random_tf_environment.RandomTFEnvironment(TSS, ACT, batch_size=B,
                                                    episode_end_probability=0.05, seed=3,
                                                    device=dev)
    rb = rb_lib.TFUniformReplayBuffer(agent.collect_data_spec, batch_size=B, max_length=T + 1,
                                      device=dev)
    drv = dynamic_step_driver.DynamicStepDriver(env, agent.collect_policy,
                                                observers=[rb.add_batch], num_steps=B * (T + 1))
    drv.run()
    return rb


def test_ppo_learner_minibatches(dev):
    agent = make_agent(num_epochs=1, compute_value_and_advantage_in_train=False,
                       update_normalizers_in_train=False, gradient_clipping=0.5)
    agent.initialize()
    B, T = 32, 15
    rb = collect_into_replay(agent, dev, B, T)
    assert "value_prediction" in agent.collect_data_spec.policy_info  # computed at collect time

    def dataset_fn():
        # TFUniformReplayBuffer itself rejects sequence_preprocess_fn, like the reference
        # (tf_uniform_replay_buffer.py:329-367); preprocess with a dataset map instead
        return rb.as_dataset(sample_batch_size=B, num_steps=T + 1).map(
            lambda traj, info: (agent.preprocess_sequence(traj), info))

    shapes = []
    train_step = common.Variable(0, name="train_step")
    lrn = ppo_learner.PPOLearner(
        None, train_step, agent, experience_dataset_fn=dataset_fn,
        normalization_dataset_fn=dataset_fn, num_samples=2, num_epochs=3, minibatch_size=100,
        shuffle_buffer_size=B * (T + 1) * 2,
        after_train_strategy_step_fn=lambda e, l: shapes.append(tuple(e[0].discount.shape)))
    before = agent.flat_params.clone()
    li = lrn.run()
    frames = 2 * B * (T + 1)
    steps = (frames // 100) * 3
    assert lrn.num_frames_for_training == frames
    assert len(shapes) == steps and all(s == (100, 1) for s in shapes)
    assert int(agent.train_step_counter.numpy()) == steps
    assert lrn.train_step_numpy == steps
    assert torch.isfinite(li.loss).item() and not torch.equal(before, agent.flat_params)
    # the preprocessed elements carry return / advantage / value_prediction
    tr, _ = next(iter(dataset_fn()))
    assert set(tr.policy_info) == {"dist_params", "value_prediction", "return", "advantage"}
    assert tr.policy_info["return"].shape == (B, T + 1)
    assert float(tr.policy_info["return"][:, -1].abs().max()) == 0.0  # padded last step


def test_ppo_learner_full_sequences(dev):
    agent = make_agent(num_epochs=2, compute_value_and_advantage_in_train=True,
                       update_normalizers_in_train=False)
    agent.initialize()
    B, T = 8, 7
    rb = collect_into_replay(agent, dev, B, T)
    fn = lambda: rb.as_dataset(sample_batch_size=B, num_steps=T + 1)
    lrn = ppo_learner.PPOLearner(None, common.Variable(0), agent, fn, fn, num_samples=3,
                                 num_epochs=2)
    lrn.run()
    assert int(agent.train_step_counter.numpy()) == 3 * 2 * 2  # samples x epochs x agent epochs


def test_ppo_learner_validation(dev):
    fn = lambda: None
    a = make_agent(update_normalizers_in_train=False, compute_value_and_advantage_in_train=False)
    with pytest.raises(ValueError, match="shuffle_buffer_size must be provided"):
        ppo_learner.PPOLearner(None, common.Variable(0), a, fn, fn, 1, minibatch_size=4)
    a2 = make_agent(update_normalizers_in_train=False)
    with pytest.raises(ValueError, match="compute_value_and_advantage_in_train"):
        ppo_learner.PPOLearner(None, common.Variable(0), a2, fn, fn, 1, minibatch_size=4,
                               shuffle_buffer_size=10)
    a3 = make_agent(compute_value_and_advantage_in_train=False)
    with pytest.raises(ValueError, match="update_normalizers_in_train"):
        ppo_learner.PPOLearner(None, common.Variable(0), a3, fn, fn, 1)


def test_fused_epochs_equal_one_train_call_per_minibatch(dev):
    """PPOLearner hands an agent that supports it (frames, permutation) and the whole epoch runs
    from one host call (PPOAgent.train_minibatches: the fused step reads its rows through the
    permutation, no gather launch).  Same bits as the reference structure -- one `agent.train` per
    gathered [minibatch, 1, ...] batch (tf_agents/train/ppo_learner.py:220-248), here forced by
    installing a (no-op) after_train_strategy_step_fn: parameters, optimizer slots, counters and
    the returned LossInfo."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "tools"))
    import bench_ppo
    with torch.cuda.device(dev):
        stacks = []
        for hook in (None, lambda *a: None):
            w = bench_ppo.build(dev, envs=64, steps=31, minibatch=256, epochs=2,
                                after_train_step_fn=hook, episode_end_probability=0.05)
            w["collect_driver"].run()
            li = w["learner"].run()
            torch.cuda.synchronize()
            stacks.append((w, li))
        (w_f, li_f), (w_s, li_s) = stacks
        a_f, a_s = w_f["agent"], w_s["agent"]
        n = (64 * 32 // 256) * 2
        from agents_amd.utils import graph
        assert graph.graphed_train(a_s).replays == n - 2      # per-step path went through train()
        assert getattr(graph.graphed_train(a_f), "replays", 0) == 0   # fused epochs: no train() calls
        assert int(a_f.train_step_counter.numpy()) == int(a_s.train_step_counter.numpy()) == n
        assert a_f._optimizer.iterations == a_s._optimizer.iterations == n
        assert torch.equal(a_f.flat_params, a_s.flat_params)
        assert torch.equal(a_f.flat_grads, a_s.flat_grads)
        for x, y in zip(a_f._optimizer.variables(), a_s._optimizer.variables()):
            assert torch.equal(x, y)
        assert float(li_f.loss) == float(li_s.loss)
        assert float(li_f.extra.clip_fraction) == float(li_s.extra.clip_fraction)
        assert w_f["learner"].train_step_numpy == w_s["learner"].train_step_numpy == n


@pytest.mark.parametrize("clip", [0.5, 0.0])
def test_merged_reduce_apply_launch_is_bit_identical(dev, clip):
    """The fused step's slab reduction and clip + Adam in ONE launch (a grid barrier that carries
    only the per-workgroup sums of squares, aa_ppo_fused_merge_apply) against the two-launch form:
    parameters, gradients, Adam slots, step counters, the global norm and the LossInfo after two
    epochs of minibatch steps, bit for bit (clip 0 = no clipping: the norm is still reported)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "tools"))
    import bench_ppo
    from agents_amd import _lib
    lib = _lib.load()
    before = lib.aa_ppo_fused_merge_apply(-1)
    try:
        with torch.cuda.device(dev):
            stacks = []
            for on in (1, 0):
                lib.aa_ppo_fused_merge_apply(on)
                w = bench_ppo.build(dev, envs=64, steps=31, minibatch=256, epochs=2,
                                    episode_end_probability=0.05)
                w["agent"]._gradient_clipping = clip
                w["collect_driver"].run()
                li = w["learner"].run()
                torch.cuda.synchronize()
                stacks.append((w, li))
    finally:
        lib.aa_ppo_fused_merge_apply(before)
    (w_m, li_m), (w_s, li_s) = stacks
    a_m, a_s = w_m["agent"], w_s["agent"]
    n = (64 * 32 // 256) * 2
    assert a_m._optimizer.iterations == a_s._optimizer.iterations == n
    assert torch.equal(a_m.flat_params, a_s.flat_params)
    assert torch.equal(a_m.flat_grads, a_s.flat_grads)
    assert bool(a_m.flat_grads.abs().sum() > 0)
    for x, y in zip(a_m._optimizer.variables(), a_s._optimizer.variables()):
        assert torch.equal(x, y)
    assert torch.equal(a_m._norm_sumsq, a_s._norm_sumsq) and float(a_m._norm_sumsq) > 0
    assert float(li_m.loss) == float(li_s.loss)
    for f in li_m.extra._fields:
        assert float(getattr(li_m.extra, f)) == float(getattr(li_s.extra, f)), f
