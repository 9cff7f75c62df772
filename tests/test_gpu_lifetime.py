"""GPU: the HIP graphs, static buffers and scratch of a graphed loop are released by REFERENCE COUNT
when their owner (agent, driver, dataset iterator) is dropped -- with Python's cyclic collector
switched off.

Round 5 ended with a hipGraphLaunch crash in the 671st test of the suite that only a
`gc.collect()` between test modules kept away: GraphedTrain <-> agent and GraphedDriverRun <->
driver were reference cycles (and a SAC agent sat in the replay hooks of its own graphs), so dead
agents' graphs lived until the collector happened to run.  Now the back-pointers are weak, a closed
graph's hipGraphExec is destroyed the next time the device is known to be idle
(agents_amd/utils/graph.py: `_GRAVEYARD`), and these tests pin it: nothing of a dropped loop is
alive, no recorded graph is live, and device memory (hipMemGetInfo) stays flat over 1,000 graphed
agents built and dropped in one process without the collector."""
import gc
import os
import sys
import weakref

import pytest
import torch

from agents_amd import optimizers
from agents_amd.agents.dqn import dqn_agent
from agents_amd.drivers import dynamic_step_driver
from agents_amd.environments import random_tf_environment
from agents_amd.networks import layers as L
from agents_amd.networks import sequential
from agents_amd.replay_buffers import tf_uniform_replay_buffer as rb_lib
from agents_amd.specs import tensor_spec
from agents_amd.train import learner
from agents_amd.trajectories import time_step as ts
from agents_amd.trajectories import trajectory
from agents_amd.utils import common, graph

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A = 4


@pytest.fixture
def no_collector():
    gc.collect()
    torch.cuda.synchronize()
    graph.release_dead()
    was = gc.isenabled()
    gc.disable()
    yield
    if was:
        gc.enable()


def _refs(objs):
    out = {}
    for k, v in objs.items():
        try:
            out[k] = weakref.ref(v)
        except TypeError:
            pass
    return out


def _specs():
    obs_spec = tensor_spec.TensorSpec((12, 12, 4), torch.uint8, "observation")
    aspec = tensor_spec.BoundedTensorSpec((), torch.int64, 0, A - 1, "action")
    return ts.time_step_spec(obs_spec), aspec


def _agent(seed=5):
    tss, aspec = _specs()
    net = sequential.Sequential([L.Rescale(255.0), L.Conv2D(8, 4, 4, "relu"), L.Flatten(),
                                 L.Dense(32, "relu"), L.Dense(A)], seed=3)
    agent = dqn_agent.DqnAgent(tss, aspec, q_network=net, optimizer=optimizers.Adam(1e-3),
                               td_errors_loss_fn=common.element_wise_huber_loss, gamma=0.9,
                               epsilon_greedy=0.3, target_update_period=3, seed=seed)
    agent.initialize()
    return agent, net, tss, aspec


def _dqn_loop(dev, iters=6, root_dir=None):
    """Driver + replay + prefetching dataset + Learner, all three programs through their graphs,
    on the overlap lanes.  Returns weak references to everything it built."""
    B = 8
    agent, net, tss, aspec = _agent()
    env = random_tf_environment.RandomTFEnvironment(tss, aspec, batch_size=B,
                                                    episode_end_probability=0.1, seed=11,
                                                    device=dev)
    rb = rb_lib.TFUniformReplayBuffer(agent.collect_data_spec, batch_size=B, max_length=64,
                                      device=dev, seed=9)
    drv = dynamic_step_driver.DynamicStepDriver(env, agent.collect_policy,
                                                observers=[rb.add_batch], num_steps=B)
    graph.enable_overlap(dev)
    run = common.function(drv.run)
    t = None
    for _ in range(4):
        t, _ = run(t)
    it = iter(rb.as_dataset(sample_batch_size=16, num_steps=2).prefetch(3))
    lrn = learner.Learner(root_dir, common.Variable(0), agent, checkpoint_interval=10 ** 9)
    for _ in range(iters):
        t, _ = run(t)
        lrn.run(iterations=1, iterator=it)
    assert run.replays > 0 and graph.graphed_train(agent).replays > 0
    torch.cuda.synchronize()
    return _refs(dict(agent=agent, net=net, env=env, rb=rb, driver=drv, run=run, learner=lrn,
                      graphed_train=agent._graphed_train, iterator=it))


def _sac_loop(dev, iters=6):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_sac
    w = bench_sac.build(dev, envs=16, max_length=16, batch=8)
    graph.enable_overlap(dev)
    it = iter(w["dataset"])
    t = None
    for _ in range(iters):
        t, _ = w["collect"](t)
        w["learner"].run(iterations=1, iterator=it)
    assert graph.graphed_train(w["agent"]).replays > 0
    torch.cuda.synchronize()
    w["iterator"], w["graphed_train"] = it, w["agent"]._graphed_train
    return _refs(w)


def _ppo_loop(dev, iters=5):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_ppo
    w = bench_ppo.build(dev, envs=16, steps=8, minibatch=32, epochs=2)
    lrn, rb = w["learner"], w["rb"]
    t = None
    for _ in range(iters):
        rb.clear()
        t, _ = w["collect"](t)
        lrn._train_iter = lrn._norm_iter = None
        lrn.run()
    torch.cuda.synchronize()
    w.pop("raw_dataset_fn", None)
    return _refs(w)


@pytest.mark.parametrize("kind", ["dqn", "dqn_checkpointed", "sac", "ppo"])
def test_a_dropped_loop_is_released_without_the_collector(dev, no_collector, kind, tmp_path):
    live0 = graph.live_graphs()[0]
    if kind == "dqn":
        refs = _dqn_loop(dev)
    elif kind == "dqn_checkpointed":       # Learner -> CheckpointTrigger -> Checkpointer -> agent
        refs = _dqn_loop(dev, root_dir=str(tmp_path))
    elif kind == "sac":
        refs = _sac_loop(dev)
    else:
        refs = _ppo_loop(dev)
    alive = sorted(k for k, r in refs.items() if r() is not None)
    assert alive == [], f"only the cyclic collector would release: {alive}"
    assert graph.live_graphs()[0] == live0, "recorded graphs outlived their owners"
    graph.release_dead()
    assert graph.live_graphs() == (live0, 0)


def test_the_graphed_functions_point_back_weakly(dev, no_collector):
    """`common.function(agent.train)` / `common.function(driver.run)` belong to the agent / driver;
    used after their owner has gone they say so instead of touching freed buffers."""
    agent, net, tss, aspec = _agent()
    train = common.function(agent.train)
    assert train is graph.graphed_train(agent) and train.agent is agent
    env = random_tf_environment.RandomTFEnvironment(tss, aspec, batch_size=4, device=dev)
    drv = dynamic_step_driver.DynamicStepDriver(env, agent.collect_policy, num_steps=4)
    run = common.function(drv.run)
    assert run is common.function(drv.run)
    del agent, drv
    with pytest.raises(ReferenceError, match="agent of this graphed train function"):
        train(None)
    with pytest.raises(ReferenceError, match="driver of this graphed run function"):
        run()


def _train_only_agent(dev, exp):
    """The cheapest graphed agent: `common.function(agent.train)` on one static batch (the third
    call records the gradient and the optimizer graph)."""
    agent, _, _, _ = _agent()
    train = common.function(agent.train)
    for _ in range(5):
        train(exp)
    assert train.replays >= 2
    return weakref.ref(agent), weakref.ref(train)


def test_a_thousand_graphed_agents_in_one_process(dev, no_collector):
    """Build and drop graphed agents with the collector off; every tenth one is a whole loop
    (driver bodies, sampler ring, per-slot train graphs, early target forwards, lanes).  Nothing
    stays alive, the graveyard never exceeds its bound, and hipMemGetInfo stays flat."""
    n_agents = int(os.environ.get("AA_LIFETIME_AGENTS", "1000"))
    B, T = 16, 2
    g = torch.Generator(device="cpu").manual_seed(0)
    exp = trajectory.Trajectory(
        step_type=torch.ones((B, T), dtype=torch.int32, device=dev),
        observation=torch.randint(0, 256, (B, T, 12, 12, 4), generator=g,
                                  dtype=torch.uint8).to(dev),
        action=torch.randint(0, A, (B, T), generator=g).to(dev),
        policy_info=(),
        next_step_type=torch.ones((B, T), dtype=torch.int32, device=dev),
        reward=torch.rand((B, T), generator=g).to(dev),
        discount=torch.ones((B, T), dtype=torch.float32, device=dev))
    live0 = graph.live_graphs()[0]
    free_at = {}
    worst_graveyard = 0
    for i in range(n_agents):
        if i % 10 == 9:
            refs = _dqn_loop(dev, iters=4)
            assert all(r() is None for r in refs.values()), i
        else:
            ra, rt = _train_only_agent(dev, exp)
            assert ra() is None and rt() is None, i
        live, parked = graph.live_graphs()
        assert live == live0, (i, live)
        worst_graveyard = max(worst_graveyard, parked)
        if i in (n_agents // 10, n_agents - 1):
            torch.cuda.synchronize()
            graph.release_dead()
            free_at[i] = torch.cuda.mem_get_info()[0]
    assert worst_graveyard <= graph._GRAVEYARD_MAX + 64
    first, last = free_at[n_agents // 10], free_at[n_agents - 1]
    assert first - last < (64 << 20), \
        f"device memory shrank by {(first - last) >> 20} MiB between agent {n_agents // 10} " \
        f"and agent {n_agents}"
