"""GPU: entry points and paths added in round 5, each against what it replaces.

  aa_rmsprop_step_slabs_pack   the LossInfo sums Learner.run returns (train/learner.py:322-337) leave
                               with the optimizer launch: same values as the copy launch it replaces,
                               storage of their own, identical training
  aa_rb_scatter_rows_count     the driver's step counter inside the replay add launch
"""
import numpy as np
import pytest
import torch

import bench
from agents_amd.agents.dqn import dqn_agent
from agents_amd.utils import common, graph

pytestmark = pytest.mark.gpu


def _loop(dev, iters, overlap=True):
    w = bench.build_workload(dev, 0, 1, 64, 16, 64, seed=3)
    w["init_driver"]._num_steps = 64 * 16
    w["init_driver"].run()
    it = iter(w["dataset"])
    collect_run = common.function(w["collect_driver"].run)
    if overlap:
        graph.enable_overlap(dev)
    ts_, out = None, []
    try:
        for _ in range(iters):
            ts_, _ = collect_run(ts_)
            out.append(w["learner"].run(iterations=1, iterator=it))
        graph.join_lanes(dev)
        torch.cuda.synchronize()
    finally:
        graph.disable_overlap(dev)
    vals = [(float(li.loss), float(li.extra.td_loss), float(li.extra.td_error)) for li in out]
    return w, out, vals


def test_loss_info_packed_by_the_optimizer_launch(dev, monkeypatch):
    monkeypatch.setattr(dqn_agent, "PACK_IN_OPTIMIZER", True)
    w1, infos, vals = _loop(dev, 12)
    assert w1["agent"].reduced_owns_storage, "the optimizer launch did not pack the LossInfo"
    # every returned LossInfo owns its storage: the values read AFTER the whole run are per step
    assert len({v[0] for v in vals}) > 6
    ptrs = {li.loss.data_ptr() for li in infos}
    assert len(ptrs) == len(infos)
    assert all(li.loss.dim() == 0 and li.extra.td_loss.dim() == 0 for li in infos)
    monkeypatch.setattr(dqn_agent, "PACK_IN_OPTIMIZER", False)
    w2, _, vals2 = _loop(dev, 12)
    assert not w2["agent"].reduced_owns_storage
    assert vals == vals2                       # same sums, bit for bit, from the copy launch
    assert torch.equal(w1["net"].flat_params, w2["net"].flat_params)
    # sums are sums: against the per-sample buffers of the last step
    wk = w1["agent"]._work[64]
    np.testing.assert_allclose(vals[-1][1], float(wk.td_loss.sum()), rtol=1e-5)


def test_step_count_inside_the_add_batch_launch_is_bit_identical(dev, monkeypatch):
    """aa_rb_scatter_rows_count: the driver's loop counter as an extra workgroup of the replay
    buffer's add launch == aa_count_steps + aa_rb_scatter_rows (same totals posted, same number of
    loop bodies, same replay contents), also when runs need make-up bodies."""
    from agents_amd.drivers import dynamic_step_driver

    def run(fused, num_steps):
        monkeypatch.setattr(graph, "COUNT_IN_ADD", fused)
        w = bench.build_workload(dev, 0, 1, 32, 24, 32, seed=5)
        env, rb, agent = w["env"], w["rb"], w["agent"]
        env._p_end = 0.3                      # many boundary steps: runs need make-up bodies
        drv = dynamic_step_driver.DynamicStepDriver(env, agent.collect_policy,
                                                    observers=[rb.add_batch], num_steps=num_steps)
        run_fn = common.function(drv.run)
        ts_ = None
        adds = []
        for _ in range(12):
            ts_, _ = run_fn(ts_)
            adds.append(rb._get_last_id())
        torch.cuda.synchronize()
        g = graph.graphed_driver_run(drv)
        return adds, [t.clone() for t in rb._data_table.variables()], \
            int(g._total.item()), g.replays

    for num_steps in (1, 32, 40):
        a1, t1, tot1, rep1 = run(True, num_steps)
        a0, t0, tot0, rep0 = run(False, num_steps)
        assert a1 == a0 and tot1 == tot0 and rep1 == rep0 and rep1 > 0, num_steps
        for x, y in zip(t1, t0):
            assert torch.equal(x, y)
    assert a1[-1] > 12        # num_steps = 40 with 32 envs: at least two bodies per run


def test_ppo_collect_loop_replayed_as_graphs_equals_the_eager_loop(dev):
    """tools/bench_ppo.py now runs the PPO collect loop through common.function(driver.run) -- HIP
    graph replays of the loop body, launched without a host wait per body (utils/graph.py) --: the
    replay buffer must hold exactly what the eager driver loop leaves, over several iterations
    with boundary steps (make-up bodies) and `clear()` in between."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "tools"))
    import bench_ppo
    from agents_amd.utils import nest_utils

    def collect(graphed):
        w = bench_ppo.build(dev, envs=64, steps=12, minibatch=64, epochs=1,
                            episode_end_probability=0.05)
        run = w["collect"] if graphed else w["collect_driver"].run
        ts_, out = None, []
        for _ in range(4):
            w["rb"].clear()
            ts_, _ = run(ts_)
            torch.cuda.synchronize()
            out.append([t.clone() for t in nest_utils.flatten(w["rb"].gather_all())])
        return out

    a, b = collect(True), collect(False)
    for it, (xa, xb) in enumerate(zip(a, b)):
        assert len(xa) == len(xb)
        for ta, tb in zip(xa, xb):
            assert ta.shape == tb.shape and torch.equal(ta, tb), it
