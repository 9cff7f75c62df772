"""GPU: entry points and paths added in round 5, each against what it replaces.

  aa_rmsprop_step_slabs_pack   the LossInfo sums Learner.run returns (train/learner.py:322-337) leave
                               with the optimizer launch: same values as the copy launch it replaces,
                               storage of their own, identical training
  EARLY_TARGET = "main"        the early target forward in stream order on the caller's stream
                               (utils/graph.py): bit-identical to the side-stream variant
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


def test_early_target_forward_in_stream_order_is_bit_identical(dev, monkeypatch):
    monkeypatch.setattr(graph, "EARLY_TARGET", "side")
    w1, _, vals1 = _loop(dev, 14)
    g1 = graph.graphed_train(w1["agent"])
    monkeypatch.setattr(graph, "EARLY_TARGET", "main")
    w2, _, vals2 = _loop(dev, 14)
    g2 = graph.graphed_train(w2["agent"])
    assert g1.early_hits > 4 and g2.early_hits > 4
    assert vals1 == vals2
    assert torch.equal(w1["net"].flat_params, w2["net"].flat_params)
    assert torch.equal(w1["agent"]._target_q_network.flat_params,
                       w2["agent"]._target_q_network.flat_params)


def test_gradient_phase_split_in_front_of_the_first_layers_weight_gradient(dev, monkeypatch):
    """AA_SPLIT_LAST_DW: the gradient phase as two graphs (everything down to the second layer |
    the first layer's weight gradient), the early target forward of the next step started
    between them: the same training, bit for bit, slabs of all three conv layers kept."""
    monkeypatch.setattr(dqn_agent, "SPLIT_LAST_DW", False)
    w1, _, vals1 = _loop(dev, 16)
    g1 = graph.graphed_train(w1["agent"])
    monkeypatch.setattr(dqn_agent, "SPLIT_LAST_DW", True)
    w2, _, vals2 = _loop(dev, 16)
    g2 = graph.graphed_train(w2["agent"])
    ents = [e for b in g2._cache.values() for e in b.values()]
    assert any(e.split_last for e in ents), "no entry was recorded in two halves"
    assert not any(e.split_last for b in g1._cache.values() for e in b.values())
    split = [e for e in ents if e.split_last]
    n_kept = {e.apply_state[0].n for b in g1._cache.values() for e in b.values()
              if e.apply_state is not None}
    assert n_kept and all(e.apply_state is not None and {e.apply_state[0].n} == n_kept
                          for e in split), \
        "the optimizer does not get the slabs the unsplit backward leaves"
    assert g1.early_hits > 4 and g2.early_hits > 4
    assert vals1 == vals2
    assert torch.equal(w1["net"].flat_params, w2["net"].flat_params)
    assert torch.equal(w1["agent"]._target_q_network.flat_params,
                       w2["agent"]._target_q_network.flat_params)


def test_actions_selected_by_the_head_launch_equal_the_select_launch(dev, monkeypatch):
    """aa_dense_small_forward_slabs_eps: the Q head's launch draws the epsilon-greedy actions of its
    own Q values -- bit-identical to head launch + aa_eps_greedy_action, with and without an action
    mask, the Philox call counter advancing alike (policies/epsilon_greedy_policy.py:120-143)."""
    from agents_amd import ops
    from agents_amd.policies import q_policy
    from agents_amd.specs import tensor_spec
    from agents_amd.trajectories import time_step as ts
    monkeypatch.setattr(q_policy, "FUSE_SELECT", True)      # (opt-in: slower inside the DQN loop)
    rng = np.random.default_rng(8)
    M, K, H, A = 256, 3136, 512, 6
    x = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(dev)
    w1 = torch.from_numpy((rng.standard_normal((K, H)) * 0.02).astype(np.float32)).to(dev)
    b1 = torch.from_numpy(rng.standard_normal(H).astype(np.float32)).to(dev)
    w2 = torch.from_numpy((rng.standard_normal((H, A)) * 0.05).astype(np.float32)).to(dev)
    b2 = torch.from_numpy(rng.standard_normal(A).astype(np.float32)).to(dev)
    mask = torch.from_numpy((rng.random((M, A)) > 0.4).astype(np.int32)).to(dev)
    mask[:, 0] = 1
    aspec = tensor_spec.BoundedTensorSpec((), torch.int64, 0, A - 1)
    tss = ts.time_step_spec(tensor_spec.TensorSpec((4,), torch.float32))
    for use_mask in (False, True):
        for eps in (0.0, 0.3, 1.0):
            pols = [q_policy._DiscretePolicy(tss, aspec, q_network=None, epsilon=eps, seed=77)
                    for _ in range(2)]
            for call in range(3):
                h, y = torch.empty(M, H, device=dev), torch.empty(M, A, device=dev)
                ops.dense_tail_forward(x, w1, b1, "relu", h, w2, b2, None, y)
                want = pols[0].select(y, mask if use_mask else None, eps)
                monkeypatch.setattr(pols[1], "_q_network", type("N", (), {"selected": False})())
                sel = pols[1]._select_args(M, mask if use_mask else None, eps, dev)
                h2, y2 = torch.empty_like(h), torch.empty_like(y)
                done = ops.dense_tail_forward(x, w1, b1, "relu", h2, w2, b2, None, y2, select=sel)
                assert done is True
                assert torch.equal(y, y2) and torch.equal(h, h2)
                assert torch.equal(want, sel["out"]), (use_mask, eps, call)
                assert int(pols[0]._call_counter[0]) == int(pols[1]._call_counter[0])
            if eps > 0:
                assert int(pols[1]._call_counter[0]) == 3
    # greedy with an all-but-one mask: the only allowed action
    one = torch.zeros((M, A), dtype=torch.int32, device=dev)
    one[:, 4] = 1
    pol = q_policy._DiscretePolicy(tss, aspec, q_network=None, epsilon=0.5, seed=1)
    monkeypatch.setattr(pol, "_q_network", type("N", (), {"selected": False})())
    sel = pol._select_args(M, one, 0.5, dev)
    ops.dense_tail_forward(x, w1, b1, "relu", torch.empty(M, H, device=dev), w2, b2, None,
                           torch.empty(M, A, device=dev), select=sel)
    assert bool((sel["out"] == 4).all())


def test_collect_loop_with_and_without_the_fused_selection_is_bit_identical(dev, monkeypatch):
    from agents_amd.policies import q_policy
    monkeypatch.setattr(q_policy, "FUSE_SELECT", True)
    w1, _, vals1 = _loop(dev, 10)
    assert w1["net"].selected is not None
    monkeypatch.setattr(q_policy, "FUSE_SELECT", False)
    w2, _, vals2 = _loop(dev, 10)
    assert vals1 == vals2
    assert torch.equal(w1["net"].flat_params, w2["net"].flat_params)
    for a, b in zip(w1["rb"]._data_table.variables(), w2["rb"]._data_table.variables()):
        assert torch.equal(a, b)          # the replay holds the same actions and frames


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


def test_dense_input_and_weight_gradient_in_one_launch(dev, monkeypatch):
    """aa_gemm_f32_pair: fc1's dX (with the ReLU mask) and dW (+ bias gradient) as one launch ==
    the two launches, bit for bit; shapes the library does not group are refused (False) and left
    untouched."""
    from agents_amd import ops
    rng = np.random.default_rng(12)
    M, K, N = 256, 3136, 512
    r = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(dev)
    dz, w, x = r(M, N), r(K, N) * 0.05, torch.relu(r(M, K))
    dx0, dw0, bg0 = torch.empty(M, K, device=dev), torch.empty(K, N, device=dev), \
        torch.empty(N, device=dev)
    ops.dense_dx(dz, w, dx0, mask_src=x, mask_act="relu")
    ops.dense_dw(x, dz, dw0, bias_grad=bg0)
    dx1, dw1, bg1 = torch.full_like(dx0, float("nan")), torch.full_like(dw0, float("nan")), \
        torch.full_like(bg0, float("nan"))
    monkeypatch.setattr(ops, "GROUP_DENSE_BWD", True)
    assert ops.dense_dx_dw(dz, w, dx1, x, dw1, mask_src=x, mask_act="relu", bias_grad=bg1) is True
    assert torch.equal(dx0, dx1) and torch.equal(dw0, dw1) and torch.equal(bg0, bg1)
    ref = (dz.double() @ w.double().T) * (x > 0)
    assert float((dx1.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    # a pair that plans otherwise (a split-K weight gradient): refused, outputs untouched
    M2, K2, N2 = 4096, 64, 64
    dz2, w2, x2 = r(M2, N2), r(K2, N2), r(M2, K2)
    dxn, dwn = torch.full((M2, K2), 7.0, device=dev), torch.full((K2, N2), 7.0, device=dev)
    assert ops.dense_dx_dw(dz2, w2, dxn, x2, dwn) is False
    assert bool((dxn == 7.0).all()) and bool((dwn == 7.0).all())
    # the whole loop with and without the grouping: identical training
    w_a, _, vals_a = _loop(dev, 8)
    monkeypatch.setattr(ops, "GROUP_DENSE_BWD", False)
    w_b, _, vals_b = _loop(dev, 8)
    assert vals_a == vals_b and torch.equal(w_a["net"].flat_params, w_b["net"].flat_params)


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
