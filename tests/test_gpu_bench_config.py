"""GPU: parity ON THE BENCHMARKED CONFIGURATION (BASELINE.json configs[1] as bench.py builds it):
256 envs, batch 256, num_steps 2, Mnih-15 Q-network on uint8 84x84x4, Huber, centred RMSProp,
`prefetch(3)` dataset -- with the replay ring shortened to 8 frames per env so that the oracle's
tables fit in host memory and ring wrap-around happens inside the test.

Three stacks run side by side on identical seeds:
  oracle   numpy replay + torch-CPU DQN step (oracle/replay.py, oracle/dqn.py)
  eager    the HIP kernels launched one by one
  graphed  what bench.py times: the collect / sample / train HIP graphs on three streams
           (`graph.enable_overlap`), train graphs bound to the sampler's ring slots
Checks, per iteration: sampled rows and ids eager == oracle bit for bit; loss eager vs oracle to
1e-5 relative and the parameters after the step to 5e-5 x max|p| (dqn_agent.py:412-449 restated
in oracle/dqn.py), BOTH taken from the same pre-step parameters -- the oracle's parameters are
reset to the GPU's before every step, because a free-running pair drifts apart chaotically (ReLU
boundaries amplify last-bit differences of the fp32 sums: measured 4e-5 on the loss after 13
steps), which says nothing about either implementation; graphed == eager bit for bit (parameters,
loss, replay tables).  At the end: replay tables eager == oracle == graphed.
"""
import numpy as np
import pytest
import torch

import bench
from agents_amd.specs import tensor_spec
from agents_amd.utils import common, graph, nest_utils
from oracle import dqn as odqn
from oracle import nets as onets
from oracle import optim as ooptim
from oracle import replay as oreplay

pytestmark = pytest.mark.gpu

B_ENV, L_RING, S, ITERS = 256, 8, 256, 24
# Parameters after ONE optimizer step from identical parameters, relative to max|p| of the tensor
# (the tolerance of tests/test_gpu_dqn_agent.py's Atari case).
TOL_PARAM = 5e-5


def _stack(dev, eager):
    w = bench.build_workload(dev, 0, 1, B_ENV, L_RING, S, seed=1)
    if eager:
        w["rb"]._dataset_ring = 0
        w["learner"]._train_fn = w["agent"].train
    return w


def test_bench_configuration_matches_oracle_and_eager(dev):
    with torch.cuda.device(dev):
        w_e, w_g = _stack(dev, True), _stack(dev, False)
        net_e, net_g = w_e["net"], w_g["net"]
        assert torch.equal(net_e.flat_params, net_g.flat_params)
        # ---- oracle twin of the eager stack --------------------------------------------------
        olayers = onets.atari_q_layers(bench.NUM_ACTIONS)
        oparams = [torch.tensor(a) for a in net_e.get_weights()]
        oagent = odqn.OracleDqnAgent(olayers, bench.OBS_SHAPE, bench.NUM_ACTIONS, oparams,
                                     optimizer=ooptim.RMSprop(2.5e-4, 0.95, 0.95, 0.01, True),
                                     gamma=0.99, loss="huber", target_update_period=2500)
        rb_e, rb_g = w_e["rb"], w_g["rb"]
        flat_specs = nest_utils.flatten(w_e["agent"].collect_data_spec)
        orb = oreplay.OracleReplayBuffer([s.shape for s in flat_specs],
                                         [tensor_spec.as_numpy_dtype(s.dtype) for s in flat_specs],
                                         B_ENV, L_RING, seed=rb_e._seed)

        def record(traj):        # second observer of the eager drivers: the oracle's add_batch
            orb.add_batch([t.cpu().numpy() for t in nest_utils.flatten(traj)])

        for drv in (w_e["init_driver"], w_e["collect_driver"]):
            drv._observers = list(drv._observers) + [record]
        # ---- prefill (random policy), as bench.py does ---------------------------------------
        for w in (w_e, w_g):
            w["init_driver"]._num_steps = B_ENV * L_RING
            w["init_driver"].run()
        assert orb.last_id == rb_e._get_last_id() == L_RING - 1

        run_g = common.function(w_g["collect_driver"].run)
        it_g = iter(w_g["dataset"])
        graph.enable_overlap(dev)
        try:
            q = []
            ts_e = ts_g = None
            max_loss_rel = worst = 0.0
            for i in range(ITERS):
                # eager + oracle
                ts_e, _ = w_e["collect_driver"].run(ts_e)
                while len(q) <= 3:           # prefetch(3): four draws in flight
                    q.append((rb_e.get_next(S, 2), orb.get_next(S, 2)))
                (exp_e, info_e), (odata, oids, oprobs) = q.pop(0)
                for g_leaf, o_leaf in zip(nest_utils.flatten(exp_e), odata):
                    assert np.array_equal(g_leaf.cpu().numpy(), o_leaf), f"rows differ, step {i}"
                assert np.array_equal(info_e.ids.cpu().numpy(), oids)
                assert np.array_equal(info_e.probabilities.cpu().numpy(), oprobs)
                with torch.no_grad():       # same pre-step parameters on both sides
                    for ov, a in zip(oagent.params, net_e.get_weights()):
                        ov.copy_(torch.from_numpy(np.asarray(a)))
                li_e = w_e["agent"].train(exp_e)
                o_st, o_obs, o_act, o_nst, o_rew, o_disc = odata
                ototal, aux, _ = oagent.train(torch.from_numpy(o_obs), o_act, o_rew, o_disc, o_st)
                got, want = float(li_e.loss), float(ototal)
                np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-7,
                                           err_msg=f"loss at step {i}")
                max_loss_rel = max(max_loss_rel, abs(got - want) / max(abs(want), 1e-12))
                for v, ov in zip(net_e.variables, oagent.params):
                    scale = max(float(ov.detach().abs().max()), 1e-12)
                    err = float((v.cpu() - ov.detach()).abs().max()) / scale
                    worst = max(worst, err)
                    assert err <= TOL_PARAM, f"param mismatch {err:.2e} after step {i}"
                # graphed (the timed configuration)
                ts_g, _ = run_g(ts_g)
                li_g = w_g["learner"].run(iterations=1, iterator=it_g)
                if i % 5 == 4 or i == ITERS - 1:
                    graph.join_lanes(dev)
                    assert torch.equal(net_e.flat_params, net_g.flat_params), f"step {i}"
                    assert float(li_g.loss) == got
                    for a, b in zip(ts_e, ts_g):
                        assert torch.equal(a, b)
            graph.join_lanes(dev)
            torch.cuda.synchronize()
        finally:
            graph.disable_overlap()
        gt = graph.graphed_train(w_g["agent"])
        assert gt.replays == ITERS - 2 and run_g.replays == ITERS - 2
        bound = next(iter(gt._cache.values()))
        assert len(bound) == 8 and None not in bound     # one train graph per ring slot, no copies
        # replay tables: graphed == eager == oracle, bit for bit (ring wrapped ITERS/8 times)
        assert rb_e._get_last_id() == rb_g._get_last_id() == orb.last_id == L_RING + ITERS - 1
        for va, vb in zip(rb_e.variables(), rb_g.variables()):
            assert torch.equal(va, vb)
        for tab, otab in zip(rb_e._data_table.variables(), orb.tables):
            got = tab.cpu().numpy().reshape(-1)
            assert np.array_equal(got.view(np.uint8), otab.reshape(-1).view(np.uint8))
        assert np.array_equal(rb_e._id_table.variables()[0].cpu().numpy(), orb.id_table)
        print(f"bench-config parity: max loss rel err {max_loss_rel:.2e}, max one-step param err "
              f"{worst:.2e} of max|p| over {ITERS} steps")
