"""GPU: the loop `bench.py --replay prioritized` times -- collect -> prioritized get_next -> DqnAgent
train -> update_priorities(td_error) through Learner(after_train_strategy_step_fn=...) -- at the
benchmark's shapes (256 envs, Atari frames, batch 256, num_steps 2, the Mnih-15 Q-network), with the
sampler checked against oracle/prioritized.py at EVERY iteration on the state the loop itself
produced: rows and probabilities bit-exact from (priorities, stored ids, last_id, Philox call),
new rows entering at the running maximum, and the fed-back priorities equal to the quantised
|td_error| of the batch that was trained on.  The reference has no prioritized buffer (parity
unpinned, see the oracle's header); its hooks are the ones used: tf_agents/train/learner.py:362-376,
agents/dqn/dqn_agent.py:50-72."""
import numpy as np
import pytest
import torch

import bench
from agents_amd.utils import common
from oracle import prioritized as op

pytestmark = pytest.mark.gpu

B_ENV, L_RING, S, ITERS = 256, 12, 256, 14


def _pq(rb):
    return (rb._prio_q.cpu().numpy().astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)


def test_prioritized_loop_matches_the_oracle_every_iteration(dev):
    w = bench.build_workload(dev, 0, 1, B_ENV, L_RING, S, seed=1, replay="prioritized")
    rb, lrn, agent = w["rb"], w["learner"], w["agent"]
    seed = rb._seed
    w["init_driver"]._num_steps = B_ENV * 5          # not full yet: the loop crosses the wrap
    w["init_driver"].run()
    it = iter(w["dataset"])
    collect_run = common.function(w["collect_driver"].run)
    time_step = None
    updated_total = 0
    for k in range(ITERS):
        max_q = int(rb._max_prio_q.cpu().numpy().astype(np.int64)[0] & 0xFFFFFFFF)
        last_before = rb._get_last_id()
        time_step, _ = collect_run(time_step)
        torch.cuda.synchronize()
        assert rb._get_last_id() == last_before + 1
        pq = _pq(rb)
        # the rows this collect step wrote entered with the running maximum priority
        new_rows = np.arange(B_ENV) * L_RING + (rb._get_last_id() % L_RING)
        assert (pq[new_rows] == max_q).all(), k
        ids = rb._id_table.variables()[0].cpu().numpy()
        call = rb._sample_calls
        want_rows, want_p, empty = op.sample(pq, ids, rb._get_last_id(), B_ENV, L_RING, S, 2,
                                             seed, call)
        assert not empty
        li = lrn.run(iterations=1, iterator=it)       # draws, trains, feeds td_error back
        torch.cuda.synchronize()
        assert np.isfinite(float(li.loss))
        np.testing.assert_array_equal(rb.last_sampled_rows.cpu().numpy(), want_rows[:, 0])
        assert rb._sample_calls == call + 1
        # priorities of the sampled rows = quantise(|td_error|) of THIS batch (duplicates: any of
        # the values written for the row; powf vs numpy power: one unit of 2^-16)
        td = agent._work[S].td_error.cpu().numpy()
        q_want = op.quantise(td, rb.priority_exponent, 1e-6).astype(np.int64)
        q_got = _pq(rb).astype(np.int64)
        by_row = {}
        for r, q in zip(want_rows[:, 0], q_want):
            by_row.setdefault(int(r), []).append(int(q))
        for r, qs in by_row.items():
            assert min(abs(int(q_got[r]) - q) for q in qs) <= 2, (k, r)
        updated_total += len(by_row)
        # rows that were neither sampled nor written keep their priority
        untouched = np.ones(pq.shape[0], bool)
        untouched[list(by_row)] = False
        np.testing.assert_array_equal(q_got[untouched], pq.astype(np.int64)[untouched])
    assert updated_total > ITERS * S // 2
    # the sampler's probabilities on the final state, through the public get_next
    pq, ids = _pq(rb), rb._id_table.variables()[0].cpu().numpy()
    call = rb._sample_calls
    _data, info = rb.get_next(sample_batch_size=S, num_steps=2)
    want_rows, want_p, _ = op.sample(pq, ids, rb._get_last_id(), B_ENV, L_RING, S, 2, seed, call)
    np.testing.assert_array_equal(info.probabilities.cpu().numpy(), want_p)
    np.testing.assert_array_equal(info.ids.cpu().numpy().reshape(S, 2), ids[want_rows])
