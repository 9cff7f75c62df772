"""CPU tests of the host-side logic: nest ordering, specs, replay index logic (vs the oracle),
dataset combinators, trajectory helpers, Periodically, loud failure without a HIP device."""
import collections

import numpy as np
import pytest
import torch

from agents_amd import _lib
from agents_amd.replay_buffers import dataset as ds_lib
from agents_amd.replay_buffers import tf_uniform_replay_buffer as rb_lib
from agents_amd.specs import tensor_spec
from agents_amd.trajectories import policy_step
from agents_amd.trajectories import time_step as ts
from agents_amd.trajectories import trajectory
from agents_amd.utils import common, nest_utils
from oracle import dqn as odqn
from oracle import replay as oreplay


def test_nest_order_matches_tf_nest():
    """namedtuples in field order, dicts in sorted-key order (table.py:54-77)."""
    info = {"value_prediction": "v", "dist_params": {"scale": "s", "loc": "l"}}
    traj = trajectory.Trajectory("st", "obs", "act", info, "nst", "rew", "disc")
    assert nest_utils.flatten(traj) == ["st", "obs", "act", "l", "s", "v", "nst", "rew", "disc"]
    packed = nest_utils.pack_sequence_as(traj, list(range(9)))
    assert packed.policy_info == {"dist_params": {"loc": 3, "scale": 4}, "value_prediction": 5}
    assert nest_utils.flatten(trajectory.Trajectory(1, 2, 3, (), 4, 5, 6)) == [1, 2, 3, 4, 5, 6]
    with pytest.raises(ValueError, match="do not match"):
        nest_utils.assert_same_structure(traj, traj._replace(policy_info=()))
    with pytest.raises(ValueError):
        nest_utils.assert_same_structure((1, 2), [1, 2][:1])
    assert nest_utils.has_lists((1, [2])) and not nest_utils.has_lists((1, (2,)))


def test_specs():
    s = tensor_spec.BoundedTensorSpec((84, 84, 4), np.uint8, 0, 255, "obs")
    assert s.dtype == torch.uint8 and s.row_bytes == 28224 and s.num_elements == 28224
    assert tensor_spec.TensorSpec((), torch.int64).row_bytes == 8
    assert s == tensor_spec.BoundedTensorSpec((84, 84, 4), torch.uint8, 0, 255)
    assert s != tensor_spec.BoundedTensorSpec((84, 84, 4), torch.uint8, 0, 254)
    t = ts.time_step_spec(s)
    assert t.step_type.dtype == torch.int32 and t.discount.maximum == 1.0
    with pytest.raises(ValueError):
        tensor_spec.BoundedTensorSpec((2,), torch.float32, [0, 0, 0], 1)


def test_valid_range_matches_oracle():
    for L in (1, 3, 10):
        for last in range(-1, 3 * L + 2):
            for T in (None, 1, 2, 3):
                assert rb_lib._valid_range_ids(last, L, T) == oreplay.valid_range_ids(last, L, T)


@pytest.mark.parametrize("args", [
    (2, 3, 3, None, None, False, None), (8, 5, 4, None, 2, False, None),
    (3, 5, 4, None, 2, True, 1), (10, 6, 4, 3, 2, False, None), (2, 5, 3, 2, None, False, None),
    (2, 5, 3, 2, None, True, None), (6, 1, 4, None, 3, False, 2)])
def test_deterministic_pass_ids_match_oracle(args):
    got = list(rb_lib.deterministic_pass_ids(*args))
    want = list(oreplay.deterministic_pass_ids(*args))
    assert len(got) == len(want)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)


def test_replay_buffer_needs_hip_device():
    with pytest.raises(_lib.AgentsAmdError, match="no CPU path"):
        rb_lib.TFUniformReplayBuffer(tensor_spec.TensorSpec((), torch.int64), batch_size=1,
                                     device="cpu")


def test_ops_refuse_cpu_tensors(lib):
    from agents_amd import ops
    x = torch.zeros(4, 4)
    with pytest.raises(_lib.AgentsAmdError, match="no CPU fallback"):
        ops.dense_forward(x, x, None, None, x.clone())


def test_dataset_combinators():
    base = ds_lib.Dataset(lambda: iter([(torch.tensor([i, i + 10]), i) for i in range(5)]))
    assert [int(b) for _, b in base.take(3)] == [0, 1, 2]
    assert [int(b) for _, b in base.prefetch(2)] == [0, 1, 2, 3, 4]
    assert [int(x) for x in base.map(lambda a, b: a[0] * 2)] == [0, 2, 4, 6, 8]
    assert [int(b) for _, b in base.filter(lambda a, b: b % 2 == 0)] == [0, 2, 4]
    assert len(list(base.repeat(2))) == 10
    un = ds_lib.Dataset(lambda: iter([torch.arange(6).reshape(3, 2)])).unbatch()
    assert [u.tolist() for u in un] == [[0, 1], [2, 3], [4, 5]]
    b = un.batch(2)
    assert [x.tolist() for x in b] == [[[0, 1], [2, 3]], [[4, 5]]]
    assert [x.tolist() for x in un.batch(2, drop_remainder=True)] == [[[0, 1], [2, 3]]]
    sh = sorted(int(b) for _, b in base.shuffle(3, seed=1))
    assert sh == [0, 1, 2, 3, 4]
    c = base.cache()
    assert len(list(c)) == 5 and len(list(c)) == 5
    inf = ds_lib.Dataset(lambda: iter(range(10**9)), infinite=True)
    it = iter(inf.prefetch(3))
    assert [next(it) for _ in range(4)] == [0, 1, 2, 3]


def test_dataset_unbatch_filter_batch_on_cpu_tensors():
    """The SAC script's `.unbatch().filter(pred).batch(n)` chain on CPU tensors takes the element-wise
    combinators (the device compaction needs HIP tensors): survivors in source order, cut into runs
    of n, non-tensor leaves passed through, remainder kept unless drop_remainder."""
    import collections
    T = collections.namedtuple("T", "a b")

    def src():
        for k in range(5):
            yield (T(torch.arange(k * 4, (k + 1) * 4).view(4, 1).repeat(1, 2),
                     torch.arange(k * 4, (k + 1) * 4)), None)
    ds = ds_lib.Dataset(src)
    chain = ds.unbatch().filter(lambda t, _: t.b % 3 != 0)
    out = list(chain.batch(5))
    assert [o[0].b.tolist() for o in out] == [[1, 2, 4, 5, 7], [8, 10, 11, 13, 14], [16, 17, 19]]
    assert all(o[1] is None for o in out) and out[0][0].a.shape == (5, 2)
    assert [o[0].b.tolist() for o in chain.batch(5, drop_remainder=True)] == \
        [[1, 2, 4, 5, 7], [8, 10, 11, 13, 14]]
    # the chain is re-iterable and the plain combinators still compose after it
    assert len(list(chain.batch(5).take(1))) == 1 and len(list(chain)) == 13


def test_time_step_constructors_and_from_transition():
    obs = torch.tensor([[1.0, 2.0], [3.0, 4.0]])
    r = ts.restart(obs, batch_size=2)
    assert r.step_type.tolist() == [0, 0] and r.reward.tolist() == [0, 0] and \
        r.discount.tolist() == [1, 1] and r.step_type.dtype == torch.int32
    m = ts.transition(obs, torch.tensor([10.0, 20.0]), 0.9)
    assert m.step_type.tolist() == [1, 1] and torch.allclose(m.discount, torch.tensor([.9, .9]))
    t = ts.termination(obs, [1.0, 2.0])
    assert t.step_type.tolist() == [2, 2] and t.discount.tolist() == [0, 0] and bool(t.is_last().all())
    tr = ts.truncation(obs, [1.0, 2.0], 0.5)
    assert tr.step_type.tolist() == [2, 2] and tr.discount.tolist() == [0.5, 0.5]
    traj = trajectory.from_transition(r, policy_step.PolicyStep(torch.tensor([0, 1]), (), ()), m)
    assert traj.next_step_type.tolist() == [1, 1] and traj.reward.tolist() == [10, 20]
    assert not bool(traj.is_boundary().any()) and bool(traj.is_first().all())
    assert ts.StepType(2) is ts.StepType.LAST
    with pytest.raises(ValueError):
        ts.StepType(5)


def test_to_n_step_transition_matches_oracle():
    rng = np.random.RandomState(0)
    B, T = 6, 4
    rew = rng.randn(B, T).astype(np.float32)
    disc = (rng.rand(B, T) > 0.2).astype(np.float32) * 0.9
    traj = trajectory.Trajectory(
        torch.zeros(B, T, dtype=torch.int32), torch.randn(B, T, 3),
        torch.zeros(B, T, dtype=torch.int64), (), torch.ones(B, T, dtype=torch.int32),
        torch.tensor(rew), torch.tensor(disc))
    tr = trajectory.to_n_step_transition(traj, gamma=0.99)
    want_r, want_d = odqn.n_step_return(rew, disc, 0.99)
    np.testing.assert_allclose(tr.next_time_step.reward.numpy(), want_r, rtol=1e-6)
    np.testing.assert_allclose(tr.next_time_step.discount.numpy(), want_d, rtol=1e-6)
    assert torch.equal(tr.next_time_step.observation, traj.observation[:, -1])
    assert torch.isnan(tr.time_step.reward).all()
    with pytest.raises(ValueError):
        trajectory.to_n_step_transition(nest_utils.map_structure(lambda x: x[:, :1], traj), 0.9)


def test_periodically_and_soft_update_validation():
    hits = []
    p = common.Periodically(lambda: hits.append(1), 3)
    for _ in range(7):
        p()
    assert len(hits) == 2          # calls 3 and 6
    p1 = common.Periodically(lambda: hits.append(1), 1)
    p1(); p1()
    assert len(hits) == 4
    assert common.Periodically(lambda: 1, None)() is None
    with pytest.raises(TypeError):
        common.Periodically(3, 1)
    with pytest.raises(ValueError):
        common.soft_variables_update([torch.zeros(1)], [torch.zeros(1)], tau=1.5)
    with pytest.raises(ValueError):
        common.soft_variables_update([torch.zeros(1)], [], tau=0.5)
    t = torch.zeros(3)
    common.soft_variables_update([torch.ones(3)], [t], tau=1.0)   # tau == 1 is a plain copy
    assert t.tolist() == [1, 1, 1]


def test_sequential_shape_inference():
    from agents_amd.networks import layers as L
    from agents_amd.networks import sequential
    net = sequential.Sequential([L.Rescale(255.0), L.Conv2D(32, 8, 4, "relu"),
                                 L.Conv2D(64, 4, 2, "relu"), L.Conv2D(64, 3, 1, "relu"),
                                 L.Flatten(), L.Dense(512, "relu"), L.Dense(6)])
    info, out = net._infer((84, 84, 4))
    assert out == (6,)
    assert [i[0] for i in info] == [(8, 8, 4, 32), (4, 4, 32, 64), (3, 3, 64, 64), (3136, 512),
                                    (512, 6)]
    n = sum(int(np.prod(k)) + int(np.prod(b)) for k, b, _, _ in info)
    assert n == 1687206        # Atari Q-net parameter count
    with pytest.raises(NotImplementedError):
        L.Conv2D(8, 3, 1, padding="same")
    with pytest.raises(TypeError):
        sequential.Sequential([object()])


def test_initializers():
    from agents_amd.networks import layers as L
    rng = np.random.default_rng(0)
    w = L.VarianceScaling(2.0)((3136, 512), rng, 3136, 512)
    assert abs(w.std() - np.sqrt(2.0 / 3136)) < 5e-4 and np.abs(w).max() <= 2 * np.sqrt(
        2.0 / 3136) / 0.87962566103423978 + 1e-6
    u = L.RandomUniform(-0.03, 0.03)((512, 6), rng, 512, 6)
    assert u.min() >= -0.03 and u.max() <= 0.03
    q = L.Orthogonal()((64, 64), rng, 64, 64)
    np.testing.assert_allclose(q.T @ q, np.eye(64), atol=1e-5)
    assert L.Constant([[2, 1], [1, 1]])((2, 2), rng, 2, 2).tolist() == [[2, 1], [1, 1]]
    assert L.Constant(-0.2)((3,), rng, 1, 3).tolist() == pytest.approx([-0.2] * 3)


def test_learner_triggers():
    from agents_amd.train import interval_trigger, triggers
    from agents_amd.utils import common
    calls = []
    t = interval_trigger.IntervalTrigger(5, lambda: calls.append(1))
    for v in range(12):
        t(v)
    assert len(calls) == 2            # at 5 and 10
    t(11, force_trigger=True)
    assert len(calls) == 3
    t(11, force_trigger=True)         # same value: not again
    assert len(calls) == 3
    never = interval_trigger.IntervalTrigger(0, lambda: calls.append(2))
    never(100)
    assert 2 not in calls
    step = common.Variable(0)
    logs = []
    sps = triggers.StepPerSecondLogTrigger(step, 3, log_fn=logs.append)
    for v in range(1, 8):
        step.assign(v)
        sps(v)
    assert len(logs) == 2 and sps.last_steps_per_sec > 0


# ---- Checkpointer (utils/common.py; reference common.py:1045-1100) -------------------------------
class _Obj:
    def __init__(self):
        self.t = torch.zeros(3)
        self.n = 0

    def state_dict(self):
        return {"t": self.t.clone(), "n": self.n, "nest": [(self.t.clone(), None)]}

    def load_state_dict(self, sd):
        self.t.copy_(sd["t"])
        self.n = int(sd["n"])


def test_checkpointer_atomic_save_prune_and_fallback(tmp_path):
    import os
    import warnings
    from agents_amd.utils import common
    d = str(tmp_path / "ck")
    o, step = _Obj(), common.Variable(0)
    ck = common.Checkpointer(d, max_to_keep=2, obj=o, step=step)
    assert not ck.checkpoint_exists
    for k in (1, 2, 3):
        o.t.fill_(float(k))
        o.n = k
        step.assign(k)
        ck.save(k)
    assert sorted(os.listdir(d)) == ["ckpt-2.pt", "ckpt-3.pt"]       # pruned, no temp files left
    o2, step2 = _Obj(), common.Variable(0)
    ck2 = common.Checkpointer(d, obj=o2, step=step2)
    assert ck2.checkpoint_exists and ck2.restored_from == "ckpt-3.pt"
    assert o2.n == 3 and float(o2.t[0]) == 3.0 and int(step2) == 3
    # a truncated newest file (crash mid-write of a non-atomic writer) falls back to the previous
    with open(os.path.join(d, "ckpt-9.pt"), "wb") as fh:
        fh.write(b"PK\x03\x04 not a checkpoint")
    o3 = _Obj()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ck3 = common.Checkpointer(d, obj=o3, step=common.Variable(0))
    assert ck3.restored_from == "ckpt-3.pt" and o3.n == 3
    assert any("could not read ckpt-9.pt" in str(x.message) for x in w)
    # stray temp files of a crashed writer are ignored
    open(os.path.join(d, "ckpt-10.pt.tmp123"), "wb").close()
    assert common.Checkpointer(d, obj=_Obj(), step=common.Variable(0)).restored_from == "ckpt-3.pt"


def test_checkpointer_does_not_hide_load_state_dict_errors(tmp_path):
    """A readable checkpoint that no longer FITS (config change, bug in load_state_dict) must
    raise: falling back would silently train from older weights -- or from scratch -- and later
    saves would prune the good files; and files that exist but cannot be read at all must not be
    mistaken for "no checkpoint"."""
    import os
    import pytest
    import warnings
    from agents_amd.utils import common
    d = str(tmp_path / "ck")
    o = _Obj()
    ck = common.Checkpointer(d, obj=o, step=common.Variable(0))
    for k in (1, 2):
        o.n = k
        ck.save(k)

    class _Other(_Obj):
        def __init__(self):
            self.t = torch.zeros(7)       # shape changed since the checkpoint was written
            self.n = 0
            self.loaded = []

        def load_state_dict(self, sd):
            self.loaded.append(int(sd["n"]))
            super().load_state_dict(sd)

    bad = _Other()
    with pytest.raises(RuntimeError):
        common.Checkpointer(d, obj=bad, step=common.Variable(0))
    assert bad.loaded == [2]              # the older file was NOT tried behind the caller's back
    assert sorted(os.listdir(d)) == ["ckpt-1.pt", "ckpt-2.pt"]
    for f in os.listdir(d):               # every file unreadable: refuse to start from scratch
        with open(os.path.join(d, f), "wb") as fh:
            fh.write(b"garbage")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(RuntimeError, match="none could be read"):
            common.Checkpointer(d, obj=_Obj(), step=common.Variable(0))


def _metric_traj(step_type, reward):
    st = torch.tensor(step_type, dtype=torch.int32)
    n = st.numel()
    return trajectory.Trajectory(step_type=st, observation=torch.zeros((n, 1)),
                                 action=torch.zeros((n,), dtype=torch.int64), policy_info=(),
                                 next_step_type=st, reward=torch.tensor(reward, dtype=torch.float32),
                                 discount=torch.ones((n,)))


def test_metrics_restored_before_first_call_keep_their_state():
    """Advisor finding (round 4): a metric restored before its first call lost its state -- the
    scripts restore the checkpoint before any driver run, so a resumed run restarted
    `while environment_steps_metric.result() < num_environment_steps` from 0
    (tf_agents/agents/ppo/examples/v2/train_eval_clip_agent.py:228-246 restores metrics with the
    checkpoint)."""
    from agents_amd.eval import metric_utils
    from agents_amd.metrics import tf_metrics

    def group():
        return metric_utils.MetricsGroup([
            tf_metrics.NumberOfEpisodes(), tf_metrics.EnvironmentSteps(),
            tf_metrics.AverageReturnMetric(batch_size=2, buffer_size=3),
            tf_metrics.AverageEpisodeLengthMetric(batch_size=2, buffer_size=3)])

    a = group()
    steps = [([0, 0], [0., 0.]), ([1, 1], [1., 2.]), ([1, 2], [1., 2.]), ([2, 0], [3., 0.]),
             ([0, 1], [0., 5.])]
    for st, r in steps:
        for m in a.metrics:
            m(_metric_traj(st, r))
    want = a.results()
    assert want["EnvironmentSteps"] > 0 and want["NumberOfEpisodes"] == 2
    sd = a.state_dict()
    b = group()
    b.load_state_dict(sd)                      # before any call: state is not allocated yet
    assert b.results() == want
    assert b.state_dict().keys() == sd.keys()  # a save before the first call keeps it, too
    c = group()
    c.load_state_dict(b.state_dict())
    more = ([1, 2], [1., 1.])
    for g in (a, b, c):
        for m in g.metrics:
            m(_metric_traj(*more))
    assert b.results() == a.results() == c.results()
    assert a.results()["EnvironmentSteps"] == want["EnvironmentSteps"] + 1


def test_episode_average_keeps_the_last_finishers_and_resets_non_finite_accumulators():
    """More environments finishing in one step than the ring holds: the LAST `buffer_size` in
    environment order stay (TFDeque adds them sequentially, tf_metrics.py:41-95); an accumulator
    that went non-finite is cleared by the is_first reset (tf.where, not a multiplication)."""
    from agents_amd.metrics import tf_metrics
    m = tf_metrics.AverageReturnMetric(batch_size=5, buffer_size=2)
    m(_metric_traj([0] * 5, [0.] * 5))
    m(_metric_traj([2] * 5, [1., 2., 3., 4., 5.]))
    assert m.result() == pytest.approx(4.5)
    m2 = tf_metrics.AverageReturnMetric(batch_size=1, buffer_size=2)
    m2(_metric_traj([1], [float("inf")]))
    m2(_metric_traj([0], [1.0]))
    m2(_metric_traj([2], [2.0]))
    assert m2.result() == pytest.approx(3.0)
