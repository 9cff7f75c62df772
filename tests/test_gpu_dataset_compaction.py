"""GPU parity: `.unbatch().filter(pred).batch(n)` compacted on the device (csrc/replay.hip:
aa_rb_compact_append / aa_rb_compact_take through replay_buffers/dataset.py) vs the element-wise
combinators, bit for bit -- the SAC script's pipeline (agents/sac/examples/v2/train_eval.py:285-296)
-- plus the numpy restatement of tf.data's unbatch / filter / batch on the same stream."""
import collections

import numpy as np
import pytest
import torch

from agents_amd import _lib
from agents_amd.replay_buffers import dataset as ds_lib
from agents_amd.replay_buffers import tf_uniform_replay_buffer as rb_lib
from agents_amd.specs import tensor_spec
from agents_amd.trajectories import trajectory
from agents_amd.utils import nest_utils

pytestmark = pytest.mark.gpu

Elem = collections.namedtuple("Elem", "step_type observation wide")


def _batches(dev, n_batches, S, seed, p_boundary=0.3, wide=40000):
    """Batched [S, T=2, ...] elements; `wide` > 32 KiB per row so a row spans several chunks."""
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n_batches):
        st = (rng.rand(S, 2) < p_boundary).astype(np.int32) * 2
        obs = rng.randn(S, 2, 5).astype(np.float32)
        w = rng.randint(0, 255, size=(S, wide)).astype(np.uint8)
        e = Elem(torch.as_tensor(st, device=dev), torch.as_tensor(obs, device=dev),
                 torch.as_tensor(w, device=dev))
        out.append((e, None))
    return out


def _pred(e, _):
    return ~(e.step_type == 2)[0]


def _run(batches, n, drop, device_path, pred=_pred):
    old = ds_lib.DEVICE_COMPACTION
    ds_lib.DEVICE_COMPACTION = device_path
    try:
        ds = ds_lib.Dataset(lambda: iter(batches))
        return list(ds.unbatch().filter(pred).batch(n, drop_remainder=drop))
    finally:
        ds_lib.DEVICE_COMPACTION = old


def _same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        fx, fy = nest_utils.flatten(x), nest_utils.flatten(y)
        assert len(fx) == len(fy)
        for l, m in zip(fx, fy):
            if isinstance(l, torch.Tensor):
                assert l.dtype == m.dtype and l.shape == m.shape
                assert torch.equal(l, m)
            else:
                assert l is m or l == m


@pytest.mark.parametrize("S,n,drop", [(16, 16, False), (16, 5, False), (7, 20, True), (64, 33, False),
                                      (1, 3, False)])
def test_device_compaction_equals_elementwise(dev, S, n, drop):
    batches = _batches(dev, 9, S, seed=S * 100 + n, wide=40000 if S <= 16 else 520)
    fast = _run(batches, n, drop, True)
    slow = _run(batches, n, drop, False)
    assert len(fast) > 0
    _same(fast, slow)


def test_device_compaction_matches_numpy_restatement(dev):
    """tf.data semantics restated in numpy: survivors in source order, cut into runs of n."""
    batches = _batches(dev, 6, 32, seed=7, wide=64)
    got = _run(batches, 10, False, True)
    st = np.concatenate([b[0].step_type.cpu().numpy() for b in batches])
    obs = np.concatenate([b[0].observation.cpu().numpy() for b in batches])
    keep = st[:, 0] != 2
    st, obs = st[keep], obs[keep]
    assert len(got) == -(-len(st) // 10)
    for i, (e, info) in enumerate(got):
        assert info is None
        np.testing.assert_array_equal(e.step_type.cpu().numpy(), st[i * 10:(i + 1) * 10])
        np.testing.assert_array_equal(e.observation.cpu().numpy(), obs[i * 10:(i + 1) * 10])


def test_all_dropped_and_all_kept(dev):
    batches = _batches(dev, 4, 8, seed=3, wide=16)
    none = _run(batches, 4, False, True, pred=lambda e, _: (e.step_type == 99)[0])
    assert none == []
    allk = _run(batches, 8, False, True, pred=lambda e, _: (e.step_type != 99)[0])
    assert len(allk) == 4
    for (e, _), (b, _) in zip(allk, batches):
        assert torch.equal(e.wide, b.wide) and torch.equal(e.observation, b.observation)


def test_source_batch_size_changes_grow_the_ring(dev):
    batches = _batches(dev, 3, 4, seed=1, wide=16) + _batches(dev, 3, 50, seed=2, wide=16) \
        + _batches(dev, 2, 3, seed=4, wide=16)
    _same(_run(batches, 7, False, True), _run(batches, 7, False, False))


def test_non_vectorisable_predicate_falls_back_to_elementwise(dev):
    """A predicate that reduces over the whole element does not yield one flag per sample with the
    batch axis moved last (or yields different flags): the pipeline must then evaluate it element by
    element -- same results as with the device path off."""
    batches = _batches(dev, 5, 8, seed=11, wide=16)

    def reduces(e, _):
        return bool((e.observation.sum() > 0).item())

    _same(_run(batches, 6, False, True, pred=reduces), _run(batches, 6, False, False, pred=reduces))

    # 5 samples x 5 features: summing the LAST axis gives 5 flags either way, but over the samples
    # when batched -- the one-time check against the element-wise flags must catch it
    batches = _batches(dev, 5, 5, seed=12, wide=16)

    def wrong_axis(e, _):
        return e.observation[0].sum(-1) > 0

    _same(_run(batches, 6, False, True, pred=wrong_axis),
          _run(batches, 6, False, False, pred=wrong_axis))


def test_c_abi_argument_checks(dev):
    import ctypes
    lib = _lib.load()
    a = torch.zeros(8, 4, device=dev)
    b = torch.zeros(4, 4, device=dev)
    keep = torch.ones(4, dtype=torch.uint8, device=dev)
    kept = torch.zeros(1, dtype=torch.int64, device=dev)
    t = (ctypes.c_void_p * 1)(a.data_ptr())
    o = (ctypes.c_void_p * 1)(b.data_ptr())
    rb = (ctypes.c_int64 * 1)(16)
    st = _lib.stream_ptr()
    # 6 pending + 4 source rows do not fit a ring of 8
    assert lib.aa_rb_compact_append(t, o, rb, 1, keep.data_ptr(), 4, 0, 6, 8, kept.data_ptr(), st) != 0
    assert lib.aa_rb_compact_append(t, o, rb, 1, keep.data_ptr(), 4, 9, 0, 8, kept.data_ptr(), st) != 0
    assert lib.aa_rb_compact_take(t, o, rb, 1, 0, 5, 4, 8, st) != 0
    # wrap-around: tail 6 of 8, 4 rows kept -> rows 6, 7, 0, 1
    src = torch.arange(16, dtype=torch.float32, device=dev).view(4, 4)
    o2 = (ctypes.c_void_p * 1)(src.data_ptr())
    assert lib.aa_rb_compact_append(t, o2, rb, 1, keep.data_ptr(), 4, 6, 0, 8, kept.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert int(kept.item()) == 4
    assert torch.equal(a[[6, 7, 0, 1]], src)
    out = torch.zeros(4, 4, device=dev)
    o3 = (ctypes.c_void_p * 1)(out.data_ptr())
    assert lib.aa_rb_compact_take(t, o3, rb, 1, 6, 4, 4, 8, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, src)


def test_sac_replay_pipeline(dev):
    """The script's pipeline on a real replay buffer: as_dataset(sample_batch_size, num_steps=2)
    .unbatch().filter(~is_boundary[0]).batch(n).prefetch(5): no emitted transition starts on a
    boundary, every batch is full, and the stream equals the element-wise pipeline's on an
    identically seeded buffer."""
    spec = trajectory.Trajectory(
        step_type=tensor_spec.TensorSpec((), torch.int32, "step_type"),
        observation=tensor_spec.TensorSpec((11,), torch.float32, "observation"),
        action=tensor_spec.TensorSpec((3,), torch.float32, "action"),
        policy_info=(),
        next_step_type=tensor_spec.TensorSpec((), torch.int32, "next_step_type"),
        reward=tensor_spec.TensorSpec((), torch.float32, "reward"),
        discount=tensor_spec.TensorSpec((), torch.float32, "discount"))

    def build():
        rb = rb_lib.TFUniformReplayBuffer(spec, batch_size=4, max_length=64, device=dev, seed=5)
        rng = np.random.RandomState(0)
        for i in range(64):
            last = (i % 9) == 8
            st = np.full(4, 2 if last else (0 if i % 9 == 0 else 1), np.int32)
            nst = np.full(4, 0 if last else (2 if i % 9 == 7 else 1), np.int32)
            rb.add_batch(trajectory.Trajectory(
                torch.as_tensor(st, device=dev),
                torch.as_tensor(rng.randn(4, 11).astype(np.float32), device=dev),
                torch.as_tensor(rng.randn(4, 3).astype(np.float32), device=dev), (),
                torch.as_tensor(nst, device=dev),
                torch.as_tensor(rng.randn(4).astype(np.float32), device=dev),
                torch.ones(4, device=dev)))
        return rb

    def pipeline(rb, device_path):
        old = ds_lib.DEVICE_COMPACTION
        ds_lib.DEVICE_COMPACTION = device_path
        try:
            ds = rb.as_dataset(sample_batch_size=32, num_steps=2).unbatch().filter(
                lambda traj, _: ~traj.is_boundary()[0]).batch(32).prefetch(5)
            it = iter(ds)
            return [next(it) for _ in range(6)]
        finally:
            ds_lib.DEVICE_COMPACTION = old

    fast = pipeline(build(), True)
    slow = pipeline(build(), False)
    for traj, info in fast:
        assert traj.step_type.shape == (32, 2) and traj.observation.shape == (32, 2, 11)
        assert not bool((traj.step_type[:, 0] == 2).any())
    _same(fast, slow)
