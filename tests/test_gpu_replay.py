"""GPU parity: TFUniformReplayBuffer (HIP kernels through the C ABI) vs oracle/replay.py, bit-exact,
on the reference's own test scenarios (tf_uniform_replay_buffer_test.py) plus seeded fuzz, and
size-independent properties at the full Atari configuration."""
import collections

import numpy as np
import pytest
import torch

from agents_amd.replay_buffers import tf_uniform_replay_buffer as rb_lib
from agents_amd.specs import tensor_spec
from agents_amd.trajectories import trajectory
from oracle import replay as oracle_replay

pytestmark = pytest.mark.gpu



def _arrivals_consistent(rb):
    """get_next's arrival shards are monotonic (csrc/replay.hip: aa_arrivals_finish): after a launch
    the eight shards add up to the `consumed` word, and no launch reported a timeout."""
    arr = rb._sample_arrival.cpu().numpy()
    assert int(arr[0:128:16].sum()) == int(arr[128]), arr[0:144:16]
    assert rb.device_error() == 0


def spec_i64():
    return tensor_spec.TensorSpec((), torch.int64, "action")


def t(x, dev, dtype=torch.int64):
    return torch.as_tensor(np.asarray(x), dtype=dtype, device=dev)


@pytest.mark.parametrize("batch_size", [1, 5])
def test_gather_all(dev, batch_size):
    rb = rb_lib.TFUniformReplayBuffer(spec_i64(), batch_size=batch_size, device=dev)
    for i in range(10):
        rb.add_batch(t(np.arange(i, i + batch_size), dev))
    expected = [list(range(i, i + 10)) for i in range(batch_size)]
    np.testing.assert_array_equal(rb.gather_all().cpu().numpy(), expected)


@pytest.mark.parametrize("batch_size", [1, 5])
def test_gather_all_over_capacity(dev, batch_size):
    rb = rb_lib.TFUniformReplayBuffer(spec_i64(), batch_size=batch_size, max_length=10, device=dev)
    for i in range(15):
        rb.add_batch(t(np.arange(0, batch_size * 100, 100) + i, dev))
    expected = [list(range(5 + x * 100, 15 + x * 100)) for x in range(batch_size)]
    np.testing.assert_array_equal(rb.gather_all().cpu().numpy(), expected)


@pytest.mark.parametrize("batch_size", [1, 5])
def test_gather_all_empty_and_num_frames(dev, batch_size):
    rb = rb_lib.TFUniformReplayBuffer(spec_i64(), batch_size=batch_size, max_length=12, device=dev)
    assert tuple(rb.gather_all().shape) == (batch_size, 0)
    for i in range(10):
        rb.add_batch(t(np.arange(i, i + batch_size), dev))
    assert rb.num_frames() == 10 * batch_size
    for i in range(10):
        rb.add_batch(t(np.arange(i, i + batch_size), dev))
    assert rb.num_frames() == rb.capacity


def test_empty_raises_and_clear(dev):
    rb = rb_lib.TFUniformReplayBuffer(spec_i64(), batch_size=2, max_length=4, device=dev)
    with pytest.raises(RuntimeError, match="TFUniformReplayBuffer is empty"):
        rb.get_next()
    rb.add_batch(t([1, 2], dev))
    with pytest.raises(RuntimeError, match="TFUniformReplayBuffer is empty"):
        rb.get_next(num_steps=2)
    item, info = rb.get_next()
    assert item.dim() == 0 and int(item) in (1, 2) and float(info.probabilities) == 0.5
    rb.clear()
    assert rb.num_frames() == 0
    with pytest.raises(RuntimeError):
        rb.get_next()
    with pytest.raises(ValueError):
        rb.add_batch(t([1, 2, 3], dev))  # wrong batch size


def test_get_next_contiguity(dev):  # testGetNext
    L = 3
    rb = rb_lib.TFUniformReplayBuffer(spec_i64(), batch_size=2, max_length=L, device=dev)
    for k in range(4):
        rb.add_batch(t([k, k + L], dev))
    exp, info = rb.get_next(sample_batch_size=256, num_steps=2, time_stacked=True)
    exp = exp.cpu().numpy()
    assert np.all(exp[:, 0] + 1 == exp[:, 1])
    assert tuple(info.ids.shape) == (256, 2) and tuple(info.probabilities.shape) == (256,)
    tup, info2 = rb.get_next(sample_batch_size=8, num_steps=2, time_stacked=False)
    assert isinstance(tup, tuple) and len(tup) == 2 and tuple(tup[0].shape) == (8,)
    assert torch.all(tup[0] + 1 == tup[1])


Leaf = collections.namedtuple("Leaf", ["shape", "dtype"])


def _traj_spec(obs_shape, obs_dtype):
    f = tensor_spec.TensorSpec
    return trajectory.Trajectory(
        step_type=f((), torch.int32), observation=f(obs_shape, obs_dtype),
        action=f((), torch.int64), policy_info=(), next_step_type=f((), torch.int32),
        reward=f((), torch.float32), discount=f((), torch.float32))


def _rand_items(rng, spec, B):
    def one(s):
        npd = tensor_spec.as_numpy_dtype(s.dtype)
        if np.issubdtype(npd, np.floating):
            return rng.randn(B, *s.shape).astype(npd)
        hi = 3 if s.shape == () and npd == np.int32 else 200
        return rng.randint(0, hi, size=(B,) + tuple(s.shape)).astype(npd)
    from agents_amd.utils import nest_utils
    return nest_utils.map_structure(one, spec)


@pytest.mark.parametrize("obs_shape,obs_dtype,B,L,adds", [
    ((4,), torch.float32, 1, 7, 5),          # CartPole-like, not full
    ((4,), torch.float32, 3, 5, 13),         # wrapped
    ((84, 84, 4), torch.uint8, 4, 6, 9),     # Atari rows (28,224 B), wrapped
    ((17,), torch.float32, 5, 3, 3),         # row_bytes not a multiple of 16
    ((3, 5), torch.uint8, 2, 4, 6),          # 15-byte rows: byte path
    ((300,), torch.float32, 5, 4, 7),        # 1,200-byte rows: sixteen rows per workgroup, ragged
    ((376,), torch.float32, 37, 3, 5),       # SAC-sized rows, three workgroups of 16 / 16 / 5 rows
    ((1028,), torch.float32, 18, 3, 4),      # 4,112 bytes: just above the medium-row limit
])
def test_fuzz_vs_oracle_bit_exact(dev, obs_shape, obs_dtype, B, L, adds):
    from agents_amd.utils import nest_utils
    spec = _traj_spec(obs_shape, obs_dtype)
    flat_specs = nest_utils.flatten(spec)
    rb = rb_lib.TFUniformReplayBuffer(spec, batch_size=B, max_length=L, device=dev, seed=1234)
    orc = oracle_replay.OracleReplayBuffer(
        [s.shape for s in flat_specs], [tensor_spec.as_numpy_dtype(s.dtype) for s in flat_specs],
        B, L, seed=1234)
    rng = np.random.RandomState(0)
    for i in range(adds):
        items = _rand_items(rng, spec, B)
        orc.add_batch(nest_utils.flatten(items))
        rb.add_batch(nest_utils.map_structure(lambda a: torch.as_tensor(a, device=dev), items))
        if i >= 1:
            for S, T in ((7, 2), (5, None), (None, 2)):
                if T is not None and i + 1 < T:
                    continue
                data, info = rb.get_next(sample_batch_size=S, num_steps=T)
                odata, oids, oprobs = orc.get_next(S, T)
                for g, o in zip(nest_utils.flatten(data), odata):
                    np.testing.assert_array_equal(g.cpu().numpy(), o)
                np.testing.assert_array_equal(info.ids.cpu().numpy(), oids)
                np.testing.assert_array_equal(info.probabilities.cpu().numpy(), oprobs)
    for g, o in zip(nest_utils.flatten(rb.gather_all()), orc.gather_all()):
        np.testing.assert_array_equal(g.cpu().numpy(), o)
    for g, o in zip(rb.variables()[:-2], orc.tables):
        np.testing.assert_array_equal(g.cpu().numpy(), o)
    np.testing.assert_array_equal(rb.variables()[-2].cpu().numpy(), orc.id_table)
    assert int(rb.variables()[-1].item()) == orc.last_id


def _collect(dev, max_length, B, num_adds, sample_batch_size, num_steps=None, **kw):
    rb = rb_lib.TFUniformReplayBuffer(spec_i64(), batch_size=B, max_length=max_length, device=dev,
                                      **kw)
    for i in range(num_adds):
        rb.add_batch(t(10 * np.arange(B) + i, dev))
    ds = rb.as_dataset(single_deterministic_pass=True, sample_batch_size=sample_batch_size,
                       num_steps=num_steps)
    return [d.cpu().numpy().tolist() for d, _ in ds]


def test_deterministic_datasets(dev):
    assert _collect(dev, 3, 5, 3, None) == \
        np.hstack([np.arange(3) + 10 * i for i in range(5)]).tolist()
    assert _collect(dev, 4, 5, 4, None, num_steps=2) == [
        [0, 1], [2, 3], [10, 11], [12, 13], [20, 21], [22, 23], [30, 31], [32, 33], [40, 41],
        [42, 43]]
    assert _collect(dev, 3, 5, 3, 5) == np.vstack(
        [10 * np.arange(5) + i for i in range(3)]).tolist()
    assert _collect(dev, 4, 6, 4, 3, num_steps=2) == [
        [[0, 1], [10, 11], [20, 21]], [[2, 3], [12, 13], [22, 23]],
        [[30, 31], [40, 41], [50, 51]], [[32, 33], [42, 43], [52, 53]]]
    rb = rb_lib.TFUniformReplayBuffer(tensor_spec.TensorSpec((), torch.int32), batch_size=2,
                                      max_length=3, dataset_drop_remainder=True, device=dev)
    with pytest.raises(ValueError, match="ALL data will be dropped"):
        rb.as_dataset(single_deterministic_pass=True, sample_batch_size=3)
    with pytest.raises(ValueError, match="ALL data will be dropped"):
        rb.as_dataset(single_deterministic_pass=True, num_steps=4)


def test_list_spec_rejected(dev):
    rb = rb_lib.TFUniformReplayBuffer([spec_i64(), spec_i64()], batch_size=1, device=dev)
    with pytest.raises(ValueError, match="contains lists"):
        rb.as_dataset()


def test_dataset_prefetch_stream_matches_get_next(dev):
    rb1 = rb_lib.TFUniformReplayBuffer(spec_i64(), batch_size=3, max_length=8, device=dev, seed=9)
    rb2 = rb_lib.TFUniformReplayBuffer(spec_i64(), batch_size=3, max_length=8, device=dev, seed=9)
    for i in range(6):
        rb1.add_batch(t([i, 100 + i, 200 + i], dev))
        rb2.add_batch(t([i, 100 + i, 200 + i], dev))
    it = iter(rb1.as_dataset(sample_batch_size=4, num_steps=2, num_parallel_calls=3).prefetch(3))
    for _ in range(5):
        a, ia = next(it)
        b, ib = rb2.get_next(4, 2)
        assert torch.equal(a, b) and torch.equal(ia.ids, ib.ids)


def test_full_size_atari_properties(dev):
    """BASELINE config 2 shapes (256 envs, 84x84x4 uint8) at a reduced ring length: every sampled
    frame's payload must equal the pattern written for (its stored id, its env block); sampled
    windows are contiguous in id and never cross a block."""
    B, L, S, T = 256, 24, 256, 2
    spec = _traj_spec((84, 84, 4), torch.uint8)
    rb = rb_lib.TFUniformReplayBuffer(spec, batch_size=B, max_length=L, device=dev, seed=3)
    env = torch.arange(B, device=dev)
    for i in range(L + 7):
        obs = ((env * 7 + i * 13) % 251).to(torch.uint8).view(B, 1, 1, 1).expand(B, 84, 84, 4)
        obs = obs.contiguous()
        obs[:, 83, 83, 3] = (i % 256)
        items = trajectory.Trajectory(
            step_type=torch.full((B,), i % 3, dtype=torch.int32, device=dev), observation=obs,
            action=(env + i).to(torch.int64), policy_info=(),
            next_step_type=torch.full((B,), (i + 1) % 3, dtype=torch.int32, device=dev),
            reward=(env.float() + 0.5 * i), discount=torch.ones(B, device=dev))
        rb.add_batch(items)
    data, info = rb.get_next(S, T)
    ids = info.ids.cpu().numpy()
    assert np.all(ids[:, 1] == ids[:, 0] + 1) and ids.min() >= 7 and ids.max() <= L + 6
    act = data.action.cpu().numpy()
    envs = act - ids
    assert np.all(envs[:, 0] == envs[:, 1]) and envs.min() >= 0 and envs.max() < B
    obs = data.observation.cpu().numpy()
    want = ((envs * 7 + ids * 13) % 251).astype(np.uint8)
    assert np.all(obs[:, :, 0, 0, 0] == want) and np.all(obs[:, :, 40, 17, 2] == want)
    assert np.all(obs[:, :, 83, 83, 3] == (ids % 256).astype(np.uint8))
    np.testing.assert_array_equal(data.reward.cpu().numpy(),
                                  (envs + 0.5 * ids).astype(np.float32))
    np.testing.assert_array_equal(data.step_type.cpu().numpy(), (ids % 3).astype(np.int32))
    assert np.float32(info.probabilities[0].item()) == np.float32(1.0) / np.float32((L - 1) * B)


@pytest.mark.parametrize("obs_elems,B,L,S,T", [
    (7056, 16, 6, 300, 2),        # Atari-sized rows: one workgroup per row, 600 workgroups
    (20000, 3, 9, 700, 3),        # 80,000-byte rows: three 32 KiB chunks, 6,300 workgroups
    (3, 1000, 4, 4096, 1),        # tiny rows, 1,000 envs: 1,000-workgroup scatter
])
def test_one_launch_get_next_and_add_batch_at_scale(dev, obs_elems, B, L, S, T):
    """add_batch and get_next are ONE launch each: the draw is recomputed per workgroup, the
    counters advance inside the launch through sharded arrival words (csrc/replay.hip).  Against
    the oracle bit for bit at grids of thousands of workgroups and multi-chunk rows; the arrival
    words are back to zero and the device counters equal the host mirrors after every call."""
    spec = tensor_spec.TensorSpec((obs_elems,), torch.float32, "obs")
    rb = rb_lib.TFUniformReplayBuffer(spec, batch_size=B, max_length=L, device=dev, seed=77)
    orc = oracle_replay.OracleReplayBuffer([(obs_elems,)], [np.float32], B, L, seed=77)
    rng = np.random.default_rng(5)
    for i in range(L + 3):
        x = rng.standard_normal((B, obs_elems)).astype(np.float32)
        rb.add_batch(torch.as_tensor(x, device=dev))
        orc.add_batch([x])
        assert int(rb._last_id.item()) == i and not bool(rb._scatter_arrival.any())
        if i + 1 >= T:
            data, info = rb.get_next(S, T)
            odata, oids, oprobs = orc.get_next(S, T)
            assert np.array_equal(data.cpu().numpy(), odata[0])
            assert np.array_equal(info.ids.cpu().numpy(), oids)
            assert np.array_equal(info.probabilities.cpu().numpy(), oprobs)
            assert int(rb._sample_calls_dev.item()) == rb._sample_calls
            _arrivals_consistent(rb)
    assert np.array_equal(rb.variables()[0].cpu().numpy(), orc.tables[0])


def test_fused_get_next_equals_the_two_launch_path(dev):
    """aa_rb_sample_gather == aa_rb_sample_rows + aa_rb_gather_rows on the same call counter."""
    from agents_amd import _lib
    lib = _lib.load()
    spec = _traj_spec((84, 84, 4), torch.uint8)
    B, L, S, T = 8, 5, 64, 2
    rb = rb_lib.TFUniformReplayBuffer(spec, batch_size=B, max_length=L, device=dev, seed=3)
    rng = np.random.RandomState(1)
    from agents_amd.utils import nest_utils
    for _ in range(L + 2):
        rb.add_batch(nest_utils.map_structure(lambda a: torch.as_tensor(a, device=dev),
                                              _rand_items(rng, spec, B)))
    for call in range(3):
        data, info = rb.get_next(S, T)
        rows = torch.empty((S, T), dtype=torch.int64, device=dev)
        probs = torch.empty((S,), dtype=torch.float32, device=dev)
        _lib.check(lib.aa_rb_sample_rows(rb._last_id.data_ptr(), B, L, S, T, rb._seed, call, None,
                                         rows.data_ptr(), probs.data_ptr(), None,
                                         _lib.stream_ptr()), "aa_rb_sample_rows")
        ids = torch.empty((S, T), dtype=torch.int64, device=dev)
        ref = rb._data_table.read(rows, rb._id_table.variables()[0], ids)
        for a, b in zip(nest_utils.flatten(data), nest_utils.flatten(ref)):
            assert torch.equal(a, b)
        assert torch.equal(info.ids, ids) and torch.equal(info.probabilities, probs)


@pytest.mark.parametrize("B,L,last_id,T", [
    (256, 3906, 5000, 2),                       # BASELINE configs[1]: both moduli far below 2^32
    (1, (1 << 32) - 1, (1 << 33) + 12345, 2),   # num_ids = 2^32 - 2: the largest 32-bit modulus
    ((1 << 32) - 5, 7, 100, 3),                 # batch just below 2^32, ids above 2^32 in rows
    ((1 << 32) + 7, 5, 1000, 4),                # batch >= 2^32: the generic 64-bit path
    (3, 1 << 33, (1 << 35) + 17, 4),            # num_ids >= 2^32: the generic 64-bit path
    (3, 1 << 31, (1 << 40) + 5, 1),             # id + t above 2^32 with a 32-bit ring length
    (65537, 65521, 70000, 5),
])
def test_draw_arithmetic_is_exact_at_the_modulus_boundaries(dev, B, L, last_id, T):
    """The draw reduces 64-bit Philox words with a float64 quotient + one correction step when the
    modulus fits 32 bits, and maps ids to ring slots with 32-bit arithmetic when both operands fit
    (csrc/replay.hip: aa_umod64 / aa_mod_nonneg).  No table is needed to check the rows: against
    numpy's uint64 arithmetic (oracle/replay.py) bit for bit, 8,192 draws per case."""
    from agents_amd import _lib
    lib = _lib.load()
    n, seed = 8192, 0x9E3779B97F4A7C15
    lid = torch.tensor([last_id], dtype=torch.int64, device=dev)
    rows = torch.empty((n, T), dtype=torch.int64, device=dev)
    probs = torch.empty((n,), dtype=torch.float32, device=dev)
    for call in (0, 1, (1 << 40) + 3):
        _lib.check(lib.aa_rb_sample_rows(lid.data_ptr(), B, L, n, T, seed, call, None,
                                         rows.data_ptr(), probs.data_ptr(), None,
                                         _lib.stream_ptr()), "aa_rb_sample_rows")
        a, c = oracle_replay.raw_draws(seed, call, n)
        orows, oprobs = oracle_replay.rows_from_draws(a, c, last_id, B, L, T)
        assert np.array_equal(rows.cpu().numpy(), orows)
        assert np.array_equal(probs.cpu().numpy(), oprobs)


@pytest.mark.parametrize("obs_shape,dtype,B,L,S,T", [
    ((84, 84, 4), torch.uint8, 8, 6, 64, 2),     # two rows of a sample per workgroup
    ((84, 84, 4), torch.uint8, 4, 9, 33, 4),     # 4 x 28 KB > 64 KB: two rows per workgroup
    ((20000,), torch.float32, 3, 9, 50, 3),      # three chunks per row, odd T: one row per group
    ((5,), torch.float32, 16, 8, 1000, 4),       # narrow leaves only: no wide-leaf workgroup
    ((1031,), torch.uint8, 5, 7, 200, 2),        # 1,031-byte rows: the byte-wise copy
])
def test_stamped_draws_equal_device_counter_draws(dev, obs_shape, dtype, B, L, S, T):
    """rb.draw_into (aa_rb_sample_gather_stamped: last_id and the call number by value, one eager
    launch) == rb.get_next (device-resident counters, the launch HIP graphs replay) on the same
    buffer state, leaf by leaf, ids and probabilities included; the two paths can be interleaved
    because each leaves both counters where the other expects them."""
    from agents_amd.utils import nest_utils
    spec = _traj_spec(obs_shape, dtype)
    rbs = [rb_lib.TFUniformReplayBuffer(spec, batch_size=B, max_length=L, device=dev, seed=11)
           for _ in range(2)]
    rng = np.random.RandomState(4)
    element, slot = None, None
    for i in range(L + 3):
        items = nest_utils.map_structure(lambda a: torch.as_tensor(a, device=dev),
                                         _rand_items(rng, spec, B))
        for rb in rbs:
            rb.add_batch(items)
        if i + 1 < T:
            continue
        ref, iref = rbs[1].get_next(S, T)
        if slot is None:
            element = rbs[0].get_next(S, T)            # allocates the static element
            slot = rbs[0].stamped_slot(element)
            got, igot = element
        elif i % 3 == 0:                               # interleave the device-counter path
            got, igot = rbs[0].get_next(S, T)
        else:
            with torch.cuda.device(dev):
                rbs[0].draw_into(slot)
            got, igot = element
        for a, b in zip(nest_utils.flatten(got), nest_utils.flatten(ref)):
            assert torch.equal(a, b)
        assert torch.equal(igot.ids, iref.ids)
        assert torch.equal(igot.probabilities, iref.probabilities)
        assert int(rbs[0]._sample_calls_dev.item()) == rbs[0]._sample_calls == rbs[1]._sample_calls
        _arrivals_consistent(rbs[0])


def test_far_rows(dev):
    """Byte copies at table offsets beyond 2^32 (the benchmarked Atari table is 28.2 GB; the
    oracle-checked cases above all live in the first megabytes).  One uint8 leaf of 16 env blocks x
    15,000 rows x 28,224 B = 6.77 GB: blocks 11..15 start past 4 GiB and the last row of block 15
    sits at byte 6.77e9.  `add_batch` writes patterns that name (id, env); `gather_all` and
    `get_next` must return exactly those bytes -- first at the START of every block (buffer not
    full), then, after the ring position has been moved next to the END of the blocks the way a
    restore does, across the wrap.  The patterns are recomputed, so no oracle holds the table.
    Reference behaviour: replay_buffers/table.py:86-137 on a [B * L, 84, 84, 4] variable,
    tf_uniform_replay_buffer.py:182-310."""
    free, _total = torch.cuda.mem_get_info(dev)
    if free < 24 * (1 << 30):
        pytest.skip("needs 24 GiB of free device memory")
    f = tensor_spec.TensorSpec
    spec = collections.OrderedDict(obs=f((84, 84, 4), torch.uint8), tag=f((), torch.int64))
    B, L, T = 16, 15000, 2
    rb = rb_lib.TFUniformReplayBuffer(spec, batch_size=B, max_length=L, device=dev, seed=21)
    assert rb._data_table.variables()[0].numel() > 6 * (1 << 30)
    assert 11 * L * 28224 > (1 << 32)
    env = torch.arange(B, device=dev, dtype=torch.int64)
    pos = torch.arange(84 * 84 * 4, device=dev, dtype=torch.int64)

    def pattern(ids, envs):
        """uint8 [n, 84, 84, 4] for int64 vectors ids, envs: every byte depends on (id, env, position)."""
        v = (pos[None, :] * (7 + ids[:, None] % 97) + 31 * envs[:, None] + ids[:, None]) % 251
        return v.to(torch.uint8).view(-1, 84, 84, 4)

    def add(i):
        ids = torch.full((B,), i, dtype=torch.int64, device=dev)
        rb.add_batch(collections.OrderedDict(obs=pattern(ids, env), tag=ids * 1000 + env + 1))

    def check_samples(S):
        data, info = rb.get_next(S, T)
        tag = data["tag"].reshape(-1)
        obs = data["obs"].reshape(-1, 84, 84, 4)
        written = tag > 0
        ids_from_tag = torch.div(tag - 1, 1000, rounding_mode="floor")
        env_from_tag = (tag - 1) % 1000
        assert torch.equal(info.ids.reshape(-1)[written], ids_from_tag[written])
        want = pattern(ids_from_tag.clamp(min=0), env_from_tag.clamp(min=0))
        want[~written] = 0                       # rows nobody wrote: the table's zero fill
        assert torch.equal(obs, want)
        return int(written.sum()), int((env_from_tag[written] >= 11).sum())

    # ---- block starts (not full) ------------------------------------------------------------------
    for i in range(6):
        add(i)
    got = rb.gather_all()
    assert tuple(got["obs"].shape) == (B, 6, 84, 84, 4)
    ids6 = torch.arange(6, device=dev, dtype=torch.int64)
    for b in range(B):
        assert torch.equal(got["obs"][b], pattern(ids6, torch.full_like(ids6, b))), b
        assert torch.equal(got["tag"][b], ids6 * 1000 + b + 1)
    n_written, n_far = check_samples(512)
    assert n_written == 512 * T and n_far > 0        # every sampled row exists; 5 of 16 blocks are far

    # ---- block ends and the wrap: ring position moved to L - 4 as a restore would ------------------
    rb._last_id.fill_(L - 4)
    rb._last_id_host = L - 4
    for i in range(L - 3, L + 5):                    # ring positions L-3, L-2, L-1, 0, 1, 2, 3, 4
        add(i)
    assert rb.num_frames() == rb.capacity
    got = rb.gather_all()                            # ids [5, L + 5): all 6.77 GB
    assert tuple(got["obs"].shape) == (B, L, 84, 84, 4)
    tail = torch.arange(L - 3, L + 5, device=dev, dtype=torch.int64)
    for b in (0, 10, 11, 15):
        assert torch.equal(got["obs"][b, L - 8:], pattern(tail, torch.full_like(tail, b))), b
        assert torch.equal(got["tag"][b, L - 8:], tail * 1000 + b + 1)
        five = torch.tensor([5], device=dev)
        assert torch.equal(got["obs"][b, :1], pattern(five, torch.full_like(five, b)))   # phase 1's id 5
        assert not bool(got["obs"][b, 1:L - 8].any())                                   # never written
    del got
    hits = far = 0
    for _ in range(4):
        h, fr = check_samples(4096)
        hits, far = hits + h, far + fr
    assert hits > 0                                  # ~9 of 15,000 ids are written: a few dozen rows
    _arrivals_consistent(rb)


def test_tf_layout_draws_match_the_oracle_twin(dev):
    """rng="tf": the two draws of `_get_next` in TensorFlow's Philox LAYOUT (SURVEY.md Appendix B:
    key = global seed, counter high words = op seed, 256 x outputs blocks reserved per execution,
    two int64 per block, minval + x mod range).  Rows / ids / probabilities bit-exact against the
    numpy twin (oracle/philox.py: tf_uniform_u64) over calls of different sizes, through get_next
    and through the dataset.  UNVERIFIED against TensorFlow itself: there is no TF here, and the
    reference's own draws are unseeded -- this pins the option to its documented layout only."""
    from oracle import replay as oreplay
    B_env, L_, = 5, 7
    spec = (tensor_spec.TensorSpec((3,), torch.float32), tensor_spec.TensorSpec((), torch.int32))
    rb = rb_lib.TFUniformReplayBuffer(spec, batch_size=B_env, max_length=L_, device=dev,
                                      seed=(1234, 7), rng="tf")
    ora = oreplay.OracleReplayBuffer([(3,), ()], [np.float32, np.int32], B_env, L_)
    rng = np.random.default_rng(0)
    with pytest.raises(RuntimeError, match="is empty"):
        rb.get_next(4, 2)
    for i in range(11):       # wraps the 7-frame ring
        a = rng.normal(size=(B_env, 3)).astype(np.float32)
        b = rng.integers(0, 100, size=(B_env,)).astype(np.int32)
        rb.add_batch((torch.tensor(a, device=dev), torch.tensor(b, device=dev)))
        ora.add_batch([a, b])
    blocks = 0
    it = iter(rb.as_dataset(sample_batch_size=6, num_steps=2))
    for k, (S, T) in enumerate([(4, 2), (9, 3), (1, 1), (6, 2), (6, 2)]):
        a_raw, c_raw = oreplay.raw_draws_tf_layout(1234, 7, 8, blocks, S)
        rows, probs = oreplay.rows_from_draws(a_raw, c_raw, ora.last_id, B_env, L_, T)
        blocks += S * 256
        data, info = next(it) if k >= 3 else rb.get_next(S, T)
        assert np.array_equal(info.ids.cpu().numpy(), ora.id_table[rows])
        assert np.array_equal(info.probabilities.cpu().numpy(), probs)
        for got, tab in zip(data, ora.tables):
            assert np.array_equal(got.cpu().numpy(), tab[rows])
    assert rb._tf_blocks == blocks
    sd = rb.state_dict()
    rb2 = rb_lib.TFUniformReplayBuffer(spec, batch_size=B_env, max_length=L_, device=dev,
                                       seed=(1234, 7, 8), rng="tf")
    rb2.load_state_dict(sd)
    x, y = rb.get_next(5, 2), rb2.get_next(5, 2)
    assert torch.equal(x[1].ids, y[1].ids)
    with pytest.raises(ValueError):
        rb_lib.TFUniformReplayBuffer(spec, batch_size=2, max_length=4, device=dev, rng="mt19937")
