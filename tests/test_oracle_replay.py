"""Pins oracle/replay.py against the reference's own replay-buffer tests
(tf_agents/replay_buffers/tf_uniform_replay_buffer_test.py): layout, ordering, ranges, probabilities,
deterministic dataset orders, empty-buffer errors.  CPU only."""
import numpy as np
import pytest

from oracle import replay


def make(batch_size, max_length=1000, dtype=np.int64):
    return replay.OracleReplayBuffer([()], [dtype], batch_size, max_length)


@pytest.mark.parametrize("batch_size", [1, 5])
def test_gather_all(batch_size):  # :313-333
    rb = make(batch_size)
    for i in range(10):
        rb.add_batch([np.arange(i, i + batch_size, dtype=np.int64)])
    expected = [list(range(i, i + 10)) for i in range(batch_size)]
    np.testing.assert_array_equal(rb.gather_all()[0], expected)


@pytest.mark.parametrize("batch_size", [1, 5])
def test_gather_all_over_capacity(batch_size):  # :339-361
    rb = make(batch_size, max_length=10)
    for i in range(15):
        rb.add_batch([np.arange(0, batch_size * 100, 100, dtype=np.int64) + i])
    expected = [list(range(5 + x * 100, 15 + x * 100)) for x in range(batch_size)]
    np.testing.assert_array_equal(rb.gather_all()[0], expected)


@pytest.mark.parametrize("batch_size", [1, 5])
def test_gather_all_empty(batch_size):  # :363-378
    rb = make(batch_size)
    assert rb.gather_all()[0].shape == (batch_size, 0)


@pytest.mark.parametrize("batch_size", [1, 5])
def test_num_frames(batch_size):  # :673-699
    rb = make(batch_size, max_length=12)
    for i in range(10):
        rb.add_batch([np.arange(i, i + batch_size, dtype=np.int64)])
    assert rb.num_frames() == 10 * batch_size
    for i in range(10):
        rb.add_batch([np.arange(i, i + batch_size, dtype=np.int64)])
    assert rb.num_frames() == rb.capacity


def test_valid_range_ids_examples():  # SURVEY appendix A.1, :610-635
    L = 10
    assert replay.valid_range_ids(-1, L) == (0, 0)
    assert replay.valid_range_ids(0, L, 1) == (0, 1)
    assert replay.valid_range_ids(0, L, 2) == (0, 0)
    assert replay.valid_range_ids(9, L, 2) == (0, 9)
    assert replay.valid_range_ids(10, L, 1) == (1, 11)
    assert replay.valid_range_ids(10, L, 2) == (1, 10)


def test_empty_buffer_raises():  # :96-109
    rb = make(2)
    with pytest.raises(RuntimeError, match="TFUniformReplayBuffer is empty"):
        rb.get_next()
    rb.add_batch([np.zeros(2, np.int64)])
    with pytest.raises(RuntimeError, match="TFUniformReplayBuffer is empty"):
        rb.get_next(num_steps=2)


def test_multi_step_contiguity():  # testGetNext :701-723
    L = 3
    rb = make(2, max_length=L)
    for t in range(4):
        rb.add_batch([np.array([t, t + L], dtype=np.int64)])
    data, ids, probs = rb.get_next(sample_batch_size=256, num_steps=2)
    exp = data[0]
    assert np.all(exp[:, 0] + 1 == exp[:, 1])
    assert ids.shape == (256, 2) and probs.shape == (256,)


def test_probabilities():  # :384-486: prob = 1 / (num valid ids * batch_size)
    rb = make(5, max_length=10)
    for i in range(7):
        rb.add_batch([np.full(5, i, np.int64)])
    _, _, p = rb.get_next(sample_batch_size=4)
    np.testing.assert_allclose(p, 1.0 / (7 * 5))
    _, _, p = rb.get_next(sample_batch_size=4, num_steps=3)
    np.testing.assert_allclose(p, 1.0 / (5 * 5))
    for i in range(10):
        rb.add_batch([np.full(5, i, np.int64)])
    _, _, p = rb.get_next(sample_batch_size=4, num_steps=2)
    np.testing.assert_allclose(p, 1.0 / (9 * 5))


def test_sampled_ids_in_valid_window_after_wrap():
    rb = make(3, max_length=4)
    for i in range(11):
        rb.add_batch([np.full(3, i, np.int64)])
    data, ids, _ = rb.get_next(sample_batch_size=512, num_steps=2)
    assert ids.min() >= 11 - 4 and ids.max() <= 10
    np.testing.assert_array_equal(data[0], ids)  # item value == its id in this fill pattern
    assert np.all(ids[:, 1] == ids[:, 0] + 1)


def _collect(max_length, B, num_adds, sample_batch_size, num_steps=None, **kw):
    rb = make(B, max_length=max_length)
    for i in range(num_adds):
        rb.add_batch([10 * np.arange(B, dtype=np.int64) + i])
    out = []
    for ids in rb.deterministic_ids(sample_batch_size, num_steps, **kw):
        out.append(rb.read_ids(ids)[0].tolist())
    return out


@pytest.mark.parametrize("B", [1, 5])
def test_deterministic_as_dataset(B):  # :548-558
    vals = _collect(3, B, 3, None)
    np.testing.assert_array_equal(vals, np.hstack([np.arange(3) + 10 * i for i in range(B)]))


def test_deterministic_with_num_steps():  # :560-589
    vals = _collect(4, 5, 4, None, num_steps=2)
    expected = [[0, 1], [2, 3], [10, 11], [12, 13], [20, 21], [22, 23], [30, 31], [32, 33],
                [40, 41], [42, 43]]
    assert vals == expected


@pytest.mark.parametrize("B", [1, 5])
def test_deterministic_with_sample_batch(B):  # :595-612
    vals = _collect(3, B, 3, B)
    np.testing.assert_array_equal(vals, np.vstack([10 * np.arange(B) + i for i in range(3)]))


def test_deterministic_with_num_steps_and_sample_batch():  # :614-641
    vals = _collect(4, 6, 4, 3, num_steps=2)
    expected = [[[0, 1], [10, 11], [20, 21]], [[2, 3], [12, 13], [22, 23]],
                [[30, 31], [40, 41], [50, 51]], [[32, 33], [42, 43], [52, 53]]]
    assert vals == expected


def test_deterministic_all_dropped_errors():  # :643-667
    rb = make(2, max_length=3, dtype=np.int32)
    rb.add_batch([np.zeros(2, np.int32)])
    with pytest.raises(ValueError, match="ALL data will be dropped"):
        rb.deterministic_ids(3, None, drop_remainder=True)
    with pytest.raises(ValueError, match="ALL data will be dropped"):
        rb.deterministic_ids(None, 4, drop_remainder=True)


def test_window_shift():  # constructor docstring :118-131
    rb = make(1, max_length=5)
    for i in range(5):
        rb.add_batch([np.array([i], np.int64)])
    w = [x.tolist() for x in rb.deterministic_ids(None, 2, window_shift=1)]
    assert w == [[0, 1], [1, 2], [2, 3], [3, 4], [4]]
    w = [x.tolist() for x in rb.deterministic_ids(None, 2)]
    assert w == [[0, 1], [2, 3], [4]]


def test_clear():  # :129-221
    rb = make(2, max_length=4)
    rb.add_batch([np.array([1, 2], np.int64)])
    rb.clear()
    assert rb.num_frames() == 0
    with pytest.raises(RuntimeError):
        rb.get_next()
    assert rb.tables[0][0] == 1  # tables untouched
    rb.clear(clear_all_variables=True)
    assert rb.tables[0].sum() == 0
