"""CPU: the prioritized-sampling oracle is self-consistent: exact fixed-point sums, validity mask
equal to the uniform buffer's window-start range, empirical frequencies proportional to priority."""
import numpy as np

from oracle import prioritized as op


def test_quantise_and_bounds():
    q = op.quantise([0.0, 1.0, 0.5, 1e9, np.nan], 1.0, 0.0)
    assert q[0] == 1 and q[1] == 65536 and q[2] == 32768 and q[3] == 4294967295 and q[4] == 1
    q2 = op.quantise([4.0], 0.5, 0.0)
    assert q2[0] == 2 * 65536


def test_sample_frequencies_and_validity():
    B, L, T = 2, 8, 2
    last_id = 5                      # not full: valid starts 0..4
    ids = np.zeros(B * L, np.int64)
    for b in range(B):
        for i in range(last_id + 1):
            ids[b * L + i] = i
    pq = np.zeros(B * L, np.uint32)
    pq[0 * L + 1] = 65536            # p = 1
    pq[1 * L + 3] = 3 * 65536        # p = 3
    pq[1 * L + 5] = 100 * 65536      # id 5 is not a valid start for T = 2: never sampled
    pq[0 * L + 7] = 50 * 65536       # never written (id 0 at an unwritten row looks valid) ...
    ids[0 * L + 7] = -1              # ... unless its id says otherwise
    counts = {}
    n = 0
    for call in range(40):
        rows, probs, empty = op.sample(pq, ids, last_id, B, L, 64, T, 1234, call)
        assert not empty
        for s in range(64):
            counts[int(rows[s, 0])] = counts.get(int(rows[s, 0]), 0) + 1
            assert rows[s, 1] == rows[s, 0] + 1
            n += 1
    assert set(counts) == {1, L + 3}
    frac = counts[L + 3] / n
    assert abs(frac - 0.75) < 0.03
    rows, probs, _ = op.sample(pq, ids, last_id, B, L, 8, T, 1, 0)
    for s in range(8):
        want = 0.25 if rows[s, 0] == 1 else 0.75
        assert probs[s] == np.float32(want)


def test_empty_buffer_flag():
    rows, probs, empty = op.sample(np.zeros(16, np.uint32), np.zeros(16, np.int64), -1, 2, 8, 4,
                                   1, 0, 0)
    assert empty and not rows.any()
