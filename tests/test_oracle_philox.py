"""Pins oracle/philox.py to the published Random123 known-answer vectors for Philox4x32-10
(Random123 kat_vectors: `philox4x32 10 ...`)."""
import numpy as np

from oracle import philox

KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_known_answers():
    for ctr, key, want in KAT:
        got = philox.philox4x32_10(*ctr, *key)
        assert tuple(int(g) for g in got) == want


def test_vectorised_matches_scalar():
    c0 = np.arange(5)
    got = philox.philox4x32_10(c0, 7, 8, 9, 1, 2)
    for i in range(5):
        one = philox.philox4x32_10(i, 7, 8, 9, 1, 2)
        assert all(int(g[i]) == int(o) for g, o in zip(got, one))


def test_u01_range():
    u = philox.u01(np.array([0, 0xffffffff, 0x80000000], dtype=np.uint32))
    assert u.dtype == np.float32 and u[0] == 0.0 and u[1] < 1.0 and u[2] == 0.5


def test_tf_layout_host_draw_matches_the_numpy_twin(lib):
    """aa_rb_draw_tf_host is host code (no GPU): rows and probability against oracle/philox.py's
    tf_uniform_u64 + oracle/replay.py's mapping, over ring states (not full / wrapped), odd sample
    counts (two int64 per Philox block) and block offsets beyond 2^32."""
    import ctypes

    from oracle import replay as oreplay
    for last_id, batch, L_, S, T, base in [(3, 4, 10, 5, 2, 0), (25, 3, 10, 8, 3, 1280),
                                           (9, 1, 10, 1, 1, 256), (99, 7, 13, 33, 4, (1 << 33) + 5)]:
        rows = (ctypes.c_int64 * (S * T))()
        prob = ctypes.c_float(0.0)
        rc = lib.aa_rb_draw_tf_host(last_id, batch, L_, S, T, 0x1234567890, 11, 12, base, rows,
                                    ctypes.byref(prob))
        assert rc == 0
        a, c = oreplay.raw_draws_tf_layout(0x1234567890, 11, 12, base, S)
        want, probs = oreplay.rows_from_draws(a, c, last_id, batch, L_, T)
        assert np.array_equal(np.asarray(rows[:]).reshape(S, T), want)
        assert np.float32(prob.value) == probs[0]
    assert lib.aa_rb_draw_tf_host(0, 4, 10, 2, 2, 1, 2, 3, 0, rows, ctypes.byref(prob)) == -34
