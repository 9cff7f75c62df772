"""CPU: oracle/perm.py is a permutation for every n, deterministic in (seed, call), different across
calls, and well mixed (no reference vector exists: the reference's shuffle order is unpinned)."""
import numpy as np
import pytest

from oracle import perm


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 16, 17, 100, 4096, 4097, 26419])
def test_is_a_permutation(n):
    p = perm.random_permutation(n, 1234, 7)
    assert p.dtype == np.int64 and np.array_equal(np.sort(p), np.arange(n))


def test_deterministic_and_call_dependent():
    a = perm.random_permutation(1000, 5, 0)
    assert np.array_equal(a, perm.random_permutation(1000, 5, 0))
    assert not np.array_equal(a, perm.random_permutation(1000, 5, 1))
    assert not np.array_equal(a, perm.random_permutation(1000, 6, 0))


def test_well_mixed():
    n = 1 << 14
    p = perm.random_permutation(n, 99, 3)
    # displacement and rank correlation of a uniform random permutation: mean |p[i] - i| ~ n/3
    assert 0.30 * n < np.abs(p - np.arange(n)).mean() < 0.37 * n
    assert abs(np.corrcoef(p, np.arange(n))[0, 1]) < 0.03
    # neighbours are not kept together
    assert (np.abs(np.diff(p)) == 1).mean() < 0.01
