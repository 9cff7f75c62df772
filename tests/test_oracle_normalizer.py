"""CPU: oracle/tensor_normalizer.py against the numeric cases of the reference's
tf_agents/utils/tensor_normalizer_test.py (ported; tolerances are tf.test's assertAllClose
defaults, rtol = atol = 1e-6, unless the reference states its own)."""
import numpy as np
import pytest

from oracle import tensor_normalizer as otn

ARR = [[1.3, 4.2, 7.5], [8.3, 2.2, 9.5], [3.3, 5.2, 6.5]]


def close(a, b, rtol=1e-6, atol=1e-6):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol,
                               atol=atol)


def test_parallel_variance_one_at_a_time():          # :32-42
    x = np.random.default_rng(0).standard_normal((5, 10))
    n, avg, m2, m2_c = 1, x[0], 0, 0
    for row in range(1, 5):
        n, avg, m2, m2_c = otn.parallel_variance_calculation(n, avg, m2, 1, x[row], 0, m2_c)
    close(avg, x.mean(axis=0))
    close(m2 / n, x.var(axis=0))


def test_parallel_variance_for_one_group():           # :44-56
    x = np.random.default_rng(1).standard_normal((5, 10))
    n, avg, var = 5, x.mean(0), x.var(0)
    new_n, new_avg, new_m2, _ = otn.parallel_variance_calculation(n, avg, var * n, 0, 0, 0, 0)
    assert new_n == 5
    close(new_avg, avg)
    close(new_m2 / n, var)


def test_parallel_variance_combines_groups():         # :58-76
    rng = np.random.default_rng(2)
    x1, x2 = rng.standard_normal((5, 10)), rng.standard_normal((15, 10))
    a1, a2 = x1.mean(0), x2.mean(0)
    m1, m2 = np.square(x1 - a1).sum(0), np.square(x2 - a2).sum(0)
    n, avg, m2_ab, _ = otn.parallel_variance_calculation(5, a1, m1, 15, a2, m2, m2 * 0.0)
    both = np.concatenate([x1, x2], 0)
    close(avg, both.mean(0))
    close(m2_ab / n, both.var(0))


@pytest.mark.parametrize("outer", [1, 2])
def test_ema_normalization(outer):                     # :157-180
    nrm = otn.EMANormalizer((3,))
    nrm.mean[:] = 10.0
    nrm.var[:] = 0.1
    vec, exp = [9.0, 10.0, 11.0], [-3.1622776601, 0.0, 3.1622776601]
    for _ in range(outer - 1):
        vec, exp = [vec] * 2, [exp] * 2
    close(nrm.normalize(np.asarray(vec, np.float32), variance_epsilon=0.0), exp, atol=1e-4)


def test_ema_should_not_center_mean():                 # :201-213
    nrm = otn.EMANormalizer((3,))
    nrm.mean[:] = 10.0
    nrm.var[:] = 0.01
    got = nrm.normalize(np.asarray([[9.0, 10.0, 11.0]], np.float32), center_mean=False,
                        variance_epsilon=0.0, clip_value=0.0)
    close(got, [[90.0, 100.0, 110.0]])


def test_ema_update_changes_variables():               # :105-121
    nrm = otn.EMANormalizer((3,))
    m0, v0 = nrm.mean.copy(), nrm.var.copy()
    nrm.update(np.asarray([[1.3, 4.2, 7.5]], np.float32))
    assert np.all(nrm.mean != m0) and np.all(nrm.var != v0)
    # :262-276 by hand: mean += 0.001 (x - 0); var += 0.001 ((x - 0)^2 - 1)
    close(nrm.mean, 0.001 * np.asarray([1.3, 4.2, 7.5]))
    close(nrm.var, 1 + 0.001 * (np.square([1.3, 4.2, 7.5]) - 1))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_streaming_reset(dtype):                       # :231-257
    nrm = otn.StreamingNormalizer((3,), dtype)
    orig = [nrm.count.copy(), nrm.avg.copy(), nrm.m2.copy(), nrm.m2_carry.copy()]
    nrm.update(np.asarray(ARR, dtype))
    nrm.reset()
    for a, b in zip(orig, [nrm.count, nrm.avg, nrm.m2, nrm.m2_carry]):
        close(a, b)
    close(nrm.count, 1e-8)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_streaming_update(dtype):                      # :263-318
    nrm = otn.StreamingNormalizer((3,), dtype)
    arr = np.asarray(ARR, dtype)
    data = arr
    for k, delta in enumerate([0.0, 1.0, -1.0]):
        nrm.update(arr + dtype(delta))
        if k:
            data = np.concatenate([data, arr + dtype(delta)], 0)
        n = data.shape[0]
        close(nrm.count, [n] * 3)
        close(nrm.avg, data.mean(0))
        close(nrm.m2, data.var(0) * n)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_streaming_normalization(dtype):               # :324-357
    rng = np.random.default_rng(3)
    nrm = otn.StreamingNormalizer((3,), dtype)
    norm_obs = rng.standard_normal((6, 2, 3))
    nrm.update(norm_obs.astype(dtype))
    view = rng.standard_normal((4, 3))
    got = nrm.normalize(view.astype(dtype), clip_value=-1, variance_epsilon=1e-6)
    close(got, (view - norm_obs.mean((0, 1))) / norm_obs.std((0, 1)),
          rtol=1e-5 if dtype == np.float32 else 1e-6, atol=1e-5)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_streaming_normalize_vs_numpy(dtype):          # :363-385
    nrm = otn.StreamingNormalizer((3,), dtype)
    arr = np.asarray(ARR, dtype)
    nrm.update(arr)
    eps = 1e-6
    close(nrm.normalize(arr, variance_epsilon=eps), (arr - arr.mean(0)) / (arr.std(0) + eps))
    mean, var = nrm.mean_var()                         # testMeanVariance :391-411
    close(mean, arr.mean(0))
    close(var, arr.var(0))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("case,iters", [("incremental_mean", 62), ("fixed_mean", 41),
                                        ("incremental_variance", 383), ("fixed_variance", 54)])
def test_streaming_long_runs(dtype, case, iters):      # :403-497, at the reference's own limits
    nrm = otn.StreamingNormalizer((3,), dtype)
    arr = np.asarray(ARR if case != "fixed_variance"
                     else [[-1.3, 4.2, 7.5], [8.3, -2.2, 9.5], [3.3, 5.2, -6.5]], dtype)
    chunks = []
    for i in range(iters):
        step = arr + dtype(100 * i) if case.startswith("incremental") else arr
        nrm.update(step)
        chunks.append(step.astype(np.float64))
        full = np.concatenate(chunks, 0)
        mean, var = nrm.mean_var()
        if case.endswith("mean"):
            close(mean, full.mean(0))
        else:
            close(var, full.var(0))
