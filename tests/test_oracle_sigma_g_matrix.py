"""Pins oracle/post_search.py (batched sigma-G, likelihood curves) to the reference's own known
answers (tests/test_sigma_g_filter.py) and to vectors generated with torch in the build container
(tests/golden/make_golden_sigma_g_matrix.py)."""

import os
import warnings

import numpy as np
import pytest

from oracle import post_search as ps

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sigma_g_matrix.npz")


def near_bound(lh, lower, upper, ulps=4):
    """Points within a few float32 ulps of a clipping bound: torch's fused lerp may move the bound by one ulp."""
    t = np.asarray(lh).astype(np.float32)
    with np.errstate(invalid="ignore"):
        tol = ulps * np.spacing(np.maximum(np.abs(lower), np.abs(upper)).astype(np.float32))[:, None]
        return (np.abs(t - lower[:, None]) <= tol) | (np.abs(t - upper[:, None]) <= tol)


def test_coeff_known_answers():
    # tests/test_sigma_g_filter.py:16, 194-199
    assert ps.find_sigma_g_coeff(25.0, 75.0) == pytest.approx(0.7413, abs=1e-4)
    for lo, hi in [(-1.0, 75.0), (25.0, 110.0), (75.0, 25.0)]:
        with pytest.raises(ValueError):
            ps.find_sigma_g_coeff(lo, hi)


def test_single_curve_known_answers():
    # tests/test_sigma_g_filter.py:24-45
    lh = np.array([(10.0 + i * 0.05) for i in range(20)])
    assert set(ps.clipped_sigma_g(lh)) == set(range(20))
    lh[2], lh[14] = 100.0, -100.0
    assert set(ps.clipped_sigma_g(lh)) == set(range(20)) - {2, 14}
    lh[0] = 50.0
    assert set(ps.clipped_sigma_g(lh)) == set(range(20)) - {0, 2, 14}
    # :78-94
    lh = np.array([(-1.0 + i * 0.2) for i in range(20)])
    lh[2], lh[14] = 20.0, -20.0
    assert set(ps.clipped_sigma_g(lh, clip_negative=True)) == {i for i in range(20) if i > 2 and i != 14}
    assert len(ps.clipped_sigma_g(np.array([(-100.0 + i * 0.2) for i in range(10)]), clip_negative=True)) == 0


def test_matrix_known_answers():
    # tests/test_sigma_g_filter.py:47-67
    lh = np.array([[(10.0 + i * 0.05) for i in range(20)] for _ in range(5)])
    lh[1, 2], lh[1, 14], lh[2, 0] = 100.0, -100.0, 50.0
    lh[3, 2], lh[3, 14], lh[3, 0] = 100.0, -100.0, 50.0
    lh[4, 7] = lh[4, 8] = lh[4, 11] = np.nan
    valid, _, _ = ps.clipped_sigma_g_matrix(lh)
    assert np.array_equal(valid, np.isfinite(lh) & (lh < 20.0) & (lh > 0.0))
    # :69-76 identical values -> the 1e-5 floor keeps all of them
    lh = np.array([[5 for _ in range(10)], [5.1 for _ in range(10)]])
    assert ps.clipped_sigma_g_matrix(lh)[0].all()
    # :96-120
    lh = np.array([[5 for _ in range(20)], [(-1.0 + i * 0.2) for i in range(20)], [(-100.0 + i * 0.2) for i in range(20)]])
    exp = np.array([[True] * 20, [False] * 3 + [True] * 17, [False] * 20])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert np.array_equal(ps.clipped_sigma_g_matrix(lh, clip_negative=True)[0], exp)


@pytest.mark.parametrize("num_obs", [10, 20, 50])
@pytest.mark.parametrize("clipped", [True, False])
@pytest.mark.parametrize("num_extreme", [0, 1, 2, 3])
def test_batch_equals_single(num_obs, clipped, num_extreme):
    # tests/test_sigma_g_filter.py:163-192
    rng = np.random.default_rng(100)
    data = 10.0 * rng.random((20, num_obs)) - 0.5
    for row in range(20):
        for _ in range(num_extreme):
            data[row, int(num_obs * rng.random())] = 100.0 * rng.random() - 50.0
    batch, _, _ = ps.clipped_sigma_g_matrix(data, clip_negative=clipped)
    for row in range(20):
        ind = ps.clipped_sigma_g(data[row], clip_negative=clipped)
        assert np.array_equal(batch[row], [(i in ind) for i in range(num_obs)])


def test_matrix_against_torch_golden():
    g = np.load(GOLD)
    n_cases = sum(1 for k in g.files if k.startswith("lh_"))
    assert n_cases >= 6
    for i in range(n_cases):
        lo, hi, ns, clip = g[f"cfg_{i}"]
        valid, lower, upper = ps.clipped_sigma_g_matrix(g[f"lh_{i}"], lo, hi, ns, bool(clip))
        ok = ~np.isnan(g[f"lower_{i}"])
        # the restated quantile arithmetic agrees with torch to two ulps on the bounds ...
        assert np.allclose(lower[ok], g[f"lower_{i}"][ok], rtol=3e-7, atol=1e-6)
        assert np.allclose(upper[ok], g[f"upper_{i}"][ok], rtol=3e-7, atol=1e-6)
        assert np.array_equal(np.isnan(lower), np.isnan(g[f"lower_{i}"]))
        # ... and the masks are identical away from one-ulp ties
        differ = valid != g[f"valid_{i}"]
        assert not (differ & ~near_bound(g[f"lh_{i}"], g[f"lower_{i}"], g[f"upper_{i}"])).any()


def test_likelihood_curves():
    psi = np.array([[1.0, 2.0, np.nan, 4.0], [1.0, 1.0, 1.0, 1.0]], dtype=np.float32)
    phi = np.array([[4.0, 0.0, 1.0, 16.0], [1.0, np.inf, 4.0, 1.0]], dtype=np.float32)
    got = ps.likelihood_curves(psi, phi, obs_valid=[[True, True, True, False], [True] * 4], mask_value=np.nan)
    exp = np.array([[0.5, np.nan, np.nan, np.nan], [1.0, np.nan, 0.5, 1.0]], dtype=np.float32)
    assert np.array_equal(got, exp, equal_nan=True)
