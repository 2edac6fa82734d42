"""Host logic of the post-search entry points (no GPU needed): constructor checks, the sigma-G coefficient,
likelihood curves, pixel prediction, calendar nights, and the loud failure of the device entry points without
a device."""

import numpy as np
import pytest

from oracle import post_search as ps


@pytest.fixture(scope="module")
def kb():
    import kbmod_amd.search as kb

    return kb


def test_sigma_g_clipping_host_side():
    from kbmod_amd.sigma_g_filter import SigmaGClipping, compute_likelihood_curves

    p = SigmaGClipping()
    assert (p.low_bnd, p.high_bnd, p.n_sigma, p.clip_negative) == (25, 75, 2, False)
    assert p.coeff == pytest.approx(0.7413, abs=1e-4) and p.coeff == pytest.approx(ps.find_sigma_g_coeff(25, 75), rel=1e-12)
    for kw in ({"n_sigma": -1.0}, {"low_bnd": 90.0, "high_bnd": 10.0}, {"high_bnd": 101.0}, {"low_bnd": -1.0}):
        with pytest.raises(ValueError):
            SigmaGClipping(**kw)
    for lo, hi in [(-1.0, 75.0), (25.0, 110.0), (75.0, 25.0)]:
        with pytest.raises(ValueError):
            SigmaGClipping.find_sigma_g_coeff(lo, hi)
    lh = np.array([(10.0 + i * 0.05) for i in range(20)])
    lh[2], lh[14] = 100.0, -100.0
    assert set(ps.clipped_sigma_g(lh)) == set(range(20)) - {2, 14}  # tests/test_sigma_g_filter.py:24-45, pinned on the oracle
    assert p.compute_clipped_sigma_g_matrix(np.zeros((0, 4))).shape == (0, 4)
    with pytest.raises(ValueError):
        p.compute_clipped_sigma_g_matrix(np.zeros(4))
    psi = np.array([[1.0, 2.0, np.nan]], dtype=np.float32)
    phi = np.array([[4.0, 0.0, 1.0]], dtype=np.float32)
    assert np.array_equal(compute_likelihood_curves(psi, phi, mask_value=np.nan),
                          ps.likelihood_curves(psi, phi, mask_value=np.nan), equal_nan=True)


def test_predict_pixel_locations():
    from kbmod_amd.stamp_utils import predict_pixel_locations

    rng = np.random.default_rng(1)
    times = np.sort(rng.random(9) * 4)
    x0, vx = rng.integers(-5, 50, 30), rng.uniform(-20, 20, 30)
    assert np.array_equal(predict_pixel_locations(times, x0, vx), ps.predict_pixel_locations(times, x0, vx))
    assert predict_pixel_locations(times, x0, vx, as_int=False).dtype == np.float64
    with pytest.raises(ValueError):
        predict_pixel_locations(times, x0, vx[:-1])


def test_calendar_nights():
    from kbmod_amd.stamp_utils import mjd_to_day

    assert mjd_to_day(60000) == "2023-02-25" == ps.mjd_to_day(60000)  # util_functions.py:63
    assert mjd_to_day(60000.99) == "2023-02-25" and mjd_to_day(60001.0) == "2023-02-26"
    assert mjd_to_day(57130.2) == "2015-04-18"


def test_device_entry_points_fail_loudly_without_a_device(kb):
    if kb.kb_has_gpu():
        pytest.skip("a device is present")
    from kbmod_amd.sigma_g_filter import SigmaGClipping
    from kbmod_amd.stamp_utils import DeviceStack

    with pytest.raises(RuntimeError):
        SigmaGClipping().compute_clipped_sigma_g_matrix(np.ones((3, 5)))
    with pytest.raises(RuntimeError):
        DeviceStack(np.zeros((2, 4, 4), dtype=np.float32))
