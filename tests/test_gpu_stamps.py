"""Stamp coadds on the device (kb_coadd_stamps behind kbmod_amd.stamp_utils) -- SURVEY.md section 8(f3).

Known answers follow the reference's tests/test_stamp_utils.py; parity is bit-exact (float32 of the
float64 result) against oracle/post_search.py's restatement of the per-trajectory loop of
append_coadds, on stacks with NaN pixels, stamps hanging over every image edge, masked epochs and
trajectories that leave the image."""

import numpy as np
import pytest

from oracle import post_search as ps

pytestmark = pytest.mark.gpu

ALL = ["sum", "mean", "median", "weighted"]


@pytest.fixture(scope="module")
def su():
    from kbmod_amd import stamp_utils

    return stamp_utils


def _kat_images():
    sci1 = np.array([[0, np.nan, np.nan], [0, np.nan, 0.5], [0, 1, 0.5]]).astype(np.float32)
    sci2 = np.array([[1, np.nan, 0.5], [1, 2, 0.5], [1, 2, 0.5]]).astype(np.float32)
    sci3 = np.array([[2, 3, 0.5], [2, 3, 0.5], [2, 3, 0.5]]).astype(np.float32)
    var = np.array([np.full((3, 3), v).astype(np.float32) for v in (0.1, 0.2, 0.5)])
    return np.array([sci1, sci2, sci3]), var


def test_make_coadds_simple(su):
    # tests/test_stamp_utils.py:143-214
    sci, var = _kat_images()
    stack = su.DeviceStack(sci, var)
    ones = np.ones((1, 3), dtype=int)
    got = stack.coadds(ones, ones, 1, ALL)
    assert got["sum"].dtype == np.float32 and got["sum"].shape == (1, 3, 3)
    assert np.allclose(got["sum"][0], [[3.0, 3.0, 1.0], [3.0, 5.0, 1.5], [3.0, 6.0, 1.5]], atol=1e-5)
    assert np.allclose(got["mean"][0], [[1.0, 3.0, 0.5], [1.0, 2.5, 0.5], [1.0, 2.0, 0.5]], atol=1e-5)
    assert np.allclose(got["median"][0], [[1.0, 3.0, 0.5], [1.0, 2.0, 0.5], [1.0, 2.0, 0.5]], atol=1e-5)
    w = 0.5294117647058824
    assert np.allclose(got["weighted"][0], [[w, 3.0, 0.5], [w, 2.2857142857142856, 0.5], [w, 1.5294117647058822, 0.5]], atol=1e-5)
    got = stack.coadds(ones, ones, 1, ALL, to_include=np.array([[True, True, False]]))
    assert np.allclose(got["sum"][0], [[1.0, 0.0, 0.5], [1.0, 2.0, 1.0], [1.0, 3.0, 1.0]], atol=1e-5)
    assert np.allclose(got["mean"][0], [[0.5, 0.0, 0.5], [0.5, 2.0, 0.5], [0.5, 1.5, 0.5]], atol=1e-5)
    assert np.allclose(got["median"][0], [[0.0, 0.0, 0.5], [0.0, 2.0, 0.5], [0.0, 1.0, 0.5]], atol=1e-5)
    t = 0.3333333333333333
    assert np.allclose(got["weighted"][0], [[t, 0.0, 0.5], [t, 2.0, 0.5], [t, 1.3333333333333333, 0.5]], atol=1e-5)
    # :216-224 nothing selected -> zeros
    got = stack.coadds(ones, ones, 1, ALL, to_include=np.zeros((1, 3), dtype=bool))
    for c in ALL:
        assert np.array_equal(got[c][0], np.zeros((3, 3), dtype=np.float32))


def _random_case(seed, T, H, W, n):
    rng = np.random.default_rng(seed)
    sci = (rng.standard_normal((T, H, W)) * 20).astype(np.float32)
    var = (rng.random((T, H, W)) * 4 + 0.5).astype(np.float32)
    sci[rng.random((T, H, W)) < 0.05] = np.nan
    var[rng.random((T, H, W)) < 0.03] = np.nan
    var[rng.random((T, H, W)) < 0.02] = 0.0
    sci[:, H // 2, W // 2] = np.nan  # a pixel without data at any epoch
    times = np.sort(rng.random(T) * 3.0)
    times[0] = 0.0
    x0 = rng.integers(-6, W + 6, n)
    y0 = rng.integers(-6, H + 6, n)
    vx = rng.uniform(-9, 9, n)
    vy = rng.uniform(-9, 9, n)
    obs_valid = rng.random((n, T)) < 0.8
    obs_valid[0] = False
    obs_valid[1] = True
    return sci, var, times, x0, y0, vx, vy, obs_valid


@pytest.mark.parametrize("radius", [1, 3, 10])
@pytest.mark.parametrize("use_mask", [False, True])
def test_coadds_equal_oracle(su, radius, use_mask):
    sci, var, times, x0, y0, vx, vy, obs_valid = _random_case(11 + radius, 17, 40, 52, 60)
    xv = ps.predict_pixel_locations(times, x0, vx)
    yv = ps.predict_pixel_locations(times, y0, vy)
    assert np.array_equal(xv, su.predict_pixel_locations(times, x0, vx))
    inc = obs_valid if use_mask else None
    got = su.DeviceStack(sci, var).coadds(xv, yv, radius, ALL, to_include=inc)
    exp = ps.coadds_for_trajectories(sci, var, xv, yv, inc, radius, ALL)
    for c in ALL:
        assert got[c].dtype == np.float32
        assert np.array_equal(got[c], exp[c], equal_nan=True), c


def test_median_between_65_and_512_epochs_uses_lds_columns(su):
    sci, var, times, x0, y0, vx, vy, obs_valid = _random_case(78, 90, 24, 30, 8)
    xv = ps.predict_pixel_locations(times, x0, vx * 0.3)
    yv = ps.predict_pixel_locations(times, y0, vy * 0.3)
    got = su.DeviceStack(sci).coadds(xv, yv, 3, ["median"], to_include=obs_valid)
    exp = ps.coadds_for_trajectories(sci, None, xv, yv, obs_valid, 3, ["median"])
    assert np.array_equal(got["median"], exp["median"])


def test_median_of_a_long_stack_uses_the_scratch_path(su):
    # more than 512 epochs: the per-pixel columns no longer fit the LDS buffer
    sci, var, times, x0, y0, vx, vy, obs_valid = _random_case(77, 600, 24, 30, 6)
    xv = ps.predict_pixel_locations(times, x0, vx * 0.2)
    yv = ps.predict_pixel_locations(times, y0, vy * 0.2)
    got = su.DeviceStack(sci).coadds(xv, yv, 2, ["median"], to_include=obs_valid)
    exp = ps.coadds_for_trajectories(sci, None, xv, yv, obs_valid, 2, ["median"])
    assert np.array_equal(got["median"], exp["median"])


def test_append_coadds_table(su):
    sci, var, times, x0, y0, vx, vy, obs_valid = _random_case(5, 30, 64, 64, 200)
    table = {"x": x0, "y": y0, "vx": vx, "vy": vy, "obs_valid": obs_valid}
    stack = su.DeviceStack(sci, var, zeroed_times=times)
    su.append_coadds(table, stack, ["mean", "median"], 4, valid_only=True)
    xv = ps.predict_pixel_locations(times, x0, vx)
    yv = ps.predict_pixel_locations(times, y0, vy)
    exp = ps.coadds_for_trajectories(sci, var, xv, yv, obs_valid, 4, ["mean", "median"])
    assert np.array_equal(table["coadd_mean"], exp["mean"]) and np.array_equal(table["coadd_median"], exp["median"])
    assert "coadd_sum" not in table
    su.append_coadds(table, stack, ["sum"], 4, valid_only=False)
    assert np.array_equal(table["coadd_sum"], ps.coadds_for_trajectories(sci, var, xv, yv, None, 4, ["sum"])["sum"])


@pytest.mark.parametrize("valid_only", [True, False])
def test_nightly_coadds(su, valid_only):
    # stamp_filters.py:85-102, 150-166: one coadd per calendar night next to the overall one
    sci, var, times, x0, y0, vx, vy, obs_valid = _random_case(21, 24, 48, 56, 80)
    mjd = 60000.3 + np.sort(np.concatenate([np.arange(8) * 0.01, 1.0 + np.arange(9) * 0.01, 3.9 + np.arange(7) * 0.03]))
    zeroed = mjd - mjd[0]
    obs_valid[2, :8] = False  # a trajectory without a valid epoch in the first night
    table = {"x": x0, "y": y0, "vx": vx, "vy": vy, "obs_valid": obs_valid}
    stack = su.DeviceStack(sci, var, zeroed_times=zeroed, times=mjd)
    su.append_coadds(table, stack, ALL, 3, valid_only=valid_only, nightly=True)
    xv = ps.predict_pixel_locations(zeroed, x0, vx)
    yv = ps.predict_pixel_locations(zeroed, y0, vy)
    exp = ps.append_coadds_columns(sci, var, mjd, xv, yv, obs_valid if valid_only else None, 3, ALL, True)
    nights = sorted(k for k in exp if k.startswith("coadd_sum_"))
    assert len(nights) == 3 and nights[0] == "coadd_sum_2023-02-25"  # 60000 is 2023-02-25 (util_functions.py:63)
    assert set(exp) == {k for k in table if k.startswith("coadd_")}
    for k, v in exp.items():
        assert table[k].dtype == np.float32
        assert np.array_equal(table[k], v, equal_nan=True), k


@pytest.mark.parametrize("radius", [1, 4, 10])
def test_all_stamps(su, radius):
    sci, var, times, x0, y0, vx, vy, _ = _random_case(31 + radius, 19, 44, 50, 70)
    table = {"x": x0, "y": y0, "vx": vx, "vy": vy}
    stack = su.DeviceStack(sci, zeroed_times=times)
    su.append_all_stamps(table, stack, radius)
    xv = ps.predict_pixel_locations(times, x0, vx)
    yv = ps.predict_pixel_locations(times, y0, vy)
    exp = ps.all_stamps_for_trajectories(sci, xv, yv, radius)
    assert table["all_stamps"].dtype == np.float32 and table["all_stamps"].shape == exp.shape
    assert np.array_equal(table["all_stamps"], exp, equal_nan=True)
    with pytest.raises(ValueError):
        su.append_all_stamps(table, stack, 0)


def test_errors(su):
    sci, var = _kat_images()
    stack = su.DeviceStack(sci)
    ones = np.ones((1, 3), dtype=int)
    with pytest.raises(ValueError):
        stack.coadds(ones, ones, 0, ["sum"])
    with pytest.raises(ValueError):
        stack.coadds(ones[:, :2], ones[:, :2], 1, ["sum"])
    with pytest.raises(ValueError):
        stack.coadds(ones, ones, 1, ["max"])
    with pytest.raises(RuntimeError):
        stack.coadds(ones, ones, 1, ["weighted"])  # no variance uploaded
    assert stack.coadds(np.zeros((0, 3), dtype=int), np.zeros((0, 3), dtype=int), 2, ["sum"])["sum"].shape == (0, 5, 5)
