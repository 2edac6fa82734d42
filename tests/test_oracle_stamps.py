"""Pins the stamp/coadd restatement of oracle/post_search.py to the reference's known answers
(tests/test_stamp_utils.py) and its numpy/torch building blocks (np.nansum, np.nanmean,
torch.nanmedian) on random stacks."""

import warnings

import numpy as np
import pytest

from oracle import post_search as ps


def test_extract_single_stamp():
    # tests/test_stamp_utils.py:20-55
    sci = np.arange(0, 120, dtype=np.single).reshape(10, 12)
    assert np.allclose(ps.extract_stamp(sci, 2, 2, 2), sci[0:5, 0:5])
    assert np.allclose(ps.extract_stamp(sci, 8, 5, 1), sci[4:7, 7:10])
    exp = np.array([[np.nan, np.nan, np.nan], [10.0, 11.0, np.nan], [22.0, 23.0, np.nan]])
    assert np.allclose(ps.extract_stamp(sci, 11, 0, 1), exp, equal_nan=True)
    assert np.isnan(ps.extract_stamp(sci, 20, 20, 1)).all() and np.isnan(ps.extract_stamp(sci, -5, -5, 1)).all()
    exp = np.full((3, 3), np.nan)
    exp[2][2] = 0.0
    assert np.allclose(ps.extract_stamp(sci, -1, -1, 1), exp, equal_nan=True)


def test_extract_stamp_stack():
    # tests/test_stamp_utils.py:57-102
    times = np.arange(4)
    data = np.arange(0, 4 * 12 * 10).reshape(4, 10, 12)
    x_vals = (-2.0 + 2.0 * times + 0.5).astype(int)
    y_vals = np.full(4, 1.0 + 0.5).astype(int)
    st = ps.extract_stamp_stack(data, x_vals, y_vals, 2)
    assert st.shape == (4, 5, 5)
    assert np.allclose(st[:, 2, 2], [np.nan, 132.0, 254.0, 376.0], equal_nan=True)
    with pytest.raises(ValueError):
        ps.extract_stamp_stack(data, x_vals, y_vals, -1)
    with pytest.raises(ValueError):
        ps.extract_stamp_stack(data, x_vals[:-1], y_vals, 2)
    st = ps.extract_stamp_stack(data, x_vals, y_vals, 2, to_include=np.array([True, True, False, True]))
    assert len(st) == 3 and np.isnan(st[0][2, 2]) and st[1][2, 2] == 132.0 and st[2][2, 2] == 376.0
    st = ps.extract_stamp_stack(data, x_vals, y_vals, 2, to_include=np.array([1, 2]))
    assert len(st) == 2 and st[0][2, 2] == 132.0 and st[1][2, 2] == 254.0
    assert ps.extract_stamp_stack(np.array([]).reshape(0, 10, 12), [], [], 2).shape == (0, 5, 5)


def _kat_images():
    sci1 = np.array([[0, np.nan, np.nan], [0, np.nan, 0.5], [0, 1, 0.5]]).astype(np.float32)
    sci2 = np.array([[1, np.nan, 0.5], [1, 2, 0.5], [1, 2, 0.5]]).astype(np.float32)
    sci3 = np.array([[2, 3, 0.5], [2, 3, 0.5], [2, 3, 0.5]]).astype(np.float32)
    var = np.array([np.full((3, 3), v).astype(np.float32) for v in (0.1, 0.2, 0.5)])
    return np.array([sci1, sci2, sci3]), var


def test_make_coadds_simple():
    # tests/test_stamp_utils.py:143-224
    sci, var = _kat_images()
    x = y = np.array([1.0, 1.0, 1.0])
    st = ps.extract_stamp_stack(sci, x, y, 1)
    vs = ps.extract_stamp_stack(var, x, y, 1)
    assert np.allclose(ps.coadd_sum(st), [[3.0, 3.0, 1.0], [3.0, 5.0, 1.5], [3.0, 6.0, 1.5]], atol=1e-5)
    assert np.allclose(ps.coadd_mean(st), [[1.0, 3.0, 0.5], [1.0, 2.5, 0.5], [1.0, 2.0, 0.5]], atol=1e-5)
    assert np.allclose(ps.coadd_median(st), [[1.0, 3.0, 0.5], [1.0, 2.0, 0.5], [1.0, 2.0, 0.5]], atol=1e-5)
    w = 0.5294117647058824
    assert np.allclose(ps.coadd_weighted(st, vs), [[w, 3.0, 0.5], [w, 2.2857142857142856, 0.5], [w, 1.5294117647058822, 0.5]],
                       atol=1e-5)
    mask = np.array([True, True, False])
    st = ps.extract_stamp_stack(sci, x, y, 1, to_include=mask)
    vs = ps.extract_stamp_stack(var, x, y, 1, to_include=mask)
    assert np.allclose(ps.coadd_sum(st), [[1.0, 0.0, 0.5], [1.0, 2.0, 1.0], [1.0, 3.0, 1.0]], atol=1e-5)
    assert np.allclose(ps.coadd_mean(st), [[0.5, 0.0, 0.5], [0.5, 2.0, 0.5], [0.5, 1.5, 0.5]], atol=1e-5)
    assert np.allclose(ps.coadd_median(st), [[0.0, 0.0, 0.5], [0.0, 2.0, 0.5], [0.0, 1.0, 0.5]], atol=1e-5)
    t = 0.3333333333333333
    assert np.allclose(ps.coadd_weighted(st, vs), [[t, 0.0, 0.5], [t, 2.0, 0.5], [t, 1.3333333333333333, 0.5]], atol=1e-5)
    empty = np.array([]).reshape(0, 3, 3).astype(np.float32)
    for got in (ps.coadd_sum(empty), ps.coadd_mean(empty), ps.coadd_median(empty), ps.coadd_weighted(empty, empty)):
        assert np.array_equal(got, np.zeros((3, 3)))


def test_extract_curve_values():
    # tests/test_stamp_utils.py:226-273
    sci = list(_kat_images()[0])
    v = ps.extract_curve_values(sci, np.array([1.0, 1.0, 1.0]), np.array([1.0, 1.0, 1.0]))
    assert np.isnan(v[0]) and v[1] == 2.0 and v[2] == 3.0
    x_vals = np.array([[1, 1, 1], [0, 0, 5], [2, 2, 2], [0, 0, 0]])
    y_vals = np.array([[1, 1, 1], [0, 0, 0], [0, 0, 0], [-1, 0, 0]])
    exp = np.array([[np.nan, 2.0, 3.0], [0.0, 1.0, np.nan], [np.nan, 0.5, 0.5], [np.nan, 1.0, 2.0]])
    assert np.allclose(ps.extract_curve_values(sci, x_vals, y_vals), exp, equal_nan=True)


def test_building_blocks_on_random_stacks():
    """The explicit sequential sums / lower median are the numpy / torch routines the reference calls."""
    import torch

    rng = np.random.default_rng(3)
    for shape in [(7, 5, 5), (64, 21, 21), (2, 3, 3), (1, 3, 3)]:
        st = rng.standard_normal(shape).astype(np.float32).astype(np.float64) * 50
        st[rng.random(shape) < 0.2] = np.nan
        st[:, 0, 0] = np.nan
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert np.array_equal(ps.coadd_sum(st), np.nansum(st, axis=0))
            masked = ps._mask_all_nans(st)
            assert np.array_equal(ps.coadd_mean(st), np.nanmean(masked, axis=0))
            med, _ = torch.nanmedian(torch.tensor(st), dim=0)
            med[torch.isnan(med)] = 0.0
            assert np.array_equal(ps.coadd_median(st), med.numpy())


def test_predict_pixel_locations():
    # truncation toward zero: -0.3 + 0.5 -> 0, -1.2 + 0.5 -> 0 (not -1)
    got = ps.predict_pixel_locations(np.array([0.0, 1.0, 2.0]), np.array([0, 5]), np.array([-0.6, 1.5]))
    assert np.array_equal(got, [[0, 0, 0], [5, 7, 8]])


def test_grid_filter_known_answers():
    # tests/test_clustering_grid.py:8-57, 93-108 (Trajectory(x, y, vx, vy, flux, lh, obs_count))
    t = [(0, 0, 0.0, 0.0, 10.0), (21, 21, 10.0, 10.0, 10.0), (21, 21, 0.0, 0.0, 10.0), (21, 21, 0.0, 0.0, 100.0),
         (0, 0, 0.0, 0.0, 5.0), (0, 0, 0.0, 0.0, 15.0)]
    x, y, vx, vy, lh = (np.array(c) for c in zip(*t))
    assert ps.grid_filter_indices(x[:1], y[:1], vx[:1], vy[:1], lh[:1]) == [0]
    assert ps.grid_filter_indices(x[:4], y[:4], vx[:4], vy[:4], lh[:4]) == [0, 1, 3]
    assert ps.grid_filter_indices(x[:5], y[:5], vx[:5], vy[:5], lh[:5]) == [0, 1, 3]
    assert ps.grid_filter_indices(x, y, vx, vy, lh, bin_width=10, max_time=1.0) == [5, 1, 3]
    with pytest.raises(ValueError):
        ps.grid_filter_indices(x, y, vx, vy, lh, bin_width=0)
    with pytest.raises(ValueError):
        ps.grid_filter_indices(x, y, vx, vy, lh, max_time=-1.0)
