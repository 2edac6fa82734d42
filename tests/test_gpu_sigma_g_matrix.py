"""Batched sigma-G clipping on the device (kb_sigma_g_clip_matrix behind
kbmod_amd.sigma_g_filter.SigmaGClipping) -- SURVEY.md section 8(f1).

The cases follow the reference's tests/test_sigma_g_filter.py; parity is against
oracle/post_search.py (bit-exact masks: both use the scalar lerp definition) and the
torch-generated golden vectors (identical away from one-ulp ties)."""

import os
import warnings

import numpy as np
import pytest

from oracle import post_search as ps
from tests.test_oracle_sigma_g_matrix import GOLD, near_bound

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sg():
    from kbmod_amd import sigma_g_filter

    return sigma_g_filter


def test_create(sg):
    # tests/test_sigma_g_filter.py:11-22
    p = sg.SigmaGClipping()
    assert (p.low_bnd, p.high_bnd, p.n_sigma, p.clip_negative) == (25.0, 75.0, 2.0, False)
    assert p.coeff == pytest.approx(0.7413, abs=1e-4)
    for kw in ({"n_sigma": -1.0}, {"low_bnd": 90.0, "high_bnd": 10.0}, {"high_bnd": 101.0}, {"low_bnd": -1.0}):
        with pytest.raises(ValueError):
            sg.SigmaGClipping(**kw)


def test_matrix_known_answers(sg):
    # tests/test_sigma_g_filter.py:47-76, 96-120
    lh = np.array([[(10.0 + i * 0.05) for i in range(20)] for _ in range(5)])
    lh[1, 2], lh[1, 14], lh[2, 0] = 100.0, -100.0, 50.0
    lh[3, 2], lh[3, 14], lh[3, 0] = 100.0, -100.0, 50.0
    lh[4, 7] = lh[4, 8] = lh[4, 11] = np.nan
    got = sg.SigmaGClipping().compute_clipped_sigma_g_matrix(lh)
    assert got.dtype == bool and np.array_equal(got, np.isfinite(lh) & (lh < 20.0) & (lh > 0.0))
    lh = np.array([[5 for _ in range(10)], [5.1 for _ in range(10)]])
    assert sg.SigmaGClipping().compute_clipped_sigma_g_matrix(lh).all()
    lh = np.array([[5 for _ in range(20)], [(-1.0 + i * 0.2) for i in range(20)], [(-100.0 + i * 0.2) for i in range(20)]])
    exp = np.array([[True] * 20, [False] * 3 + [True] * 17, [False] * 20])
    assert np.array_equal(sg.SigmaGClipping(clip_negative=True).compute_clipped_sigma_g_matrix(lh), exp)


@pytest.mark.parametrize("num_obs", [10, 20, 50])
@pytest.mark.parametrize("clipped", [True, False])
def test_batch_equals_single(sg, num_obs, clipped):
    # tests/test_sigma_g_filter.py:163-192
    for num_extreme in range(4):
        rng = np.random.default_rng(100)
        data = 10.0 * rng.random((20, num_obs)) - 0.5
        for row in range(20):
            for _ in range(num_extreme):
                data[row, int(num_obs * rng.random())] = 100.0 * rng.random() - 50.0
        clipper = sg.SigmaGClipping(25, 75, clip_negative=clipped)
        batch = clipper.compute_clipped_sigma_g_matrix(data)
        for row in range(20):
            ind = ps.clipped_sigma_g(data[row], 25, 75, 2, clip_negative=clipped)  # the reference's single-curve helper
            assert np.array_equal(batch[row], [(i in ind) for i in range(num_obs)])


@pytest.mark.parametrize("shape", [(1, 1), (3, 2), (257, 64), (1000, 65), (64, 999), (9, 4096)])
@pytest.mark.parametrize("clip_negative", [False, True])
def test_matrix_equals_oracle(sg, shape, clip_negative):
    rng = np.random.default_rng(shape[0] * 7919 + shape[1])
    lh = (8.0 * rng.standard_normal(shape) + 3.0).astype(np.float32)
    lh[rng.random(shape) < 0.07] = np.nan
    lh[rng.random(shape) < 0.02] *= 30.0
    if shape[0] > 4:
        lh[2, :] = np.nan
        lh[3, :] = np.float32(1.25)
        lh[4, :shape[1] // 2] = np.inf
    clipper = sg.SigmaGClipping(20, 80, 2.5, clip_negative=clip_negative)
    got = clipper.compute_clipped_sigma_g_matrix(lh)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp, _, _ = ps.clipped_sigma_g_matrix(lh, 20, 80, 2.5, clip_negative)
    assert np.array_equal(got, exp)


def test_matrix_against_torch_golden(sg):
    g = np.load(GOLD)
    for i in range(sum(1 for k in g.files if k.startswith("lh_"))):
        lo, hi, ns, clip = g[f"cfg_{i}"]
        got = sg.SigmaGClipping(lo, hi, ns, clip_negative=bool(clip)).compute_clipped_sigma_g_matrix(g[f"lh_{i}"])
        differ = got != g[f"valid_{i}"]
        assert not (differ & ~near_bound(g[f"lh_{i}"], g[f"lower_{i}"], g[f"upper_{i}"])).any()


def test_errors_and_empty(sg, kb):
    assert sg.SigmaGClipping().compute_clipped_sigma_g_matrix(np.zeros((0, 5))).shape == (0, 5)
    with pytest.raises(ValueError):
        sg.SigmaGClipping().compute_clipped_sigma_g_matrix(np.zeros(5))
    with pytest.raises(RuntimeError):
        kb.sigma_g_clip_matrix(np.zeros((2, 5000), dtype=np.float32))
    with pytest.raises(RuntimeError):
        kb.sigma_g_clip_matrix(np.zeros((2, 5), dtype=np.float32), low_bnd=80.0, high_bnd=20.0)


def test_search_curves_to_clip_pipeline(sg, kb):
    """search_all -> get_all_psi_phi_curves -> likelihood curves -> batched clip, as load_and_filter_results
    chains them (run_search.py:251-337), against the oracle's restatement of the same chain."""
    from kbmod_amd import fake_data as fd
    from tests import util

    st = util.make_stack(24, 60, 70, seed=9, noise=2.0, psf=1.0, objects=[(20, 15, 12.0, 7.0, 300.0)], mask_fraction=0.02)
    s = kb.StackSearch(st.sci, st.var, st.psfs, st.zeroed_times)
    s.set_min_obs(10)
    s.set_min_lh(3.0)
    vx, vy = fd.kbmod_v1_candidates(16, 2.0, 20.0, 8, 0.0, 1.2)
    s.search_all(util.trajectories(kb, vx, vy), True)
    trjs = s.get_results(0, 500)
    assert len(trjs) > 50
    curves = np.asarray(s.get_all_psi_phi_curves(trjs))
    T = 24
    lh = sg.compute_likelihood_curves(curves[:, :T], curves[:, T:], mask_value=np.nan)
    assert np.array_equal(lh, ps.likelihood_curves(curves[:, :T], curves[:, T:], mask_value=np.nan), equal_nan=True)
    got = sg.SigmaGClipping().compute_clipped_sigma_g_matrix(lh)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp, _, _ = ps.clipped_sigma_g_matrix(lh)
    assert np.array_equal(got, exp) and 0 < got.sum() < got.size
