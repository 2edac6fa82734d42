"""Known answers observed from the real reference and held by its own tests, asserted on the
oracle AND on the product's host code paths (no GPU needed): the trajectory evaluator with the in-search sigma-G clip,
the sigma-G keep-set, the online cluster grid and the single-curve sigma-G clip."""

import numpy as np
import pytest

from oracle import post_search as ps

COEFF = 0.7413


@pytest.fixture(scope="module")
def kb():
    import kbmod_amd.search as kb

    return kb


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as o

    o.build()
    return o


# 8 epochs, phi = 0.5, psi = 1 + 0.1 i with epoch 3 set to 100, sigma-G [25, 75] on
# -> lh 5.077963, flux 2.714286, obs_count 8 (values the survey observed from kernels.cu:154-242 run on the host).
LITERAL_LH, LITERAL_FLUX, LITERAL_OBS = 5.077963, 2.714286, 8


def test_appendix_b_literals_on_the_oracle(orc):
    T = 8
    psi = [np.full((4, 4), 1 + 0.1 * i, np.float32) for i in range(T)]
    psi[3][:] = 100
    phi = [np.full((4, 4), 0.5, np.float32) for _ in range(T)]
    pp = orc.PsiPhi(psi, phi, np.arange(T, dtype=float))
    p = pp.default_params(do_sigmag_filter=1, sgl_L=0.25, sgl_H=0.75, sigmag_coeff=COEFF, min_lh=0.0)
    r = pp.evaluate_kernel(1, 1, 0.0, 0.0, p)
    assert int(r["obs_count"]) == LITERAL_OBS
    assert float(r["lh"]) == pytest.approx(LITERAL_LH, abs=2e-6)
    assert float(r["flux"]) == pytest.approx(LITERAL_FLUX, abs=2e-6)
    # ... and the keep-set of SigmaGFilteredIndicesCU([-1,-1,-1,0,1,2,2,2,5.46], .25, .75, .7413, 2.0): sorted slots 0..7
    # = original indices 0..7 (the 5.46 goes)
    assert list(orc.sigmag_filtered_indices([-1, -1, -1, 0, 1, 2, 2, 2, 5.46], 0.25, 0.75, COEFF, 2.0)) == list(range(8))


def test_appendix_b_literals_on_the_product_host_evaluator(kb, orc):
    """kb_evaluate_trajectory_host through the C ABI: the host instantiation of the evaluator that the device kernels
    share (csrc/search_math.h); needs no device.  The array bytes come from the product's own fill routine."""
    import ctypes as C

    from kbmod_amd.capi import Meta, Params, load_lib

    T = 8
    psi = [np.full((4, 4), 1 + 0.1 * i, np.float32) for i in range(T)]
    psi[3][:] = 100
    phi = [np.full((4, 4), 0.5, np.float32) for _ in range(T)]
    times = np.arange(T, dtype=np.float64)
    arr = kb.PsiPhiArray()
    kb.fill_psi_phi_array(arr, 4, psi, phi, list(times))
    host = np.array([[arr.read_psi_phi(t, r, c).psi, arr.read_psi_phi(t, r, c).phi] for t in range(T) for r in range(4)
                     for c in range(4)], dtype=np.float32).ravel()
    meta = Meta(T, 4, 4, 16, 2 * 16 * T, 4, 2 * 16 * T * 4, 4, 0.0, 0.0, 1.0, 0.0, 0.0, 1.0)
    lib = load_lib()
    lib.kb_evaluate_trajectory_host.argtypes = [C.POINTER(Meta), C.c_void_p, C.c_void_p, Params, C.c_void_p]
    t = np.zeros(1, dtype=orc.TRJ_DTYPE)
    t["x"], t["y"] = 1, 1
    p = Params(0, 0.0, 1, 0.25, 0.75, COEFF, -1, 0, 0, 0, 0, 8, 0)
    assert lib.kb_evaluate_trajectory_host(C.byref(meta), host.ctypes.data, times.ctypes.data, p, t.ctypes.data) == 0
    assert int(t["obs_count"][0]) == LITERAL_OBS
    assert float(t["lh"][0]) == pytest.approx(LITERAL_LH, abs=2e-6) and float(t["flux"][0]) == pytest.approx(LITERAL_FLUX, abs=2e-6)
    u = np.zeros(1, dtype=orc.TRJ_DTYPE)
    u["x"], u["y"] = 1, 1
    p.do_sigmag_filter = 0
    assert lib.kb_evaluate_trajectory_host(C.byref(meta), host.ctypes.data, times.ctypes.data, p, u.ctypes.data) == 0
    assert int(u["obs_count"][0]) == 8 and float(u["flux"][0]) == pytest.approx((9.5 + 100.0) / 4.0, rel=1e-6)  # unclipped
    assert kb.sigmag_filtered_indices([-1.0, -1.0, -1.0, 0.0, 1.0, 2.0, 2.0, 2.0, 5.46], 0.25, 0.75, COEFF, 2.0) == list(range(8))


@pytest.mark.gpu
def test_appendix_b_literals_through_stack_search(kb):
    """The same literals through StackSearch.evaluate_single_trajectory(trj, use_kernel=True) (needs a device, as in the
    reference) and through a one-pixel device search with the in-search sigma-G filter."""
    T = 8
    # identity PSF, var = 2 -> phi = 0.5 exactly; sci = 2 psi -> psi = sci / 2 exactly
    sci = [np.full((4, 4), 2.0 * np.float32(1 + 0.1 * i), np.float32) for i in range(T)]
    sci[3][:] = 200.0
    var = [np.full((4, 4), 2.0, np.float32) for _ in range(T)]
    psf = [np.ones((1, 1), np.float32)] * T
    s = kb.StackSearch(sci, var, psf, [float(i) for i in range(T)])
    s.enable_gpu_sigmag_filter([0.25, 0.75], COEFF, 0.0)
    t = kb.Trajectory(x=1, y=1, vx=0.0, vy=0.0)
    s.evaluate_single_trajectory(t, True)
    assert t.obs_count == LITERAL_OBS
    assert t.lh == pytest.approx(LITERAL_LH, abs=2e-6) and t.flux == pytest.approx(LITERAL_FLUX, abs=2e-6)
    s.set_start_bounds_x(1, 2)
    s.set_start_bounds_y(1, 2)
    s.set_results_per_pixel(1)
    s.search_all([kb.Trajectory(vx=0.0, vy=0.0)], True)
    (r,) = s.get_results(0, 1)
    assert (r.x, r.y, r.obs_count) == (1, 1, LITERAL_OBS)
    assert r.lh == pytest.approx(LITERAL_LH, abs=2e-6) and r.flux == pytest.approx(LITERAL_FLUX, abs=2e-6)


# --------------------------------------------------------------------------------------------------------------
# TrajectoryClusterGrid: the known answers of the reference's tests/test_clustering_grid.py:8-88
# --------------------------------------------------------------------------------------------------------------
def _trj(kb, x, y, vx, vy, flux, lh, obs):
    return kb.Trajectory(x, y, vx, vy, flux, lh, obs)


def test_trajectory_cluster_grid_online(kb):
    from kbmod_amd.clustering_grid import TrajectoryClusterGrid

    g = TrajectoryClusterGrid(10, 1.0)
    assert len(g) == 0 and g.total_count == 0
    g.add_trajectory(_trj(kb, 0, 0, 0.0, 0.0, 1.0, 10.0, 10))
    assert len(g) == 1 and g.total_count == 1 and g.count[(0, 0, 0, 0)] == 1 and g.get_indices() == [0]
    g.add_trajectory(_trj(kb, 21, 21, 10.0, 10.0, 1.0, 10.0, 10))
    g.add_trajectory(_trj(kb, 21, 21, 0.0, 0.0, 1.0, 10.0, 10))
    g.add_trajectory(_trj(kb, 21, 21, 0.0, 0.0, 1.0, 100.0, 9))
    assert len(g) == 3 and g.total_count == 4
    assert set(g.table) == {(0, 0, 0, 0), (2, 2, 3, 3), (2, 2, 2, 2)}
    assert (g.count[(0, 0, 0, 0)], g.count[(2, 2, 3, 3)], g.count[(2, 2, 2, 2)]) == (1, 1, 2)
    assert g.table[(2, 2, 2, 2)].obs_count == 9 and set(g.get_indices()) == {0, 1, 3} and len(g.get_trajectories()) == 3
    g.add_trajectory(_trj(kb, 0, 0, 0.0, 0.0, 1.0, 5.0, 5))  # worse: counted, not kept
    assert len(g) == 3 and g.count[(0, 0, 0, 0)] == 2 and g.table[(0, 0, 0, 0)].obs_count == 10
    g.add_trajectory(_trj(kb, 0, 0, 0.0, 0.0, 1.0, 15.0, 15), idx=10)  # better, explicit index
    assert g.count[(0, 0, 0, 0)] == 3 and g.table[(0, 0, 0, 0)].obs_count == 15 and set(g.get_indices()) == {10, 1, 3}
    for bad in ({"bin_width": 0}, {"bin_width": float("nan")}, {"max_time": -1.0}):
        with pytest.raises(ValueError):
            TrajectoryClusterGrid(**bad)


def test_trajectory_cluster_grid_list_and_oracle(kb):
    from kbmod_amd.clustering_grid import TrajectoryClusterGrid

    rows = [(0, 0, 0.0, 0.0, 1.0, 10.0, 10), (21, 21, 10.0, 10.0, 1.0, 10.0, 10), (21, 21, 0.0, 0.0, 1.0, 10.0, 10),
            (21, 21, 0.0, 0.0, 1.0, 100.0, 9), (0, 0, 0.0, 0.0, 1.0, 5.0, 5)]
    g = TrajectoryClusterGrid(10, 1.0)
    g.add_trajectory_list([_trj(kb, *r) for r in rows])
    assert len(g) == 3 and g.total_count == 5
    assert (g.count[(0, 0, 0, 0)], g.count[(2, 2, 3, 3)], g.count[(2, 2, 2, 2)]) == (2, 1, 2)
    assert g.table[(0, 0, 0, 0)].obs_count == 10 and g.table[(2, 2, 2, 2)].obs_count == 9
    assert set(g.get_indices()) == {0, 1, 3}
    # random lists: the online grid agrees with the oracle's restatement of the filter (same keys, same winners)
    rng = np.random.default_rng(5)
    n = 400
    x, y = rng.integers(-30, 200, n), rng.integers(-30, 200, n)
    vx, vy = rng.normal(0, 30, n).astype(np.float32), rng.normal(0, 30, n).astype(np.float32)
    lh = rng.integers(0, 6, n).astype(np.float32)  # many equal likelihoods: the earliest of equals stays
    g = TrajectoryClusterGrid(7, 0.8)
    for i in range(n):
        g.add_trajectory(kb.Trajectory(int(x[i]), int(y[i]), float(vx[i]), float(vy[i]), 1.0, float(lh[i]), 3), idx=i)
    assert g.get_indices() == [int(i) for i in ps.grid_filter_indices(x, y, vx, vy, lh, bin_width=7, max_time=0.8)]


# --------------------------------------------------------------------------------------------------------------
# single-curve sigma-G clip: tests/test_sigma_g_filter.py:24-45 and the oracle on random curves
# --------------------------------------------------------------------------------------------------------------
def test_compute_clipped_sigma_g_single_curve():
    from kbmod_amd.sigma_g_filter import SigmaGClipping, invert_gauss_cdf, sigma_g_coefficient

    p = SigmaGClipping()
    lh = np.array([(10.0 + i * 0.05) for i in range(20)])
    assert set(p.compute_clipped_sigma_g(lh)) == set(range(20))
    lh[2], lh[14] = 100.0, -100.0
    assert set(p.compute_clipped_sigma_g(lh)) == set(range(20)) - {2, 14}
    lh[0] = 50.0
    assert set(p.compute_clipped_sigma_g(lh)) == set(range(20)) - {0, 2, 14}
    rng = np.random.default_rng(11)
    for clip_negative in (False, True):
        q = SigmaGClipping(20, 80, 3, clip_negative)
        for _ in range(50):
            curve = rng.normal(3.0, 4.0, int(rng.integers(1, 40)))
            got = q.compute_clipped_sigma_g(curve)
            exp = ps.clipped_sigma_g(curve, 20, 80, 3, clip_negative)
            assert np.array_equal(np.asarray(got, dtype=np.int64), np.asarray(exp, dtype=np.int64))
    assert SigmaGClipping(clip_negative=True).compute_clipped_sigma_g(np.array([-1.0, -2.0])).size == 0
    # the quantile helper and the coefficient: the reference's erfinv formulation, endpoints included
    assert SigmaGClipping.invert_gauss_cdf(0.5) == 0.0 and invert_gauss_cdf(0.0) == -np.inf and invert_gauss_cdf(1.0) == np.inf
    assert invert_gauss_cdf(0.975) == pytest.approx(1.959964, abs=1e-6) and invert_gauss_cdf(0.975) == ps.invert_gauss_cdf(0.975)
    assert sigma_g_coefficient(0, 100) == 0.0 and sigma_g_coefficient(0, 50) == 0.0
    assert sigma_g_coefficient(25, 75) == ps.find_sigma_g_coeff(25, 75)  # bit-equal, not approximately
