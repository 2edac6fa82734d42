"""The reference's end-to-end regression test as a property test of the device path.

/root/reference/tests/test_regression_test.py:96-229 -- the one end-to-end property the reference itself asserts for this
path: a 20 x 1024 x 512 stack (noise 4, per-epoch Gaussian PSFs 0.95 / 1.05 / 1.15, four exposures a night), 20 injected
movers of flux 500 (two of them starting off the chip), an ecliptic-centred grid of 26 angles x 52 speeds, start pixels
10 beyond every edge, at least 15 observations, likelihood 25, the in-search sigma-G filter (`gpu_filter`), eight results
per pixel; every injected trajectory must come back within 3 pixels (averaged over t = 0 and t = 2 days).

The control plane of that test (SearchConfiguration, WorkUnit, SearchRunner, Results tables, DBSCAN) is out of scope; what
it drives on the hot path is called here directly, in its order (run_search.py:25-72, 251-337, 339-390): StackSearch ->
search_all on the device -> the near-duplicate grid filter (kb_grid_filter) -> psi/phi curves -> batched sigma-G
(kb_sigma_g_clip_matrix) -> the re-test of observation count and likelihood.  Independent of the oracle: the assertion is
recovery of what was injected.
"""

import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

from kbmod_amd import fake_data as fd

pytestmark = pytest.mark.gpu

FLUX = 500.0
INJECTED = [  # (x, y, vx, vy): test_regression_test.py:147-168
    (357, 997, -15.814404, -172.098450), (477, 777, -70.858154, -117.137817), (408, 533, -53.721024, -106.118118),
    (425, 740, -32.865086, -132.898575), (515, 881, -73.831688, -93.251732), (412, 980, -79.985207, -192.813080),
    (443, 923, -36.977375, -103.556976), (368, 1015, -43.644382, -176.487488), (510, 1011, -125.422997, -166.863983),
    (398, 939, -51.037308, -107.434616), (491, 925, -74.266739, -104.155556), (366, 824, -18.041782, -153.808197),
    (477, 870, -45.608849, -90.093689), (447, 993, -38.152031, -196.087646), (481, 882, -96.767357, -143.192352),
    (423, 912, -104.900154, -125.859169), (409, 803, -99.066856, -173.469589), (328, 797, -33.212299, -196.984467),
    (466, 1026, -67.892105, -118.881493),  # off chip in y
    (514, 795, -20.134245, -171.646683),   # off chip in x
]


def regression_stack():
    """make_fake_images + the time / PSF schedule of run_full_test (test_regression_test.py:26-55, 172-190)."""
    times, psf_vals = [], []
    seen_on_day = day_num = 0
    for i in range(20):
        times.append(57130.2 + day_num + seen_on_day * 0.01)
        seen_on_day += 1
        if seen_on_day == 4:
            seen_on_day = 0
            day_num += 1
        psf_vals.append(1.05 - 0.1 + 0.1 * (i % 3))
    psfs = [fd.make_gaussian_kernel(v) for v in psf_vals]
    stack = fd.make_fake_image_stack(1024, 512, times, noise_level=4.0, psfs=psfs, rng=np.random.default_rng(1001))
    for x, y, vx, vy in INJECTED:
        fd.add_fake_object(stack, x, y, vx, vy, flux=FLUX)
    return stack


def match_sets(truth, found, threshold, times):
    """trajectory_utils.py:361-440: optimal one-to-one matching on the mean distance of the predicted positions."""
    times = np.asarray(times, dtype=np.float64)
    fx = found[:, 0:1] + times[None, :] * found[:, 2:3]
    fy = found[:, 1:2] + times[None, :] * found[:, 3:4]
    dists = np.zeros((len(truth), len(found)))
    for q, (x, y, vx, vy) in enumerate(truth):
        dists[q] = np.mean(np.hypot((x + times * vx)[None, :] - fx, (y + times * vy)[None, :] - fy), axis=1)
    rows, cols = linear_sum_assignment(dists)
    out = np.full(len(truth), -1)
    for r, c in zip(rows, cols):
        if dists[r, c] < threshold:
            out[r] = c
    return out


def test_reference_regression_shape_recovers_every_injected_trajectory(kb):
    from kbmod_amd.clustering_grid import apply_trajectory_grid_filter
    from kbmod_amd.sigma_g_filter import SigmaGClipping, compute_likelihood_curves

    stack = regression_stack()
    num_obs, lh_level, lims = 15, 25.0, [25, 75]
    search = kb.StackSearch(stack.sci, stack.var, stack.psfs, stack.zeroed_times, -1)
    assert search.get_psi_phi_array().device_resident
    # configure_kb_search_stack (run_search.py:25-72)
    search.set_min_obs(num_obs)
    search.set_min_lh(lh_level)
    search.set_start_bounds_x(-10, search.get_image_width() + 10)
    search.set_start_bounds_y(-10, search.get_image_height() + 10)
    search.set_results_per_pixel(8)
    search.enable_gpu_sigmag_filter(np.array(lims) / 100.0, SigmaGClipping.find_sigma_g_coeff(*lims), lh_level)

    vx, vy = fd.ecliptic_centered_candidates([92.0, 550.0, 52], [np.pi - np.pi / 10.0, np.pi + np.pi / 10.0, 26],
                                             given_ecliptic=1.1901106654050821)
    assert len(vx) == 26 * 52
    search.search_all([kb.Trajectory(vx=float(a), vy=float(b)) for a, b in zip(vx, vy)], True)
    stats = search.last_search_stats()
    # a device search kernel ran (no CPU path behind on_gpu=True); with shifts of up to 2 000 pixels it is the direct one
    assert "kb_search" in stats["kernel_name"], stats

    results = search.get_all_results()
    assert len(results) > 0
    assert all(results[i].lh >= results[i + 1].lh for i in range(min(len(results), 2000) - 1))  # sorted by likelihood
    assert results[-1].lh >= lh_level and min(r.obs_count for r in results) >= num_obs

    # load_and_filter_results (run_search.py:251-337): near-duplicate grid filter, curves, clipped sigma-G, re-test
    max_dt = float(np.max(stack.zeroed_times) - np.min(stack.zeroed_times))
    kept, _ = apply_trajectory_grid_filter(results, 10, max_dt)
    assert 0 < len(kept) <= len(results)
    T = len(stack.zeroed_times)
    curves = search.get_all_psi_phi_curves(kept)
    psi, phi = curves[:, :T], curves[:, T:]
    lh_curves = compute_likelihood_curves(psi, phi)
    valid = SigmaGClipping(lims[0], lims[1], 2, True).compute_clipped_sigma_g_matrix(lh_curves)
    usable = valid & np.isfinite(psi) & np.isfinite(phi) & (phi != 0)
    psi_sum = np.where(usable, psi, 0.0).sum(axis=1)
    phi_sum = np.where(usable, phi, 0.0).sum(axis=1)
    new_lh = np.where(phi_sum > 0, psi_sum / np.sqrt(np.where(phi_sum > 0, phi_sum, 1.0)), 0.0)
    rows = (usable.sum(axis=1) >= num_obs) & (new_lh >= lh_level)
    found = np.array([[t.x, t.y, t.vx, t.vy] for t, ok in zip(kept, rows) if ok], dtype=np.float64)
    assert len(found) >= len(INJECTED)

    matches = match_sets(INJECTED, found, 3.0, [0.0, 2.0])
    missing = [INJECTED[i] for i in np.flatnonzero(matches == -1)]
    assert not missing, f"not recovered: {missing} ({len(found)} trajectories survived the filters)"
    # the recovered fluxes are the injected ones (the reference's own recovery tests allow 15-25 %)
    by_key = {(t.x, t.y, round(t.vx, 3), round(t.vy, 3)): t for t in kept}
    for i, c in enumerate(matches):
        t = by_key[(int(found[c, 0]), int(found[c, 1]), round(found[c, 2], 3), round(found[c, 3], 3))]
        assert t.flux / FLUX == pytest.approx(1.0, abs=0.35), (INJECTED[i], t.flux)
