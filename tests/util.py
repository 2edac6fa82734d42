"""Shared helpers of the parity tests (seeded inputs, oracle comparison)."""

import numpy as np

from kbmod_amd import fake_data as fd

FIELDS = ("x", "y", "vx", "vy", "lh", "flux", "obs_count")


def make_stack(T, H, W, seed, noise=2.0, psf=1.0, objects=(), mask_fraction=0.0, times=None):
    rng = np.random.default_rng(seed)
    if times is None:
        times = np.arange(T) / float(T)
    stack = fd.make_fake_image_stack(H, W, times, noise_level=noise, psf_val=psf, rng=rng)
    if mask_fraction > 0:
        fd.add_random_masks(stack, mask_fraction, rng)
    for (x, y, vx, vy, flux) in objects:
        fd.add_fake_object(stack, x, y, vx, vy, flux=flux)
    return stack


def as_table(res):
    """Oracle structured array -> (N, 7) float64 table in results_to_numpy() column order."""
    return np.stack([res[k] for k in FIELDS], 1).astype(np.float64)


def trajectories(kb, vx, vy):
    return [kb.Trajectory(vx=float(a), vy=float(b)) for a, b in zip(vx, vy)]


def oracle_params(pp, search_cfg):
    """Params for the oracle from a dict of StackSearch-style settings."""
    kw = {}
    if "min_obs" in search_cfg:
        kw["min_observations"] = search_cfg["min_obs"]
    if "min_lh" in search_cfg:
        kw["min_lh"] = search_cfg["min_lh"]
    if "K" in search_cfg:
        kw["results_per_pixel"] = search_cfg["K"]
    if "xb" in search_cfg:
        kw["x_start_min"], kw["x_start_max"] = search_cfg["xb"]
    if "yb" in search_cfg:
        kw["y_start_min"], kw["y_start_max"] = search_cfg["yb"]
    if "sigmag" in search_cfg:
        lo, hi, coeff, min_lh = search_cfg["sigmag"]
        kw.update(do_sigmag_filter=1, sgl_L=lo, sgl_H=hi, sigmag_coeff=coeff, min_lh=min_lh)
    return pp.default_params(**kw)


def configure(search, cfg):
    if "min_obs" in cfg:
        search.set_min_obs(cfg["min_obs"])
    if "min_lh" in cfg:
        search.set_min_lh(cfg["min_lh"])
    if "K" in cfg:
        search.set_results_per_pixel(cfg["K"])
    if "xb" in cfg:
        search.set_start_bounds_x(*cfg["xb"])
    if "yb" in cfg:
        search.set_start_bounds_y(*cfg["yb"])
    if "sigmag" in cfg:
        lo, hi, coeff, min_lh = cfg["sigmag"]
        search.enable_gpu_sigmag_filter([lo, hi], coeff, min_lh)


def run_both(kb, orc, stack, vx, vy, cfg, num_bytes=-1, on_gpu=True, flags=0):
    """(product table, oracle table) for one configuration."""
    search = kb.StackSearch(stack.sci, stack.var, stack.psfs, stack.zeroed_times, num_bytes)
    configure(search, cfg)
    search.set_search_flags(flags)
    search.search_all(trajectories(kb, vx, vy), on_gpu)
    got = search.results_to_numpy()

    pp = orc.PsiPhi.from_images(stack.sci, stack.var, stack.psfs, stack.zeroed_times, num_bytes)
    params = oracle_params(pp, cfg)
    cands = orc.make_candidates(vx, vy)
    raw = pp.search_kernel_semantics(cands, params) if on_gpu else pp.search_cpu(cands, params)
    exp = as_table(orc.filter_sort(raw, params.min_lh, params.min_observations))
    return got, exp, search
