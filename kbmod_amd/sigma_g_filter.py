"""Batched sigma-G clipping of likelihood curves on the device -- SURVEY.md section 8(f1).

The product side of ``kbmod.filters.sigma_g_filter.SigmaGClipping`` (src/kbmod/filters/sigma_g_filter.py:
19-168): the object carries the same parameters, and ``compute_clipped_sigma_g_matrix`` -- what
``SearchRunner.load_and_filter_results`` calls on up to S*K likelihood curves (run_search.py:251-337) -- runs
``kb_sigma_g_clip_matrix`` through the C ABI.  Like the rest of the product it raises ``RuntimeError`` without
a GPU.  The single-curve form (``compute_clipped_sigma_g``) and the quantile helper (``invert_gauss_cdf``) are a
few lines of host numpy, kept because callers of the reference use them.
"""

import math

import numpy as np
from scipy.special import erfinv

from . import search as _search


def invert_gauss_cdf(z):
    """Standard normal quantile of probability ``z`` through the inverse error function, the reference's formulation
    (sigma_g_filter.py:85-92): -inf / +inf at 0 / 1, so that percentile bounds on the edge give a coefficient of 0."""
    side = -1.0 if z < 0.5 else 1.0
    return float(side * math.sqrt(2.0) * erfinv(side * (2.0 * z - 1.0)))


def sigma_g_coefficient(low_pct, high_pct):
    """1 / (z(high) - z(low)) with z the standard normal quantile function: the factor that turns an
    inter-percentile range into a standard deviation (0.7413 for [25, 75]; sigma_g_filter.py:49-83)."""
    if not (0 <= low_pct < high_pct <= 100):
        raise ValueError(f"Invalid percentiles for sigma G coefficient [{low_pct}, {high_pct}]")
    return 1.0 / (invert_gauss_cdf(high_pct / 100.0) - invert_gauss_cdf(low_pct / 100.0))


class SigmaGClipping:
    """Parameters of the clip (percentile bounds on the reference's [0, 100] scale, width in sigma-G, whether
    non-positive likelihoods are left out of the percentiles) and the batched device call."""

    def __init__(self, low_bnd=25, high_bnd=75, n_sigma=2, clip_negative=False):
        if not (0.0 < low_bnd <= high_bnd < 100.0):
            raise ValueError(f"Invalid bounds [{low_bnd}, {high_bnd}]")
        if not n_sigma > 0.0:
            raise ValueError(f"Invalid n_sigma {n_sigma}")
        self.low_bnd, self.high_bnd = low_bnd, high_bnd
        self.n_sigma = n_sigma
        self.clip_negative = clip_negative
        self.coeff = sigma_g_coefficient(low_bnd, high_bnd)

    find_sigma_g_coeff = staticmethod(sigma_g_coefficient)
    invert_gauss_cdf = staticmethod(invert_gauss_cdf)

    def compute_clipped_sigma_g(self, lh):
        """Indices of ONE likelihood curve that lie strictly within ``n_sigma`` sigma-G of its median
        (sigma_g_filter.py:94-123): percentiles by numpy's linear interpolation over all points -- over the positive
        ones with ``clip_negative``, and nothing survives a curve without any --, the inter-percentile range floored
        at 1e-8.  Host numpy: one curve is not device work; batches go through ``compute_clipped_sigma_g_matrix``."""
        curve = np.asarray(lh)
        basis = curve[curve > 0] if self.clip_negative else curve
        if basis.size == 0:
            return np.array([])
        low, middle, high = np.percentile(basis, [self.low_bnd, 50, self.high_bnd])
        reach = self.n_sigma * self.coeff * max(high - low, 1e-8)
        return np.flatnonzero((curve > middle - reach) & (curve < middle + reach))

    def compute_clipped_sigma_g_matrix(self, lh):
        """N x T curves -> N x T bool matrix (True = kept), computed on the device."""
        curves = np.ascontiguousarray(np.asarray(lh), dtype=np.float32)
        if curves.ndim != 2:
            raise ValueError("expected an N x T matrix of likelihood curves")
        if curves.size == 0:
            return np.zeros(curves.shape, dtype=bool)
        return _search.sigma_g_clip_matrix(curves, float(self.low_bnd), float(self.high_bnd), float(self.n_sigma),
                                           float(self.coeff), bool(self.clip_negative))


def compute_likelihood_curves(psi_curves, phi_curves, obs_valid=None, mask_value=0.0):
    """psi / sqrt(phi) per epoch where both are finite, phi is non-zero and the epoch is marked valid;
    ``mask_value`` elsewhere (``Results.compute_likelihood_curves``, src/kbmod/results.py:568-606).  Feeds
    ``compute_clipped_sigma_g_matrix`` from the output of ``StackSearch.get_all_psi_phi_curves``."""
    psi, phi = np.asarray(psi_curves), np.asarray(phi_curves)
    usable = np.isfinite(psi) & np.isfinite(phi) & (phi != 0)
    if obs_valid is not None:
        usable &= np.asarray(obs_valid)
    out = np.full(psi.shape, mask_value, dtype=np.float32)
    np.divide(psi, np.sqrt(phi, where=usable, out=np.ones(phi.shape, dtype=phi.dtype)), out=out, where=usable)
    return out
