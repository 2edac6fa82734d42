"""Host-side mirror of ``kbmod.filters.sigma_g_filter.SigmaGClipping``
(src/kbmod/filters/sigma_g_filter.py:19-168) for the post-search step of SURVEY.md section 8(f1).

Same constructor, attributes, methods and errors.  ``compute_clipped_sigma_g_matrix`` -- the
batched form ``SearchRunner.load_and_filter_results`` calls on up to S*K likelihood curves
(run_search.py:251-337) -- runs ``kb_sigma_g_clip_matrix`` on the device through the C ABI and, like
the rest of the product, raises ``RuntimeError`` without a GPU; the single-curve helper is the
reference's few lines of numpy.
"""

import numpy as np

from . import search as _search


class SigmaGClipping:
    def __init__(self, low_bnd=25, high_bnd=75, n_sigma=2, clip_negative=False):
        if low_bnd > high_bnd or low_bnd <= 0.0 or high_bnd >= 100.0:
            raise ValueError(f"Invalid bounds [{low_bnd}, {high_bnd}]")
        if n_sigma <= 0.0:
            raise ValueError(f"Invalid n_sigma {n_sigma}")
        self.low_bnd = low_bnd
        self.high_bnd = high_bnd
        self.n_sigma = n_sigma
        self.coeff = SigmaGClipping.find_sigma_g_coeff(low_bnd, high_bnd)
        self.clip_negative = clip_negative

    @staticmethod
    def find_sigma_g_coeff(low_bnd, high_bnd):
        if (high_bnd <= low_bnd) or (low_bnd < 0) or (high_bnd > 100):
            raise ValueError(f"Invalid percentiles for sigma G coefficient [{low_bnd}, {high_bnd}]")
        x1 = SigmaGClipping.invert_gauss_cdf(low_bnd / 100.0)
        x2 = SigmaGClipping.invert_gauss_cdf(high_bnd / 100.0)
        return 1 / (x2 - x1)

    @staticmethod
    def invert_gauss_cdf(z):
        from scipy.special import erfinv

        sign = -1 if z < 0.5 else 1
        return float(sign * np.sqrt(2) * erfinv(sign * (2 * z - 1)))

    def compute_clipped_sigma_g(self, lh):
        """Indices of one likelihood curve within n_sigma * sigmaG of its median."""
        lh = np.asarray(lh)
        if self.clip_negative:
            if np.count_nonzero(lh > 0) == 0:
                return np.array([])
            lower_per, median, upper_per = np.percentile(lh[lh > 0], [self.low_bnd, 50, self.high_bnd])
        else:
            lower_per, median, upper_per = np.percentile(lh, [self.low_bnd, 50, self.high_bnd])
        delta = max(upper_per - lower_per, 1e-8)
        n_sigma_g = self.n_sigma * self.coeff * delta
        return np.where(np.logical_and(lh > median - n_sigma_g, lh < median + n_sigma_g))[0]

    def compute_clipped_sigma_g_matrix(self, lh):
        """N x T curves -> N x T bool matrix (True = kept), computed on the device."""
        lh = np.ascontiguousarray(np.asarray(lh), dtype=np.float32)
        if lh.ndim != 2:
            raise ValueError("expected an N x T matrix of likelihood curves")
        if lh.size == 0:
            return np.zeros(lh.shape, dtype=bool)
        return _search.sigma_g_clip_matrix(lh, float(self.low_bnd), float(self.high_bnd), float(self.n_sigma),
                                           float(self.coeff), bool(self.clip_negative))


def compute_likelihood_curves(psi_curves, phi_curves, obs_valid=None, mask_value=0.0):
    """``Results.compute_likelihood_curves`` (src/kbmod/results.py:568-606) on plain arrays."""
    psi = np.asarray(psi_curves)
    phi = np.asarray(phi_curves)
    valid = (phi != 0) & np.isfinite(psi) & np.isfinite(phi)
    if obs_valid is not None:
        valid = valid & np.asarray(obs_valid)
    lh_matrix = np.full(psi.shape, mask_value, dtype=np.float32)
    lh_matrix[valid] = psi[valid] / np.sqrt(phi[valid])
    return lh_matrix
