"""ORACLE (test infrastructure only -- never imported by the product).

numpy restatements of the post-search Python stages of SURVEY.md section 8(f):

* batched sigma-G clipping: ``SigmaGClipping`` (src/kbmod/filters/sigma_g_filter.py:19-168)
* likelihood curves from psi/phi curves (src/kbmod/results.py:568-606)

The matrix form of the reference delegates to a third-party routine that is not part of
/root/reference: ``torch.nanquantile`` (PyTorch 2.10, ATen/native/Sorting.cpp ``quantile_compute``,
linear interpolation).  Its published algorithm is restated here in float32:
sort ascending with NaN last; ``rank = q * (n_valid - 1)`` (float32; 0 when the row is all NaN);
``below = floor(rank)``, ``above = ceil(rank)``, ``w = rank - below``; result =
``lerp(v[below], v[above], w)`` with ATen's scalar definition (native/Lerp.h)
``w < 0.5 ? a + w * (b - a) : b - (b - a) * (1 - w)``.  ATen's vectorised CPU kernel and its GPU
kernel contract these into fused multiply-adds, so torch itself is only defined to about one ulp
here; parity is therefore pinned on the reference's own known answers
(tests/test_sigma_g_filter.py:24-120, 163-199) and, within two ulps of the bounds, on vectors
generated with torch in this container (tests/golden/make_golden_sigma_g_matrix.py).
"""

import math

import numpy as np


def invert_gauss_cdf(z):
    """sigma_g_filter.py:77-83 (scipy.special.erfinv, double precision)."""
    from scipy.special import erfinv

    sign = -1 if z < 0.5 else 1
    return float(sign * np.sqrt(2) * erfinv(sign * (2 * z - 1)))


def find_sigma_g_coeff(low_bnd, high_bnd):
    """sigma_g_filter.py:49-75."""
    if (high_bnd <= low_bnd) or (low_bnd < 0) or (high_bnd > 100):
        raise ValueError(f"Invalid percentiles for sigma G coefficient [{low_bnd}, {high_bnd}]")
    return 1 / (invert_gauss_cdf(high_bnd / 100.0) - invert_gauss_cdf(low_bnd / 100.0))


def clipped_sigma_g(lh, low_bnd=25, high_bnd=75, n_sigma=2, clip_negative=False):
    """Single curve, sigma_g_filter.py:85-112 (numpy percentile in double)."""
    lh = np.asarray(lh)
    coeff = find_sigma_g_coeff(low_bnd, high_bnd)
    if clip_negative:
        if np.count_nonzero(lh > 0) == 0:
            return np.array([])
        lower_per, median, upper_per = np.percentile(lh[lh > 0], [low_bnd, 50, high_bnd])
    else:
        lower_per, median, upper_per = np.percentile(lh, [low_bnd, 50, high_bnd])
    delta = max(upper_per - lower_per, 1e-8)
    n_sigma_g = n_sigma * coeff * delta
    return np.where(np.logical_and(lh > median - n_sigma_g, lh < median + n_sigma_g))[0]


def _lerp32(a, b, w):
    a, b, w = np.float32(a), np.float32(b), np.float32(w)
    if w < np.float32(0.5):
        return np.float32(a + np.float32(w * np.float32(b - a)))
    return np.float32(b - np.float32(np.float32(b - a) * np.float32(np.float32(1.0) - w)))


def nanquantile_rows_f32(x, qs):
    """torch.nanquantile(x, qs, dim=1) for a float32 matrix; returns [len(qs)][rows] float32."""
    x = np.asarray(x, dtype=np.float32)
    out = np.empty((len(qs), x.shape[0]), dtype=np.float32)
    for r in range(x.shape[0]):
        row = np.sort(x[r])  # numpy also sorts NaN last
        n_valid = int(np.count_nonzero(~np.isnan(row)))
        for i, q in enumerate(qs):
            rank = np.float32(np.float32(q) * np.float32(n_valid - 1))
            if rank < 0:
                rank = np.float32(0.0)
            below = int(math.floor(float(rank)))
            above = int(math.ceil(float(rank)))
            w = np.float32(rank - np.float32(below))
            out[i, r] = _lerp32(row[below], row[above], w)
    return out


def clipped_sigma_g_matrix(lh, low_bnd=25, high_bnd=75, n_sigma=2, clip_negative=False, coeff=None):
    """N x T matrix form, sigma_g_filter.py:114-168.  Returns the N x T bool matrix and the
    (lower bound, upper bound) rows so that tests can tell a real mismatch from a one-ulp tie."""
    if coeff is None:
        coeff = find_sigma_g_coeff(low_bnd, high_bnd)
    with np.errstate(invalid="ignore", over="ignore"):
        t = np.asarray(lh).astype(np.float32)
        masked = np.where(t > np.float32(0.0), t, np.float32(np.nan)) if clip_negative else t
        lower_per, median, upper_per = nanquantile_rows_f32(masked, [low_bnd / 100.0, 0.5, high_bnd / 100.0])
        delta = (upper_per - lower_per).astype(np.float32)
        delta[delta < np.float32(1e-5)] = np.float32(1e-5)
        n_sigma_g = (np.float32(n_sigma * coeff) * delta).astype(np.float32)
        lower = (median - n_sigma_g).astype(np.float32)
        upper = (median + n_sigma_g).astype(np.float32)
        valid = np.isfinite(t) & (t < upper[:, None]) & (t > lower[:, None])
    return valid, lower, upper


def likelihood_curves(psi, phi, obs_valid=None, mask_value=0.0):
    """results.py:596-606."""
    psi = np.asarray(psi)
    phi = np.asarray(phi)
    valid = (phi != 0) & np.isfinite(psi) & np.isfinite(phi)
    if obs_valid is not None:
        valid = valid & np.asarray(obs_valid)
    out = np.full(psi.shape, mask_value, dtype=np.float32)
    out[valid] = psi[valid] / np.sqrt(phi[valid])
    return out
