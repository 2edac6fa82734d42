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


# ---------------------------------------------------------------------------
# stamps and coadds: src/kbmod/core/stamp_utils.py
# ---------------------------------------------------------------------------
# Pinned by the reference's known answers (tests/test_stamp_utils.py:20-273).  coadd_median delegates
# to the third-party torch.nanmedian (PyTorch 2.10): per pixel the LOWER median of the non-NaN values
# (index (n - 1) // 2 of the ascending order); checked against torch itself in
# tests/test_oracle_stamps.py.


def predict_pixel_locations(times, x0, vx, centered=True, as_int=True):
    """src/kbmod/trajectory_utils.py:28-75 (truncation toward zero, not floor)."""
    times, x0, vx = np.asarray(times), np.asarray(x0), np.asarray(vx)
    if len(x0) != len(vx):
        raise ValueError(f"x0 and vx must be same size. Found {len(x0)} vs {len(vx)}")
    pos = vx[:, np.newaxis] * times[np.newaxis, :] + x0[:, np.newaxis]
    if centered:
        pos = pos + 0.5
    if as_int:
        pos = pos.astype(int)
    return pos


def extract_stamp(img, x_val, y_val, radius):
    """stamp_utils.py:352-397: float64, NaN where there is no data."""
    h, w = img.shape
    s = 2 * radius + 1
    stamp = np.full((s, s), np.nan)
    for j in range(s):
        yy = y_val - radius + j
        if yy < 0 or yy >= h:
            continue
        for i in range(s):
            xx = x_val - radius + i
            if 0 <= xx < w:
                stamp[j, i] = img[yy, xx]
    return stamp


def extract_stamp_stack(imgs, x_vals, y_vals, radius, to_include=None):
    """stamp_utils.py:16-84 for an array of images (mask or index list in to_include)."""
    num_times = len(imgs)
    if radius < 1:
        raise ValueError("Radius must be at least 1.")
    if len(x_vals) != num_times or len(y_vals) != num_times:
        raise ValueError("X and Y values must have the same length as the number of times.")
    mask = np.full(num_times, True)
    if to_include is not None:
        to_include = np.asarray(to_include)
        if to_include.dtype == bool:
            if len(to_include) != num_times:
                raise ValueError("Time mask must have the same length as the number of times.")
            mask = to_include
        else:
            mask = np.full(num_times, False)
            mask[to_include] = True
    s = 2 * radius + 1
    if num_times == 0 or np.count_nonzero(mask) == 0:
        return np.empty((0, s, s))
    x_vals = np.asarray(x_vals, dtype=int)
    y_vals = np.asarray(y_vals, dtype=int)
    return np.array([extract_stamp(np.asarray(imgs[t]), x_vals[t], y_vals[t], radius) for t in range(num_times) if mask[t]])


def extract_curve_values(imgs, x_vals, y_vals):
    """stamp_utils.py:87-141, 478-512."""
    num_times = len(imgs)
    x_vals = np.asanyarray(x_vals, dtype=int)
    y_vals = np.asanyarray(y_vals, dtype=int)
    single = x_vals.ndim == 1
    if single:
        x_vals, y_vals = x_vals[np.newaxis, :], y_vals[np.newaxis, :]
    if x_vals.shape[1] != num_times or y_vals.shape[1] != num_times:
        raise ValueError(f"X and Y values must have the same length as times ({num_times}).")
    h, w = np.asarray(imgs[0]).shape
    values = np.full(x_vals.shape, np.nan)
    for r in range(x_vals.shape[0]):
        for t in range(num_times):
            x_i, y_i = x_vals[r, t], y_vals[r, t]
            if 0 <= x_i < w and 0 <= y_i < h:
                values[r, t] = imgs[t][y_i, x_i]
    return values.flatten() if single else values


def _sequential_sum(a):
    """np.add.reduce over axis 0 of a C-contiguous (T, H, W) array adds the slices one after the other
    starting from slice 0 (no pairwise blocking across the reduced outer axis); restated explicitly."""
    out = np.array(a[0], dtype=np.float64, copy=True)
    for t in range(1, a.shape[0]):
        out = out + a[t]
    return out


def _mask_all_nans(stack):
    """stamp_utils.py:214-238."""
    stack = np.asarray(stack)
    none_valid = np.all(np.isnan(stack), axis=0)
    if np.any(none_valid):
        stack = stack.copy()
        stack[:, none_valid] = 0.0
    return stack


def coadd_sum(stack):
    """stamp_utils.py:241-255 (np.nansum)."""
    stack = np.asarray(stack, dtype=np.float64)
    if stack.shape[0] == 0:
        return np.zeros(stack.shape[1:])
    return _sequential_sum(np.where(np.isnan(stack), 0.0, stack))


def coadd_mean(stack):
    """stamp_utils.py:258-275 (np.nanmean after masking all-NaN pixels to 0)."""
    stack = np.asarray(stack, dtype=np.float64)
    if stack.shape[0] == 0:
        return np.zeros(stack.shape[1:])
    stack = _mask_all_nans(stack)
    cnt = np.sum(~np.isnan(stack), axis=0)
    return _sequential_sum(np.where(np.isnan(stack), 0.0, stack)) / cnt


def coadd_median(stack):
    """stamp_utils.py:278-303 (torch.nanmedian: lower median; all-NaN pixel -> 0)."""
    stack = np.asarray(stack, dtype=np.float64)
    if stack.shape[0] == 0:
        return np.zeros(stack.shape[1:])
    srt = np.sort(stack, axis=0)  # NaN last
    n = np.sum(~np.isnan(stack), axis=0)
    idx = np.maximum((n - 1) // 2, 0)
    out = np.take_along_axis(srt, idx[np.newaxis], axis=0)[0]
    out[n == 0] = 0.0
    return out


def coadd_weighted(stack, var_stack):
    """stamp_utils.py:306-344."""
    stack = np.asarray(stack, dtype=np.float64)
    var_stack = np.asarray(var_stack, dtype=np.float64)
    if stack.shape[0] == 0:
        return np.zeros(stack.shape[1:])
    stack = _mask_all_nans(stack)
    pix_valid = ~(np.isnan(stack) | np.isnan(var_stack) | (var_stack == 0.0))
    weights = np.zeros(stack.shape)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        weights[pix_valid] = 1.0 / var_stack[pix_valid]
        weighted_sci = np.zeros(stack.shape)
        weighted_sci[pix_valid] = stack[pix_valid] * weights[pix_valid]
        weighted_sum = _sequential_sum(weighted_sci)
        sum_of_weights = _sequential_sum(weights)
        sum_of_weights[sum_of_weights == 0.0] = 1e24
        return weighted_sum / sum_of_weights


COADDS = {"sum": coadd_sum, "mean": coadd_mean, "median": coadd_median}


def coadds_for_trajectories(sci, var, xvals, yvals, obs_valid, radius, coadd_types):
    """The loop of append_coadds (src/kbmod/filters/stamp_filters.py:72-168, nightly=False) on plain
    arrays: float32 [N][S][S] per coadd type."""
    if radius <= 0:
        raise ValueError(f"Invalid stamp radius {radius}")
    s = 2 * radius + 1
    n = len(xvals)
    out = {c: np.zeros((n, s, s), dtype=np.float32) for c in coadd_types}
    for idx in range(n):
        inc = None if obs_valid is None else obs_valid[idx]
        import warnings

        with warnings.catch_warnings(), np.errstate(invalid="ignore", divide="ignore"):
            warnings.simplefilter("ignore")
            sci_stack = extract_stamp_stack(sci, xvals[idx], yvals[idx], radius, to_include=inc)
            for c in coadd_types:
                if c == "weighted":
                    var_stack = extract_stamp_stack(var, xvals[idx], yvals[idx], radius, to_include=inc)
                    out[c][idx] = coadd_weighted(sci_stack, var_stack)
                else:
                    out[c][idx] = COADDS[c](sci_stack)
    return out


def mjd_to_day(mjd):
    """util_functions.py:52-65 (astropy Time(mjd, format="mjd").strftime("%Y-%m-%d"), UTC): the calendar date
    floor(mjd) days after 1858-11-17.  Known answer: mjd 60000 -> 2023-02-25."""
    import datetime

    return (datetime.date(1858, 11, 17) + datetime.timedelta(days=int(np.floor(mjd)))).isoformat()


def append_coadds_columns(sci, var, times_mjd, xvals, yvals, obs_valid, radius, coadd_types, nightly):
    """The whole loop of append_coadds (stamp_filters.py:72-168) on plain arrays, nightly columns included:
    {column name: float32 [N][S][S]}."""
    import warnings

    s = 2 * radius + 1
    n = len(xvals)
    day_strs = np.array([f"_{mjd_to_day(t)}" for t in times_mjd])
    days_to_use = np.unique(day_strs) if nightly else []
    out = {f"coadd_{c}": np.zeros((n, s, s), dtype=np.float32) for c in coadd_types}
    for day in days_to_use:
        for c in coadd_types:
            out[f"coadd_{c}{day}"] = np.zeros((n, s, s), dtype=np.float32)
    to_include = np.full(len(times_mjd), True)
    for idx in range(n):
        if obs_valid is not None:
            to_include = obs_valid[idx]
        with warnings.catch_warnings(), np.errstate(invalid="ignore", divide="ignore"):
            warnings.simplefilter("ignore")
            sci_stack = np.asanyarray(extract_stamp_stack(sci, xvals[idx], yvals[idx], radius, to_include=to_include))
            var_stack = None
            if "weighted" in coadd_types:
                var_stack = np.asanyarray(extract_stamp_stack(var, xvals[idx], yvals[idx], radius, to_include=to_include))
            for c in coadd_types:
                out[f"coadd_{c}"][idx] = coadd_weighted(sci_stack, var_stack) if c == "weighted" else COADDS[c](sci_stack)
            for day in days_to_use:
                day_mask = day == day_strs[to_include]
                sci_day = sci_stack[day_mask]
                for c in coadd_types:
                    if c == "weighted":
                        out[f"coadd_{c}{day}"][idx] = coadd_weighted(sci_day, var_stack[day_mask])
                    else:
                        out[f"coadd_{c}{day}"][idx] = COADDS[c](sci_day)
    return out


def all_stamps_for_trajectories(sci, xvals, yvals, radius):
    """append_all_stamps (stamp_filters.py:171-211): float32 [N][T][S][S]."""
    n, T = len(xvals), len(sci)
    s = 2 * radius + 1
    out = np.zeros((n, T, s, s), dtype=np.float32)
    for idx in range(n):
        out[idx] = extract_stamp_stack(sci, xvals[idx], yvals[idx], radius)
    return out


# ---------------------------------------------------------------------------
# near-duplicate grid filter: src/kbmod/filters/clustering_grid.py
# ---------------------------------------------------------------------------


def grid_filter_indices(x, y, vx, vy, lh, bin_width=10, max_time=1.0):
    """apply_trajectory_grid_filter (clustering_grid.py:152-175) / TrajectoryClusterGrid.add_trajectory
    (:58-92) on columns: the index of the best trajectory (strictly larger lh replaces) of every
    (start bin, end bin) key, in the order in which the keys were first seen."""
    if bin_width < 1 or not np.isfinite(bin_width):
        raise ValueError(f"Bin width must be at least 1. Got {bin_width}.")
    if max_time < 0 or not np.isfinite(max_time):
        raise ValueError(f"Max time must be >= 0. Got {max_time}.")
    table, idx_table = {}, {}
    for i in range(len(x)):
        xi, yi, vxi, vyi = int(x[i]), int(y[i]), float(vx[i]), float(vy[i])
        key = (int(xi / bin_width), int(yi / bin_width), int((xi + max_time * vxi) / bin_width),
               int((yi + max_time * vyi) / bin_width))
        if key not in table:
            table[key] = lh[i]
            idx_table[key] = i
        elif lh[i] > table[key]:
            table[key] = lh[i]
            idx_table[key] = i
    return list(idx_table.values())
