"""Stamp coadds for result trajectories on the device -- SURVEY.md section 8(f3).

Mirrors the batched entry point of the reference, ``append_coadds(result_data, im_stack,
coadd_types, radius, valid_only)`` (src/kbmod/filters/stamp_filters.py:72-168), on plain arrays:
the image stack is uploaded once (``DeviceStack``), the stamp centres are predicted with the
reference's ``predict_pixel_locations`` arithmetic (src/kbmod/trajectory_utils.py:28-75) and all
trajectories are coadded in one ``kb_coadd_stamps`` launch per coadd type instead of the reference's
per-trajectory ``extract_stamp_stack`` + ``coadd_*`` loop (src/kbmod/core/stamp_utils.py).
Raises ``RuntimeError`` without a GPU; there is no host fallback.
"""

import numpy as np

from . import search as _search

COADD_TYPES = ("sum", "mean", "median", "weighted")


def predict_pixel_locations(times, x0, vx, centered=True, as_int=True):
    """R x T predicted pixel positions (trajectory_utils.py:28-75; ``astype(int)`` truncates)."""
    times = np.asarray(times)
    x0 = np.asarray(x0)
    vx = np.asarray(vx)
    if len(x0) != len(vx):
        raise ValueError(f"x0 and vx must be same size. Found {len(x0)} vs {len(vx)}")
    pos = vx[:, np.newaxis] * times[np.newaxis, :] + x0[:, np.newaxis]
    if centered:
        pos = pos + 0.5
    if as_int:
        pos = pos.astype(int)
    return pos


class DeviceStack:
    """Science (and variance) images of an ``ImageStackPy`` resident in HBM."""

    def __init__(self, sci, var=None, zeroed_times=None):
        sci = np.ascontiguousarray(np.asarray(sci, dtype=np.float32))
        if sci.ndim != 3:
            raise ValueError("expected T images of the same H x W shape")
        var_arr = None if var is None else np.ascontiguousarray(np.asarray(var, dtype=np.float32))
        self._dev = _search.DeviceImageStack(sci, var_arr)
        self.zeroed_times = None if zeroed_times is None else np.asarray(zeroed_times, dtype=np.float64)

    num_times = property(lambda self: self._dev.num_times)
    height = property(lambda self: self._dev.height)
    width = property(lambda self: self._dev.width)

    def coadds(self, xvals, yvals, radius, coadd_types, to_include=None):
        """{type: N x (2r+1) x (2r+1) float32} for N x T integer stamp centres."""
        if radius <= 0:
            raise ValueError(f"Invalid stamp radius {radius}")
        for c in coadd_types:
            if c not in COADD_TYPES:
                raise ValueError(f"Unknown coadd type {c}")
        xvals = np.ascontiguousarray(np.asarray(xvals, dtype=int).astype(np.int32))
        yvals = np.ascontiguousarray(np.asarray(yvals, dtype=int).astype(np.int32))
        if xvals.ndim != 2 or xvals.shape != yvals.shape or xvals.shape[1] != self.num_times:
            raise ValueError("X and Y values must have the same length as the number of times.")
        inc = None
        if to_include is not None:
            inc = np.ascontiguousarray(np.asarray(to_include, dtype=bool))
            if inc.shape != xvals.shape:
                raise ValueError("Time mask must have the same length as the number of times.")
        return dict(self._dev.coadds(xvals, yvals, inc, int(radius), list(coadd_types)))


def append_coadds(result_data, im_stack, coadd_types, radius, valid_only=True):
    """``append_coadds`` on a dict-like table of columns ``x, y, vx, vy`` (and ``obs_valid``): adds
    ``coadd_<type>`` columns of float32 stamps, computed on the device.  ``im_stack``: a ``DeviceStack``
    with ``zeroed_times``."""
    if radius <= 0:
        raise ValueError(f"Invalid stamp radius {radius}")
    times = im_stack.zeroed_times
    if times is None:
        raise ValueError("the stack needs its zeroed times")
    valid_only = valid_only and "obs_valid" in result_data
    xvals = predict_pixel_locations(times, result_data["x"], result_data["vx"], centered=True, as_int=True)
    yvals = predict_pixel_locations(times, result_data["y"], result_data["vy"], centered=True, as_int=True)
    inc = np.asarray(result_data["obs_valid"], dtype=bool) if valid_only else None
    for name, stamps in im_stack.coadds(xvals, yvals, radius, coadd_types, to_include=inc).items():
        result_data[f"coadd_{name}"] = stamps
    return result_data
