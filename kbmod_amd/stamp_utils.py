"""Stamps of result trajectories on the device -- SURVEY.md section 8(f3).

The reference builds them on the host, one trajectory at a time (src/kbmod/filters/stamp_filters.py:
``append_coadds`` :72-168 -- all epochs or one coadd per calendar night -- and ``append_all_stamps``
:171-211, both loops of ``extract_stamp_stack`` + ``coadd_*`` from src/kbmod/core/stamp_utils.py).  Here the
image stack is uploaded once (``DeviceStack``), the stamp centres of every trajectory are predicted in one
batch with the reference's ``predict_pixel_locations`` arithmetic (src/kbmod/trajectory_utils.py:28-75), and
one launch per coadd type (``kb_coadd_stamps``) or one launch in all (``kb_extract_stamps``) covers every
trajectory.  A nightly coadd is the same launch with the epoch mask narrowed to the night.
Raises ``RuntimeError`` without a GPU; there is no host fallback.
"""

import datetime

import numpy as np

from . import search as _search

COADD_TYPES = ("sum", "mean", "median", "weighted")
_MJD_EPOCH = datetime.date(1858, 11, 17)


def mjd_to_day(mjd):
    """Calendar date (UTC) of a modified Julian date as YYYY-MM-DD (util_functions.py:52-65: mjd 60000 is
    2023-02-25)."""
    return (_MJD_EPOCH + datetime.timedelta(days=int(np.floor(mjd)))).isoformat()


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
    """Science (and variance) images of an ``ImageStackPy`` resident in HBM, with the epoch times the stamp
    functions need (``zeroed_times`` for the positions, ``times`` -- MJD -- for the nights)."""

    def __init__(self, sci, var=None, zeroed_times=None, times=None):
        sci = np.ascontiguousarray(np.asarray(sci, dtype=np.float32))
        if sci.ndim != 3:
            raise ValueError("expected T images of the same H x W shape")
        var_arr = None if var is None else np.ascontiguousarray(np.asarray(var, dtype=np.float32))
        self._dev = _search.DeviceImageStack(sci, var_arr)
        self.zeroed_times = None if zeroed_times is None else np.asarray(zeroed_times, dtype=np.float64)
        self.times = None if times is None else np.asarray(times, dtype=np.float64)

    @classmethod
    def from_device(cls, sci, var=None, zeroed_times=None, times=None):
        """The same over stacks that already are on the device: ``sci`` / ``var`` = contiguous float32 CUDA tensors
        [T][H][W] (e.g. ``kbmod_amd.fits_ingest.DeviceWorkUnit.sci`` / ``.var``).  Nothing is copied; the tensors are kept
        alive with the stack."""
        for t in (sci, var):
            if t is not None and not (t.is_cuda and t.is_contiguous() and t.dim() == 3 and str(t.dtype) == "torch.float32"):
                raise ValueError("expected contiguous float32 CUDA tensors [T][H][W]")
        if var is not None and tuple(var.shape) != tuple(sci.shape):
            raise ValueError("science and variance stacks differ in shape")
        self = cls.__new__(cls)
        T, H, W = (int(n) for n in sci.shape)
        self._dev = _search.DeviceImageStack.from_device(sci.data_ptr(), 0 if var is None else var.data_ptr(), T, H, W, (sci, var))
        self.zeroed_times = None if zeroed_times is None else np.asarray(zeroed_times, dtype=np.float64)
        self.times = None if times is None else np.asarray(times, dtype=np.float64)
        return self

    num_times = property(lambda self: self._dev.num_times)
    height = property(lambda self: self._dev.height)
    width = property(lambda self: self._dev.width)

    def _centres(self, xvals, yvals):
        xvals = np.ascontiguousarray(np.asarray(xvals, dtype=int).astype(np.int32))
        yvals = np.ascontiguousarray(np.asarray(yvals, dtype=int).astype(np.int32))
        if xvals.ndim != 2 or xvals.shape != yvals.shape or xvals.shape[1] != self.num_times:
            raise ValueError("X and Y values must have the same length as the number of times.")
        return xvals, yvals

    def coadds(self, xvals, yvals, radius, coadd_types, to_include=None):
        """{type: N x (2r+1) x (2r+1) float32} for N x T integer stamp centres."""
        if radius <= 0:
            raise ValueError(f"Invalid stamp radius {radius}")
        for c in coadd_types:
            if c not in COADD_TYPES:
                raise ValueError(f"Unknown coadd type {c}")
        xvals, yvals = self._centres(xvals, yvals)
        inc = None
        if to_include is not None:
            inc = np.ascontiguousarray(np.asarray(to_include, dtype=bool))
            if inc.shape != xvals.shape:
                raise ValueError("Time mask must have the same length as the number of times.")
        return dict(self._dev.coadds(xvals, yvals, inc, int(radius), list(coadd_types)))

    def all_stamps(self, xvals, yvals, radius):
        """N x T x (2r+1) x (2r+1) float32: every epoch's stamp of every trajectory, NaN outside the image."""
        if radius < 1:
            raise ValueError(f"Invalid stamp radius: {radius}")
        xvals, yvals = self._centres(xvals, yvals)
        return self._dev.all_stamps(xvals, yvals, int(radius))


def _positions(result_data, im_stack):
    times = im_stack.zeroed_times
    if times is None:
        raise ValueError("the stack needs its zeroed times")
    xvals = predict_pixel_locations(times, result_data["x"], result_data["vx"], centered=True, as_int=True)
    yvals = predict_pixel_locations(times, result_data["y"], result_data["vy"], centered=True, as_int=True)
    return xvals, yvals


def append_coadds(result_data, im_stack, coadd_types, radius, valid_only=True, nightly=False):
    """``append_coadds`` (stamp_filters.py:72-168) on a dict-like table of columns ``x, y, vx, vy`` (and
    ``obs_valid``): adds ``coadd_<type>`` columns of float32 stamps and, with ``nightly``, one
    ``coadd_<type>_<YYYY-MM-DD>`` column per calendar night of ``im_stack.times``, all computed on the
    device.  ``im_stack``: a ``DeviceStack``."""
    if radius <= 0:
        raise ValueError(f"Invalid stamp radius {radius}")
    valid_only = valid_only and "obs_valid" in result_data
    xvals, yvals = _positions(result_data, im_stack)
    num_res = len(xvals)
    inc = np.asarray(result_data["obs_valid"], dtype=bool) if valid_only else None
    for name, stamps in im_stack.coadds(xvals, yvals, radius, coadd_types, to_include=inc).items():
        result_data[f"coadd_{name}"] = stamps
    if nightly:
        if im_stack.times is None:
            raise ValueError("nightly coadds need the stack's epoch times (MJD)")
        day_strs = np.array([f"_{mjd_to_day(t)}" for t in im_stack.times])
        base = inc if inc is not None else np.ones((num_res, len(day_strs)), dtype=bool)
        for day in np.unique(day_strs):
            night = base & (day_strs == day)[np.newaxis, :]
            for name, stamps in im_stack.coadds(xvals, yvals, radius, coadd_types, to_include=night).items():
                result_data[f"coadd_{name}{day}"] = stamps
    return result_data


def append_all_stamps(result_data, im_stack, stamp_radius):
    """``append_all_stamps`` (stamp_filters.py:171-211): adds the column ``all_stamps``, N x T x (2r+1) x
    (2r+1) float32, extracted on the device."""
    if stamp_radius < 1:
        raise ValueError(f"Invalid stamp radius: {stamp_radius}")
    xvals, yvals = _positions(result_data, im_stack)
    result_data["all_stamps"] = im_stack.all_stamps(xvals, yvals, stamp_radius)
    return result_data
