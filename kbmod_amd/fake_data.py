"""Synthetic inputs for the shift-and-stack search (tests and bench).

Own restatement of the input generators the reference's tests and README use,
so that the same shapes and value distributions are searched:

* ``create_fake_times``      -- fake_data/fake_data_creator.py:23-59
* ``make_gaussian_kernel``   -- core/psf.py:49-74 (radius ``int(3*sigma)``)
* ``make_fake_image_stack``  -- fake_data/fake_data_creator.py:62-100
* ``add_fake_object``        -- fake_data/fake_data_creator.py:128-172
* ``kbmod_v1_candidates``    -- trajectory_generator.py:416-456 (upper bounds exclusive)
* ``velocity_grid_candidates`` -- trajectory_generator.py:232-268 (inclusive grid)
* ``ecliptic_centered_candidates`` -- trajectory_generator.py:496-625 (angle offsets around an ecliptic angle, inclusive grid)

(paths relative to /root/reference/src/kbmod/).  Pure numpy; no GPU work here.
"""

import math

import numpy as np


def create_fake_times(num_times, t0=0.0, obs_per_day=1, intra_night_gap=0.01, inter_night_gap=1):
    if num_times <= 0:
        raise ValueError(f"Invalid number of times {num_times}")
    out = []
    seen_on_day = 0
    day_num = 0
    for _ in range(num_times):
        out.append(t0 + day_num + seen_on_day * intra_night_gap)
        seen_on_day += 1
        if seen_on_day == obs_per_day:
            seen_on_day = 0
            day_num += inter_night_gap
    return out


def make_gaussian_kernel(stddev, normalize=True):
    if stddev < 0:
        raise ValueError("Standard deviation must be non-negative.")
    radius = int(3 * stddev)
    x = np.arange(-radius, radius + 1)
    xx, yy = np.meshgrid(x, x)
    kernel = np.exp(-0.5 * (xx**2 + yy**2) / stddev**2)
    if normalize:
        kernel /= np.sum(kernel)
    return kernel.astype(np.float32)


class FakeStack:
    """Minimal stand-in for ``ImageStackPy``: lists of float32 images + times."""

    def __init__(self, times, sci, var, psfs):
        self.times = np.asarray(times, dtype=np.float64)
        self.zeroed_times = self.times - self.times[0]  # core/image_stack_py.py:102-104
        self.sci = sci
        self.var = var
        self.psfs = psfs
        self.height, self.width = sci[0].shape

    def __len__(self):
        return len(self.sci)


def make_fake_image_stack(height, width, times, noise_level=2.0, psf_val=0.5, psfs=None, rng=None):
    if rng is None:
        rng = np.random.default_rng()
    times = np.asarray(times)
    sci = [rng.normal(0.0, noise_level, (height, width)).astype(np.float32) for _ in range(len(times))]
    var = [np.full((height, width), noise_level**2).astype(np.float32) for _ in range(len(times))]
    if psfs is None:
        k = make_gaussian_kernel(psf_val)
        psfs = [k for _ in range(len(times))]
    elif len(psfs) != len(times):
        raise ValueError("The number of PSFs must be the same as times.")
    return FakeStack(times, sci, var, psfs)


def add_random_masks(stack, mask_fraction, rng):
    for idx in range(len(stack.sci)):
        mask = rng.random(stack.sci[idx].shape) < mask_fraction
        stack.sci[idx][mask] = np.nan
        stack.var[idx][mask] = np.nan


def add_fake_object(stack, x, y, vx, vy, flux=100.0):
    for idx, t in enumerate(stack.zeroed_times):
        k = stack.psfs[idx]
        dim = k.shape[0]
        rad = dim // 2
        px = int(x + vx * t + 0.5)
        py = int(y + vy * t + 0.5)
        for ky in range(dim):
            for kx in range(dim):
                ix = px + kx - rad
                iy = py + ky - rad
                if 0 <= ix < stack.width and 0 <= iy < stack.height and np.isfinite(stack.sci[idx][iy, ix]):
                    stack.sci[idx][iy, ix] += flux * k[ky, kx]


def kbmod_v1_candidates(vel_steps, min_vel, max_vel, ang_steps, min_ang, max_ang):
    """(vx, vy) float32 arrays in the reference generator's order: angle outer, velocity inner."""
    vel_step = (max_vel - min_vel) / float(vel_steps)
    ang_step = (max_ang - min_ang) / float(ang_steps)
    vxs, vys = [], []
    for ang_i in range(ang_steps):
        for vel_i in range(vel_steps):
            ang = min_ang + ang_i * ang_step
            vel = min_vel + vel_i * vel_step
            vxs.append(math.cos(ang) * vel)
            vys.append(math.sin(ang) * vel)
    return np.asarray(vxs, dtype=np.float32), np.asarray(vys, dtype=np.float32)


def velocity_grid_candidates(vx_steps, min_vx, max_vx, vy_steps, min_vy, max_vy):
    sx = (max_vx - min_vx) / float(vx_steps - 1)
    sy = (max_vy - min_vy) / float(vy_steps - 1)
    vxs, vys = [], []
    for vy_i in range(vy_steps):
        for vx_i in range(vx_steps):
            vxs.append(min_vx + vx_i * sx)
            vys.append(min_vy + vy_i * sy)
    return np.asarray(vxs, dtype=np.float32), np.asarray(vys, dtype=np.float32)


def sigmag_coeff(lo=25.0, hi=75.0):
    """1 / (Phi^-1(hi) - Phi^-1(lo)); 0.7413 for [25, 75] (filters/sigma_g_filter.py:49-83)."""
    from statistics import NormalDist

    nd = NormalDist()
    return 1.0 / (nd.inv_cdf(hi / 100.0) - nd.inv_cdf(lo / 100.0))


def ecliptic_centered_candidates(velocities, angles, given_ecliptic=0.0):
    """``EclipticCenteredSearch``: velocities = [min, max, steps] in pixels per day, angles = [min offset, max offset,
    steps] in radians around ``given_ecliptic``; both grids inclusive, angle outer, velocity inner."""
    v_lo, v_hi, v_n = velocities
    a_lo, a_hi, a_n = angles
    if a_n < 1 or v_n < 1 or v_hi < v_lo:
        raise ValueError("invalid ecliptic-centred grid")
    vel_step = (v_hi - v_lo) / float(v_n - 1)
    min_ang = given_ecliptic + a_lo
    ang_step = ((given_ecliptic + a_hi) - min_ang) / float(a_n - 1)
    vxs, vys = [], []
    for ang_i in range(int(a_n)):
        for vel_i in range(int(v_n)):
            ang = min_ang + ang_i * ang_step
            vel = v_lo + vel_i * vel_step
            vxs.append(math.cos(ang) * vel)
            vys.append(math.sin(ang) * vel)
    return np.asarray(vxs, dtype=np.float32), np.asarray(vys, dtype=np.float32)
