"""GPU parity around the shift table's proof of uniform shifts (kb_shift_table_kernel, uniform_shift) and the kernel
instances the host picks from its verdict.

The reference predicts a sample's pixel per start pixel as floor((x + v * t) + 0.5) in double precision
(kernels.cu:33-35).  The staged kernel sums a (candidate, epoch) with ONE shift for all start pixels when the table
proves that floor cannot depend on x: the fraction of v * t + 0.5 keeps 2^-27 from both ends, or v * t has so few
fractional bits (dyadic times) that both sums are exact.  Everything else is summed per lane (special_epoch).  These
tests build all three cases on purpose, compare bit for bit with the oracle, and read the table's verdict from
last_search_stats()["special_epochs"] and the kernel instance from ["kernel_name"].
"""

import numpy as np
import pytest

from kbmod_amd import fake_data as fd
from tests import util

pytestmark = pytest.mark.gpu

LDS, TALL = 4, 64


def _check(got, exp):
    assert got.shape == exp.shape, (got.shape, exp.shape)
    bad = np.nonzero(np.any(got != exp, axis=1))[0]
    assert len(bad) == 0, f"{len(bad)} rows differ, first {bad[:3]}: {got[bad[:3]]} vs {exp[bad[:3]]}"


def _near_half_velocities(times, want, band=5e-9, v_lo=2.0, v_hi=30.0):
    """float32 velocities v with frac(v * t + 0.5) within `band` of 0 or 1 for some t of `times`, v * t not dyadic."""
    found = []
    for t in times[1:]:
        for k in range(int(v_lo * t), int(v_hi * t) + 1):
            centre = np.float32((k + 0.5) / t)
            for step in range(-40, 41):
                v = np.float32(centre + step * np.spacing(centre))
                a = float(v) * float(t)
                g = a + 0.5
                frac = g - np.floor(g)
                exact = (a * 2.0**29) == np.floor(a * 2.0**29)
                if not exact and (frac < band or frac > 1.0 - band) and v_lo <= v <= v_hi:
                    found.append(float(v))
                    break
            if len(found) >= want:
                return np.array(found, dtype=np.float32)
    return np.array(found, dtype=np.float32)


def test_guard_band_cases_are_summed_per_lane(kb, orc):
    """Non-dyadic times: velocities whose v * t + 0.5 comes within 5e-9 of an integer at some epoch cannot be proven
    uniform (the band is 2^-27 = 7.5e-9) and go through the per-lane path; the result is the oracle's."""
    times = np.arange(12) * 0.3  # 0.3 is not dyadic: products carry 50-odd fractional bits
    near = _near_half_velocities(times, want=2, v_hi=9.0)  # (inside the grid's own range: the chunks still fit their slabs)
    assert len(near) >= 1, "no velocity inside the guard band found for these times"
    st = util.make_stack(12, 40, 150, seed=5, objects=[(30, 10, 9.0, 2.0, 200.0)], mask_fraction=0.02, times=times)
    vx, vy = fd.kbmod_v1_candidates(8, 2.0, 9.0, 8, 0.02, 0.9)
    vx, vy = vx.copy(), vy.copy()
    for i, v in enumerate(near):
        vx[5 + 16 * i] = v
        vy[9 + 16 * i] = v
    got, exp, s = util.run_both(kb, orc, st, vx, vy, {"min_lh": -1e30, "xb": (-5, 155), "yb": (-3, 44)}, flags=LDS)
    stats = s.last_search_stats()
    assert stats["kernel_name"].startswith("kb::kb_search_lds<"), stats
    assert stats["special_epochs"] >= 1, stats
    _check(got, exp)


def test_exact_half_pixels_with_dyadic_times_are_uniform(kb, orc):
    """v * t + 0.5 lands exactly on integers, but with times k / 4 and these velocities every sum of the reference's
    prediction is exact: one shift serves every start pixel, no epoch is summed per lane -- and the oracle agrees."""
    times = np.array([0.0, 0.25, 0.5, 0.75, 1.0, 1.25, 1.5])  # (short enough for 16 directions to share a slab)
    st = util.make_stack(7, 48, 140, seed=11, times=times)
    vx = np.array([2.0, 6.0, -2.0, 1.0, 3.0, -5.0, 10.5, 0.5, 2.0, 6.0, -2.0, 1.0, 3.0, -5.0, 10.5, 0.5], dtype=np.float32)
    vy = np.array([2.0, -6.0, 1.0, 0.0, 1.0, 7.0, -3.5, 0.5, -2.0, 6.0, 1.5, 0.0, -1.0, 7.0, 3.5, 0.25], dtype=np.float32)
    got, exp, s = util.run_both(kb, orc, st, vx, vy, {"min_lh": -1e30, "xb": (-3, 143), "yb": (-3, 51)}, flags=LDS)
    stats = s.last_search_stats()
    assert stats["kernel_name"].startswith("kb::kb_search_lds<"), stats
    assert stats["special_epochs"] == 0, stats
    _check(got, exp)


def test_wide_chunks_for_clean_lists_narrow_ones_otherwise(kb, orc):
    """Lists of up to 8 results per pixel, float staging, nothing to sum per lane: the instance for chunks of 16
    candidates runs (half the passes over the stack).  One candidate inside the guard band, and the tables are rebuilt
    for chunks of 8 -- that instance has the per-lane path.  Both are the oracle's result."""
    times = np.arange(10) * 0.2
    st = util.make_stack(10, 64, 160, seed=8, objects=[(20, 12, 12.0, 5.0, 300.0)], times=times)
    vx, vy = fd.kbmod_v1_candidates(8, 3.0, 11.0, 6, 0.05, 1.2)  # 48 candidates: three wide chunks
    cfg = {"min_obs": 2, "K": 8}
    got, exp, s = util.run_both(kb, orc, st, vx, vy, cfg, flags=LDS | TALL)
    stats = s.last_search_stats()
    assert stats["special_epochs"] == 0, stats
    assert stats["kernel_name"].startswith("kb::kb_search_lds<8, 16,"), stats
    _check(got, exp)

    near = _near_half_velocities(times, want=1, v_lo=3.0, v_hi=11.0)
    assert len(near) == 1
    vx2, vy2 = vx.copy(), vy.copy()
    vx2[17] = near[0]
    got, exp, s = util.run_both(kb, orc, st, vx2, vy2, cfg, flags=LDS | TALL)
    stats = s.last_search_stats()
    assert stats["special_epochs"] >= 1, stats
    assert stats["kernel_name"].startswith("kb::kb_search_lds<8, 8,"), stats
    _check(got, exp)


@pytest.mark.parametrize("n_cands", [17, 31, 32, 33, 100])
def test_wide_chunks_ragged_candidate_counts(kb, orc, n_cands):
    """Candidate counts that are not multiples of 16 (nor of 8): the last wide chunk is padded with its first shift."""
    times = np.arange(9) * 0.3
    st = util.make_stack(9, 40, 130, seed=31 + n_cands, objects=[(11, 7, 6.0, 3.0, 150.0)], mask_fraction=0.01, times=times)
    vx, vy = fd.kbmod_v1_candidates(10, 1.0, 9.0, 10, -0.4, 0.8)
    vx, vy = vx[:n_cands].copy(), vy[:n_cands].copy()
    got, exp, s = util.run_both(kb, orc, st, vx, vy, {"min_obs": 3, "K": 5}, flags=LDS)
    stats = s.last_search_stats()
    assert stats["kernel_name"].startswith("kb::kb_search_lds<"), stats
    if stats["special_epochs"] == 0:
        assert stats["kernel_name"].startswith("kb::kb_search_lds<8, 16,"), stats
    _check(got, exp)
