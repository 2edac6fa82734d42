"""Observation counts of tiles at the image's edge out of tables (edge_counts, csrc/search_lds.h; kb_edge_count_kernel)
instead of one vector instruction per sample: the result must be what the counting loops and the oracle give, wherever
the tables apply -- and the tables must not apply where their premises fail (NO_DATA pixels in the stack, epochs out of
time order, start pixels off the image)."""

import os

import numpy as np
import pytest

from kbmod_amd import fake_data as fd
from tests import util

pytestmark = pytest.mark.gpu


def _stack(T, H, W, seed, mask_fraction=0.0, times=None):
    return util.make_stack(T, H, W, seed=seed, noise=2.0, psf=1.0, objects=[(W // 3, H // 3, 9.0, 6.0, 300.0)],
                           mask_fraction=mask_fraction, times=times)


def _oracle(orc, stack, vx, vy, cfg):
    pp = orc.PsiPhi.from_images(stack.sci, stack.var, stack.psfs, stack.zeroed_times)
    return pp.search_kernel_semantics(orc.make_candidates(vx, vy), util.oracle_params(pp, cfg))


def _same(got, exp, what):
    g = util.as_records(got)
    for name in util.FIELDS:
        assert np.array_equal(g[name], exp[name]), (name, what)


# all four directions, mixed within chunks of 16, zero velocity, a long reach along one axis (the padded copy of the staged
# kernel may not outweigh the image 4 : 1, or the direct kernel runs instead)
GRIDS = {
    "towards +x +y": fd.kbmod_v1_candidates(8, 2.0, 30.0, 8, 0.0, 1.5),
    "all directions": fd.velocity_grid_candidates(9, -12.0, 12.0, 7, -10.0, 10.0),
}
# a long reach along x; the 16 candidates of a chunk stay close together (a slab holds 64 + 24 columns at most)
_bases = np.repeat(np.array([-36.0, -18.0, 18.0, 36.0], dtype=np.float32), 16)
GRIDS["fast along x"] = (_bases + np.tile(np.arange(16, dtype=np.float32) * 0.5, 4),
                         np.tile(np.array([-3.0, 3.0, 0.0, 1.5], dtype=np.float32), 16))
SHAPE = (12, 96, 200)


@pytest.mark.parametrize("grid", list(GRIDS))
@pytest.mark.parametrize("cfg", [dict(K=8), dict(K=4, min_obs=9, min_lh=1.0)])
def test_counts_from_tables_equal_oracle_and_counting_loops(orc, grid, cfg):
    stack = _stack(*SHAPE, seed=5)
    ds = util.DeviceStack(stack)
    try:
        vx, vy = GRIDS[grid]
        got, st = ds.search(ds.params(**cfg), ds.candidates(vx, vy), 0)
        assert st.edge_count_tables == 1 and st.kernel_name.decode().startswith("kb::kb_search_lds<8, 16,"), st.kernel_name
        exp = _oracle(orc, stack, vx, vy, cfg)
        _same(got, exp, (grid, cfg))
        rec = util.as_records(got)
        assert (rec["obs_count"][rec["lh"] > -1] < 12).any()  # trajectories that leave the image are among the results
        os.environ["KBMOD_EDGE_COUNTS"] = "0"
        try:
            counted, st0 = ds.search(ds.params(**cfg), ds.candidates(vx, vy), 0)
        finally:
            del os.environ["KBMOD_EDGE_COUNTS"]
        assert st0.edge_count_tables == 0
        assert counted.cpu().numpy().tobytes() == got.cpu().numpy().tobytes()
    finally:
        ds.close()


def test_stack_with_masked_pixels_keeps_counting(orc):
    stack = _stack(10, 96, 200, seed=6, mask_fraction=0.02)
    ds = util.DeviceStack(stack)
    try:
        vx, vy = GRIDS["all directions"]
        got, st = ds.search(ds.params(K=8, min_obs=3), ds.candidates(vx, vy), 0)
        assert st.edge_count_tables == 1  # built (the host cannot know) -- and ignored by every tile: the copy holds NO_DATA
        _same(got, _oracle(orc, stack, vx, vy, dict(K=8, min_obs=3)), "masked")
    finally:
        ds.close()


def test_epochs_out_of_time_order_keep_counting(orc):
    # the in-bounds epochs of a trajectory are then no leading run of the epochs: the table kernel clears its flag
    times = np.array([0.0, 0.5, 0.1, 0.9, 0.3, 0.7, 0.2, 1.0])
    stack = _stack(8, 96, 200, seed=7, times=times)
    ds = util.DeviceStack(stack)
    try:
        vx, vy = GRIDS["all directions"]
        got, st = ds.search(ds.params(K=8), ds.candidates(vx, vy), 0)
        assert st.edge_count_tables == 1
        _same(got, _oracle(orc, stack, vx, vy, dict(K=8)), "unsorted epochs")
    finally:
        ds.close()


def test_start_pixels_off_the_image_build_no_tables(orc):
    stack = _stack(9, 96, 200, seed=8)
    ds = util.DeviceStack(stack)
    try:
        vx, vy = GRIDS["towards +x +y"]
        cfg = dict(K=8, xb=(-10, 210), yb=(-5, 99))
        got, st = ds.search(ds.params(**cfg), ds.candidates(vx, vy), 0)
        assert st.edge_count_tables == 0
        _same(got, _oracle(orc, stack, vx, vy, cfg), "off-image start pixels")
        cfg = dict(K=8, xb=(20, 200), yb=(3, 96))  # a window that touches the right and bottom edges
        got, st = ds.search(ds.params(**cfg), ds.candidates(vx, vy), 0)
        assert st.edge_count_tables == 1
        _same(got, _oracle(orc, stack, vx, vy, cfg), "window at the edge")
    finally:
        ds.close()


def test_stable_lists_of_the_exchange_use_the_tables(orc):
    """flag 512 (what every rank of the multi-device search runs): 16 stable records per pixel, pooled lists."""
    stack = _stack(10, 96, 200, seed=9)
    ds = util.DeviceStack(stack)
    try:
        vx, vy = GRIDS["all directions"]
        p = ds.params(K=16)
        got, st = ds.search_compact(p, ds.candidates(vx, vy), 0, 512)
        assert st.edge_count_tables == 1 and ", 4>" in st.kernel_name.decode(), st.kernel_name
        os.environ["KBMOD_EDGE_COUNTS"] = "0"
        try:
            counted, st0 = ds.search_compact(p, ds.candidates(vx, vy), 0, 512)
        finally:
            del os.environ["KBMOD_EDGE_COUNTS"]
        assert st0.edge_count_tables == 0
        assert counted.cpu().numpy().tobytes() == got.cpu().numpy().tobytes()
        direct, _ = ds.search_compact(p, ds.candidates(vx, vy), 0, 512 | 2)
        assert direct.cpu().numpy().tobytes() == got.cpu().numpy().tobytes()
        rec = util.as_records(got, util.COMPACT_DTYPE)
        assert (rec["obs_count"][rec["cand"] >= 0] < 10).any()
    finally:
        ds.close()


def test_sigma_g_emitting_instance_uses_the_tables(orc):
    """In-search sigma-G: the counts only decide which trajectories are emitted for clipping (min_obs on the unclipped count)."""
    stack = _stack(12, 96, 200, seed=10)
    for num_bytes in (-1, 1):
        ds = util.DeviceStack(stack, num_bytes)
        try:
            vx, vy = GRIDS["all directions"]
            cfg = dict(K=4, min_obs=8, sigmag=(0.25, 0.75, 0.7413, 2.0))
            got, st = ds.search(ds.params(**cfg), ds.candidates(vx, vy), 0)
            assert st.edge_count_tables == 1 and ", true, 0>" in st.kernel_name.decode(), st.kernel_name
            pp = orc.PsiPhi.from_images(stack.sci, stack.var, stack.psfs, stack.zeroed_times, num_bytes)
            exp = pp.search_kernel_semantics(orc.make_candidates(vx, vy), util.oracle_params(pp, cfg))
            _same(got, exp, ("sigma-G", num_bytes))
            os.environ["KBMOD_EDGE_COUNTS"] = "0"
            try:
                counted, st0 = ds.search(ds.params(**cfg), ds.candidates(vx, vy), 0)
            finally:
                del os.environ["KBMOD_EDGE_COUNTS"]
            assert st0.edge_count_tables == 0 and counted.cpu().numpy().tobytes() == got.cpu().numpy().tobytes()
        finally:
            ds.close()


def test_fast_movers_beyond_the_table_cap_keep_counting(orc):
    # shifts of more than 200 pixels: no tables (their size grows with the largest shift), same results
    stack = util.make_stack(5, 420, 520, seed=11, noise=2.0, psf=1.0, objects=[(40, 30, 300.0, 120.0, 400.0)])
    ds = util.DeviceStack(stack)
    try:
        vx = np.repeat(np.array([296.0, 304.0], dtype=np.float32), 16) + np.tile(np.arange(16, dtype=np.float32) * 0.25, 2)
        vy = np.tile(np.array([118.0, 120.0, 122.0, 124.0], dtype=np.float32), 8)
        cfg = dict(K=8, min_obs=2)
        got, st = ds.search(ds.params(**cfg), ds.candidates(vx, vy), 0)
        assert st.kernel_name.decode().startswith("kb::kb_search_lds<8, 16,"), st.kernel_name
        assert st.edge_count_tables == 0
        _same(got, _oracle(orc, stack, vx, vy, cfg), "fast movers")
    finally:
        ds.close()


def test_sigma_g_batches_share_the_tables(orc):
    # a work-item store too small for the candidate list: the emit runs in batches of chunks, every batch reads its rows
    stack = _stack(12, 96, 200, seed=12)
    ds = util.DeviceStack(stack)
    os.environ["KBMOD_SIGMAG_CAP"] = "4096"
    try:
        vx, vy = fd.velocity_grid_candidates(16, -12.0, 12.0, 12, -10.0, 10.0)
        cfg = dict(K=4, min_obs=8, sigmag=(0.25, 0.75, 0.7413, 2.0))
        got, st = ds.search(ds.params(**cfg), ds.candidates(vx, vy), 0)
        assert st.edge_count_tables == 1 and st.num_search_launches > 1, st.num_search_launches
        _same(got, _oracle(orc, stack, vx, vy, cfg), "sigma-G batches")
    finally:
        del os.environ["KBMOD_SIGMAG_CAP"]
        ds.close()
