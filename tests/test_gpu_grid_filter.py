"""Near-duplicate grid filter on the device (kb_grid_filter) -- SURVEY.md section 8(f2).
Known answers from the reference's tests/test_clustering_grid.py; parity (same indices, same order)
against oracle/post_search.py's dictionary restatement on search results and on adversarial lists."""

import numpy as np
import pytest

from oracle import post_search as ps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cg():
    from kbmod_amd import clustering_grid

    return clustering_grid


def _trjs(kb, rows):
    return [kb.Trajectory(int(x), int(y), float(vx), float(vy), 1.0, float(lh), 10) for x, y, vx, vy, lh in rows]


def _oracle(trjs, bw, dt):
    cols = [np.array([getattr(t, f) for t in trjs]) for f in ("x", "y", "vx", "vy", "lh")]
    return ps.grid_filter_indices(*cols, bin_width=bw, max_time=dt)


def test_known_answers(cg, kb):
    # tests/test_clustering_grid.py:93-108
    rows = [(0, 0, 0.0, 0.0, 10.0), (21, 21, 10.0, 10.0, 10.0), (21, 21, 0.0, 0.0, 10.0), (21, 21, 0.0, 0.0, 100.0),
            (0, 0, 0.0, 0.0, 5.0), (0, 0, 0.0, 0.0, 15.0)]
    trjs = _trjs(kb, rows)
    res, idx = cg.apply_trajectory_grid_filter(trjs, bin_width=10, max_dt=1.0)
    assert idx == [5, 1, 3] and [t.lh for t in res] == [15.0, 10.0, 100.0]
    # :8-57: after the first five trajectories the online structure holds indices {0, 1, 3}
    assert set(cg.apply_trajectory_grid_filter(trjs[:5], bin_width=10, max_dt=1.0)[1]) == {0, 1, 3}
    assert cg.apply_trajectory_grid_filter([], 10, 1.0) == ([], [])
    with pytest.raises(ValueError):
        cg.apply_trajectory_grid_filter(trjs, 0, 1.0)
    with pytest.raises(ValueError):
        cg.apply_trajectory_grid_filter(trjs, 10, -1.0)


@pytest.mark.parametrize("n,bw,dt", [(1, 10, 1.0), (5000, 10, 1.0), (20000, 3, 2.5), (3000, 1, 0.0), (4000, 50.5, 7.0)])
def test_random_lists_equal_oracle(cg, kb, n, bw, dt):
    rng = np.random.default_rng(n)
    rows = zip(rng.integers(-40, 300, n), rng.integers(-40, 200, n), rng.uniform(-30, 30, n).astype(np.float32),
               rng.uniform(-30, 30, n).astype(np.float32),
               # few distinct likelihoods: ties must go to the earliest trajectory
               rng.integers(0, 12, n).astype(np.float32) * 0.5 - 1.0)
    trjs = _trjs(kb, rows)
    res, idx = cg.apply_trajectory_grid_filter(trjs, bw, dt)
    assert idx == _oracle(trjs, bw, dt)
    assert [(t.x, t.y, t.lh) for t in res] == [(trjs[i].x, trjs[i].y, trjs[i].lh) for i in idx]


def test_on_search_results(cg, kb):
    from kbmod_amd import fake_data as fd
    from tests import util

    st = util.make_stack(16, 60, 70, seed=4, noise=2.0, psf=1.0, objects=[(20, 15, 12.0, 7.0, 300.0)])
    s = kb.StackSearch(st.sci, st.var, st.psfs, st.zeroed_times)
    s.set_min_lh(2.0)
    vx, vy = fd.kbmod_v1_candidates(16, 2.0, 20.0, 8, 0.0, 1.2)
    s.search_all(util.trajectories(kb, vx, vy), True)
    trjs = s.get_results(0, 20000)
    assert len(trjs) > 1000
    res, idx = cg.apply_trajectory_grid_filter(trjs, 5, float(st.zeroed_times[-1]))
    assert idx == _oracle(trjs, 5, float(st.zeroed_times[-1])) and 0 < len(idx) < len(trjs)
    assert idx[0] == 0  # the list is sorted by likelihood: its head always survives


def test_filter_sort_results_checked_through_the_c_abi():
    """kb_filter_sort_results / _checked on a crafted result buffer: survivors of (lh >= min_lh, obs >= min_obs) in stable
    descending likelihood order (stack_search.cpp:266-277, trajectory_list.cpp:96-126) and the lowest index of a record that
    fails Trajectory::is_valid (the scan of trajectory_list.cpp:155-164), -1 when there is none."""
    import ctypes as C

    import torch

    from kbmod_amd import capi

    lib = capi.load_lib()
    lib.kb_filter_sort_results.argtypes = [C.c_void_p, C.c_uint64, C.c_float, C.c_int32, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]
    lib.kb_filter_sort_results_checked.argtypes = [C.c_void_p, C.c_uint64, C.c_float, C.c_int32, C.c_void_p, C.POINTER(C.c_uint64),
                                                   C.POINTER(C.c_int64), C.c_void_p]
    dt = np.dtype([("vx", "<f4"), ("vy", "<f4"), ("lh", "<f4"), ("flux", "<f4"), ("x", "<i4"), ("y", "<i4"), ("obs", "<i4")])
    rng = np.random.default_rng(5)
    for n in (1, 1000, 300_000):
        rec = np.zeros(n, dtype=dt)
        rec["lh"] = np.round(rng.normal(5, 3, n), 1)     # many equal likelihoods: the sort must be stable
        rec["obs"] = rng.integers(0, 20, n)
        rec["x"] = np.arange(n)
        rec["flux"] = rng.normal(0, 1, n)
        keep = ~(rec["lh"] < np.float32(4.0)) & ~(rec["obs"] < 5)
        want = rec[keep][np.argsort(-rec["lh"][keep], kind="stable")]
        d = torch.from_numpy(rec.view(np.uint8)).cuda()
        out = torch.zeros_like(d)
        cnt, bad = C.c_uint64(0), C.c_int64(7)
        capi.check(lib.kb_filter_sort_results(d.data_ptr(), n, 4.0, 5, out.data_ptr(), C.byref(cnt), None))
        assert cnt.value == len(want) and out.cpu().numpy().view(dt)[:len(want)].tobytes() == want.tobytes()
        out.zero_()
        capi.check(lib.kb_filter_sort_results_checked(d.data_ptr(), n, 4.0, 5, out.data_ptr(), C.byref(cnt), C.byref(bad), None))
        assert cnt.value == len(want) and bad.value == -1
        assert out.cpu().numpy().view(dt)[:len(want)].tobytes() == want.tobytes()
        if len(want) > 10:
            # an infinite flux and (further down the sorted list) a negative count: the first one in OUTPUT order is reported
            i_a, i_b = int(want["x"][len(want) // 3]), int(want["x"][len(want) // 2])
            rec2 = rec.copy()
            rec2["flux"][i_a] = np.inf
            rec2["vx"][i_b] = np.nan
            d2 = torch.from_numpy(rec2.view(np.uint8)).cuda()
            capi.check(lib.kb_filter_sort_results_checked(d2.data_ptr(), n, 4.0, 5, out.data_ptr(), C.byref(cnt), C.byref(bad), None))
            assert bad.value == len(want) // 3 and cnt.value == len(want)
