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
