"""kb_search_lds with 32 candidates per staged slab (XWIDE_CHUNK; `kb::kb_search_lds<8, 32, 16, 4, true, false, 3>`): the
instance the host takes for arrays beyond the Infinity Cache.  Pinned here on small stacks with KBMOD_CHUNK=32 (which asks for
the instance wherever it can run) against the oracle, the direct kernel and the chunks-of-16 instance, bit for bit; and the
cases the host must keep away from it (NO_DATA pixels, start pixels off the image, sigma-G, K > 8) must fall back and still
equal the oracle."""

import numpy as np
import pytest

from kbmod_amd import fake_data as fd
from tests import util

pytestmark = pytest.mark.gpu

DIRECT, LDS = 2, 4
XWIDE = "kb::kb_search_lds<8, 32, 16, 4, true, false, 3>"


def _clean_stack(T, H, W, seed, times=None, objects=()):
    return util.make_stack(T, H, W, seed=seed, objects=list(objects), times=times)


def _name(s):
    return s.last_search_stats()["kernel_name"]


@pytest.mark.parametrize("shape", [(16, 70, 130), (40, 33, 64), (7, 100, 200), (33, 48, 257)])
@pytest.mark.parametrize("grid", [(16, 4, 0.0, 1.5), (32, 3, -0.4, 0.4), (11, 6, 0.2, 1.2)])
@pytest.mark.parametrize("K", [1, 5, 8])
def test_instance_against_the_oracle(kb, orc, shape, grid, K, monkeypatch):
    T, H, W = shape
    vel, ang, a0, a1 = grid
    st = _clean_stack(T, H, W, seed=T * 1000 + H, times=np.arange(T) / 32.0,   # (dyadic: no shift on a rounding boundary)
                      objects=[(W // 3, H // 3, 12.0, 7.0, 300.0), (5, 4, 20.0, 2.0, 250.0)])
    vx, vy = fd.kbmod_v1_candidates(vel, 2.0, 30.0, ang, a0, a1)
    cfg = {"K": K, "min_obs": T // 3}
    monkeypatch.setenv("KBMOD_CHUNK", "32")
    got, exp, s = util.run_both(kb, orc, st, vx, vy, cfg, flags=LDS)
    assert _name(s) == XWIDE, _name(s)
    assert got.shape == exp.shape and np.array_equal(got, exp)
    assert len(got) > 100
    monkeypatch.setenv("KBMOD_CHUNK", "16")
    wide, _, s16 = util.run_both(kb, orc, st, vx, vy, cfg, flags=LDS)
    assert _name(s16) != XWIDE and np.array_equal(wide, exp)
    monkeypatch.delenv("KBMOD_CHUNK")
    direct, _, _ = util.run_both(kb, orc, st, vx, vy, cfg, flags=DIRECT)
    assert np.array_equal(direct, exp)


@pytest.mark.parametrize("n_cands", [32, 33, 63, 64, 65, 100, 250])
def test_ragged_candidate_lists(kb, orc, n_cands, monkeypatch):
    """Candidate lists that do not fill their last chunk, one chunk only, long groups: T = 96 epochs on a stack whose slabs fit
    four and more to a group buffer; irregular (non-dyadic) time stamps."""
    rng = np.random.default_rng(n_cands)
    T, H, W = 96, 40, 150
    times = np.sort(rng.random(T) * 2.0)
    times[0] = 0.0
    st = _clean_stack(T, H, W, seed=n_cands, times=times, objects=[(30, 10, 9.0, 4.0, 200.0)])
    vx = (3.0 + np.cumsum(rng.uniform(0.0, 0.25, n_cands))).astype(np.float32)
    vy = (1.0 + np.cumsum(rng.uniform(-0.05, 0.12, n_cands))).astype(np.float32)
    cfg = {"K": 8, "min_obs": 10, "min_lh": 0.0}
    monkeypatch.setenv("KBMOD_CHUNK", "32")
    got, exp, s = util.run_both(kb, orc, st, vx, vy, cfg, flags=LDS)
    assert np.array_equal(got, exp)
    # (irregular time stamps can put a shift on a rounding boundary: then the wide instances are refused and chunks of 8 run)
    if s.last_search_stats()["special_epochs"] == 0:
        assert _name(s) == XWIDE


def test_ties_stable_lists_and_the_list_floor(kb, orc, monkeypatch):
    """Slow candidates that coincide sample for sample (ties by the dozen), searched by the reference's insertion, with the lists'
    floor (flag 1024) and as stable lists (flag 512, through the compact entry point) -- the three forms the packed lists take."""
    T, H, W = 24, 64, 128
    st = _clean_stack(T, H, W, seed=5, times=np.arange(T) / 32.0)
    vx, vy = fd.kbmod_v1_candidates(16, 0.1, 6.0, 4, 0.0, 1.0)   # slow: many candidates share every sample
    monkeypatch.setenv("KBMOD_CHUNK", "32")
    for cfg in ({"K": 8}, {"K": 8, "min_lh": 1.0}, {"K": 3, "min_obs": 5}):
        got, exp, s = util.run_both(kb, orc, st, vx, vy, cfg, flags=LDS)
        assert _name(s) == XWIDE and np.array_equal(got, exp)
        floor, _, s2 = util.run_both(kb, orc, st, vx, vy, cfg, flags=LDS | 1024)
        assert _name(s2) == XWIDE and np.array_equal(floor, exp)
    ds = util.DeviceStack(st)
    cands = ds.candidates(vx, vy)
    p = ds.params(K=8)
    a, st_a = ds.search_compact(p, cands, 0, 512 | LDS)
    assert st_a.kernel_name.decode() == XWIDE
    monkeypatch.setenv("KBMOD_CHUNK", "16")
    b, st_b = ds.search_compact(p, cands, 0, 512 | LDS)
    assert st_b.kernel_name.decode() != XWIDE
    assert ds.torch.equal(a.view(ds.torch.int32), b.view(ds.torch.int32))
    c, _ = ds.search_compact(p, cands, 0, 512 | DIRECT)
    assert ds.torch.equal(a.view(ds.torch.int32), c.view(ds.torch.int32))
    ds.close()


def test_start_bounds_inside_the_image_and_partial_tiles(kb, orc, monkeypatch):
    T, H, W = 20, 90, 300
    st = _clean_stack(T, H, W, seed=9, times=np.arange(T) / 16.0, objects=[(100, 40, 10.0, 5.0, 300.0)])
    vx, vy = fd.kbmod_v1_candidates(16, 2.0, 25.0, 4, -0.2, 1.3)
    monkeypatch.setenv("KBMOD_CHUNK", "32")
    for xb, yb in (((0, 300), (0, 90)), ((17, 203), (5, 38)), ((250, 300), (80, 90)), ((0, 64), (0, 16))):
        cfg = {"K": 4, "xb": xb, "yb": yb}
        got, exp, s = util.run_both(kb, orc, st, vx, vy, cfg, flags=LDS)
        assert _name(s) == XWIDE and np.array_equal(got, exp), (xb, yb)


@pytest.mark.parametrize("case", ["masked", "off_image", "sigmag", "K16", "encoded_staging"])
def test_what_the_instance_cannot_take_falls_back(kb, orc, case, monkeypatch):
    """The instance is count-free, keeps packed lists of up to 8 and has no emit: a stack with NO_DATA pixels, start pixels off
    the image, the in-search sigma-G filter, longer lists and encoded staging all run on other instances -- same bits."""
    T, H, W = 16, 70, 130
    times = np.arange(T) / 16.0
    objects = [(20, 30, 15.0, 6.0, 300.0)]
    num_bytes, flags = -1, LDS
    st = util.make_stack(T, H, W, seed=31, objects=objects, times=times, mask_fraction=0.02 if case == "masked" else 0.0)
    cfg = {"K": 8, "min_obs": 3}
    if case == "off_image":
        cfg.update(xb=(-10, 100), yb=(-3, 50))
    if case == "sigmag":
        cfg["sigmag"] = (0.25, 0.75, 0.7413, 2.0)
    if case == "K16":
        cfg["K"] = 16
    if case == "encoded_staging":
        num_bytes, flags = 2, LDS | 16
    vx, vy = fd.kbmod_v1_candidates(16, 2.0, 25.0, 4, 0.0, 1.2)
    monkeypatch.setenv("KBMOD_CHUNK", "32")
    got, exp, s = util.run_both(kb, orc, st, vx, vy, cfg, num_bytes=num_bytes, flags=flags)
    assert _name(s) != XWIDE and "kb_search_lds" in _name(s), _name(s)
    assert got.shape == exp.shape and np.array_equal(got, exp)
    # ... and a second search of the same (masked) array does not try again
    if case == "masked":
        again, _, s2 = util.run_both(kb, orc, st, vx, vy, cfg, flags=flags)
        assert np.array_equal(again, exp) and _name(s2) != XWIDE


@pytest.mark.parametrize("case", ["epochs_out_of_order", "shifts_beyond_the_tables", "slabs_of_three_rounds"])
def test_tables_that_refuse_the_instance(kb, orc, case, monkeypatch):
    """What only the tables can tell: epochs out of time order (shifts not monotone: no edge tables), shifts of more than 200
    pixels, slabs of more than two staging rounds.  The instance has no counting loop and traps if a tile needed one -- so the
    host must refuse it here, run another instance, and still equal the oracle."""
    T, H, W = 16, 80, 160
    times = np.arange(T) / 16.0
    vel = (16, 2.0, 25.0, 4, 0.0, 1.2)
    if case == "epochs_out_of_order":
        times = times[np.random.default_rng(3).permutation(T)].copy()
        times[0], times[np.argmin(times)] = times[np.argmin(times)], times[0]   # (zeroed times start at the first epoch)
    if case == "shifts_beyond_the_tables":
        H, W, vel = 300, 400, (32, 150.0, 260.0, 1, 0.2, 0.3)
    if case == "slabs_of_three_rounds":
        vel = (32, 2.0, 60.0, 2, 0.7, 0.9)   # one chunk = the whole speed range at 45 degrees: (16 + 40) x (64 + 40) pixels
    st = util.make_stack(T, H, W, seed=41, objects=[(30, 30, 15.0, 6.0, 300.0)], times=times - times[0])
    vx, vy = fd.kbmod_v1_candidates(*vel)
    monkeypatch.setenv("KBMOD_CHUNK", "32")
    got, exp, s = util.run_both(kb, orc, st, vx, vy, {"K": 8, "min_obs": 2}, flags=LDS)
    assert _name(s) != XWIDE, _name(s)
    assert got.shape == exp.shape and np.array_equal(got, exp)
