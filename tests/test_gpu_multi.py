"""GPU tests of the multi-GPU exchange on one device: the compact-record search, the merge kernels
against their host twins, and the sharded search (slices searched one after the other on the same GPU,
merged by kb_merge_compact) against the unsharded one."""

import numpy as np
import pytest

from kbmod_amd import fake_data as fd
from tests import util

pytestmark = pytest.mark.gpu

OBJ = [(17, 12, 21.0, 16.0, 250.0), (60, 40, -8.0, 11.0, 180.0)]
EMPTY = np.float32(-3.4028234663852886e38)


@pytest.fixture(scope="module")
def ds():
    st = util.make_stack(20, 60, 100, seed=100, noise=4.0, psf=1.0, objects=OBJ, mask_fraction=0.01)
    d = util.DeviceStack(st)
    yield d
    d.close()


@pytest.fixture(scope="module")
def grid():
    return fd.kbmod_v1_candidates(12, 5.0, 40.0, 11, 0.0, 1.5)  # 132 candidates


@pytest.mark.parametrize("flags", [2, 4])
@pytest.mark.parametrize("cfg", [dict(K=8), dict(K=5, min_obs=10, min_lh=2.0), dict(K=16),
                                 dict(K=4, min_obs=8, sigmag=(0.25, 0.75, 0.7413, 3.0))])
def test_compact_records_equal_full_results(ds, grid, cfg, flags):
    vx, vy = grid
    cands = ds.candidates(vx, vy)
    p = ds.params(**cfg)
    full, _ = ds.search(p, cands, flags)
    comp, _ = ds.search_compact(p, cands, 1000, flags)
    f, c = util.as_records(full), util.as_records(comp, util.COMPACT_DTYPE)
    assert np.array_equal(f["lh"].view(np.uint32), c["lh"].view(np.uint32))
    assert np.array_equal(f["flux"].view(np.uint32), c["flux"].view(np.uint32))
    assert np.array_equal(f["obs_count"], c["obs_count"])
    filled = f["lh"] != EMPTY
    assert filled.any() and (c["cand"][~filled] == -1).all()
    idx = c["cand"][filled] - 1000
    assert idx.min() >= 0 and idx.max() < len(vx)
    assert np.array_equal(vx[idx], f["vx"][filled]) and np.array_equal(vy[idx], f["vy"][filled])


def test_compact_search_with_sigma_g_batches(ds, grid, monkeypatch):
    vx, vy = grid
    cands = ds.candidates(vx, vy)
    p = ds.params(K=4, min_obs=8, sigmag=(0.25, 0.75, 0.7413, 3.0))
    one, st1 = ds.search_compact(p, cands, 0, 4)
    monkeypatch.setenv("KBMOD_SIGMAG_CAP", "4000")
    many, st2 = ds.search_compact(p, cands, 0, 4)
    assert st2.num_search_launches > st1.num_search_launches
    assert ds.torch.equal(one, many)


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("cfg", [dict(K=8), dict(K=4, min_obs=8, sigmag=(0.25, 0.75, 0.7413, 3.0))])
def test_sharded_search_equals_unsharded(kb, ds, grid, world, cfg):
    from kbmod_amd import distributed as kdist

    vx, vy = grid
    all_cands = ds.candidates(vx, vy)
    p = ds.params(**cfg)
    K = p.results_per_pixel
    parts = []
    for r in range(world):
        lo, hi = kdist.shard_bounds(len(vx), r, world)
        rec, _ = ds.search_compact(p, all_cands[lo:hi], lo, 0)
        parts.append(rec)
    gathered = ds.torch.stack(parts)
    merged = kdist.merge_compact(gathered, (0, ds.W), (0, ds.H), K, all_cands)
    ds.torch.cuda.synchronize()
    # device merge == host twin, bit for bit
    host = kdist.merge_compact(gathered.cpu(), (0, ds.W), (0, ds.H), K, all_cands.cpu())
    assert ds.torch.equal(merged.cpu().view(ds.torch.int32), host.view(ds.torch.int32))
    # == the unsharded search wherever no two candidates share a likelihood at the pixel
    full, _ = ds.search(p, all_cands, 0)
    g = util.as_records(merged).reshape(-1, K)
    f = util.as_records(full).reshape(-1, K)
    for name in ("lh", "x", "y"):
        assert np.array_equal(g[name], f[name]), name
    # (the K + 1 best likelihoods of the pixel are distinct: the top-K set and its order are then unique)
    every, _ = ds.search(ds.params(**{**cfg, "K": 32}), all_cands, 0)
    e = util.as_records(every).reshape(-1, 32)["lh"][:, :K + 1]
    tied = ((e[:, :-1] == e[:, 1:]) & (e[:, :-1] != EMPTY)).any(axis=1)
    assert (~tied).sum() > 1000
    for name in ("vx", "vy", "flux", "obs_count"):
        assert np.array_equal(g[name][~tied], f[name][~tied]), name


@pytest.mark.parametrize("world", [2, 5])
def test_merge_topk_kernel_equals_host_twin(kb, ds, grid, world):
    from kbmod_amd import distributed as kdist

    vx, vy = grid
    p = ds.params(K=6)
    parts = []
    for r in range(world):
        lo, hi = kdist.shard_bounds(len(vx), r, world)
        res, _ = ds.search(p, ds.candidates(vx[lo:hi], vy[lo:hi]), 0)
        parts.append(res)
    gathered = ds.torch.stack(parts)
    dev = kdist.merge_topk(gathered, ds.W * ds.H, 6)
    ds.torch.cuda.synchronize()
    host = kdist.merge_topk(gathered.cpu(), ds.W * ds.H, 6)
    assert ds.torch.equal(dev.cpu().view(ds.torch.int32), host.view(ds.torch.int32))
    full, _ = ds.search(p, ds.candidates(vx, vy), 0)
    assert np.array_equal(util.as_records(dev)["lh"], util.as_records(full)["lh"])
