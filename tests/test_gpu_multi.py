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
def ds_dyadic():
    """Time stamps i / 16: every velocity x time product is exact in double precision, the shift table proves every epoch
    uniform (half pixels included) and the search takes its wide-chunk instances -- `ds` (i / 20) has rounding-boundary
    epochs, which only the chunk-of-8 instances handle."""
    st = util.make_stack(16, 60, 100, seed=101, noise=4.0, psf=1.0, objects=OBJ, mask_fraction=0.01)
    d = util.DeviceStack(st)
    yield d
    d.close()


@pytest.fixture(scope="module")
def grid_dense():
    # 128 candidates whose chunks of 16 (one speed, sixteen angles) spread over few pixels: staged as wide chunks
    return fd.kbmod_v1_candidates(8, 5.0, 20.0, 16, 0.0, 0.5)


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


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("cfg", [dict(K=8), dict(K=3, min_obs=5, min_lh=1.0), dict(K=16),
                                 dict(K=4, min_obs=8, sigmag=(0.25, 0.75, 0.7413, 3.0))])
def test_tie_exact_sharded_search_equals_unsharded(kb, ds, grid, world, cfg):
    """Every slice searched with 2 K stable lists (flag 512), merged by kb_merge_compact_exact: the result IS the
    unsharded search, field for field, at every pixel -- including the border pixels whose trajectories leave the
    image over the same samples and tie (the plain merge above only promises their likelihoods)."""
    from kbmod_amd import distributed as kdist

    vx, vy = grid
    all_cands = ds.candidates(vx, vy)
    p = ds.params(**cfg)
    K = p.results_per_pixel
    p2 = ds.params(**{**cfg, "K": 2 * K})
    parts = []
    for r in range(world):
        lo, hi = kdist.shard_bounds(len(vx), r, world)
        rec, _ = ds.search_compact(p2, all_cands[lo:hi], lo, 512)
        parts.append(rec)
    gathered = ds.torch.stack(parts)
    merged = kdist.merge_compact_exact(gathered, (0, ds.W), (0, ds.H), K, 2 * K, all_cands)
    ds.torch.cuda.synchronize()
    host = kdist.merge_compact_exact(gathered.cpu(), (0, ds.W), (0, ds.H), K, 2 * K, all_cands.cpu())
    assert ds.torch.equal(merged.cpu().view(ds.torch.int32), host.view(ds.torch.int32))  # kernel == host twin
    full, _ = ds.search(p, all_cands, 0)
    assert ds.torch.equal(merged.view(ds.torch.int32), full.view(ds.torch.int32))
    # the test means something: some pixels do hold ties among their K + 1 best (where the plain merge may differ)
    every, _ = ds.search(ds.params(**{**cfg, "K": 32}), all_cands, 0)
    e = util.as_records(every).reshape(-1, 32)["lh"][:, :K + 1]
    assert ((e[:, :-1] == e[:, 1:]) & (e[:, :-1] != EMPTY)).any()


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("cfg", [dict(K=8), dict(K=3, min_obs=5), dict(K=16), dict(K=1)])
@pytest.mark.parametrize("num_bytes", [-1, 2])
def test_k_record_exchange_with_repair_equals_unsharded(kb, ds, grid, world, cfg, num_bytes):
    """The exchange with K records per rank (round 6): every slice searched with its NORMAL lists -- the reference's
    insertion, no flag 512 --, merged by kb_merge_compact_repairable, the pixels the records do not decide re-made by
    kb_repair_pixels: the result IS the unsharded search, field for field, at every pixel (ties included: the border pixels
    of this stack tie among their K + 1 best, see the test above).  The merge kernel names the same hazards as its host twin."""
    from kbmod_amd import distributed as kdist

    d = ds if num_bytes == -1 else util.DeviceStack(util.make_stack(20, 60, 100, seed=100, noise=4.0, psf=1.0, objects=OBJ,
                                                                    mask_fraction=0.01), num_bytes)
    try:
        vx, vy = grid
        all_cands = d.candidates(vx, vy)
        p = d.params(**cfg)
        K = p.results_per_pixel
        parts = []
        for r in range(world):
            lo, hi = kdist.shard_bounds(len(vx), r, world)
            rec, _ = d.search_compact(p, all_cands[lo:hi], lo, 0)
            parts.append(rec)
        gathered = d.torch.stack(parts)
        stack = (d.meta, d.arr.value, d.times.data_ptr())
        merged = kdist.merge_compact_repair(gathered, (0, d.W), (0, d.H), K, all_cands, stack, min_obs=p.min_observations)
        d.torch.cuda.synchronize()
        hazards = kdist.last_repair()["hazards"]
        full, _ = d.search(p, all_cands, 0)
        assert d.torch.equal(merged.view(d.torch.int32), full.view(d.torch.int32))
        # ... and with the repair told which candidates each list covers (only the suspect slices are evaluated again)
        begin = [kdist.shard_bounds(len(vx), r, world)[0] for r in range(world)] + [len(vx)]
        again = kdist.merge_compact_repair(gathered, (0, d.W), (0, d.H), K, all_cands, stack, min_obs=p.min_observations,
                                           list_begin=begin)
        d.torch.cuda.synchronize()
        assert kdist.last_repair()["hazards"] == hazards
        assert d.torch.equal(again.view(d.torch.int32), full.view(d.torch.int32))
        # the host twin names the same pixels, and agrees wherever the lists decide
        raw = np.ascontiguousarray(gathered.cpu().numpy()).view(np.uint8).reshape(-1)
        cands = [kb.Trajectory(vx=float(a), vy=float(b)) for a, b in zip(vx, vy)]
        host, hz = kb.merge_compact_repairable_host(raw, world, K, 0, d.W, 0, d.H, cands)
        assert len(hz) == hazards
        decided = np.ones(d.W * d.H, dtype=bool)
        decided[np.asarray(hz, dtype=np.int64)] = False
        got = merged.cpu().numpy().view(np.int32).reshape(d.W * d.H, K * 7)
        want = np.asarray(host).view(np.int32).reshape(d.W * d.H, K * 7)
        assert np.array_equal(got[decided], want[decided])
        if world == 1:
            assert hazards == 0  # one list IS the answer
    finally:
        if d is not ds:
            d.close()


def test_repair_pixels_alone_is_the_search(ds, grid):
    """kb_repair_pixels over EVERY pixel of a sub-area == kb_device_search_filter (the batched evaluator has the bits of
    evaluate_trajectory_full and the wave-ordered insertion is the reference's)."""
    import ctypes as C

    from kbmod_amd import capi

    vx, vy = grid
    all_cands = ds.candidates(vx, vy)
    for cfg in (dict(K=8), dict(K=5, min_obs=12), dict(K=32)):
        p = ds.params(**cfg, xb=(-3, 70), yb=(2, 40))
        K = p.results_per_pixel
        S = 73 * 38
        full, _ = ds.search(p, all_cands, 0)
        out = ds.torch.full((S * K, 7), float("nan"), dtype=ds.torch.float32, device="cuda")
        pixels = ds.torch.arange(S, dtype=ds.torch.int32, device="cuda")
        capi.check(ds.lib.kb_repair_pixels(C.byref(ds.meta), ds.arr, ds.times.data_ptr(), p, all_cands.data_ptr(), len(vx),
                                           pixels.data_ptr(), S, None, 0, None, out.data_ptr(), ds.stream))
        ds.torch.cuda.synchronize()
        assert ds.torch.equal(out.view(ds.torch.int32), full.view(ds.torch.int32)), cfg


def test_stable_lists_are_the_top_by_likelihood_then_candidate(ds, grid):
    """Flag 512 alone: the per-pixel list is the first K of the candidates ordered by (likelihood descending, index
    ascending) -- checked against a list long enough to hold every candidate."""
    vx, vy = grid
    cands = ds.candidates(vx, vy)
    K = 8
    for flags in (512 | 2, 512 | 4):
        got, _ = ds.search_compact(ds.params(K=K), cands, 0, flags)
        g = util.as_records(got, util.COMPACT_DTYPE).reshape(-1, K)
        every, _ = ds.search_compact(ds.params(K=32), cands, 0, flags)  # 132 candidates > 32: compare values only ...
        e = util.as_records(every, util.COMPACT_DTYPE).reshape(-1, 32)
        assert np.array_equal(g["lh"], e["lh"][:, :K]) and np.array_equal(g["cand"], e["cand"][:, :K])
        filled = g["cand"] >= 0
        lh_a, lh_b, c_a, c_b = g["lh"][:, :-1], g["lh"][:, 1:], g["cand"][:, :-1], g["cand"][:, 1:]
        both = filled[:, :-1] & filled[:, 1:]
        assert ((lh_a > lh_b) | ((lh_a == lh_b) & (c_a < c_b)))[both].all()
        assert ((lh_a == lh_b) & both).any()  # ties are present in this stack


@pytest.mark.parametrize("K,min_obs,n_cands", [(9, 0, 128), (12, 14, 121), (16, 16, 128), (16, 0, 39)])
def test_pooled_stable_lists_equal_the_stored_ones(ds_dyadic, grid_dense, K, min_obs, n_cands, monkeypatch):
    """Stable lists of 9 to 16 (what the tie-exact exchange asks every device for): the wide-chunk instance with the pooled
    list store (kb_search_lds<16, 16, ..., 4>) against the chunk-of-8 instances with (likelihood, candidate) lists and
    against the direct kernel -- every record, bit for bit, ties included."""
    ds = ds_dyadic
    vx, vy = grid_dense
    cands = ds.candidates(vx[:n_cands], vy[:n_cands])  # (121, 39: a last chunk that is not full)
    p = ds.params(K=K, min_obs=min_obs)
    got, st = ds.search_compact(p, cands, 0, 512 | 4)
    assert "kb_search_lds<16, 16," in st.kernel_name.decode() and st.kernel_name.decode().rstrip(">").endswith(" 4")
    monkeypatch.setenv("KBMOD_CHUNK", "8")
    narrow, st8 = ds.search_compact(p, cands, 0, 512 | 4)
    assert "kb_search_lds<16, 8," in st8.kernel_name.decode()
    direct, _ = ds.search_compact(p, cands, 0, 512 | 2)
    assert ds.torch.equal(got.view(ds.torch.int32), narrow.view(ds.torch.int32))
    assert ds.torch.equal(got.view(ds.torch.int32), direct.view(ds.torch.int32))
    g = util.as_records(got, util.COMPACT_DTYPE).reshape(-1, K)
    assert (g["cand"] >= 0).all(axis=1).any()
    if min_obs > 0:  # (trajectories that leave the image or cross masked pixels fall short: lists with empty slots)
        assert (g["cand"] < 0).any()
    lh_a, lh_b = g["lh"][:, :-1], g["lh"][:, 1:]
    assert ((lh_a == lh_b) & (g["cand"][:, 1:] >= 0)).any()  # ties are present


@pytest.mark.parametrize("world", [2, 8])
def test_tie_exact_exchange_on_the_wide_chunk_instances(ds_dyadic, grid_dense, world):
    """The tie-exact exchange as bench.py --gpus N runs it for K = 8 -- every slice through the pooled 16-slot instance -- ==
    the unsharded search (packed register lists, reference insertion), every field at every pixel."""
    from kbmod_amd import distributed as kdist

    ds = ds_dyadic
    vx, vy = grid_dense
    all_cands = ds.candidates(vx, vy)
    p, p2 = ds.params(K=8), ds.params(K=16)
    parts = []
    for r in range(world):
        lo, hi = kdist.shard_bounds(len(vx), r, world)
        rec, st = ds.search_compact(p2, all_cands[lo:hi], lo, 512)
        assert "kb_search_lds<16, 16," in st.kernel_name.decode()
        parts.append(rec)
    merged = kdist.merge_compact_exact(ds.torch.stack(parts), (0, ds.W), (0, ds.H), 8, 16, all_cands)
    full, st = ds.search(p, all_cands, 0)
    assert "kb_search_lds<8, 16," in st.kernel_name.decode()
    assert ds.torch.equal(merged.view(ds.torch.int32), full.view(ds.torch.int32))


def test_two_ranks_on_one_gpu_through_gloo(tmp_path):
    """bench.py --gpus 2 end to end on this box's single GPU: two processes, the gloo backend carrying the records
    through host memory (KBMOD_DIST_BACKEND; RCCL refuses two ranks on one device), tie-exact merge on rank 0,
    verified against one search over the job-wide candidate list."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KBMOD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    # (256 candidates per rank: from there on the dense exchange is K records per rank + repair; below, 2 K stable lists)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames", "16",
           "--size", "128", "--vel-steps", "16", "--ang-steps", "16", "--verify", "--no-cpu-baseline"]
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    line = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["verify"] == {"merged_equals_single_device_ok": True, "merged_likelihoods_equal_ok": True, "tie_exact": True,
                              "backend": "gloo", "exchange": "dense", "world": 2}
    assert line["config"]["candidates_per_gpu"] == 256
    assert line["exchange"]["lists"] == "K records + repair" and line["exchange"]["repair"]["pixels"] == 128 * 128


def _bench_line(extra_args, env_extra, timeout=900):
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--frames", "16", "--size", "128",
           "--vel-steps", "8", "--ang-steps", "4", "--verify", "--no-cpu-baseline", "--no-live-traffic", "--no-masked"] + extra_args
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert run.returncode == 0, run.stderr[-3000:]
    return json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1])


def test_two_ranks_on_one_gpu_sparse_exchange_through_gloo():
    """bench.py --gpus 2 with a likelihood threshold: the sparse exchange (count byte per pixel + surviving records, one
    gather of headers + one message per rank) end to end as two processes, merged == the single-device search after the
    reference's post-filter."""
    line = _bench_line(["--gpus", "2", "--min-lh", "6"], {"KBMOD_DIST_BACKEND": "gloo"})
    assert line["n_gpus"] == 2
    assert line["verify"] == {"merged_equals_single_device_ok": True, "merged_likelihoods_equal_ok": True, "tie_exact": True,
                              "backend": "gloo", "exchange": "sparse", "world": 2}
    ex = line["exchange"]
    assert ex["form"] == "sparse" and ex["wire_bytes_per_rank"] < ex["dense_bytes_per_rank"] // 8
    assert 0 < line["verify_survivors"] < 128 * 128 * 8 // 4 and sum(ex["records_per_rank"]) >= line["verify_survivors"]


@pytest.mark.parametrize("extra", [[], ["--no-overlap"], ["--min-lh", "6"], ["--sigmag", "--num-bytes", "1"]])
def test_rccl_backend_at_world_size_one(extra):
    """The N > 1 branch of bench.py on the REAL backend (nccl = RCCL) with the one GPU there is: KBMOD_FORCE_DIST=1 makes a
    world of one rank initialise the process group on the device and run compact search -> dist.gather of device tensors
    (async, finished behind the next search) / sparse exchange -> merge kernel on the RCCL-ordered stream -> --verify."""
    line = _bench_line(["--gpus", "1"] + extra, {"KBMOD_FORCE_DIST": "1"})
    sparse = "--min-lh" in extra or "--sigmag" in extra
    assert line["n_gpus"] == 1
    assert line["verify"] == {"merged_equals_single_device_ok": True, "merged_likelihoods_equal_ok": True, "tie_exact": True,
                              "backend": "nccl", "exchange": "sparse" if sparse else "dense", "world": 1}
    assert line["exchange"]["backend"] == "nccl" and line["exchange"]["form"] == ("sparse" if sparse else "dense")
    assert line["exchange"]["overlapped"] == (not sparse and "--no-overlap" not in extra)


@pytest.mark.parametrize("world", [1, 3, 8, 11])  # (11: the merge instance for more than eight lists)
@pytest.mark.parametrize("cfg", [dict(K=8, min_lh=5.0), dict(K=3, min_obs=5, min_lh=1.0), dict(K=16, min_lh=8.0),
                                 dict(K=4, min_obs=8, sigmag=(0.25, 0.75, 0.7413, 3.0)), dict(K=8, min_lh=1.0e9), dict(K=5)])
def test_sparse_exchange_kernels(kb, ds, grid, world, cfg):
    """kb_sparsify_compact + kb_merge_sparse_exact on the device == their host twins, byte for byte; == the dense tie-exact
    merge and == the unsharded search wherever a slot survives the reference's post-filter (lh >= min_lh), placeholders
    elsewhere."""
    from kbmod_amd import distributed as kdist

    torch = ds.torch
    vx, vy = grid
    all_cands = ds.candidates(vx, vy)
    p = ds.params(**cfg)
    K = p.results_per_pixel
    S = ds.W * ds.H
    min_lh = float(p.min_lh) if ("min_lh" in cfg or "sigmag" in cfg) else None
    p2 = ds.params(**{**cfg, "K": 2 * K})
    headers, packed, dense = [], [], []
    for r in range(world):
        lo, hi = kdist.shard_bounds(len(vx), r, world)
        rec, _ = ds.search_compact(p2, all_cands[lo:hi], lo, 512)
        h, pk, total = kdist.sparsify_compact(rec, S, 2 * K, min_lh)
        hh, hp, htotal = kdist.sparsify_compact(rec.cpu(), S, 2 * K, min_lh)
        assert total == htotal == pk.shape[0] and torch.equal(h.cpu(), hh) and torch.equal(pk.cpu(), hp)
        # a preallocated buffer that is too small: an error that names the count, and the header is complete
        if total > 1:
            with pytest.raises(RuntimeError, match=f"{total} records kept, room for 1"):
                kdist.sparsify_compact(rec, S, 2 * K, min_lh, packed=torch.empty((1, 4), dtype=torch.int32, device="cuda"))
        headers.append(h)
        packed.append(pk)
        dense.append(rec)
    headers = torch.stack(headers)
    merged = kdist.merge_sparse_exact(headers, packed, (0, ds.W), (0, ds.H), K, 2 * K, all_cands)
    torch.cuda.synchronize()
    host = kdist.merge_sparse_exact(headers.cpu(), [q.cpu() for q in packed], (0, ds.W), (0, ds.H), K, 2 * K, all_cands.cpu())
    assert torch.equal(merged.cpu().view(torch.int32), host.view(torch.int32))
    want = kdist.merge_compact_exact(torch.stack(dense), (0, ds.W), (0, ds.H), K, 2 * K, all_cands)
    full, _ = ds.search(p, all_cands, 0)
    assert torch.equal(want.view(torch.int32), full.view(torch.int32))
    if min_lh is not None:
        gone = want[:, 2] < min_lh
        want[gone, 0:2] = 0.0
        want[gone, 2] = float(EMPTY)
        want[gone, 3] = 0.0
        want.view(torch.int32)[gone, 6] = 0
        if cfg.get("min_lh", 0) < 1.0e8:
            assert 0 < int((~gone).sum()) < want.shape[0]
        else:
            assert int((want[:, 2] != float(EMPTY)).sum()) == 0  # nothing survives: every count is zero, every slot a placeholder
    assert torch.equal(merged.view(torch.int32), want.view(torch.int32))
    # the counted form: the merged lists' lengths next to them, no slot written for a wave nothing reaches -- and the filter
    # that reads through the counts returns what the filter over every slot returns
    import ctypes as C

    from kbmod_amd import capi

    counts = torch.full((S,), 0xEE, dtype=torch.uint8, device="cuda")
    out_c = torch.full((S * K, 7), float("nan"), dtype=torch.float32, device="cuda")
    kdist.merge_sparse_exact(headers, packed, (0, ds.W), (0, ds.H), K, 2 * K, all_cands, out=out_c, counts_out=counts)
    n_valid = (want[:, 2] != float(EMPTY)).view(S, K).sum(dim=1)
    assert torch.equal(counts.to(torch.int64), n_valid)
    covered = torch.arange(K, device="cuda").repeat(S) < n_valid.repeat_interleave(K)
    same_row = (out_c.view(torch.int32) == want.view(torch.int32)).all(dim=1)
    assert bool(same_row[covered].all()) and bool((same_row | torch.isnan(out_c).all(dim=1)).all())
    with pytest.raises(ValueError, match="host twin"):
        kdist.merge_sparse_exact(headers.cpu(), [q.cpu() for q in packed], (0, ds.W), (0, ds.H), K, 2 * K, all_cands.cpu(),
                                 counts_out=counts.cpu())
    if min_lh is not None and min_lh > -1e30:
        lib = ds.lib
        res = [torch.empty((S * K, 7), dtype=torch.float32, device="cuda") for _ in range(2)]
        cnt, bad = [C.c_uint64(0), C.c_uint64(0)], [C.c_int64(-2), C.c_int64(-2)]
        lib.kb_filter_sort_results_checked.argtypes = [C.c_void_p, C.c_uint64, C.c_float, C.c_int32, C.c_void_p, C.POINTER(C.c_uint64),
                                                       C.POINTER(C.c_int64), C.c_void_p]
        capi.check(lib.kb_filter_sort_results_checked(want.data_ptr(), S * K, min_lh, p.min_observations, res[0].data_ptr(),
                                                      C.byref(cnt[0]), C.byref(bad[0]), None))
        capi.check(lib.kb_filter_sort_results_counted(out_c.data_ptr(), S, K, counts.data_ptr(), min_lh, p.min_observations,
                                                      res[1].data_ptr(), C.byref(cnt[1]), C.byref(bad[1]), None))
        assert cnt[0].value == cnt[1].value and bad[0].value == bad[1].value == -1
        assert torch.equal(res[0][:cnt[0].value].view(torch.int32), res[1][:cnt[1].value].view(torch.int32))


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


def _mask_below(torch, t, min_lh):
    """Result tensor [S*K, 7] with every slot below min_lh (and every empty one) turned into the empty-slot placeholder."""
    t = t.clone()
    gone = t[:, 2] < min_lh
    t[gone, 0:2] = 0.0
    t[gone, 2] = float(EMPTY)
    t[gone, 3] = 0.0
    t.view(torch.int32)[gone, 6] = 0
    return t, int((~gone).sum())


@pytest.mark.parametrize("stable", [0, 512])
@pytest.mark.parametrize("flags", [2, 4, 4 | 64, 4 | 128])
@pytest.mark.parametrize("cfg", [dict(K=8, min_lh=5.0), dict(K=16, min_lh=3.0), dict(K=4, min_obs=10, min_lh=6.0),
                                 dict(K=32, min_lh=4.0), dict(K=8, min_lh=-2.0)])
@pytest.mark.parametrize("which", ["chunks_of_8", "wide_chunks"])
def test_list_floor_flag_keeps_every_survivor(ds, ds_dyadic, grid, grid_dense, which, cfg, flags, stable):
    """Flag 1024 (nothing below min_lh need enter a list: the caller post-filters like stack_search.cpp:266-270): every slot
    at or above min_lh is bit for bit the slot of the default search, in every kernel and list form, with the reference's
    insertion and with stable lists."""
    d, (vx, vy) = (ds, grid) if which == "chunks_of_8" else (ds_dyadic, grid_dense)
    torch = d.torch
    cands = d.candidates(vx, vy)
    p = d.params(**cfg)
    plain, st0 = d.search(p, cands, flags | stable)
    floor, st1 = d.search(p, cands, flags | stable | 1024)
    assert st0.kernel_name == st1.kernel_name
    a, n_a = _mask_below(torch, plain, cfg["min_lh"])
    b, n_b = _mask_below(torch, floor, cfg["min_lh"])
    assert n_a == n_b and 0 < n_a
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    if cfg["min_lh"] > 0:
        # ... and the lists really are spared: hardly anything below the floor is left in them (only what the approximate
        # screen cannot decide)
        r = util.as_records(floor)
        below = (r["lh"] != EMPTY) & (r["lh"] < np.float32(cfg["min_lh"]))
        assert below.sum() <= max(4, n_a // 50)


@pytest.mark.parametrize("flags", [2, 4, 4 | 64, 4 | 128, 4 | 512, 4 | 512 | 1024, 4 | 1024])
@pytest.mark.parametrize("cfg", [dict(K=8, min_lh=5.0), dict(K=16, min_lh=3.0), dict(K=4, min_obs=10, min_lh=6.0),
                                 dict(K=32, min_lh=4.0), dict(K=8, min_lh=-2.0), dict(K=8, min_lh=1e9),
                                 dict(K=4, min_obs=8, sigmag=(0.25, 0.75, 0.7413, 3.0)), dict(K=8, sigmag=(0.25, 0.75, 0.7413, 6.0))])
@pytest.mark.parametrize("which", ["chunks_of_8", "wide_chunks"])
def test_counted_search_writes_the_sparse_header(ds, ds_dyadic, grid, grid_dense, which, cfg, flags):
    """kb_device_search_counted: the count bytes the search writes are the ones kb_sparsify_compact counts from the full
    record lists, every record a count covers is the record of kb_device_search_compact, records of waves that keep nothing
    are NOT written (the buffer keeps its fill), and kb_sparsify_counted packs the same bytes.  Where the kernel instance
    cannot write counts, it says so and the records are all there."""
    from kbmod_amd import distributed as kdist

    d, (vx, vy) = (ds, grid) if which == "chunks_of_8" else (ds_dyadic, grid_dense)
    torch = d.torch
    cands = d.candidates(vx, vy)
    p = d.params(**cfg)
    K, S = cfg["K"], d.H * d.W
    want, st0 = d.search_compact(p, cands, 7, flags)
    h_want, pk_want, total = kdist.sparsify_compact(want, S, K, float(p.min_lh))
    POISON = 0x5a5a5a5a
    got, header, written, st1 = d.search_counted(p, cands, 7, flags, poison=POISON)
    assert st0.kernel_name == st1.kernel_name
    if not written:
        assert torch.equal(got, want)
        return
    assert torch.equal(header[:S], h_want[:S])  # (the padding and the total behind the counts are kb_sparsify_counted's)
    counts = header[:S].to(torch.int64)
    slot = torch.arange(K, device=got.device).repeat(S)
    covered = slot < counts.repeat_interleave(K)
    assert torch.equal(got[covered], want[covered])
    # a pixel's uncovered records are either the search's own (its wave kept something) or untouched
    untouched = (got == POISON).all(dim=1)
    assert bool(((got == want).all(dim=1) | untouched).all())
    assert not bool((untouched & covered).any())
    if total == 0:
        assert bool(untouched.all())
    packed = torch.empty((max(1, total), 4), dtype=torch.int32, device=got.device)
    _, _, n = kdist.sparsify_counted(got, S, K, header, packed)
    assert n == total and torch.equal(packed[:total], pk_want[:total]) and torch.equal(header, h_want)


def test_counted_sigmag_search_in_batches(ds_dyadic, grid_dense, monkeypatch):
    """The in-search sigma-G filter with its work-item store capped so that the candidates go in several batches: only the
    LAST batch's select kernel may leave rows unwritten (the lists of the others are read back slot by slot), and the counts
    are those of the finished lists."""
    from kbmod_amd import distributed as kdist

    d, (vx, vy) = ds_dyadic, grid_dense
    torch = d.torch
    monkeypatch.setenv("KBMOD_SIGMAG_CAP", "4000")
    cands = d.candidates(vx, vy)
    p = d.params(K=8, sigmag=(0.25, 0.75, 0.7413, 6.0))
    S = d.H * d.W
    want, st0 = d.search_compact(p, cands, 0, 4)
    assert st0.num_search_launches > 3  # (three launches per batch)
    h_want, pk_want, total = kdist.sparsify_compact(want, S, 8, float(p.min_lh))
    got, header, written, st1 = d.search_counted(p, cands, 0, 4, poison=0x5a5a5a5a)
    assert written == 1 and st1.num_search_launches == st0.num_search_launches and 0 < total < S * 8
    assert torch.equal(header[:S], h_want[:S])
    packed = torch.empty((total, 4), dtype=torch.int32, device=got.device)
    _, _, n = kdist.sparsify_counted(got, S, 8, header, packed)
    assert n == total and torch.equal(packed, pk_want[:total]) and torch.equal(header, h_want)
    # what the counts do not cover is either untouched or the search's own record (an earlier batch has written the buffer
    # the last one finishes in: placeholders where nothing had passed yet)
    covered = torch.arange(8, device=got.device).repeat(S) < header[:S].to(torch.int64).repeat_interleave(8)
    untouched = (got == 0x5a5a5a5a).all(dim=1)
    assert bool(((got == want).all(dim=1) | untouched)[~covered].all()) and not bool((untouched & covered).any())


@pytest.mark.parametrize("flags", [2, 4, 4 | 64, 4 | 128, 4 | 1024, 4 | 512 | 1024])
@pytest.mark.parametrize("cfg", [dict(K=8, min_lh=5.0), dict(K=8, min_lh=5.0, min_obs=30), dict(K=16, min_lh=3.0),
                                 dict(K=4, min_obs=10, min_lh=6.0), dict(K=8, min_lh=-2.0), dict(K=8, min_lh=1e9),
                                 dict(K=1, min_lh=0.5), dict(K=32, min_lh=4.0),
                                 dict(K=4, min_obs=8, sigmag=(0.25, 0.75, 0.7413, 3.0)), dict(K=8, sigmag=(0.25, 0.75, 0.7413, 6.0))])
@pytest.mark.parametrize("which", ["chunks_of_8", "wide_chunks"])
def test_counted_filter_sort_equals_the_filter_over_every_record(ds, ds_dyadic, grid, grid_dense, which, cfg, flags):
    """kb_device_search_filter_counted + kb_filter_sort_results_counted (what StackSearch.search_all runs for a search with a
    likelihood threshold): the filtered, sorted trajectories are bit for bit those of kb_device_search_filter +
    kb_filter_sort_results_checked, although the record runs of waves that keep nothing were never written (the buffer is
    filled with NaN patterns first: a record read by mistake would surface as an invalid trajectory or a mismatch)."""
    import ctypes as C

    from kbmod_amd import capi

    d, (vx, vy) = (ds, grid) if which == "chunks_of_8" else (ds_dyadic, grid_dense)
    torch, lib = d.torch, d.lib
    cands = d.candidates(vx, vy)
    p = d.params(**cfg)
    K, S = cfg["K"], d.H * d.W
    n = S * K
    want_raw, st0 = d.search(p, cands, flags)
    out0 = torch.empty((n, 7), dtype=torch.float32, device="cuda")
    cnt0, bad0 = C.c_uint64(0), C.c_int64(-2)
    lib.kb_filter_sort_results_checked.argtypes = [C.c_void_p, C.c_uint64, C.c_float, C.c_int32, C.c_void_p, C.POINTER(C.c_uint64),
                                                   C.POINTER(C.c_int64), C.c_void_p]
    capi.check(lib.kb_filter_sort_results_checked(want_raw.data_ptr(), n, p.min_lh, p.min_observations, out0.data_ptr(),
                                                  C.byref(cnt0), C.byref(bad0), None))
    got_raw = torch.full((n, 7), float("nan"), dtype=torch.float32, device="cuda")
    counts = torch.full((S,), 0xEE, dtype=torch.uint8, device="cuda")
    st1, written = capi.Stats(), C.c_int32(-1)
    capi.check(lib.kb_device_search_filter_counted(C.byref(d.meta), d.arr, d.times.data_ptr(), p, cands.data_ptr(), cands.shape[0],
                                                   got_raw.data_ptr(), n, counts.data_ptr(), flags, d.stream, C.byref(st1),
                                                   C.byref(written)))
    torch.cuda.synchronize()
    assert st0.kernel_name == st1.kernel_name
    if not written.value:
        assert torch.equal(got_raw.view(torch.int32), want_raw.view(torch.int32))
        return
    out1 = torch.full((n, 7), float("nan"), dtype=torch.float32, device="cuda")
    cnt1, bad1 = C.c_uint64(0), C.c_int64(-2)
    capi.check(lib.kb_filter_sort_results_counted(got_raw.data_ptr(), S, K, counts.data_ptr(), p.min_lh, p.min_observations,
                                                  out1.data_ptr(), C.byref(cnt1), C.byref(bad1), None))
    assert cnt1.value == cnt0.value and bad1.value == bad0.value == -1
    k = int(cnt0.value)
    assert torch.equal(out1[:k].view(torch.int32), out0[:k].view(torch.int32))
    if cfg.get("min_lh", 0) >= 1e9:
        assert k == 0 and bool(torch.isnan(got_raw).all())
    with pytest.raises(RuntimeError, match="above -FLT_MAX"):
        capi.check(lib.kb_filter_sort_results_counted(got_raw.data_ptr(), S, K, counts.data_ptr(), float("-inf"), 0, out1.data_ptr(),
                                                      C.byref(cnt1), C.byref(bad1), None))


def test_exchange_budget_tool_at_reduced_size():
    """tools/exchange_budget.py (the 8-rank exchange measured piece by piece on one GPU; DESIGN.md section 5's table) on a stack
    small enough for the suite: every rank's sparse lists merged == one search over the job-wide list after the post-filter,
    sparse bytes on the wire far below dense, and the pieces it reports are all there."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, os.path.join(root, "tools", "exchange_budget.py"), "--frames", "32", "--size", "512",
                          "--vel-steps", "16", "--ang-steps", "2", "--world", "4", "--min-lh", "8", "--reps", "1", "--dense"],
                         capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    d = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["verify"]["merged_equals_single_device_after_post_filter_ok"] is True and d["verify"]["survivors"] > 0
    assert max(d["per_rank"]["wire_bytes"]) < d["per_rank"]["dense_wire_bytes"] // 8
    assert len(d["per_rank"]["search_call_ms"]) == 4 and d["root"]["merge_sparse_ms"] > 0 and d["root"]["merge_dense_ms"] > 0
    assert d["predicted_no_overlap"]["aggregate_vs_one_gpu_sparse"] > 1.0
