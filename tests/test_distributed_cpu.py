"""The multi-GPU path on CPU: world_size 2 over gloo.  Each rank searches its own
contiguous candidate slice (the per-rank search is played by the oracle here --
the device search itself is covered by the -m gpu suite) and packs its lists into
the 16-byte exchange records, then the product's exchange step runs unchanged:
kbmod_amd.distributed.gather_and_merge_compact = ONE gather to rank 0 + per-pixel
K-way merge.  The merged lists must equal the unsharded search wherever a pixel's
likelihoods are distinct; where equal likelihoods compete the slots carry the same
likelihoods but possibly another member of the tie (kbmod_amd/distributed.py
explains why no merge of per-rank lists can do better)."""

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stable_rank_lists(rank, world, K):
    """What every rank's device search leaves with flag 512 -- the top 2 K of ITS candidates by (likelihood descending,
    candidate ascending); here the oracle evaluates every candidate and the order is imposed with a stable sort."""
    from kbmod_amd import distributed as kdist
    from kbmod_amd import fake_data as fd
    from oracle import oracle as orc
    from tests import util

    st = util.make_stack(12, 24, 40, seed=21, objects=[(8, 6, 14.0, 9.0, 200.0)], mask_fraction=0.02)
    vx, vy = fd.kbmod_v1_candidates(7, 1.0, 40.0, 5, 0.0, 1.5)  # 35 candidates: uneven split, slow ones that coincide
    S = 24 * 40
    pp = orc.PsiPhi.from_images(st.sci, st.var, st.psfs, st.zeroed_times)
    lo, hi = kdist.shard_bounds(len(vx), rank, world)
    n_local = hi - lo
    every = pp.search_kernel_semantics(orc.make_candidates(vx[lo:hi], vy[lo:hi]),
                                       pp.default_params(results_per_pixel=n_local)).reshape(S, n_local)
    index = {(float(a), float(b)): i for i, (a, b) in enumerate(zip(vx, vy))}
    rec = np.zeros((S, 2 * K), dtype=[("lh", "<f4"), ("flux", "<f4"), ("cand", "<i4"), ("obs_count", "<i4")])
    rec["lh"], rec["cand"] = np.float32(-3.4028234663852886e38), -1
    for p in range(S):
        rows = [(r["lh"], index[(float(r["vx"]), float(r["vy"]))], r["flux"], r["obs_count"]) for r in every[p]
                if r["lh"] != np.float32(-3.4028234663852886e38)]
        rows.sort(key=lambda r: (-r[0], r[1]))
        for s, r in enumerate(rows[:2 * K]):
            rec[p, s] = (r[0], r[2], r[1], r[3])
    local_t = torch.from_numpy(rec.reshape(-1).view(np.int32).reshape(S * 2 * K, 4).copy())
    all_cands = np.zeros((len(vx), 7), dtype=np.float32)
    all_cands[:, 0], all_cands[:, 1] = vx, vy
    return pp, vx, vy, local_t, torch.from_numpy(all_cands)


def _exact_worker(rank, world, port, out_dir):
    """Tie-exact exchange: rank 0 merges the ranks' stable lists with kb_merge_compact_exact's host twin."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kbmod_amd import distributed as kdist
        from oracle import oracle as orc

        K, S = 4, 24 * 40
        pp, vx, vy, local_t, all_cands = _stable_rank_lists(rank, world, K)
        if world == 3:
            # the exchange in two halves (what bench.py --gpus N overlaps with the next search): two under way at once
            first = kdist.start_gather_compact(local_t, (0, 40), (0, 24), K, all_cands, list_len=2 * K)
            second = kdist.start_gather_compact(local_t.clone(), (0, 40), (0, 24), K, all_cands, list_len=2 * K)
            merged, again = first.finish(), second.finish()
            assert (again is None) == (rank != 0)
            if rank == 0:
                assert torch.equal(merged, again)
        else:
            merged = kdist.gather_and_merge_compact(local_t, (0, 40), (0, 24), K, all_cands, list_len=2 * K)
        assert (merged is None) == (rank != 0)
        if rank == 0:
            full = pp.search_kernel_semantics(orc.make_candidates(vx, vy), pp.default_params(results_per_pixel=K))
            np.save(os.path.join(out_dir, "got.npy"), merged.numpy().reshape(-1).view(orc.TRJ_DTYPE))
            np.save(os.path.join(out_dir, "full.npy"), full)
            more = pp.search_kernel_semantics(orc.make_candidates(vx, vy), pp.default_params(results_per_pixel=K + 1))
            np.save(os.path.join(out_dir, "more_lh.npy"), more["lh"].reshape(S, K + 1))
        dist.barrier()  # (nobody leaves -- rank 0 holds the rendezvous store -- before everybody is through)
    finally:
        dist.destroy_process_group()


def _repair_worker(rank, world, port, out_dir):
    """The exchange with K records per rank (round 6): every rank's list is the reference's insertion over ITS slice -- the
    oracle's kernel-semantics search with results_per_pixel = K --, rank 0 folds the lists in candidate order with
    kb_merge_compact_repairable's host twin and re-makes the pixels the records do not decide (the evaluator is the oracle's
    here; on devices kb_repair_pixels)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kbmod_amd import distributed as kdist
        from kbmod_amd import fake_data as fd
        from oracle import oracle as orc
        from tests import util

        K, S = 4, 24 * 40
        st = util.make_stack(12, 24, 40, seed=21, objects=[(8, 6, 14.0, 9.0, 200.0)], mask_fraction=0.02)
        vx, vy = fd.kbmod_v1_candidates(7, 1.0, 40.0, 5, 0.0, 1.5)  # 35 candidates: uneven split, slow ones that coincide
        pp = orc.PsiPhi.from_images(st.sci, st.var, st.psfs, st.zeroed_times)
        lo, hi = kdist.shard_bounds(len(vx), rank, world)
        mine = pp.search_kernel_semantics(orc.make_candidates(vx[lo:hi], vy[lo:hi]), pp.default_params(results_per_pixel=K))
        index = {(float(a), float(b)): i for i, (a, b) in enumerate(zip(vx, vy))}
        rec = np.zeros(S * K, dtype=[("lh", "<f4"), ("flux", "<f4"), ("cand", "<i4"), ("obs_count", "<i4")])
        empty = mine["lh"] == np.float32(-3.4028234663852886e38)
        rec["lh"], rec["flux"], rec["obs_count"] = mine["lh"], mine["flux"], mine["obs_count"]
        rec["cand"] = [-1 if e else index[(float(a), float(b))] for e, a, b in zip(empty, mine["vx"], mine["vy"])]
        local_t = torch.from_numpy(rec.view(np.int32).reshape(S * K, 4).copy())
        all_np = np.zeros((len(vx), 7), dtype=np.float32)
        all_np[:, 0], all_np[:, 1] = vx, vy
        all_cands = torch.from_numpy(all_np)
        params = pp.default_params(results_per_pixel=K)

        def evaluate(x, y, cvx, cvy):
            t = pp.evaluate_kernel(x, y, cvx, cvy, params)
            return float(t["lh"]), float(t["flux"]), int(t["obs_count"])

        begin = [kdist.shard_bounds(len(vx), r, world)[0] for r in range(world)] + [len(vx)]
        if world == 3:
            flight = kdist.start_gather_compact(local_t, (0, 40), (0, 24), K, all_cands, list_len=K, repair_stack=evaluate,
                                                list_begin=begin)
            merged = flight.finish()
        else:
            merged = kdist.gather_and_merge_compact(local_t, (0, 40), (0, 24), K, all_cands, list_len=K, repair_stack=evaluate,
                                                    list_begin=begin)
        assert (merged is None) == (rank != 0)
        if rank == 0:
            full = pp.search_kernel_semantics(orc.make_candidates(vx, vy), pp.default_params(results_per_pixel=K))
            np.save(os.path.join(out_dir, "got.npy"), merged.numpy().reshape(-1).view(orc.TRJ_DTYPE))
            np.save(os.path.join(out_dir, "full.npy"), full)
            np.save(os.path.join(out_dir, "hazards.npy"), np.array([kdist.last_repair()["hazards"]]))
        dist.barrier()  # (nobody leaves -- rank 0 holds the rendezvous store -- before everybody is through)
    finally:
        dist.destroy_process_group()


def _sparse_worker(rank, world, port, out_dir, min_lh):
    """The sparse form of the tie-exact exchange (kbmod_amd.distributed.gather_and_merge_sparse): one count byte per pixel
    + the records that pass min_lh, one gather of the headers, one message per rank with records, merge on rank 0."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kbmod_amd import distributed as kdist
        from oracle import oracle as orc

        K, S = 4, 24 * 40
        pp, vx, vy, local_t, all_cands = _stable_rank_lists(rank, world, K)
        stats = {}
        merged = kdist.gather_and_merge_sparse(local_t, (0, 40), (0, 24), K, 2 * K, min_lh, all_cands, stats=stats)
        assert (merged is None) == (rank != 0)
        assert stats["wire_bytes"] >= S and (stats["wire_bytes"] - (S + 15) // 16 * 16 - 16) % 16 == 0
        dense = kdist.gather_and_merge_compact(local_t, (0, 40), (0, 24), K, all_cands, list_len=2 * K)
        if rank == 0:
            assert len(stats["totals"]) == world
            full = pp.search_kernel_semantics(orc.make_candidates(vx, vy), pp.default_params(results_per_pixel=K))
            np.save(os.path.join(out_dir, "got.npy"), merged.numpy().reshape(-1).view(orc.TRJ_DTYPE))
            np.save(os.path.join(out_dir, "dense.npy"), dense.numpy().reshape(-1).view(orc.TRJ_DTYPE))
            np.save(os.path.join(out_dir, "full.npy"), full)
            np.save(os.path.join(out_dir, "wire.npy"), np.array([stats["wire_bytes"], local_t.numel() * 4]))
        dist.barrier()  # (nobody leaves -- rank 0 holds the rendezvous store -- before everybody is through)
    finally:
        dist.destroy_process_group()


def _subgroup_worker(rank, world, port, out_dir):
    """Both exchanges inside a sub-group whose ranks do not start at 0: global ranks 1 and 2 of a world of 3 search the two
    halves of the candidate list, the root is GLOBAL rank 1 (group rank 0); global rank 0 takes no part."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kbmod_amd import distributed as kdist
        from oracle import oracle as orc

        group = dist.new_group([1, 2])   # (every rank of the world calls this)
        if rank != 0:
            K = 4
            pp, vx, vy, local_t, all_cands = _stable_rank_lists(dist.get_rank(group), 2, K)
            sparse = kdist.gather_and_merge_sparse(local_t, (0, 40), (0, 24), K, 2 * K, 2.5, all_cands, group=group, dst=1)
            dense = kdist.gather_and_merge_compact(local_t, (0, 40), (0, 24), K, all_cands, list_len=2 * K, group=group, dst=1)
            assert (sparse is None) == (rank != 1) and (dense is None) == (rank != 1)
            if rank == 1:
                full = pp.search_kernel_semantics(orc.make_candidates(vx, vy), pp.default_params(results_per_pixel=K))
                np.save(os.path.join(out_dir, "sparse.npy"), sparse.numpy().reshape(-1).view(orc.TRJ_DTYPE))
                np.save(os.path.join(out_dir, "dense.npy"), dense.numpy().reshape(-1).view(orc.TRJ_DTYPE))
                np.save(os.path.join(out_dir, "full.npy"), full)
        # rank 0 hosts the rendezvous store: it stays until the sub-group is through (leaving early took the store away under
        # the other two now and then)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kbmod_amd import distributed as kdist
        from kbmod_amd import fake_data as fd
        from oracle import oracle as orc
        from tests import util

        st = util.make_stack(12, 24, 40, seed=21, objects=[(8, 6, 14.0, 9.0, 200.0)], mask_fraction=0.02)
        vx, vy = fd.kbmod_v1_candidates(7, 5.0, 40.0, 5, 0.0, 1.5)  # 35 candidates: uneven split
        K, S = 4, 24 * 40
        pp = orc.PsiPhi.from_images(st.sci, st.var, st.psfs, st.zeroed_times)
        params = pp.default_params(results_per_pixel=K)

        lo, hi = kdist.shard_bounds(len(vx), rank, world)
        local = pp.search_kernel_semantics(orc.make_candidates(vx[lo:hi], vy[lo:hi]), params)
        # pack into kb_compact_result records: the candidate grid has no duplicate (vx, vy)
        index = {(float(a), float(b)): i for i, (a, b) in enumerate(zip(vx, vy))}
        rec = np.zeros(S * K, dtype=[("lh", "<f4"), ("flux", "<f4"), ("cand", "<i4"), ("obs_count", "<i4")])
        rec["lh"], rec["flux"], rec["obs_count"] = local["lh"], local["flux"], local["obs_count"]
        empty = local["lh"] == np.float32(-3.4028234663852886e38)
        rec["cand"] = [-1 if e else index[(float(a), float(b))] for e, a, b in zip(empty, local["vx"], local["vy"])]
        local_t = torch.from_numpy(rec.view(np.int32).reshape(S * K, 4).copy())
        all_cands = np.zeros((len(vx), 7), dtype=np.float32)
        all_cands[:, 0], all_cands[:, 1] = vx, vy
        merged = kdist.gather_and_merge_compact(local_t, (0, 40), (0, 24), K, torch.from_numpy(all_cands))
        assert (merged is None) == (rank != 0)

        if rank == 0:
            full = pp.search_kernel_semantics(orc.make_candidates(vx, vy), params)
            got = merged.numpy().reshape(-1).view(orc.TRJ_DTYPE)
            np.save(os.path.join(out_dir, "got.npy"), got)
            np.save(os.path.join(out_dir, "full.npy"), full)
            # every candidate's likelihood per pixel (a list as long as the candidate list keeps them all)
            every = pp.search_kernel_semantics(orc.make_candidates(vx, vy), pp.default_params(results_per_pixel=len(vx)))
            np.save(os.path.join(out_dir, "every_lh.npy"), every["lh"].reshape(S, len(vx)))
        dist.barrier()  # (nobody leaves -- rank 0 holds the rendezvous store -- before everybody is through)
    finally:
        dist.destroy_process_group()


def test_shard_bounds():
    from kbmod_amd.distributed import shard_bounds

    for n in (0, 1, 7, 35, 1024):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gather_and_merge(tmp_path, orc, kb):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "got.npy")
    full = np.load(tmp_path / "full.npy")
    assert got.shape == full.shape
    K = 4
    g, f = got.reshape(-1, K), full.reshape(-1, K)
    # every slot carries the likelihood (and position) the unsharded search puts there
    for name in ("lh", "x", "y"):
        assert np.array_equal(g[name], f[name])
    # pixels where no two candidates share a likelihood: identical records, field for field
    every = np.load(tmp_path / "every_lh.npy")  # sorted descending per pixel
    tied = ((every[:, :-1] == every[:, 1:]) & (every[:, :-1] > np.float32(-3.0e38))).any(axis=1)
    assert (~tied).sum() > 500 and tied.sum() > 0
    for name in ("vx", "vy", "flux", "obs_count"):
        assert np.array_equal(g[name][~tied], f[name][~tied]), name
    # tied pixels (trajectories that leave the image over the same samples): same flux / obs_count per slot --
    # the competing candidates sample the same pixels -- but possibly another member of the tie
    for name in ("flux", "obs_count"):
        assert np.array_equal(g[name][tied], f[name][tied]), name


@pytest.mark.parametrize("world", [2, 3])
def test_tie_exact_gather_and_merge(tmp_path, orc, kb, world):
    mp.spawn(_exact_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "got.npy")
    full = np.load(tmp_path / "full.npy")
    assert got.tobytes() == full.tobytes()  # every field of every slot, ties included
    more = np.load(tmp_path / "more_lh.npy")
    assert ((more[:, :-1] == more[:, 1:]) & (more[:, :-1] > np.float32(-3.0e38))).any()  # and there were ties to get right


@pytest.mark.parametrize("world", [2, 3])
def test_k_record_exchange_with_repair(tmp_path, orc, kb, world):
    """The ranks' NORMAL K-record lists (no stable lists of 2 K) folded in candidate order, hazards re-made: the unsharded
    search, every field of every slot, ties included -- over gloo, world 2 and 3 (3: the exchange in two halves)."""
    mp.spawn(_repair_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "got.npy")
    full = np.load(tmp_path / "full.npy")
    assert got.tobytes() == full.tobytes()
    assert np.load(tmp_path / "hazards.npy")[0] < len(full) // 4 // 2  # most pixels are decided by the records alone


@pytest.mark.parametrize("world,min_lh", [(2, 6.0), (3, 2.5), (2, None)])
def test_sparse_gather_and_merge(tmp_path, orc, kb, world, min_lh):
    """Sparse exchange over gloo == the unsharded search followed by the reference's likelihood filter, slot for slot."""
    mp.spawn(_sparse_worker, args=(world, _free_port(), str(tmp_path), min_lh), nprocs=world, join=True)
    got = np.load(tmp_path / "got.npy").reshape(-1, 4)
    dense = np.load(tmp_path / "dense.npy").reshape(-1, 4)
    full = np.load(tmp_path / "full.npy").reshape(-1, 4)
    assert dense.tobytes() == full.tobytes()
    thr = -np.inf if min_lh is None else np.float32(min_lh)
    empty = np.float32(-3.4028234663852886e38)
    n_pass = 0
    for p in range(full.shape[0]):
        want = full[p][(full[p]["lh"] != empty) & ~(full[p]["lh"] < thr)]
        n = len(want)
        n_pass += n
        assert got[p, :n].tobytes() == want.tobytes()          # every field of every surviving slot, ties included
        assert (got[p, n:]["lh"] == empty).all() and (got[p, n:]["obs_count"] == 0).all()
        assert (got[p]["x"] == full[p]["x"]).all() and (got[p]["y"] == full[p]["y"]).all()
    wire, dense_bytes = np.load(tmp_path / "wire.npy")
    if min_lh is not None:
        assert 0 < n_pass < full.size // 2 and wire < dense_bytes // 2  # a thresholded search: fewer bytes than the dense lists
    else:
        assert n_pass > full.size // 2 and wire > dense_bytes // 2


def test_exchange_inside_a_subgroup(tmp_path, orc, kb):
    """`dst` is a global rank, the lists are ordered by group rank: a sub-group {1, 2} with root 1 must gather, receive from the
    right peers and merge exactly as the default group does."""
    mp.spawn(_subgroup_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    full = np.load(tmp_path / "full.npy").reshape(-1, 4)
    dense = np.load(tmp_path / "dense.npy").reshape(-1, 4)
    sparse = np.load(tmp_path / "sparse.npy").reshape(-1, 4)
    assert dense.tobytes() == full.tobytes()
    empty = np.float32(-3.4028234663852886e38)
    for p in range(full.shape[0]):
        want = full[p][(full[p]["lh"] != empty) & ~(full[p]["lh"] < np.float32(2.5))]
        assert sparse[p, :len(want)].tobytes() == want.tobytes()
        assert (sparse[p, len(want):]["lh"] == empty).all()
