"""One-off sweep of the sparse multi-GPU exchange on one GPU: python tools/fuzz_exchange.py FIRST LAST
Random stacks, candidate lists, K, world sizes, thresholds and partitions: per-rank searches with 2 K stable records and the
list floor -> kb_sparsify_compact -> kb_merge_sparse_exact must equal ONE search over the whole list after the post-filter
(and the dense tie-exact merge must equal that search as it is); and where the search can write the counts itself
(kb_device_search_counted), header and packed records through kb_sparsify_counted must be the same bytes."""
import sys

sys.path.insert(0, ".")
import numpy as np

import torch  # noqa: F401  (first: one HIP runtime per process)
from kbmod_amd import distributed as kdist
from tests import util

EMPTY = np.float32(-3.4028234663852886e38)
bad = []
n_counted = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    T = int(rng.integers(3, 40))
    H, W = int(rng.integers(8, 80)), int(rng.integers(8, 140))
    times = (np.arange(T) / 16.0) if rng.random() < 0.5 else np.sort(rng.random(T) * 3.0)
    times = times - times[0]
    n_c = int(rng.choice([8, 17, 40, 64, 100, 160]))
    vmax = float(rng.choice([2.0, 8.0, 25.0]))
    vx = (rng.uniform(-vmax, vmax) + np.cumsum(rng.uniform(-0.4, 0.6, n_c))).astype(np.float32)
    vy = (rng.uniform(-vmax, vmax) + np.cumsum(rng.uniform(-0.5, 0.4, n_c))).astype(np.float32)
    if rng.random() < 0.5:
        vx[: n_c // 3] = np.round(vx[: n_c // 3])  # slow, coinciding candidates: ties
        vy[: n_c // 3] = 0.0
    K = int(rng.choice([1, 2, 5, 8, 11, 16]))
    world = min(int(rng.choice([1, 2, 3, 5, 8, 12])), n_c - 1)
    min_lh = float(rng.choice([-1e30, 0.0, 1.5, 4.0, 50.0]))
    min_obs = int(rng.integers(0, T))
    st = util.make_stack(T, H, W, seed=seed, noise=float(rng.uniform(0.5, 4.0)), psf=float(rng.choice([0.5, 1.0])),
                         objects=[(int(rng.integers(0, W)), int(rng.integers(0, H)), float(vx[0]), float(vy[0]), 300.0)],
                         mask_fraction=float(rng.choice([0.0, 0.02, 0.2])), times=times)
    d = util.DeviceStack(st, int(rng.choice([-1, -1, 1, 2])))
    try:
        tt = d.torch
        all_cands = d.candidates(vx, vy)
        p, p2 = d.params(K=K, min_obs=min_obs, min_lh=min_lh), d.params(K=2 * K, min_obs=min_obs, min_lh=min_lh)
        flags = int(rng.choice([0, 2, 4, 4 | 64, 4 | 128]))
        cuts = np.sort(rng.choice(np.arange(1, n_c), world - 1, replace=False)) if world > 1 else np.array([], int)
        lo_hi = list(zip(np.concatenate([[0], cuts]), np.concatenate([cuts, [n_c]])))
        headers, packed, dense = [], [], []
        for lo, hi in lo_hi:
            rec, _ = d.search_compact(p2, all_cands[int(lo):int(hi)], int(lo), flags | 512 | 1024)
            h, pk, total = kdist.sparsify_compact(rec, H * W, 2 * K, min_lh)
            headers.append(h), packed.append(pk)
            # the same through the counted search: header and packed records must be the same bytes
            recc, hc, written, _ = d.search_counted(p2, all_cands[int(lo):int(hi)], int(lo), flags | 512 | 1024, poison=0x5a5a5a5a)
            if written:
                n_counted += 1
                pkc = tt.empty((max(1, total), 4), dtype=tt.int32, device="cuda")
                _, _, totc = kdist.sparsify_counted(recc, H * W, 2 * K, hc, pkc)
                if not (totc == total and tt.equal(hc, h) and tt.equal(pkc[:total], pk[:total])):
                    bad.append(("counted", seed))
            rec0, _ = d.search_compact(p2, all_cands[int(lo):int(hi)], int(lo), flags | 512)
            dense.append(rec0)
        merged = kdist.merge_sparse_exact(tt.stack(headers), packed, (0, W), (0, H), K, 2 * K, all_cands)
        want, _ = d.search(p, all_cands, flags)
        exact = kdist.merge_compact_exact(tt.stack(dense), (0, W), (0, H), K, 2 * K, all_cands)
        ok = tt.equal(exact.view(tt.int32), want.view(tt.int32))
        gone = want[:, 2] < min_lh
        want[gone, 0:2] = 0.0
        want[gone, 2] = float(EMPTY)
        want[gone, 3] = 0.0
        want.view(tt.int32)[gone, 6] = 0
        ok = ok and tt.equal(merged.view(tt.int32), want.view(tt.int32))
        # the counted merge: list lengths per pixel, every counted slot the single search's, the rest placeholder or untouched
        S = H * W
        counts = tt.full((S,), 0xEE, dtype=tt.uint8, device="cuda")
        out_c = tt.full((S * K, 7), float("nan"), dtype=tt.float32, device="cuda")
        kdist.merge_sparse_exact(tt.stack(headers), packed, (0, W), (0, H), K, 2 * K, all_cands, out=out_c, counts_out=counts)
        n_valid = (want[:, 2] != float(EMPTY)).view(S, K).sum(dim=1)
        covered = tt.arange(K, device="cuda").repeat(S) < n_valid.repeat_interleave(K)
        same_row = (out_c.view(tt.int32) == want.view(tt.int32)).all(dim=1)
        ok_c = tt.equal(counts.to(tt.int64), n_valid) and bool(same_row[covered].all()) and bool((same_row | tt.isnan(out_c).all(dim=1)).all())
        if not ok:
            bad.append(seed)
        if not ok_c:
            bad.append(("counted merge", seed))
    finally:
        d.close()
print("seeds", sys.argv[1], sys.argv[2], "mismatches", bad, "counted searches", n_counted)
