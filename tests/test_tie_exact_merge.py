"""Tie-exact merge of per-device result lists (kb_merge_compact_exact; host twin here, device kernel in
tests/test_gpu_multi.py).  The claim (csrc/search_kernels.hip): per-device lists of 2K entries
built by STABLE insertion, merged under (likelihood descending, candidate ascending) and replayed over the
candidates above the K-th value plus the first K equal to it, reproduce the reference's sequential swap-down
insertion over the whole candidate list -- for any partition of the candidates.  Checked by brute force on
likelihoods drawn from a handful of levels (ties everywhere)."""

import numpy as np
import pytest

EMPTY_LH = np.float32(-3.4028234663852886e38)
REC = np.dtype([("lh", "<f4"), ("flux", "<f4"), ("cand", "<i4"), ("obs", "<i4")])


def swap_down(seq, K):
    """kernels.cu:323-330 over (lh, cand) pairs in list order."""
    slots = [(EMPTY_LH, -1)] * K
    for item in seq:
        cur = item
        for s in range(K):
            if cur[0] > slots[s][0]:
                cur, slots[s] = slots[s], cur
    return slots


def stable_top(seq, n):
    """What a device's search leaves with flag 512: the top n by (lh descending, candidate ascending)."""
    slots = [(EMPTY_LH, -1)] * n
    for item in seq:
        cur, placed = item, False
        for s in range(n):
            if placed or cur[0] > slots[s][0]:
                cur, slots[s] = slots[s], cur
                placed = True
    return slots


@pytest.fixture(scope="module")
def kb():
    import kbmod_amd.search as kb

    return kb


@pytest.mark.parametrize("K", [1, 2, 3, 8, 16])
def test_merge_reproduces_sequential_insertion(kb, K):
    rng = np.random.default_rng(100 + K)
    n_pixels, n_cands = 300, 70
    K2 = 2 * K
    for n_lists, interleaved, levels in [(1, False, 3), (2, False, 2), (3, True, 4), (8, False, 3), (8, True, 60), (5, False, 1)]:
        lh = rng.integers(0, levels + 1, (n_pixels, n_cands)).astype(np.float32)
        lh[rng.random((n_pixels, n_cands)) < 0.1] = -1.0  # the value of a trajectory without data
        keep = rng.random((n_pixels, n_cands)) < 0.9       # candidates the thresholds drop never reach a list
        if interleaved:
            owner = np.arange(n_cands) % n_lists
        else:
            cuts = np.sort(rng.choice(np.arange(1, n_cands), n_lists - 1, replace=False)) if n_lists > 1 else []
            owner = np.searchsorted(cuts, np.arange(n_cands), side="right")
        lists = np.zeros((n_lists, n_pixels, K2), dtype=REC)
        truth = []
        for p in range(n_pixels):
            seq = [(lh[p, c], c) for c in range(n_cands) if keep[p, c]]
            truth.append(swap_down(seq, K))
            for r in range(n_lists):
                top = stable_top([it for it in seq if owner[it[1]] == r], K2)
                lists[r, p]["lh"] = [t[0] for t in top]
                lists[r, p]["cand"] = [t[1] for t in top]
                lists[r, p]["flux"] = [0.5 * t[1] for t in top]
                lists[r, p]["obs"] = [t[1] + 1 if t[1] >= 0 else 0 for t in top]
        cands = [kb.Trajectory(vx=float(c), vy=float(-c)) for c in range(n_cands)]
        raw = np.ascontiguousarray(lists).view(np.uint8).reshape(-1)
        out = kb.merge_compact_exact_host(raw, n_lists, K2, K, 0, n_pixels, 0, 1, cands)
        out = out.view(np.dtype([("vx", "<f4"), ("vy", "<f4"), ("lh", "<f4"), ("flux", "<f4"), ("x", "<i4"), ("y", "<i4"),
                                 ("obs", "<i4")])).reshape(n_pixels, K)
        for p in range(n_pixels):
            for s in range(K):
                t_lh, t_c = truth[p][s]
                got = out[p, s]
                assert got["x"] == p and got["y"] == 0
                if t_c < 0:
                    assert got["lh"] == EMPTY_LH and got["obs"] == 0 and got["vx"] == 0.0
                else:
                    assert (got["lh"], got["vx"], got["vy"], got["flux"], got["obs"]) == (t_lh, t_c, -t_c, np.float32(0.5 * t_c), t_c + 1), \
                        (K, n_lists, interleaved, p, s, truth[p], out[p])


def test_merge_argument_checks(kb):
    cands = [kb.Trajectory()]
    with pytest.raises(RuntimeError):
        kb.merge_compact_exact_host(np.zeros(16 * 4, np.uint8), 1, 4, 5, 0, 1, 0, 1, cands)  # K > list length
    with pytest.raises(RuntimeError):
        kb.merge_compact_exact_host(np.zeros(16 * 3, np.uint8), 1, 4, 2, 0, 1, 0, 1, cands)  # wrong buffer size
    with pytest.raises(RuntimeError):
        kb.merge_compact_exact_host(np.zeros(16 * 40, np.uint8), 1, 40, 2, 0, 1, 0, 1, cands)  # list length > 32


TRJ = np.dtype([("vx", "<f4"), ("vy", "<f4"), ("lh", "<f4"), ("flux", "<f4"), ("x", "<i4"), ("y", "<i4"), ("obs", "<i4")])


@pytest.mark.parametrize("K", [1, 3, 8, 16])
def test_sparse_exchange_equals_filtered_sequential_insertion(kb, K):
    """The sparse form of the exchange (kb_sparsify_compact + kb_merge_sparse_exact; host twins here): records below min_lh
    are dropped BEFORE the merge.  Claim (csrc/exchange_kernels.hip): what survives the reference's post-filter
    (stack_search.cpp:266-270, lh < min_lh removed) of the sequential swap-down insertion over ALL candidates is, slot
    for slot, what the sparse merge leaves -- ties at, above and below the threshold included; and the sparse merge
    equals the dense tie-exact merge wherever the dense result passes the filter."""
    rng = np.random.default_rng(700 + K)
    n_pixels, n_cands = 260, 60
    K2 = 2 * K
    for n_lists, interleaved, levels, min_lh in [(1, False, 3, 2.0), (2, False, 2, 1.0), (3, True, 4, 3.0), (8, False, 3, 0.0),
                                                 (8, True, 40, 20.0), (5, False, 1, 1.0), (4, True, 3, 9.0), (4, False, 3, None)]:
        lh = rng.integers(0, levels + 1, (n_pixels, n_cands)).astype(np.float32)
        lh[rng.random((n_pixels, n_cands)) < 0.1] = -1.0
        keep = rng.random((n_pixels, n_cands)) < 0.9
        keep[: n_pixels // 8] = False                       # pixels nothing reaches: the common case of a thresholded search
        if interleaved:
            owner = np.arange(n_cands) % n_lists
        else:
            cuts = np.sort(rng.choice(np.arange(1, n_cands), n_lists - 1, replace=False)) if n_lists > 1 else []
            owner = np.searchsorted(cuts, np.arange(n_cands), side="right")
        lists = np.zeros((n_lists, n_pixels, K2), dtype=REC)
        truth = []
        for p in range(n_pixels):
            seq = [(lh[p, c], c) for c in range(n_cands) if keep[p, c]]
            full = swap_down(seq, K)
            thr = -np.inf if min_lh is None else min_lh
            truth.append([t for t in full if t[1] >= 0 and not (t[0] < thr)])
            for r in range(n_lists):
                top = stable_top([it for it in seq if owner[it[1]] == r], K2)
                lists[r, p]["lh"] = [t[0] for t in top]
                lists[r, p]["cand"] = [t[1] for t in top]
                lists[r, p]["flux"] = [0.5 * t[1] for t in top]
                lists[r, p]["obs"] = [t[1] + 1 if t[1] >= 0 else 0 for t in top]
        cands = [kb.Trajectory(vx=float(c), vy=float(-c)) for c in range(n_cands)]
        hb = kb.sparse_header_bytes(n_pixels)
        headers = np.zeros((n_lists, hb), np.uint8)
        packed = []
        for r in range(n_lists):
            h, pk = kb.sparsify_compact_host(np.ascontiguousarray(lists[r]).view(np.uint8).reshape(-1), n_pixels, K2,
                                             float("-inf") if min_lh is None else min_lh)
            headers[r] = h
            recs = pk.view(REC)
            assert int(h[(n_pixels + 15) // 16 * 16:].view(np.uint64)[0]) == len(recs) == int(h[:n_pixels].sum())
            thr = -np.inf if min_lh is None else min_lh
            want = lists[r][(lists[r]["cand"] >= 0) & ~(lists[r]["lh"] < thr)]
            assert recs.tobytes() == want.tobytes()           # kept records, pixel after pixel, list order
            packed.append(pk)
        out = kb.merge_sparse_exact_host(headers.reshape(-1), hb, packed, K2, K, 0, n_pixels, 0, 1, cands).view(TRJ).reshape(n_pixels, K)
        dense = kb.merge_compact_exact_host(np.ascontiguousarray(lists).view(np.uint8).reshape(-1), n_lists, K2, K, 0, n_pixels,
                                            0, 1, cands).view(TRJ).reshape(n_pixels, K)
        for p in range(n_pixels):
            assert (out[p]["x"] == p).all() and (out[p]["y"] == 0).all()
            got = [(o["lh"], int(o["vx"])) for o in out[p] if o["lh"] != EMPTY_LH]
            assert got == truth[p], (K, n_lists, min_lh, p, got, truth[p])
            n = len(got)
            assert (out[p, n:]["lh"] == EMPTY_LH).all() and (out[p, n:]["obs"] == 0).all() and (out[p, n:]["vx"] == 0).all()
            thr = -np.inf if min_lh is None else min_lh
            passing = dense[p][(dense[p]["lh"] != EMPTY_LH) & ~(dense[p]["lh"] < thr)]
            assert out[p, :n].tobytes() == passing.tobytes()    # field for field what the dense exchange + post-filter gives


def test_sparse_argument_checks(kb):
    cands = [kb.Trajectory()]
    with pytest.raises(RuntimeError):
        kb.sparsify_compact_host(np.zeros(16 * 3, np.uint8), 1, 4, 0.0)        # wrong buffer size
    with pytest.raises(RuntimeError):
        kb.sparsify_compact_host(np.zeros(16 * 40, np.uint8), 1, 40, 0.0)      # list length > 32
    h = np.zeros(kb.sparse_header_bytes(1), np.uint8)
    h[0] = 3
    with pytest.raises(RuntimeError):
        kb.merge_sparse_exact_host(h, len(h), [np.zeros(16 * 2, np.uint8)], 4, 2, 0, 1, 0, 1, cands)  # counts say 3, 2 records
    with pytest.raises(RuntimeError):
        kb.merge_sparse_exact_host(h, len(h) - 16, [np.zeros(16 * 3, np.uint8)], 4, 2, 0, 1, 0, 1, cands)  # stride too short


def test_counted_forms_are_device_only():
    """The counted forms of the exchange (the search wrote the counts; the merge leaves list lengths) exist on the device only:
    CPU tensors are refused with a message instead of being handed to the C ABI."""
    import torch

    from kbmod_amd import distributed as kdist

    S, L, K = 4, 4, 2
    records = torch.zeros((S * L, 4), dtype=torch.int32)
    header = torch.zeros(32, dtype=torch.uint8)
    with pytest.raises(ValueError, match="device tensor"):
        kdist.sparsify_counted(records, S, L, header, torch.zeros((8, 4), dtype=torch.int32))
    headers = torch.zeros((1, 32), dtype=torch.uint8)
    cands = torch.zeros((3, 7), dtype=torch.float32)
    with pytest.raises(ValueError, match="host twin"):
        kdist.merge_sparse_exact(headers, [torch.zeros((0, 4), dtype=torch.int32)], (0, 2), (0, 2), K, L, cands,
                                 counts_out=torch.zeros(S, dtype=torch.uint8))


def test_exact_merges_refuse_lists_shorter_than_2k_minus_1(kb):
    """The tie-exact merge needs the first 2 K - 1 entries of every list (search_math.h); shorter lists are refused, not merged
    approximately."""
    cands = [kb.Trajectory(vx=1.0, vy=1.0)]
    K, L = 4, 6
    raw = np.zeros(1 * 1 * L * 16, dtype=np.uint8)
    with pytest.raises(RuntimeError, match="2 K - 1"):
        kb.merge_compact_exact_host(raw, 1, L, K, 0, 1, 0, 1, cands)
    header = np.zeros(kb.sparse_header_bytes(1), dtype=np.uint8)
    with pytest.raises(RuntimeError, match="2 K - 1"):
        kb.merge_sparse_exact_host(header, len(header), [np.zeros(0, dtype=np.uint8)], L, K, 0, 1, 0, 1, cands)
