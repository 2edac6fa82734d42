#!/usr/bin/env python3
"""The multi-GPU exchange of one job, measured piece by piece on ONE GPU (no 8-GPU node needed):

    python tools/exchange_budget.py --frames 128 --size 4096 --vel-steps 32 --ang-steps 2 --world 8 --min-lh 10

plays every rank's part one after the other on this device -- the search of its (v, theta) slice with 2 K stable records
per pixel (flag 512), kb_sparsify_compact -- keeps what each rank would put on the wire, then runs the root's merge over the
`world` lists (sparse: kb_merge_sparse_exact; dense: kb_merge_compact_exact, when --dense) and checks the merged result
against ONE search over the job-wide candidate list on the same device.  Prints one JSON object: per-rank search /
sparsify times, bytes on the wire per rank (sparse and dense), merge times, and the step and aggregate an N-GPU run would
see WITHOUT any overlap of exchange and search (xGMI at 153 GB/s per link into the root, one link per peer)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

XGMI_LINK_GBPS = 153.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--vel-steps", type=int, default=32)
    ap.add_argument("--ang-steps", type=int, default=2)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--min-lh", type=float, default=10.0)
    ap.add_argument("--results-per-pixel", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--rank-flags", type=int, default=1024,
                    help="extra kb_device_search_compact flags of the per-rank searches (1024: nothing below min_lh enters a list; 0 to compare)")
    ap.add_argument("--contiguous", action="store_true", help="contiguous angle bands per rank (default: angle rows dealt boustrophedon)")
    ap.add_argument("--single-flags", type=int, default=0,
                    help="flags of the single-GPU step the aggregate is quoted against (1024: with the list floor, what "
                         "StackSearch.search_all and bench.py pass for a thresholded search)")
    ap.add_argument("--counted", action="store_true",
                    help="per-rank searches through kb_device_search_counted (the search writes the count bytes and skips the "
                         "record runs of waves that keep nothing) + kb_sparsify_counted")
    ap.add_argument("--dense", action="store_true", help="also gather and merge the dense lists (world x S x 2K x 16 bytes on this device)")
    args = ap.parse_args()

    import torch

    import bench
    from kbmod_amd import distributed as kdist
    from kbmod_amd import fake_data as fd
    from kbmod_amd.capi import Meta, Params, Stats, load_lib

    lib = load_lib()
    dev = torch.device("cuda", 0)
    T, H, W, K, world = args.frames, args.size, args.size, args.results_per_pixel, args.world
    L = 2 * K
    S = H * W
    sci, var, times, psf = bench.synthetic_stack(torch, dev, T, H, W)
    psf_all = np.ascontiguousarray(np.tile(psf.ravel(), T), dtype=np.float32)
    psf_dims = np.full(T, psf.shape[0], dtype=np.int32)
    meta, arr = Meta(), C.c_void_p()
    stream = torch.cuda.current_stream().cuda_stream
    bench.check(lib, lib.kb_build_psi_phi_from_device_ex(sci.data_ptr(), var.data_ptr(), psf_all.ctypes.data, psf_dims.ctypes.data,
                                                         T, H, W, -1, 0, C.byref(meta), C.byref(arr), stream))
    torch.cuda.synchronize()
    del sci, var

    n_local = args.vel_steps * args.ang_steps
    vx, vy = fd.kbmod_v1_candidates(args.vel_steps, 5.0, 40.0, args.ang_steps * world, 0.0, 1.5)
    if not args.contiguous:
        # angle rows dealt boustrophedon (bench.py --gpus N): every rank gets the same mix of flat and steep trajectories
        order = np.array([j * world + (r if j % 2 == 0 else world - 1 - r) for r in range(world) for j in range(args.ang_steps)])
        vx = vx.reshape(args.ang_steps * world, args.vel_steps)[order].reshape(-1)
        vy = vy.reshape(args.ang_steps * world, args.vel_steps)[order].reshape(-1)
    all_np = np.zeros((n_local * world, 7), dtype=np.float32)
    all_np[:, 0], all_np[:, 1] = vx, vy
    all_cands = torch.from_numpy(all_np).to(dev)
    params = Params(0, args.min_lh, 0, 0.25, 0.75, -1.0, -1, 0, W, 0, H, K, 0)
    rank_params = Params.from_buffer_copy(params)
    rank_params.results_per_pixel = L

    def timed(fn, reps):
        """Median wall time of `reps` synchronised calls after one warm-up, in ms (a single hiccup does not set a rank's time)."""
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(ts))

    records = torch.empty((S * L, 4), dtype=torch.int32, device=dev)
    packed_buf = torch.empty((max(1024, S * L // 16), 4), dtype=torch.int32, device=dev)
    header_buf = torch.empty(int(lib.kb_sparse_header_bytes(S)), dtype=torch.uint8, device=dev)
    headers, packed, dense_lists = [], [], []
    search_ms, kernel_ms, sparsify_ms, wire = [], [], [], []
    for r in range(world):
        cands = all_cands[r * n_local:(r + 1) * n_local]
        st = Stats()

        counted = C.c_int32(0)

        def search():
            if args.counted:
                bench.check(lib, lib.kb_device_search_counted(C.byref(meta), arr, times.data_ptr(), rank_params, cands.data_ptr(),
                                                              n_local, r * n_local, records.data_ptr(), S * L, header_buf.data_ptr(),
                                                              512 | args.rank_flags, stream, C.byref(st), C.byref(counted)))
            else:
                bench.check(lib, lib.kb_device_search_compact(C.byref(meta), arr, times.data_ptr(), rank_params, cands.data_ptr(),
                                                              n_local, r * n_local, records.data_ptr(), S * L, 512 | args.rank_flags, stream, C.byref(st)))

        search_ms.append(timed(search, args.reps))
        kernel_ms.append(float(st.search_kernel_ms))
        out = {}

        def sparsify():
            if args.counted and counted.value:
                out["h"], out["p"], out["n"] = kdist.sparsify_counted(records, S, L, header_buf, packed_buf)
            else:
                out["h"], out["p"], out["n"] = kdist.sparsify_compact(records, S, L, args.min_lh, header_buf, packed_buf)

        sparsify_ms.append(timed(sparsify, args.reps))
        headers.append(out["h"].clone())
        packed.append(out["p"][:out["n"]].clone())
        wire.append(int(header_buf.numel()) + 16 * out["n"])
        if args.dense:
            dense_lists.append(records.clone())
    headers = torch.stack(headers)
    results = torch.empty((S * K, 7), dtype=torch.float32, device=dev)

    merge_sparse_ms = timed(lambda: kdist.merge_sparse_exact(headers, packed, (0, W), (0, H), K, L, all_cands, out=results),
                            args.reps)
    merged = results.clone()
    # ... and with the merged lists' lengths written next to them: no slot is written for a wave nothing reaches
    # (kb_merge_sparse_exact_counted: what StackSearch.search_all's filter reads through behind a multi-device search)
    merged_counts = torch.empty(S, dtype=torch.uint8, device=dev)
    results.fill_(float("nan"))
    merge_counted_ms = timed(lambda: kdist.merge_sparse_exact(headers, packed, (0, W), (0, H), K, L, all_cands, out=results,
                                                              counts_out=merged_counts), args.reps)
    merged_c = results.clone()
    merge_dense_ms = None
    if args.dense:
        gathered = torch.stack(dense_lists)
        del dense_lists
        merge_dense_ms = timed(lambda: kdist.merge_compact_exact(gathered, (0, W), (0, H), K, L, all_cands, out=results), args.reps)
        del gathered

    # one search over the job-wide list on this device: the answer the merged lists must give, and the single-GPU step
    single = torch.empty((S * K, 7), dtype=torch.float32, device=dev)
    st1 = Stats()

    def one():
        bench.check(lib, lib.kb_device_search_filter(C.byref(meta), arr, times.data_ptr(), params, all_cands.data_ptr(),
                                                     n_local * world, single.data_ptr(), S * K, 0, stream, C.byref(st1)))

    one_all_ms = timed(one, 1)
    gone = single[:, 2] < args.min_lh
    survivors = int((~gone).sum().item())
    single[gone, 0:2] = 0.0
    single[gone, 2] = torch.finfo(torch.float32).min
    single[gone, 3] = 0.0
    single.view(torch.int32)[gone, 6] = 0
    ok = bool(torch.equal(merged.view(torch.int32), single.view(torch.int32)))
    # the counted merge: the counts are the survivors per pixel, every counted slot is the single search's, the rest is either
    # the placeholder or untouched
    want_counts = (~gone).view(S, K).sum(dim=1).to(torch.uint8)
    covered = (torch.arange(K, device=dev).repeat(S) < merged_counts.to(torch.int64).repeat_interleave(K))
    untouched = torch.isnan(merged_c).all(dim=1)
    same_row = (merged_c.view(torch.int32) == single.view(torch.int32)).all(dim=1)
    ok_counted = bool(torch.equal(merged_counts, want_counts)) and bool(same_row[covered].all()) and bool((same_row | untouched).all())
    ok = ok and ok_counted
    del single, covered, untouched, same_row

    # the single-GPU step of the weak-scaling series: n_local candidates, K records per pixel, the reference's insertion
    st0 = Stats()
    res1 = torch.empty((S * K, 7), dtype=torch.float32, device=dev)

    def one_rank_plain():
        bench.check(lib, lib.kb_device_search_filter(C.byref(meta), arr, times.data_ptr(), params, all_cands.data_ptr(), n_local,
                                                     res1.data_ptr(), S * K, args.single_flags, stream, C.byref(st0)))

    single_gpu_ms = timed(one_rank_plain, args.reps)
    # ... and the same with the list floor (flag 1024): what a single GPU does for a thresholded search through StackSearch
    st2 = Stats()

    def one_rank_floor():
        bench.check(lib, lib.kb_device_search_filter(C.byref(meta), arr, times.data_ptr(), params, all_cands.data_ptr(), n_local,
                                                     res1.data_ptr(), S * K, 1024, stream, C.byref(st2)))

    single_gpu_floor_ms = timed(one_rank_floor, args.reps)

    dense_bytes = S * L * 16
    wire_ms_sparse = max(wire) / (XGMI_LINK_GBPS * 1e6)
    wire_ms_dense = dense_bytes / (XGMI_LINK_GBPS * 1e6)
    rank_ms = max(a + b for a, b in zip(search_ms, sparsify_ms))
    step_sparse = rank_ms + wire_ms_sparse + merge_sparse_ms
    out = {
        "workload": f"{T}x{H}x{W} f32, {n_local} candidates per rank x {world} ranks, K={K}, lists of {L} stable records, min_lh={args.min_lh:g}",
        "counted_search": bool(args.counted),
        "per_rank": {"search_call_ms": search_ms, "search_kernel_ms": kernel_ms, "sparsify_ms": sparsify_ms, "wire_bytes": wire,
                     "dense_wire_bytes": dense_bytes},
        "root": {"merge_sparse_ms": merge_sparse_ms, "merge_sparse_counted_ms": merge_counted_ms, "merge_dense_ms": merge_dense_ms,
                 "merge_sparse_reads_bytes": int(sum(wire)), "merge_dense_reads_bytes": dense_bytes * world,
                 "merge_writes_bytes": S * K * 28},
        "verify": {"merged_equals_single_device_after_post_filter_ok": ok, "counted_merge_ok": ok_counted, "survivors": survivors},
        "single_gpu": {"step_ms": single_gpu_ms, "kernel_ms": float(st0.search_kernel_ms), "kernel": st0.kernel_name.decode(),
                       "step_ms_with_list_floor": single_gpu_floor_ms, "job_wide_list_on_one_gpu_ms": one_all_ms},
        "predicted_no_overlap": {
            "xgmi_link_GBps": XGMI_LINK_GBPS,
            "wire_ms_sparse": wire_ms_sparse, "wire_ms_dense": wire_ms_dense,
            "step_ms_sparse": step_sparse,
            "step_ms_sparse_counted_merge": rank_ms + wire_ms_sparse + merge_counted_ms,
            "step_ms_dense": None if merge_dense_ms is None else max(search_ms) + wire_ms_dense + merge_dense_ms,
            "aggregate_vs_one_gpu_sparse": world * single_gpu_ms / step_sparse,
            "aggregate_vs_one_gpu_with_list_floor_sparse": world * single_gpu_floor_ms / step_sparse,
            "aggregate_vs_one_gpu_dense": None if merge_dense_ms is None else world * single_gpu_ms / (max(search_ms) + wire_ms_dense + merge_dense_ms),
        },
    }
    print(json.dumps(out))
    lib.kb_free_gpu_block(arr)
    sys.exit(0 if ok else 3)


if __name__ == "__main__":
    main()
