#!/usr/bin/env python3
"""The dense exchange with K records per rank + repair (round 6), measured piece by piece on ONE GPU:

    python tools/exchange_repair_budget.py --frames 64 --size 512 --vel-steps 32 --ang-steps 4 --world 8

plays every rank's part one after the other on this device -- the search of its (v, theta) slice with its NORMAL K-record
lists (the reference's insertion: the fastest single-GPU instance, no flag 512) --, keeps the dense lists each rank would
put on the wire, runs the root's kb_merge_compact_repairable + kb_repair_pixels, and checks the result against ONE search
over the job-wide candidate list on the same device, bit for bit.  Next to it the same job through the 2 K stable lists of
kb_merge_compact_exact (what rounds 3-5 ran).  Prints one JSON object: per-rank search times of both forms, bytes on the
wire, merge / repair times, hazards, and the step and aggregate an N-GPU run would see WITHOUT overlap of exchange and search
(xGMI at 153 GB/s per link into the root, one link per peer)."""
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
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--vel-steps", type=int, default=32)
    ap.add_argument("--ang-steps", type=int, default=4, help="angle rows per rank")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--results-per-pixel", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--contiguous", action="store_true", help="contiguous angle bands per rank (default: angle rows dealt boustrophedon)")
    ap.add_argument("--mask-fraction", type=float, default=0.0)
    args = ap.parse_args()

    import torch

    import bench
    from kbmod_amd import distributed as kdist
    from kbmod_amd import fake_data as fd
    from kbmod_amd.capi import Meta, Params, Stats, load_lib

    lib = load_lib()
    dev = torch.device("cuda", 0)
    T, H, W, K, world = args.frames, args.size, args.size, args.results_per_pixel, args.world
    S = H * W
    sci, var, times, psf = bench.synthetic_stack(torch, dev, T, H, W, args.mask_fraction)
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
        order = np.array([j * world + (r if j % 2 == 0 else world - 1 - r) for r in range(world) for j in range(args.ang_steps)])
        vx = vx.reshape(args.ang_steps * world, args.vel_steps)[order].reshape(-1)
        vy = vy.reshape(args.ang_steps * world, args.vel_steps)[order].reshape(-1)
    all_np = np.zeros((n_local * world, 7), dtype=np.float32)
    all_np[:, 0], all_np[:, 1] = vx, vy
    all_cands = torch.from_numpy(all_np).to(dev)
    params = Params(0, 0.0, 0, 0.25, 0.75, -1.0, -1, 0, W, 0, H, K, 0)

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(ts))

    def rank_lists(list_len, flags):
        p = Params.from_buffer_copy(params)
        p.results_per_pixel = list_len
        lists, call_ms, kern_ms, names = [], [], [], set()
        rec = torch.empty((S * list_len, 4), dtype=torch.int32, device=dev)
        for r in range(world):
            cands = all_cands[r * n_local:(r + 1) * n_local]
            st = Stats()

            def search():
                bench.check(lib, lib.kb_device_search_compact(C.byref(meta), arr, times.data_ptr(), p, cands.data_ptr(), n_local,
                                                              r * n_local, rec.data_ptr(), S * list_len, flags, stream, C.byref(st)))

            call_ms.append(timed(search, args.reps))
            kern_ms.append(float(st.search_kernel_ms))
            names.add(st.kernel_name.decode())
            lists.append(rec.clone())
        return torch.stack(lists), call_ms, kern_ms, sorted(names)

    results = torch.empty((S * K, 7), dtype=torch.float32, device=dev)
    # ---- K records, repaired ----
    gathered, k_call, k_kern, k_names = rank_lists(K, 0)
    stack = (meta, arr.value, times.data_ptr())
    hazard_buf = torch.empty(S, dtype=torch.int32, device=dev)
    begin = [r * n_local for r in range(world + 1)]
    merge_repair_ms = timed(lambda: kdist.merge_compact_repair(gathered, (0, W), (0, H), K, all_cands, stack, out=results,
                                                               hazard_buf=hazard_buf, list_begin=begin), args.reps)
    hazards = kdist.last_repair()["hazards"]
    merged_k = results.clone()
    # (the merge alone: a stack that is never asked for -- only legal when there is no hazard -- or the timing split below)
    n_h = C.c_uint64(0)
    p_k = Params.from_buffer_copy(params)

    def merge_only():
        bench.check(lib, lib.kb_merge_compact_repairable(gathered.data_ptr(), world, p_k, all_cands.data_ptr(), n_local * world,
                                                         results.data_ptr(), hazard_buf.data_ptr(), C.byref(n_h), stream))

    merge_only_ms = timed(merge_only, args.reps)
    del gathered
    # ---- 2 K stable records (rounds 3-5) ----
    gathered2, s_call, s_kern, s_names = rank_lists(2 * K, 512)
    merge_exact_ms = timed(lambda: kdist.merge_compact_exact(gathered2, (0, W), (0, H), K, 2 * K, all_cands, out=results), args.reps)
    merged_2k = results.clone()
    del gathered2

    # one search over the job-wide list on this device: the answer, and the single-GPU step of the weak-scaling series
    single = torch.empty((S * K, 7), dtype=torch.float32, device=dev)
    st1 = Stats()
    bench.check(lib, lib.kb_device_search_filter(C.byref(meta), arr, times.data_ptr(), params, all_cands.data_ptr(), n_local * world,
                                                 single.data_ptr(), S * K, 0, stream, C.byref(st1)))
    torch.cuda.synchronize()
    ok_k = bool(torch.equal(merged_k.view(torch.int32), single.view(torch.int32)))
    ok_2k = bool(torch.equal(merged_2k.view(torch.int32), single.view(torch.int32)))
    st0 = Stats()
    res1 = torch.empty((S * K, 7), dtype=torch.float32, device=dev)

    def one_rank_plain():
        bench.check(lib, lib.kb_device_search_filter(C.byref(meta), arr, times.data_ptr(), params, all_cands.data_ptr(), n_local,
                                                     res1.data_ptr(), S * K, 0, stream, C.byref(st0)))

    single_gpu_ms = timed(one_rank_plain, args.reps)
    wire_k = S * K * 16
    wire_ms_k = wire_k / (XGMI_LINK_GBPS * 1e6)
    step_k = max(k_call) + wire_ms_k + merge_repair_ms
    step_2k = max(s_call) + 2 * wire_ms_k + merge_exact_ms
    out = {
        "workload": f"{T}x{H}x{W} f32, {n_local} candidates per rank x {world} ranks, K={K}, no threshold (dense exchange)"
                    + (f", {args.mask_fraction:g} masked" if args.mask_fraction else ""),
        "k_records_repaired": {"per_rank_search_call_ms": k_call, "per_rank_search_kernel_ms": k_kern, "kernels": k_names,
                               "wire_bytes_per_rank": wire_k, "merge_ms": merge_only_ms, "merge_plus_repair_ms": merge_repair_ms,
                               "hazard_pixels": hazards, "hazard_fraction": hazards / S,
                               "merged_equals_single_device_ok": ok_k},
        "two_k_stable": {"per_rank_search_call_ms": s_call, "per_rank_search_kernel_ms": s_kern, "kernels": s_names,
                         "wire_bytes_per_rank": 2 * wire_k, "merge_ms": merge_exact_ms, "merged_equals_single_device_ok": ok_2k},
        "single_gpu": {"step_ms": single_gpu_ms, "kernel_ms": float(st0.search_kernel_ms), "kernel": st0.kernel_name.decode()},
        "predicted_no_overlap": {"xgmi_link_GBps": XGMI_LINK_GBPS, "step_ms_k_records": step_k, "step_ms_two_k": step_2k,
                                 "aggregate_vs_one_gpu_k_records": world * single_gpu_ms / step_k,
                                 "aggregate_vs_one_gpu_two_k": world * single_gpu_ms / step_2k,
                                 "aggregate_k_records_exchange_overlapped": world * single_gpu_ms / max(max(k_call), wire_ms_k + merge_repair_ms),
                                 "aggregate_two_k_exchange_overlapped": world * single_gpu_ms / max(max(s_call), 2 * wire_ms_k + merge_exact_ms)},
    }
    print(json.dumps(out))
    lib.kb_free_gpu_block(arr)
    sys.exit(0 if (ok_k and ok_2k) else 3)


if __name__ == "__main__":
    main()
