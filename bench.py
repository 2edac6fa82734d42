#!/usr/bin/env python3
"""Headline benchmark: shift-and-stack trajectory search on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): trajectory-epoch evaluations / second (+ achieved GB/s of
algorithmic bytes against the HBM roofline) on a 64-frame 512x512 float32 stack,
full per-pixel start grid x 1024 (v, theta) candidates, sigma-G off (configs[1]).

A "step" is one pass of the hot path -- kb_device_search_filter through the C
ABI of libkbmod_hip.so -- over one batch of synthetic input that is already
resident in HBM (the psi/phi array is built on the device by the HIP builder
before the timed region).  torch is plumbing only: device memory, the stream,
and torch.distributed (RCCL) for the multi-GPU gather.

Multi-GPU (N > 1): one process per GPU (started by the driver's torch.distributed.run
command, or by this script itself when WORLD_SIZE is not set); the candidate list is
sharded into N contiguous (v, theta) slices (weak scaling: every rank searches the full
start grid over its own 1024 candidates, so per-GPU work is fixed), psi/phi is
replicated, every rank leaves 16-byte records per slot (kb_device_search_compact) and
ONE RCCL gather to rank 0 followed by a per-pixel K-way merge there (kb_merge_compact)
produces the job's result lists.

roofline: the 134 MB psi/phi array of the headline configuration lives in L2 / Infinity
Cache and the sums read LDS, so the report names the binding resource it measured --
LDS read bytes against the aggregate LDS rate for kb_search_lds -- next to the
algorithmic rate, the nominal and the measured HBM peak (a device copy
kernel timed in this run) and, where a rocprofv3 PMC profile of the same configuration
exists under profiles/, the fabric-side bytes per launch.
"""

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBPS = 6300.0  # what a float4 streaming copy reaches on this part (same guide; measured again in every run)
LDS_PEAK_GBPS = 157000.0      # 256 B / clk / CU x 256 CUs x 2.4 GHz for ds_read_b64 (same guide, LDS table)
INFINITY_CACHE_BYTES = 256 << 20

from kbmod_amd.capi import Meta, Params, Stats, load_lib  # noqa: E402


def check(lib, rc):
    if rc != 0:
        raise RuntimeError(lib.kb_last_error().decode())


def synthetic_stack(torch, dev, T, H, W, mask_fraction=0.0):
    """The benchmark's synthetic image stack, generated on the device: the distributions of
    fake_data.make_fake_image_stack (sci ~ N(0, 2^2), var = 4, Gaussian PSF sigma = 1, epochs i / T days) with ten
    injected movers.  Seeded: identical on every rank (psi/phi is replicated) and in the tests that re-create it."""
    from kbmod_amd import fake_data as fd

    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    sci = torch.randn((T, H, W), generator=gen, device=dev, dtype=torch.float32) * 2.0
    var = torch.full((T, H, W), 4.0, device=dev, dtype=torch.float32)
    times = torch.arange(T, dtype=torch.float64, device=dev) / T
    psf = fd.make_gaussian_kernel(1.0)
    # ~10 injected movers (x, y, vx, vy, flux)
    obj_rng = np.random.default_rng(99)
    tcpu = times.cpu().numpy()
    for _ in range(10):
        x0, y0 = obj_rng.integers(20, W - 60), obj_rng.integers(20, H - 60)
        v, ang = obj_rng.uniform(8, 35), obj_rng.uniform(0.1, 1.3)
        for t in range(T):
            px = int(x0 + v * np.cos(ang) * tcpu[t] + 0.5)
            py = int(y0 + v * np.sin(ang) * tcpu[t] + 0.5)
            r = psf.shape[0] // 2
            if r <= px < W - r and r <= py < H - r:
                sci[t, py - r:py + r + 1, px - r:px + r + 1] += torch.from_numpy(300.0 * psf).to(dev)
    if mask_fraction > 0.0:
        sci[torch.rand((T, H, W), generator=gen, device=dev) < mask_fraction] = float("nan")
    return sci, var, times, psf


def self_launch(args_list, n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the
    same command the driver would issue) and pass their output through."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + args_list
    return subprocess.call(cmd, env=env)


def roofline_block(*, kernel_ms, algorithmic_bytes, lds_read_bytes, evals_per_step, traffic, traffic_source, traffic_rejected,
                   compulsory_bytes, padded_copy_bytes_guess, is_lds_kernel, kernel, edge_count_tables, mask_fraction,
                   padded_copy_reused, env_overrides, copy_gbps, read_gbps, lds_gbps, psi_phi_bytes, traced=None):
    """The `roofline` object of the line, from plain measured values (no device, no library: tests/test_bench_contract.py
    calls this with stubbed numbers).  What bounds the dominant kernel, and every fraction with its yardstick and its clock
    (every rate divides by kernel_ms: the HIP-event duration of the search launch on its own stream, averaged over the run's
    un-profiled timed steps):
      cache-resident arrays (the float copy fits the 256 MiB Infinity Cache: BASELINE configs[1], [2]) -- DRAM sees the array
        once; the fabric bytes are Infinity-Cache hits; what the kernel is bound by is the rate at which the sums read LDS
        and instructions issue.  SURVEY 8(d)'s "8 B per evaluation" ARE those LDS reads (one ds_read_b64 per sample):
        achieved = algorithmic bytes / kernel_ms against the LDS read peak of the guide; frac_algorithmic, the same bytes
        against the 8 TB/s of HBM, is above 1 for that reason and is not a roofline fraction.
      arrays beyond the Infinity Cache (configs[3], [4]) -- bound "hbm": achieved = fabric bytes / kernel_ms against 8 TB/s.
    `traced` (live_traffic): what the run's own rocprofv3 sub-runs saw of the same kernel instance ON THIS BOX -- its mean
    dispatch duration under the tracer (traced_kernel_ms), LDS instructions (lds_insts: one ds_read_b64 per 64 evaluations,
    so lds_insts x 64 ~ evals per launch) and LDS-array cycles (lds_cycles) per launch: with them `frac` and the
    one-read-per-64-evaluations identity can be recomputed from the line alone."""
    k_s = kernel_ms * 1e-3
    alg_rate = float(algorithmic_bytes) / k_s / 1e9
    cache_resident = padded_copy_bytes_guess <= INFINITY_CACHE_BYTES
    lds_rate = float(lds_read_bytes) / k_s / 1e9
    fabric_rate = None if traffic is None else traffic / k_s / 1e9
    note = []
    if cache_resident and is_lds_kernel:
        bound, achieved, peak = "lds+issue", lds_rate, LDS_PEAK_GBPS
        note.append(f"the array's float copy ({padded_copy_bytes_guess >> 20} MiB) lives in the 256 MiB Infinity Cache: fabric bytes are cache "
                    "hits, DRAM traffic is about compulsory_bytes; the kernel is bound by LDS reads and instruction issue.  SURVEY "
                    "8(d)'s 8 B per evaluation are LDS bytes here (one ds_read_b64 per sample): achieved = those bytes / kernel_ms, "
                    "peak = the guide's LDS read rate; frac_of_measured_lds uses the ds_read_b64 rate measured in this run")
    else:
        bound, peak = "hbm", HBM_PEAK_GBPS
        if fabric_rate is None:
            # (the algorithmic rate of SURVEY 8(d) is no HBM rate -- staged pixels are reused out of LDS, it exceeds the peak
            # several times over; it stays in the line as algorithmic_GBps)
            achieved = compulsory_bytes / k_s / 1e9
            note.append("no fabric-byte measurement for this workload and kernel instance: achieved = compulsory_bytes / kernel_ms, "
                        "a LOWER bound of the HBM rate (what any implementation must move: the array once + the result slots)")
        else:
            achieved = fabric_rate
            note.append("achieved = fabric bytes per launch (traffic) / kernel_ms")
    note.append("clock of every rate: kernel_ms = HIP events around the search launch, un-profiled timed region of this run; "
                "traffic = bytes per launch from PMC passes (byte counts do not depend on the profiler's slowdown)")
    return {
        "bound": bound,
        "achieved": achieved,
        "peak": peak,
        "unit": "GB/s",
        "frac": achieved / peak,
        "traffic": traffic,
        "traffic_source": traffic_source,
        "traffic_rejected": traffic_rejected,
        "compulsory_bytes": compulsory_bytes,
        "traffic_over_compulsory": None if traffic is None else traffic / compulsory_bytes,
        "note": "; ".join(note),
        "kernel": kernel,
        "kernel_ms": kernel_ms,
        "obs_counts": ("border tiles from tables of epochs per shift (the stack has no masked pixel), none needed elsewhere"
                       if edge_count_tables and mask_fraction == 0.0 else "counted per sample where NO_DATA can occur"),
        "padded_copy": ("kept from the first search of this array (the library built the array and nothing has written into it): "
                        "the decode-and-pad pass is OUTSIDE the timed steps, see first_search"
                        if padded_copy_reused else "made by this search (decode-and-pad pass inside the step)"),
        "env_overrides": env_overrides,   # KBMOD_* switches set in this process (0: none; include/kbmod_hip.h)
        "kernel_evals_per_s": evals_per_step / k_s,
        "frac_algorithmic": alg_rate / HBM_PEAK_GBPS,
        "algorithmic_bytes_per_launch": int(algorithmic_bytes),
        "algorithmic_GBps": alg_rate,
        "fabric_GBps": fabric_rate,
        "fabric_frac_of_hbm_peak": None if fabric_rate is None else fabric_rate / HBM_PEAK_GBPS,
        "fabric_frac_of_achievable": None if fabric_rate is None else fabric_rate / HBM_ACHIEVABLE_GBPS,
        "hbm_peak_GBps": HBM_PEAK_GBPS,
        "hbm_achievable_GBps": HBM_ACHIEVABLE_GBPS,
        "hbm_measured_copy_GBps": float(copy_gbps) or None,
        "hbm_measured_read_GBps": float(read_gbps) or None,
        "lds_read_bytes_per_launch": int(lds_read_bytes),
        "lds_read_GBps": lds_rate,
        "lds_guide_peak_GBps": LDS_PEAK_GBPS,
        "lds_measured_peak_GBps": float(lds_gbps) or None,
        "frac_of_guide_lds": None if not lds_read_bytes else lds_rate / LDS_PEAK_GBPS,
        "frac_of_measured_lds": None if not (lds_read_bytes and lds_gbps) else lds_rate / float(lds_gbps),
        "psi_phi_bytes": int(psi_phi_bytes),
        "cache_resident": bool(cache_resident),
        "traced_kernel_ms": (traced or {}).get("traced_kernel_ms"),
        "traced_launches": (traced or {}).get("traced_launches"),
        "frac_at_traced_kernel_ms": (None if not (traced or {}).get("traced_kernel_ms")
                                     else achieved * kernel_ms / traced["traced_kernel_ms"] / peak),
        "lds_insts": (traced or {}).get("lds_insts"),
        "lds_cycles": (traced or {}).get("lds_cycles"),
        "evals_per_lds_inst": (None if not (traced or {}).get("lds_insts") else evals_per_step / traced["lds_insts"]),
        "lds_cycles_per_inst": (None if not ((traced or {}).get("lds_insts") and (traced or {}).get("lds_cycles"))
                                else traced["lds_cycles"] / traced["lds_insts"]),
    }


def headline_line(*, total_evals, elapsed_s, world, steps, warmup, dtype, config, roofline):
    """The driver's contract fields around `config` and `roofline` (pure: tests/test_bench_contract.py)."""
    return {
        "metric": "trajectory-epoch evals/sec",
        "value": total_evals / elapsed_s,
        "unit": "evals/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": elapsed_s / steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": dtype,
        "data": "synthetic",
        "config": config,
        "roofline": roofline,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--vel-steps", type=int, default=32)
    ap.add_argument("--ang-steps", type=int, default=32)
    ap.add_argument("--num-bytes", type=int, default=-1, choices=[-1, 1, 2, 4])
    ap.add_argument("--min-vel", type=float, default=5.0)
    ap.add_argument("--max-vel", type=float, default=40.0)
    ap.add_argument("--min-ang", type=float, default=0.0)
    ap.add_argument("--max-ang", type=float, default=1.5)
    ap.add_argument("--results-per-pixel", type=int, default=8, help="K (BASELINE: 8); other values for kernel work only")
    ap.add_argument("--inset", type=int, default=0, help="shrink the start-pixel grid by this many pixels per side")
    ap.add_argument("--flags", type=int, default=0, help="kb_device_search_filter flags (1 exact positions, 4 LDS-staged kernel)")
    ap.add_argument("--sigmag", action="store_true",
                    help="BASELINE configs[2]: in-kernel sigma-G ([25, 75] percentiles, coeff 0.7413, min_lh 10), min_obs T/2")
    ap.add_argument("--mask-fraction", type=float, default=0.0,
                    help="fraction of science pixels set to NaN before psi/phi is built (default 0: the BASELINE stack "
                         "has no masked pixels, so most tiles of kb_search_lds take its count-free path)")
    ap.add_argument("--min-lh", type=float, default=None, help="override the likelihood threshold (default 0, 10 with --sigmag)")
    ap.add_argument("--verify", action="store_true",
                    help="after timing: size-independent checks of the last result buffer (both kernels agree bit for "
                         "bit, per-pixel lists sorted, a start window re-done with exact per-lane positions agrees)")
    ap.add_argument("--reuse-padded-copy", action="store_true",
                    help="from the second step on tell the library that the array is unchanged (flag 256), as a StackSearch "
                         "with a resident array does: the decode-and-pad pass is then skipped (not the default: a step is a whole search)")
    ap.add_argument("--remake-padded-copy", action="store_true",
                    help="flag 2048: every step re-makes the padded canonical copy of the array (the decode-and-pad pass).  Default: "
                         "the library keeps the copy of an array it built itself for as long as nothing writes into the array -- "
                         "made once, by the first search (a warm-up step here), like psi/phi itself is built before the timed region")
    ap.add_argument("--no-overlap", action="store_true",
                    help="--gpus N > 1: finish every step's gather and merge before the next search starts (default: the "
                         "gather of step i travels while step i + 1 searches; all K steps complete inside the timed region)")
    ap.add_argument("--plain-ties", action="store_true",
                    help="multi-GPU: exchange K records per pixel and break ties by candidate index (the default exchanges "
                         "2 K records built by stable insertion and reproduces the single-GPU result exactly, ties included)")
    ap.add_argument("--stable-lists", action="store_true",
                    help="multi-GPU, dense exchange: every rank searches with 2 K stable records per pixel (flag 512) and the root "
                         "merges them exactly (rounds 3-5); the default since round 6 exchanges the K records of the ranks' normal "
                         "searches and re-makes the few pixels they do not decide (kb_merge_compact_repairable + kb_repair_pixels)")
    ap.add_argument("--separable-psf", action="store_true",
                    help="build psi/phi with the separable PSF kernel (KB_BUILD_SEPARABLE: <= 1e-4 relative to the reference's "
                         "tap loop instead of bit-identical)")
    ap.add_argument("--exchange", choices=["auto", "dense", "sparse"], default="auto",
                    help="multi-GPU: what travels to rank 0.  dense = every slot of every per-rank list (one gather, hidden under "
                         "the next search); sparse = one count byte per pixel + the records that pass min_lh (kb_sparsify_compact: "
                         "the reference drops everything below min_lh right after its kernel, stack_search.cpp:266-270, so nothing "
                         "that survives is lost); auto = sparse when the search has a likelihood threshold (min_lh > 0 or --sigmag)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target duration of the CPU baseline samples (both together)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="skip the two rocprofv3 --pmc sub-runs (FETCH_SIZE, WRITE_SIZE) of this command that measure the dominant "
                         "kernel's fabric bytes per launch; the roofline block then falls back to profiles/traffic.json")
    ap.add_argument("--no-masked", action="store_true",
                    help="skip the second timing on the same stack with 1 %% of the science pixels masked (reported as `masked`)")
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)  # a profiled sub-run of this script: timing loop only
    args = ap.parse_args()
    if args.child:
        args.no_cpu_baseline = args.no_live_traffic = args.no_masked = True

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(self_launch(sys.argv[1:], args.gpus))

    import torch
    import torch.distributed as dist

    from kbmod_amd import distributed as kdist
    from kbmod_amd import fake_data as fd

    world = int(world_env or "1")
    # KBMOD_FORCE_DIST=1: take the N > 1 branch -- process group, compact search, exchange, merge -- at whatever world size
    # this is, world size 1 included (tests/test_gpu_multi.py drives the real RCCL backend that way on a single GPU).
    dist_mode = world > 1 or os.environ.get("KBMOD_FORCE_DIST", "0") not in ("", "0")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU: the search has no CPU fallback")
    if world != args.gpus:
        raise RuntimeError(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    # KBMOD_DIST_BACKEND=gloo (default nccl = RCCL): the N > 1 path end to end where there are fewer GPUs than ranks --
    # ranks share the devices round-robin and the gather goes through host memory (tests/test_gpu_multi.py runs two
    # ranks on one GPU this way; RCCL itself refuses two ranks on one device).  Not a measurement configuration.
    backend = os.environ.get("KBMOD_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if dist_mode:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket

            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            sk.close()
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    lib = load_lib()
    T, H, W = args.frames, args.size, args.size
    K = args.results_per_pixel

    sci, var, times, psf = synthetic_stack(torch, dev, T, H, W, args.mask_fraction)
    tcpu = times.cpu().numpy()

    psf_all = np.ascontiguousarray(np.tile(psf.ravel(), T), dtype=np.float32)
    psf_dims = np.full(T, psf.shape[0], dtype=np.int32)
    meta = Meta()
    arr = C.c_void_p()
    stream = torch.cuda.current_stream().cuda_stream
    t0 = time.perf_counter()
    build_flags = 1 if args.separable_psf else 0
    check(lib, lib.kb_build_psi_phi_from_device_ex(sci.data_ptr(), var.data_ptr(), psf_all.ctypes.data, psf_dims.ctypes.data,
                                                   T, H, W, args.num_bytes, build_flags, C.byref(meta), C.byref(arr), stream))
    torch.cuda.synchronize()
    build_ms = (time.perf_counter() - t0) * 1e3
    # the builder again, device time only (events on the stream it runs on): correlation + range scan + encode
    build_kernel_ms = None
    if rank == 0 and T * H * W * 16 < (8 << 30):
        meta2, arr2 = Meta(), C.c_void_p()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib, lib.kb_build_psi_phi_from_device_ex(sci.data_ptr(), var.data_ptr(), psf_all.ctypes.data, psf_dims.ctypes.data,
                                                       T, H, W, args.num_bytes, build_flags, C.byref(meta2), C.byref(arr2), stream))
        e1.record()
        torch.cuda.synchronize()
        build_call_ms = e0.elapsed_time(e1)           # whole call on the device timeline (allocation gaps included)
        build_kernel_ms = float(lib.kb_last_build_kernel_ms())  # the correlation launch alone (HIP events inside the library)
        # ... and the OTHER builder kernel on the same stack (the north star's separable kernel when the default 2-D one was
        # measured above, and the other way round), with the largest relative difference between the two arrays where both
        # hold data (float arrays; the NO_DATA pattern must be identical) -- so that one driver record holds both
        other_build = None
        if not args.child and args.num_bytes in (-1, 4):
            meta3, arr3 = Meta(), C.c_void_p()
            check(lib, lib.kb_build_psi_phi_from_device_ex(sci.data_ptr(), var.data_ptr(), psf_all.ctypes.data, psf_dims.ctypes.data,
                                                           T, H, W, args.num_bytes, build_flags ^ 1, C.byref(meta3), C.byref(arr3), stream))
            torch.cuda.synchronize()
            other_ms = float(lib.kb_last_build_kernel_ms())
            n_el = int(meta3.num_entries)
            a_t = torch.empty(n_el, dtype=torch.float32, device=dev)
            b_t = torch.empty(n_el, dtype=torch.float32, device=dev)
            hip_rt = C.CDLL("libamdhip64.so")
            hip_rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            assert hip_rt.hipMemcpy(a_t.data_ptr(), arr2, n_el * 4, 3) == 0 and hip_rt.hipMemcpy(b_t.data_ptr(), arr3, n_el * 4, 3) == 0
            same_nan = bool(torch.equal(torch.isnan(a_t), torch.isnan(b_t)))
            ok = ~torch.isnan(a_t) & ~torch.isnan(b_t)
            # relative to the array's own scale (psi crosses zero: a per-element ratio there says nothing)
            scale = float(torch.abs(a_t[ok]).max().item()) if bool(ok.any()) else 1.0
            max_rel = float((torch.abs(a_t[ok] - b_t[ok]).max() / scale).item()) if bool(ok.any()) else 0.0
            other_build = {"kernel": "2-D strip (bit-identical)" if args.separable_psf else "separable strip (<= 1e-4)",
                           "kernel_ms": other_ms, "same_no_data_pattern": same_nan,
                           "max_abs_diff_over_array_max": max_rel}
            del a_t, b_t
            lib.kb_free_gpu_block(arr3)
        lib.kb_free_gpu_block(arr2)
    del sci, var

    # ---- candidates: KBMODV1Search(32, 5, 40, 32, 0, 1.5) = 1024 per GPU; rank r
    # takes the r-th contiguous slice of an N*1024 candidate (v, theta) grid ----
    vx, vy = fd.kbmod_v1_candidates(args.vel_steps, args.min_vel, args.max_vel, args.ang_steps * world, args.min_ang,
                                    args.max_ang)
    if world > 1:
        # The job-wide list in rank-major order with the angles dealt boustrophedon: rank r searches angle rows r, 2 N - 1 - r, 2 N + r, ... of the
        # N * ang_steps-row grid (back and forth, so that every rank pairs flat rows with steep ones).  A chunk still holds the speeds of ONE angle (what its slab width follows), but every rank
        # gets the same mix of flat and steep trajectories -- with contiguous angle bands the rank with the steepest band set
        # the step (cfg4: 17.0 ... 22.5 ms per rank).  The merged result is checked against one search over THIS list.
        order = np.array([j * world + (r if j % 2 == 0 else world - 1 - r) for r in range(world) for j in range(args.ang_steps)])
        vx = vx.reshape(args.ang_steps * world, args.vel_steps)[order].reshape(-1)
        vy = vy.reshape(args.ang_steps * world, args.vel_steps)[order].reshape(-1)
    n_local = args.vel_steps * args.ang_steps
    sl = slice(rank * n_local, (rank + 1) * n_local)
    all_np = np.zeros((n_local * world, 7), dtype=np.float32)
    all_np[:, 0], all_np[:, 1] = vx, vy
    all_cands = torch.from_numpy(all_np).to(dev)   # the job-wide list (the merge on rank 0 indexes it)
    cands = all_cands[sl]                           # this rank's slice

    ins = args.inset
    S = (H - 2 * ins) * (W - 2 * ins)
    nb_param = -1 if args.num_bytes in (-1, 4) else args.num_bytes
    if args.sigmag:
        params = Params(T // 2, 10.0 if args.min_lh is None else args.min_lh, 1, 0.25, 0.75, 0.7413, nb_param, ins, W - ins,
                        ins, H - ins, K, 0)
    else:
        params = Params(0, 0.0 if args.min_lh is None else args.min_lh, 0, 0.25, 0.75, -1.0, nb_param, ins, W - ins, ins,
                        H - ins, K, 0)
    # multi-GPU: tie-exact exchange -- every rank keeps 2 K records per pixel by stable insertion (flag 512), the merge
    # on rank 0 replays the reference's insertion (kbmod_amd/distributed.py); K > 16 or --plain-ties: K records, ties by index
    # ... and in what form: a search with a likelihood threshold leaves nearly every slot empty or below it, and the
    # reference drops those right after its kernel -- one count byte per pixel + the survivors travel instead (sparse)
    thresholded = args.sigmag or float(params.min_lh) > 0.0
    want_exact = dist_mode and not args.plain_ties
    sparse = want_exact and K <= 16 and (args.exchange == "sparse" or (args.exchange == "auto" and thresholded))
    if args.exchange == "sparse" and dist_mode and not sparse:
        raise RuntimeError("--exchange sparse goes through the tie-exact merge (K <= 16, not --plain-ties)")
    # Round 6, dense exchange: the ranks search with their NORMAL K-record lists (the fastest single-GPU instance, half the
    # bytes on the wire), the root folds the lists in candidate order and re-makes the few pixels the records do not decide
    # from its replica of the stack.  Same result as the 2 K stable lists, bit for bit (--verify).  Not with the in-search
    # sigma-G filter (the repair evaluates plain trajectories) -- those searches exchange sparsely anyway.
    # ... and not for short candidate lists per rank (unless K > 16, where it is the only exact form): the pixels whose
    # trajectories mostly leave the image tie in bulk, and with 64 candidates per rank on a 4096 x 4096 grid 4 % of the
    # pixels went to the repair (100 ms against a 20 ms search, profiles/r06_exchange_repair_cfg4.json).
    repair = (want_exact and not sparse and not args.stable_lists and not args.sigmag and K <= 32 and
              (n_local >= 256 or K > 16))
    exact_ties = want_exact and not repair and K <= 16   # 2 K stable lists (flag 512) + kb_merge_compact_exact
    list_len = 2 * K if exact_ties else K
    wire = {}
    if dist_mode:
        rank_params = Params.from_buffer_copy(params)
        rank_params.results_per_pixel = list_len
        # two sets of exchange buffers: the records of step i are on the wire while step i + 1 fills the other set
        n_sets = 1 if sparse else 2
        records2 = [torch.empty((S * list_len, 4), dtype=torch.int32, device=dev) for _ in range(n_sets)]  # kb_compact_result per slot
        gathered2 = [torch.empty((world, S * list_len, 4), dtype=torch.int32, device=dev) if (rank == 0 and not sparse) else None
                     for _ in range(n_sets)]
        results = torch.empty((S * K, 7), dtype=torch.float32, device=dev) if rank == 0 else None
        if sparse:
            sp_header = torch.empty(int(lib.kb_sparse_header_bytes(S)), dtype=torch.uint8, device=dev)
            sp_packed = [None]   # sized by the first step (grown when a later one keeps more)
    else:
        results = torch.empty((S * K, 7), dtype=torch.float32, device=dev)

    # A search with a likelihood threshold is launched the way StackSearch.search_all launches it (flag 1024): what the
    # reference's post-filter removes anyway need not enter a list.  The sparse exchange relies on the same post-filter;
    # the dense one carries whole lists, sub-threshold slots included, and keeps the default.
    base_flags = args.flags | (1024 if (thresholded and (sparse or not dist_mode)) else 0) | (2048 if args.remake_padded_copy else 0)

    def run_timed(meta, arr, n_warmup, n_steps):
        """n_warmup untimed + n_steps timed whole searches of the array at `arr`; (elapsed s, kernel ms per step, last stats)."""
        kernel_ms = []
        call_ms = []   # host wall time of the search call alone (tables + sync + kernel; the call returns when its kernel has)
        searched = [False]
        in_flight = [None]   # the exchange of the previous step (N > 1, dense, overlapped)
        which_set = [0]

        def drain():
            if in_flight[0] is not None:
                in_flight[0].finish()
                in_flight[0] = None

        def step(record):
            st = Stats()
            # Every step is a whole search: tables, decode-and-pad pass, search kernel.  --reuse-padded-copy adds flag 256
            # from the second step on (the array has not changed since the previous search: what a StackSearch with a
            # resident array passes), which lets the library keep the padded float copy of the last search.
            flags = base_flags | (256 if (searched[0] and args.reuse_padded_copy) else 0)
            searched[0] = True
            t_call = time.perf_counter()
            if dist_mode:
                b = which_set[0]
                which_set[0] = (b + 1) % n_sets
                records, gathered = records2[b], gathered2[b]
                if sparse:
                    # the search writes the count bytes of the sparse header itself where its kernel instance can, and then
                    # leaves out the record runs of waves that keep nothing (kb_device_search_counted)
                    counted = C.c_int32(0)
                    check(lib, lib.kb_device_search_counted(C.byref(meta), arr, times.data_ptr(), rank_params, cands.data_ptr(),
                                                            n_local, rank * n_local, records.data_ptr(), S * list_len,
                                                            sp_header.data_ptr(), flags | (512 if exact_ties else 0), stream,
                                                            C.byref(st), C.byref(counted)))
                    t_call = time.perf_counter() - t_call
                    # nothing to hide: the wire carries a count byte per pixel and the few records above the threshold
                    stats = {}
                    try:
                        kdist.gather_and_merge_sparse(records, (ins, W - ins), (ins, H - ins), K, list_len, float(params.min_lh),
                                                      all_cands, out=results, header=sp_header, packed=sp_packed[0], stats=stats,
                                                      counted=bool(counted.value))
                    except RuntimeError as err:
                        if "records kept, room for" not in str(err):
                            raise
                        # (the header is complete and holds the total: make room and go again)
                        need = int(sp_header[(S + 15) // 16 * 16:].view(torch.int64)[0].item())
                        sp_packed[0] = torch.empty((need + need // 4 + 1024, 4), dtype=torch.int32, device=dev)
                        kdist.gather_and_merge_sparse(records, (ins, W - ins), (ins, H - ins), K, list_len, float(params.min_lh),
                                                      all_cands, out=results, header=sp_header, packed=sp_packed[0], stats=stats,
                                                      counted=bool(counted.value))
                    stats["search_wrote_counts"] = bool(counted.value)
                    wire.update(stats)
                else:
                    check(lib, lib.kb_device_search_compact(C.byref(meta), arr, times.data_ptr(), rank_params, cands.data_ptr(),
                                                            n_local, rank * n_local, records.data_ptr(), S * list_len,
                                                            flags | (512 if exact_ties else 0), stream, C.byref(st)))
                    t_call = time.perf_counter() - t_call
                    nxt = kdist.start_gather_compact(records, (ins, W - ins), (ins, H - ins), K, all_cands, gathered=gathered,
                                                     out=results, list_len=list_len,
                                                     repair_stack=(meta, arr.value, times.data_ptr()) if repair else None,
                                                     min_obs=int(params.min_observations),
                                                     list_begin=[r * n_local for r in range(world + 1)] if repair else None)
                    drain()               # the previous step's gather has had this step's search to travel in; merge it now
                    in_flight[0] = nxt
                    if args.no_overlap:
                        drain()
                    wire["wire_bytes"] = int(records.numel()) * 4
            else:
                check(lib, lib.kb_device_search_filter(C.byref(meta), arr, times.data_ptr(), params, cands.data_ptr(), n_local,
                                                       results.data_ptr(), S * K, flags, stream, C.byref(st)))
                t_call = time.perf_counter() - t_call
            if record:
                kernel_ms.append(st.search_kernel_ms)
                call_ms.append(t_call * 1e3)
            return st

        for _ in range(n_warmup):
            step(False)
        drain()

        def barrier():
            if dist_mode:
                dist.barrier()
            torch.cuda.synchronize()

        barrier()
        t0 = time.perf_counter()
        last = None
        for _ in range(n_steps):
            last = step(True)
        drain()   # the last step's gather and merge belong to the timed region
        barrier()
        elapsed = time.perf_counter() - t0
        if dist_mode:
            tt = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed, kernel_ms, last, call_ms

    if sparse:
        sp_packed[0] = torch.empty((max(1024, S * list_len // 64), 4), dtype=torch.int32, device=dev)
    elapsed, kernel_ms, last, call_ms = run_timed(meta, arr, args.warmup, args.steps)

    evals_per_step_rank = int(last.num_evals)
    total_evals = evals_per_step_rank * world * args.steps
    value = total_evals / elapsed
    k_ms = float(np.mean(kernel_ms))
    alg_rate = float(last.algorithmic_bytes) / (k_ms * 1e-3) / 1e9
    dtype = {-1: "f32", 4: "f32", 1: "u8", 2: "u16"}[args.num_bytes]
    instance = last.kernel_name.decode()
    is_lds_kernel = int(last.kernel_variant) // 10000 != 0   # 0xxxx = kb_search_direct, 1xxxx / 2xxxx = kb_search_lds

    copy_gbps, read_gbps, lds_gbps = C.c_double(0.0), C.c_double(0.0), C.c_double(0.0)
    if not args.child:
        # measured HBM peak: a streaming device copy (4 GiB each way) timed with HIP events in this run
        check(lib, lib.kb_measure_copy_bandwidth(4 << 30, 10, stream, C.byref(copy_gbps)))
        # ... what a read-only stream over a block of the array's size reaches (no write traffic; for cfg2's 134 MB this is
        # the rate at which the Infinity Cache feeds the L2s): the ceiling of the search's fabric traffic, which is 99 % reads
        check(lib, lib.kb_measure_read_bandwidth(min(int(T) * H * W * 8, 4 << 30), 20, stream, C.byref(read_gbps)))
        # ... and the aggregate LDS read rate (ds_read_b64 on every CU): the yardstick of the sums' LDS traffic
        check(lib, lib.kb_measure_lds_bandwidth(4096, stream, C.byref(lds_gbps)))

    # Fabric-side traffic of the dominant kernel (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE per launch; the x 2 is the gfx950
    # correction of MI355X_MICROARCH.md, HBM section).  Counters cannot be read from inside this process, so rank 0 of a
    # single-GPU run profiles two short sub-runs of this very command (one --pmc pass per counter, as that guide
    # prescribes) and takes the per-dispatch average of the instance that ran here.  Without rocprofv3, or with
    # --no-live-traffic: the stored profile of the same workload AND instance (profiles/traffic.json), said in traffic_source.
    traffic = traffic_source = traffic_rejected = None
    traced = {}
    if rank == 0 and world == 1 and not dist_mode and not args.no_live_traffic:
        live = live_traffic(sys.argv[1:], instance)
        if "bytes" in live:
            traffic, traffic_source = live["bytes"], live["source"]
            traced = {k: live.get(k) for k in ("traced_kernel_ms", "traced_launches", "lds_insts", "lds_cycles", "lds_error")
                      if live.get(k) is not None}
        else:
            traffic_rejected = live["error"]
    if traffic is None:
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
                table = json.load(fh)
            key = f"{dtype}:{T}x{H}x{W}:{n_local}" + (":sigmag" if args.sigmag else "")
            entry = table.get(key)
            if entry is not None and entry.get("kernel_instance") == instance:
                traffic, traffic_source = entry["bytes"], "stored profile, not this run: " + entry["source"]
            elif entry is not None:
                traffic_rejected = (traffic_rejected + "; " if traffic_rejected else "") + \
                    f"profiles/traffic.json[{key}] was measured on {entry.get('kernel_instance')}, this run launched {instance}"
        except (OSError, ValueError):
            pass

    padded_guess = int(meta.total_array_size) * (8 // int(meta.block_size * 2) if meta.num_bytes != 4 else 1)
    # compulsory = what any implementation must move: the array once, the candidates and times in, the result slots out
    compulsory = int(meta.total_array_size) + n_local * 28 + T * 8 + S * K * 28
    roof = roofline_block(kernel_ms=k_ms, algorithmic_bytes=int(last.algorithmic_bytes), lds_read_bytes=int(last.lds_read_bytes),
                          evals_per_step=evals_per_step_rank, traffic=traffic, traffic_source=traffic_source,
                          traffic_rejected=traffic_rejected, compulsory_bytes=compulsory, padded_copy_bytes_guess=padded_guess,
                          is_lds_kernel=is_lds_kernel, kernel=instance, edge_count_tables=int(last.edge_count_tables),
                          mask_fraction=args.mask_fraction, padded_copy_reused=int(last.padded_copy_reused),
                          env_overrides=int(last.env_overrides), copy_gbps=copy_gbps.value, read_gbps=read_gbps.value,
                          lds_gbps=lds_gbps.value, psi_phi_bytes=int(meta.total_array_size), traced=traced)
    config = {
        "workload": f"{T}x{H}x{W} psi/phi ({'float32' if args.num_bytes in (-1, 4) else 'uint%d' % (8 * args.num_bytes)}), "
                    f"full {W}x{H} start grid x {n_local} (v,theta) candidates per GPU, K={K}, "
                    f"sigma-G {'on, min_obs %d' % (T // 2) if args.sigmag else 'off'}"
                    + (f", min_lh {float(params.min_lh):g}" if float(params.min_lh) > 0 else "")
                    + (f", {args.mask_fraction:g} of the science pixels masked" if args.mask_fraction > 0 else "")
                    + ("; a step = one whole search of the resident array (tables + search kernel); the one-time decode-and-pad "
                       "pass of the array's first search is outside the steps and reported in first_search"
                       if int(last.padded_copy_reused) else ""),
        "frames": T, "height": H, "width": W, "candidates_per_gpu": n_local, "results_per_pixel": K,
        "sharding": ("candidates (v,theta) by rank (angle rows dealt boustrophedon); psi/phi replicated; "
                     + (f"sparse exchange (one count byte per pixel + the records with lh >= {float(params.min_lh):g} of {list_len} "
                        "per pixel: one RCCL gather of the headers + one message per rank) + per-pixel merge, tie-exact" if sparse
                        else f"one RCCL gather of 16-byte records to rank 0 ({list_len} per pixel) + per-pixel merge, "
                        + ("tie-exact (2 K stable lists)" if exact_ties else
                           ("tie-exact (the ranks' normal K-record lists folded in candidate order; the pixels they do not decide "
                            "re-made on rank 0)" if repair else "ties by candidate index"))))
                    if dist_mode else "none",
        "psi_phi_build_ms": build_ms,
    }
    out = headline_line(total_evals=total_evals, elapsed_s=elapsed, world=world, steps=args.steps, warmup=args.warmup, dtype=dtype,
                        config=config, roofline=roof)
    if dist_mode:
        # what a SCALE record needs to explain itself: every rank's own search time (device, HIP events) gathered to rank 0,
        # what one rank put on the wire, the root's merge, the form of the exchange and how many ranks the backend ran
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "search_kernel_ms": k_ms, "search_call_ms": float(np.mean(call_ms))})
        out["exchange"] = {"form": "sparse" if sparse else "dense", "exchange_form": "sparse" if sparse else "dense",
                           "list_len": list_len, "backend": backend, "rccl_ranks": world if backend == "nccl" else 0,
                           "wire_bytes_per_rank": wire.get("wire_bytes"), "dense_bytes_per_rank": S * list_len * 16,
                           "records_per_rank": wire.get("totals"), "overlapped": bool(not sparse and not args.no_overlap),
                           "search_wrote_counts": wire.get("search_wrote_counts"),
                           "per_rank_search_ms": [p["search_call_ms"] for p in per_rank],
                           "per_rank_search_kernel_ms": [p["search_kernel_ms"] for p in per_rank],
                           "merge_ms": kdist.last_merge_ms() if rank == 0 else None,
                           "lists": "2K stable" if exact_ties else ("K records + repair" if repair else "K records"),
                           "repair": kdist.last_repair() if (rank == 0 and repair) else None}
    if build_kernel_ms is not None:
        in_out = float(T) * H * W * 8 + float(meta.total_array_size)  # sci + var in, the array out
        out["psi_phi_build"] = {"kernel": "separable strip (<= 1e-4)" if args.separable_psf else "2-D strip (bit-identical)",
                                "kernel_ms": build_kernel_ms, "call_device_ms": build_call_ms, "bytes_in_plus_out": in_out,
                                "GBps": in_out / (build_kernel_ms * 1e-3) / 1e9,
                                "frac_of_achievable": in_out / (build_kernel_ms * 1e-3) / 1e9 / HBM_ACHIEVABLE_GBPS}
        if other_build is not None:
            other_build.update(GBps=in_out / (other_build["kernel_ms"] * 1e-3) / 1e9,
                               frac_of_achievable=in_out / (other_build["kernel_ms"] * 1e-3) / 1e9 / HBM_ACHIEVABLE_GBPS)
            out["psi_phi_build"]["other_kernel"] = other_build
    if args.sigmag:
        out["config"]["sigmag_work_items"] = int(last.sigmag_work_items)
        out["config"]["sigmag_trajectories_clipped"] = int(last.sigmag_trajectories)

    if args.verify and world == 1 and not dist_mode:
        out["verify"] = verify(lib, torch, meta, arr, times, params, cands, n_local, results, S, K, ins, W, H, last, stream,
                               base_flags & 1024)
    if args.verify and dist_mode and rank == 0:
        # the merged lists of the job against ONE search over the job-wide candidate list on this rank's device
        single = torch.empty((S * K, 7), dtype=torch.float32, device=dev)
        st1 = Stats()
        check(lib, lib.kb_device_search_filter(C.byref(meta), arr, times.data_ptr(), params, all_cands.data_ptr(),
                                               n_local * world, single.data_ptr(), S * K, args.flags, stream, C.byref(st1)))  # (default lists: no floor)
        torch.cuda.synchronize()
        if sparse:
            # the sparse exchange carries what survives the reference's post-filter (stack_search.cpp:266-270): those
            # slots must equal the single-device search field for field, every other slot is the empty-slot placeholder
            gone = single[:, 2] < float(params.min_lh)
            single[gone, 0:2] = 0.0
            single[gone, 2] = torch.finfo(torch.float32).min
            single[gone, 3] = 0.0
            single.view(torch.int32)[gone, 6] = 0
            out["verify_survivors"] = int((~gone).sum().item())
        same = torch.equal(results.view(torch.int32), single.view(torch.int32))
        lh_same = torch.equal(results[:, 2].contiguous().view(torch.int32), single[:, 2].contiguous().view(torch.int32))
        out["verify"] = {"merged_equals_single_device_ok": bool(same) if (exact_ties or repair or sparse) else bool(lh_same),
                         "merged_likelihoods_equal_ok": bool(lh_same), "tie_exact": bool(exact_ties or repair or sparse), "backend": backend,
                         "exchange": "sparse" if sparse else "dense", "world": world}

    # What a pipeline that builds ONE StackSearch and calls search_all once pays (the reference's, run_search.py:363-378): the
    # builder + the first search of a FRESH array, whose padded canonical copy does not exist yet -- the timed steps above search
    # an array whose copy the warm-up search made.  Wall clock, every piece synchronised; the second search of the same fresh
    # array is the steady step again, the difference is the decode-and-pad pass.
    if rank == 0 and world == 1 and not dist_mode and not args.child:
        sci_f, var_f, _, _ = synthetic_stack(torch, dev, T, H, W, args.mask_fraction)
        meta_f, arr_f = Meta(), C.c_void_p()
        res_f = torch.empty((S * K, 7), dtype=torch.float32, device=dev)
        # (one untimed build + free first: the device allocator then holds a block of the array's size, and what is timed below
        # is the builder and the search, not a fresh multi-gigabyte hipMalloc)
        check(lib, lib.kb_build_psi_phi_from_device_ex(sci_f.data_ptr(), var_f.data_ptr(), psf_all.ctypes.data, psf_dims.ctypes.data,
                                                       T, H, W, args.num_bytes, build_flags, C.byref(meta_f), C.byref(arr_f), stream))
        torch.cuda.synchronize()
        lib.kb_free_gpu_block(arr_f)
        arr_f = C.c_void_p()
        t0 = time.perf_counter()
        check(lib, lib.kb_build_psi_phi_from_device_ex(sci_f.data_ptr(), var_f.data_ptr(), psf_all.ctypes.data, psf_dims.ctypes.data,
                                                       T, H, W, args.num_bytes, build_flags, C.byref(meta_f), C.byref(arr_f), stream))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        st_f = [Stats(), Stats()]
        t_s = [t1]
        for i in range(2):
            check(lib, lib.kb_device_search_filter(C.byref(meta_f), arr_f, times.data_ptr(), params, cands.data_ptr(), n_local,
                                                   res_f.data_ptr(), S * K, base_flags, stream, C.byref(st_f[i])))
            torch.cuda.synchronize()
            t_s.append(time.perf_counter())
        out["first_search"] = {
            "first_search_ms": (t_s[1] - t0) * 1e3,                 # builder + (tables + decode-and-pad + search kernel)
            "build_ms": (t1 - t0) * 1e3,
            "first_search_call_ms": (t_s[1] - t1) * 1e3,
            "second_search_call_ms": (t_s[2] - t_s[1]) * 1e3,
            "pad_pass_ms": ((t_s[1] - t1) - (t_s[2] - t_s[1])) * 1e3 if int(st_f[1].padded_copy_reused) else None,
            "padded_copy_made_by_first": not bool(st_f[0].padded_copy_reused),
            "padded_copy_reused_by_second": bool(st_f[1].padded_copy_reused),
            "results_equal_the_timed_steps": bool(torch.equal(res_f.view(torch.int32), results.view(torch.int32))),
        }
        lib.kb_free_gpu_block(arr_f)
        del sci_f, var_f, res_f

    # The same workload on a stack with 1 % of its science pixels masked -- what every real survey stack looks like: then
    # every tile of kb_search_lds counts observations per sample instead of taking them from tables.
    if rank == 0 and world == 1 and not dist_mode and not args.no_masked and args.mask_fraction == 0.0 and T * H * W * 16 < (8 << 30):
        sci_m, var_m, _, _ = synthetic_stack(torch, dev, T, H, W, 0.01)
        meta_m, arr_m = Meta(), C.c_void_p()
        check(lib, lib.kb_build_psi_phi_from_device_ex(sci_m.data_ptr(), var_m.data_ptr(), psf_all.ctypes.data, psf_dims.ctypes.data,
                                                       T, H, W, args.num_bytes, build_flags, C.byref(meta_m), C.byref(arr_m), stream))
        torch.cuda.synchronize()
        del sci_m, var_m
        el_m, k_m, last_m, _ = run_timed(meta_m, arr_m, args.warmup, args.steps)
        out["masked"] = {"mask_fraction": 0.01, "ms_per_step": el_m / args.steps * 1e3, "kernel_ms": float(np.mean(k_m)),
                         "value": int(last_m.num_evals) * args.steps / el_m, "unit": "evals/s",
                         "kernel": last_m.kernel_name.decode(), "steps": args.steps,
                         "obs_counts": "counted per sample in every tile"}
        lib.kb_free_gpu_block(arr_m)

    if rank == 0 and world == 1 and not dist_mode and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(lib, meta, arr, tcpu, vx[sl], vy[sl], args.cpu_seconds / 2)
        if meta.num_bytes == 4:
            out["cpu_baseline_full_sort"] = cpu_baseline_full_sort(lib, meta, arr, tcpu, vx[sl], vy[sl], args.cpu_seconds / 2)

    if os.environ.get("KBMOD_EXP_PROFILE") and hasattr(lib, "kb_exp_read_profile"):
        prof = (C.c_ulonglong * 8)()
        lib.kb_exp_read_profile(prof)
        waves = max(1, prof[6])
        print("phase ticks per wave:", [round(prof[i] / waves) for i in range(6)], "waves", prof[6], file=sys.stderr)
    if rank == 0:
        print(json.dumps(out))
    lib.kb_free_gpu_block(arr)
    if args.verify and rank == 0 and not all(v for k, v in out["verify"].items() if k.endswith("_ok")):
        sys.exit(3)
    if dist_mode:
        dist.destroy_process_group()


def live_traffic(argv, instance):
    """Counters and traced durations of the search kernel instance `instance`, measured now: three short sub-runs of this
    command under `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE need separate passes: MI355X_MICROARCH.md, PMC slots; the
    third collects SQ_INSTS_LDS and SQ_LDS_IDX_ACTIVE), per-dispatch averages over the launches of that instance.
    bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: the counters are in KiB and gfx950's FETCH_SIZE tallies 128-byte requests
    at 64 bytes (same guide, HBM section).  Every pass also records each dispatch's start and end, so the kernel's duration
    ON THIS BOX under the tracer comes with the counters (traced_kernel_ms: the mean over the three passes' launches).
    {"error": ...} when rocprofv3 is missing or a byte pass fails; the LDS pass is optional (its keys are then absent)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile

    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return {"error": "rocprofv3 not found"}
    child = [a for a in argv if a not in ("--verify",)]
    for flag in ("--steps", "--warmup", "--gpus", "--cpu-seconds"):
        while flag in child:
            i = child.index(flag)
            del child[i:i + 2]
    cmd_tail = [sys.executable, os.path.abspath(__file__)] + child + ["--steps", "3", "--warmup", "1", "--child"]
    found, durations, lds_error = {}, [], None
    env = dict(os.environ, TMPDIR="/tmp")
    for counters in (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE")):
        optional = counters[0].startswith("SQ_")
        d = tempfile.mkdtemp(prefix="kb_pmc_", dir="/tmp")
        try:
            r = subprocess.run([prof, "--pmc"] + list(counters) + ["-d", d, "-o", "r", "--"] + cmd_tail, cwd="/tmp", env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=180)
            dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                raise RuntimeError(f"rocprofv3 --pmc {' '.join(counters)} failed (rc {r.returncode}): "
                                   f"{r.stderr.decode(errors='replace')[-300:]}")
            con = sqlite3.connect(dbs[0])
            rows = con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                               "group by kernel_name, counter_name").fetchall()
            # (one row per dispatch and counter: the duration of a dispatch once)
            dur = con.execute("select kernel_name, dispatch_id, max(duration) from counters_collection "
                              "group by kernel_name, dispatch_id").fetchall()
            con.close()
            for counter in counters:
                hit = [(c, v) for n, cn, c, v in rows if instance in n and cn == counter]
                if not hit:
                    raise RuntimeError(f"no dispatch of {instance} in the --pmc {counter} pass")
                found[counter] = (hit[0][0], float(hit[0][1]))
            durations += [float(t) * 1e-6 for n, _, t in dur if instance in n and t]
        except (subprocess.TimeoutExpired, sqlite3.Error, OSError, RuntimeError) as err:
            if not optional:
                return {"error": f"--pmc {' '.join(counters)} pass: {err}"}
            lds_error = str(err)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch, write = found["FETCH_SIZE"][1], found["WRITE_SIZE"][1]
    out = {"bytes": int((2.0 * fetch + write) * 1024), "fetch_kib": fetch, "write_kib": write,
           "source": f"live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE sub-runs of this command, average of "
                     f"{found['FETCH_SIZE'][0]} launches of the instance; (2 x FETCH_SIZE + WRITE_SIZE) x 1024",
           "traced_kernel_ms": float(np.mean(durations)) if durations else None, "traced_launches": len(durations)}
    if "SQ_INSTS_LDS" in found:
        out["lds_insts"] = found["SQ_INSTS_LDS"][1]
        out["lds_cycles"] = found["SQ_LDS_IDX_ACTIVE"][1]
    elif lds_error:
        out["lds_error"] = lds_error
    return out


def verify(lib, torch, meta, arr, times, params, cands, n_cands, results, S, K, ins, W, H, last, stream, list_flags=0):
    """Size-independent checks of one finished search (results = its buffer, on the device; list_flags = the flags of that
    search that shape its lists, passed on to the searches it is compared with)."""
    dev = results.device
    sw = W - 2 * ins

    def run(p, n_slots, flags):
        buf = torch.empty((n_slots, 7), dtype=torch.float32, device=dev)
        st = Stats()
        check(lib, lib.kb_device_search_filter(C.byref(meta), arr, times.data_ptr(), p, cands.data_ptr(), n_cands,
                                               buf.data_ptr(), n_slots, flags, stream, C.byref(st)))
        torch.cuda.synchronize()
        return buf, st

    out = {}
    # 1. the other kernel, same search, bit for bit
    was_lds = int(last.kernel_variant) // 10000 != 0
    other, st = run(params, S * K, (2 if was_lds else 4) | list_flags)
    out["other_kernel"] = "kb_search_direct" if int(st.kernel_variant) // 10000 == 0 else "kb_search_lds"
    out["kernels_agree_ok"] = bool(torch.equal(results.view(torch.int32), other.view(torch.int32)))
    del other
    # 2. every per-pixel list is in descending likelihood order (the swap-down insertion's invariant)
    lh = results[:, 2].view(S, K)
    out["lists_sorted_ok"] = bool((lh[:, :-1] >= lh[:, 1:]).all().item())
    # 3. a start window re-done with exact per-lane positions (no shift table, no staging) gives the same slots
    x0 = ins + max(0, (sw - 96) // 2)
    y0 = ins + max(0, (H - 2 * ins - 8) // 2)
    x1, y1 = min(x0 + 96, W - ins), min(y0 + 8, H - ins)
    wp = Params.from_buffer_copy(params)
    wp.x_start_min, wp.x_start_max, wp.y_start_min, wp.y_start_max = x0, x1, y0, y1
    win, _ = run(wp, (x1 - x0) * (y1 - y0) * K, 1 | list_flags)
    full = results.view(H - 2 * ins, sw, K, 7)[y0 - ins:y1 - ins, x0 - ins:x1 - ins].reshape(-1, 7)
    out["exact_window"] = [x0, x1, y0, y1]
    out["exact_window_ok"] = bool(torch.equal(full.contiguous().view(torch.int32), win.view(torch.int32)))
    return out


def cpu_baseline(lib, meta, arr, times, vx, vy, target_s):
    """The oracle's restatement of the reference CPU search (cpu_search_algorithms.cpp:93-124),
    OpenMP over the host cores, on a bounded start-pixel window of the SAME stack and candidates."""
    from oracle import oracle as orc

    host = np.empty(int(meta.num_entries), dtype={4: np.float32, 2: np.uint16, 1: np.uint8}[meta.num_bytes])
    check(lib, lib.kb_copy_block_to_cpu(host.ctypes.data, arr, meta.total_array_size))
    pp = orc.PsiPhi.__new__(orc.PsiPhi)
    pp.meta = orc.Meta(meta.num_times, meta.width, meta.height, meta.num_bytes, meta.psi_min_val, meta.psi_max_val,
                       meta.psi_scale, meta.phi_min_val, meta.phi_max_val, meta.phi_scale)
    pp.array = host
    pp.times = np.ascontiguousarray(times, dtype=np.float64)
    pp.T, pp.H, pp.W, pp.nb = int(meta.num_times), int(meta.height), int(meta.width), int(meta.num_bytes)
    cands = orc.make_candidates(vx, vy)
    H, W, T = pp.H, pp.W, pp.T

    def run(rows):
        y0 = (H - rows) // 2
        p = pp.default_params(y_start_min=y0, y_start_max=y0 + rows)
        t0 = time.perf_counter()
        pp.search_cpu(cands, p)
        return time.perf_counter() - t0, rows * W * len(cands) * T

    dt, ev = run(16)  # calibration
    rate = ev / dt
    rows = int(max(4, min(H, target_s * rate / (W * len(cands) * T))))
    dt, ev = run(rows)
    return {
        "value": ev / dt,
        "unit": "evals/s",
        "cores": orc.num_threads(),
        "kind": "port",
        "sample": f"{rows} of {H} start rows (full width) x {len(cands)} candidates x {T} epochs of the same stack, "
                  f"{dt:.1f} s, oracle/kbmod_oracle.c orc_search_cpu (OpenMP)",
    }


def cpu_baseline_full_sort(lib, meta, arr, times, vx, vy, target_s):
    """The reference CPU search's own shape (cpu_search_algorithms.cpp:57-124: per start pixel every candidate evaluated
    into a list, the WHOLE list sorted, its head kept), as the product's host layer restates it (search_cpu_only of
    kbmod_amd/csrc/host/stack_search.h, OpenMP over the host cores), on a bounded window of the same stack and candidates."""
    import kbmod_amd.search as kb

    T, H, W = int(meta.num_times), int(meta.height), int(meta.width)
    host = np.empty((T, H, W, 2), dtype=np.float32)
    check(lib, lib.kb_copy_block_to_cpu(host.ctypes.data, arr, meta.total_array_size))
    pp = kb.PsiPhiArray()
    kb.fill_psi_phi_array(pp, 4, [np.ascontiguousarray(host[t, :, :, 0]) for t in range(T)],
                          [np.ascontiguousarray(host[t, :, :, 1]) for t in range(T)], [float(t) for t in times])
    del host
    cands = kb.TrajectoryList([kb.Trajectory(vx=float(a), vy=float(b)) for a, b in zip(vx, vy)])

    def run(rows):
        p = kb.SearchParameters()
        p.min_observations, p.min_lh, p.do_sigmag_filter = 0, 0.0, False
        p.x_start_min, p.x_start_max = 0, W
        p.y_start_min = (H - rows) // 2
        p.y_start_max = p.y_start_min + rows
        p.results_per_pixel = 8
        res = kb.TrajectoryList(0)
        t0 = time.perf_counter()
        kb.search_cpu_only(pp, p, cands, res)
        return time.perf_counter() - t0, rows * W * len(vx) * T

    dt, ev = run(8)  # calibration
    rows = int(max(4, min(H, target_s * (ev / dt) / (W * len(vx) * T))))
    dt, ev = run(rows)
    return {
        "value": ev / dt,
        "unit": "evals/s",
        "cores": int(kb.omp_max_threads()),
        "kind": "restatement",
        "sample": f"{rows} of {H} start rows (full width) x {len(vx)} candidates x {T} epochs of the same stack, {dt:.1f} s, "
                  "kbmod_amd.search.search_cpu_only (per-pixel full sort like cpu_search_algorithms.cpp:57-86, OpenMP)",
    }


if __name__ == "__main__":
    main()
