"""Device time of the post-search kernels of SURVEY.md section 8(f) (not the headline bench):
kb_sigma_g_clip_matrix on N x T likelihood curves resident in HBM, HIP-event timed on the stream it
is launched on (the null stream), with its algorithmic bytes (N*T*(4 read + 1 written)) against the
HBM peak, and the reference's torch sequence (filters/sigma_g_filter.py:132-165) timed on the host
cores on a bounded sample.  Usage: python tools/bench_post_search.py [--rows N] [--cols T]"""

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2_000_000)
    ap.add_argument("--cols", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import torch

    from bench import check, load_lib

    lib = load_lib()
    lib.kb_sigma_g_clip_matrix.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float,
                                           C.c_int32, C.c_void_p, C.c_void_p]
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    lh = torch.randn((args.rows, args.cols), generator=gen, device=dev) * 3.0 + 5.0
    lh[torch.rand((args.rows, args.cols), generator=gen, device=dev) < 0.05] = float("nan")
    valid = torch.empty((args.rows, args.cols), dtype=torch.uint8, device=dev)

    def step():
        check(lib, lib.kb_sigma_g_clip_matrix(lh.data_ptr(), args.rows, args.cols, 25.0, 75.0, 2.0, 0.7413, 0,
                                              valid.data_ptr(), None))

    step()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # the library launches on the null stream; torch's default stream is the null stream as well
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / args.steps
    alg = args.rows * args.cols * 5
    # CPU: the reference's torch sequence on a bounded sample
    n_cpu = min(args.rows, 200_000)
    sample = lh[:n_cpu].cpu()
    t0 = time.perf_counter()
    q = torch.tensor([0.25, 0.5, 0.75], dtype=torch.float32)
    lo, med, hi = torch.nanquantile(sample, q, dim=1)
    delta = hi - lo
    delta[delta < 1e-5] = 1e-5
    ns = 2.0 * 0.7413 * delta
    ref = torch.isfinite(sample) & (sample < (med + ns).reshape(-1, 1)) & (sample > (med - ns).reshape(-1, 1))
    cpu_s = time.perf_counter() - t0
    same = float((ref.numpy() == valid[:n_cpu].cpu().numpy().astype(bool)).mean())
    print(json.dumps({
        "kernel": "kb_sigma_g_clip_kernel", "rows": args.rows, "cols": args.cols, "ms": ms,
        "curves_per_s": args.rows / (ms * 1e-3),
        "roofline": {"bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": alg / (ms * 1e-3) / 1e9 / 8000.0},
        "cpu_baseline": {"value": n_cpu / cpu_s, "unit": "curves/s", "cores": torch.get_num_threads(), "kind": "reference",
                         "sample": f"torch.nanquantile sequence of sigma_g_filter.py:132-165 on {n_cpu} curves"},
        "agreement_with_torch_cpu": same,
    }))


def bench_coadds(n=100_000, T=64, H=512, W=512, radius=10, steps=3):
    """kb_coadd_stamps on device-resident stacks and positions; algorithmic bytes = the stamp pixels
    read (4 B each, twice for the weighted coadd) + the float32 coadd written."""
    import torch

    from bench import check, load_lib
    from oracle import post_search as ps

    lib = load_lib()
    lib.kb_coadd_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(2)
    sci = (rng.standard_normal((T, H, W)) * 2).astype(np.float32)
    sci[rng.random((T, H, W)) < 0.01] = np.nan
    var = np.full((T, H, W), 4.0, dtype=np.float32)
    times = np.arange(T) / T
    x0, y0 = rng.integers(0, W, n), rng.integers(0, H, n)
    vx, vy = rng.uniform(-40, 40, n), rng.uniform(-40, 40, n)
    xv = ps.predict_pixel_locations(times, x0, vx).astype(np.int32)
    yv = ps.predict_pixel_locations(times, y0, vy).astype(np.int32)
    d_sci, d_var = torch.from_numpy(sci).to(dev), torch.from_numpy(var).to(dev)
    d_x, d_y = torch.from_numpy(xv).to(dev), torch.from_numpy(yv).to(dev)
    S = 2 * radius + 1
    out = torch.empty((n, S, S), dtype=torch.float32, device=dev)
    res = {}
    for name, code in (("sum", 0), ("mean", 1), ("median", 2), ("weighted", 3)):
        def step():
            check(lib, lib.kb_coadd_stamps(d_sci.data_ptr(), d_var.data_ptr(), T, H, W, d_x.data_ptr(), d_y.data_ptr(),
                                           None, n, radius, code, out.data_ptr(), None))
        step()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(steps):
            step()
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / steps
        alg = n * T * S * S * 4 * (2 if name == "weighted" else 1) + n * S * S * 4
        res[name] = {"ms": ms, "trajectories_per_s": n / (ms * 1e-3),
                     "roofline": {"bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                  "frac": alg / (ms * 1e-3) / 1e9 / 8000.0}}
    n_cpu = 100
    t0 = time.perf_counter()
    exp = ps.coadds_for_trajectories(sci, var, xv[:n_cpu], yv[:n_cpu], None, radius, ["sum", "mean", "median", "weighted"])
    cpu_s = time.perf_counter() - t0
    check(lib, lib.kb_coadd_stamps(d_sci.data_ptr(), d_var.data_ptr(), T, H, W, d_x.data_ptr(), d_y.data_ptr(), None, n,
                                   radius, 3, out.data_ptr(), None))
    same = bool(np.array_equal(out[:n_cpu].cpu().numpy(), exp["weighted"]))
    print(json.dumps({"kernel": "kb_coadd_kernel", "trajectories": n, "frames": T, "radius": radius, "types": res,
                      "cpu_baseline": {"value": n_cpu / cpu_s, "unit": "trajectories/s (all four coadds)", "cores": 1,
                                       "kind": "port", "sample": f"oracle loop of append_coadds on {n_cpu} trajectories"},
                      "weighted_equals_oracle_on_sample": same}))


def bench_grid_filter(n=2_000_000):
    """kb_grid_filter on n result trajectories resident in HBM (28 B read per trajectory, 4 B per survivor)."""
    import torch

    from bench import check, load_lib
    from oracle import post_search as ps

    lib = load_lib()
    lib.kb_grid_filter.argtypes = [C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]
    rng = np.random.default_rng(8)
    rec = np.zeros(n, dtype=[("vx", "f4"), ("vy", "f4"), ("lh", "f4"), ("flux", "f4"), ("x", "i4"), ("y", "i4"), ("obs", "i4")])
    rec["x"], rec["y"] = rng.integers(0, 2048, n), rng.integers(0, 2048, n)
    rec["vx"], rec["vy"] = rng.uniform(-40, 40, n), rng.uniform(-40, 40, n)
    rec["lh"] = rng.uniform(5, 50, n)
    dev = torch.device("cuda", 0)
    d = torch.from_numpy(rec.view(np.uint8)).to(dev)
    kept = torch.empty(n, dtype=torch.int32, device=dev)
    cnt = C.c_uint64(0)

    def step():
        check(lib, lib.kb_grid_filter(d.data_ptr(), n, 10.0, 1.0, kept.data_ptr(), C.byref(cnt), None))

    step()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(5):
        step()
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 5
    n_cpu = 200_000
    t0 = time.perf_counter()
    exp = ps.grid_filter_indices(rec["x"][:n_cpu], rec["y"][:n_cpu], rec["vx"][:n_cpu], rec["vy"][:n_cpu], rec["lh"][:n_cpu])
    cpu_s = time.perf_counter() - t0
    check(lib, lib.kb_grid_filter(d.data_ptr(), n_cpu, 10.0, 1.0, kept.data_ptr(), C.byref(cnt), None))
    same = kept[:cnt.value].cpu().numpy().tolist() == exp
    print(json.dumps({"kernel": "kb_grid_filter (3 radix sorts + scan)", "trajectories": n, "ms": ms,
                      "trajectories_per_s": n / (ms * 1e-3),
                      "roofline": {"bound": "hbm", "achieved": n * 32 / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                   "frac": n * 32 / (ms * 1e-3) / 1e9 / 8000.0},
                      "cpu_baseline": {"value": n_cpu / cpu_s, "unit": "trajectories/s", "cores": 1, "kind": "port",
                                       "sample": f"oracle dictionary loop on {n_cpu} trajectories"},
                      "equals_oracle_on_sample": same}))


def bench_filter_sort(n=2_097_152, keep_fraction=1.0):
    """kb_filter_sort_results (stack_search.cpp:266-281 in HBM) on n result slots: select + radix sort of (lh, index) + gather.
    Algorithmic bytes: 28 read + 28 written per kept record, 28 read per dropped one."""
    import torch

    from bench import check, load_lib

    lib = load_lib()
    lib.kb_filter_sort_results.argtypes = [C.c_void_p, C.c_uint64, C.c_float, C.c_int32, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]
    rng = np.random.default_rng(9)
    rec = np.zeros(n, dtype=[("vx", "f4"), ("vy", "f4"), ("lh", "f4"), ("flux", "f4"), ("x", "i4"), ("y", "i4"), ("obs", "i4")])
    rec["lh"] = rng.normal(5, 3, n)
    rec["obs"] = 64
    min_lh = float(np.quantile(rec["lh"], 1.0 - keep_fraction)) if keep_fraction < 1.0 else -1.0e30
    dev = torch.device("cuda", 0)
    d = torch.from_numpy(rec.view(np.uint8)).to(dev)
    out = torch.empty_like(d)
    cnt = C.c_uint64(0)

    def step():
        check(lib, lib.kb_filter_sort_results(d.data_ptr(), n, min_lh, 0, out.data_ptr(), C.byref(cnt), None))

    step()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(5):
        step()
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 5
    kept = int(cnt.value)
    got = out.cpu().numpy().view(rec.dtype)[:kept]
    order = np.argsort(-rec["lh"][rec["lh"] >= min_lh], kind="stable")
    same = bool(np.array_equal(got["lh"], rec["lh"][rec["lh"] >= min_lh][order]))
    alg = n * 28 + kept * 28
    print(json.dumps({"kernel": "kb_filter_sort_results (select + radix sort + gather)", "slots": n, "kept": kept, "ms": ms,
                      "roofline": {"bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                   "frac": alg / (ms * 1e-3) / 1e9 / 8000.0, "frac_of_achievable": alg / (ms * 1e-3) / 1e9 / 6300.0},
                      "sorted_like_numpy_stable": same}))


if __name__ == "__main__":
    main()
    bench_coadds()
    bench_grid_filter()
    bench_filter_sort()
    bench_filter_sort(keep_fraction=0.02)
