"""End-to-end time of StackSearch.search_all(on_gpu=True) at cfg2 (what a caller of the pybind surface sees: search,
filter + sort in HBM, download of the survivors into the host list).  Other sizes: search_all_timing.py T H W VEL_STEPS ANG_STEPS."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (first: two copies of the HIP runtime cannot both initialise)
import kbmod_amd.search as kb
from kbmod_amd import fake_data as fd

rng = np.random.default_rng(1)
T, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 512, 512)
VS, AS = (int(v) for v in sys.argv[4:6]) if len(sys.argv) > 5 else (32, 32)
sci = (rng.standard_normal((T, H, W)) * 2).astype(np.float32)
var = np.full((T, H, W), 4.0, dtype=np.float32)
psf = fd.make_gaussian_kernel(1.0)
s = kb.StackSearch.from_image_stacks(sci, var, [psf] * T, list(np.arange(T) / T))
s.preload_psi_phi_array()
vx, vy = fd.kbmod_v1_candidates(VS, 5.0, 40.0, AS, 0.0, 1.5)
cands = [kb.Trajectory(0, 0, float(a), float(b)) for a, b in zip(vx, vy)]
for min_lh in (0.0, 10.0, "sigmag 10"):
    if isinstance(min_lh, str):
        s.enable_gpu_sigmag_filter([0.25, 0.75], 0.7413, 10.0)  # the in-search sigma-G filter (BASELINE configs[2])
    else:
        s.set_min_lh(min_lh)
    for rep in range(4):
        t0 = time.perf_counter()
        s.search_all(cands, True)
        t1 = time.perf_counter()
        n = s.get_number_total_results()
        st = s.last_search_stats()
        print(f"min_lh {min_lh}: search_all {1e3 * (t1 - t0):.2f} ms, {n} results kept, kernel {st['search_kernel_ms']:.2f} ms; inside the call: "
              f"search {st['host_search_ms']:.2f} + filter/sort {st['host_filter_sort_ms']:.2f} + download {st['host_download_ms']:.2f} + "
              f"validity scan {st['host_validate_ms']:.2f} = {st['host_total_ms']:.2f} ms")
