import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import kbmod_amd.search as kb
from kbmod_amd import fake_data as fd
rng = np.random.default_rng(1)
T, H, W = 64, 512, 512
sci = (rng.standard_normal((T, H, W)) * 2).astype(np.float32)
var = np.full((T, H, W), 4.0, dtype=np.float32)
psf = fd.make_gaussian_kernel(1.0)
vx, vy = fd.kbmod_v1_candidates(32, 5.0, 40.0, 32, 0.0, 1.5)
for rep in range(2):
    t0 = time.perf_counter()
    s = kb.StackSearch.from_image_stacks(sci, var, [psf] * T, list(np.arange(T) / T))
    t1 = time.perf_counter()
    cands = [kb.Trajectory(0, 0, float(a), float(b)) for a, b in zip(vx, vy)]
    t2 = time.perf_counter()
    s.search_all(cands, True)
    t3 = time.perf_counter()
    st = s.last_search_stats()
    print(f"rep {rep}: constructor {1e3*(t1-t0):.1f} ms, candidates {1e3*(t2-t1):.1f} ms, search_all {1e3*(t3-t2):.1f} ms (inside {st['host_total_ms']:.1f}), total {1e3*(t3-t0):.1f} ms")
    del s
