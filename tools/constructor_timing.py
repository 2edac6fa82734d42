import time, numpy as np, sys
sys.path.insert(0, '/root/repo')
import kbmod_amd.search as kb
from kbmod_amd import fake_data as fd
rng = np.random.default_rng(1)
T, H, W = 64, 512, 512
sci = (rng.standard_normal((T, H, W)) * 2).astype(np.float32)
var = np.full((T, H, W), 4.0, dtype=np.float32)
psf = fd.make_gaussian_kernel(1.0)
times = list(np.arange(T) / T)
for rep in range(3):
    t0 = time.perf_counter(); s = kb.StackSearch.from_image_stacks(sci, var, [psf]*T, times); t1 = time.perf_counter()
    s2 = kb.StackSearch([x for x in sci], [x for x in var], [psf]*T, times); t2 = time.perf_counter()
    s3 = kb.StackSearch.from_image_stacks(sci, var, [psf]*T, times, separable_psf=True); t3 = time.perf_counter()
    print(f"from_image_stacks {1e3*(t1-t0):.1f} ms   list constructor {1e3*(t2-t1):.1f} ms   separable {1e3*(t3-t2):.1f} ms")
    del s, s2, s3
