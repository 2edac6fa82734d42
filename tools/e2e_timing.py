import sys, time
sys.path.insert(0, '.')
import numpy as np
import kbmod_amd.search as kb
from kbmod_amd import fake_data as fd
rng = np.random.default_rng(1)
T, H, W = 64, 512, 512
st = fd.make_fake_image_stack(H, W, np.arange(T) / T, 2.0, 1.0, rng=rng)
t0 = time.perf_counter(); s = kb.StackSearch(st.sci, st.var, st.psfs, st.zeroed_times); t1 = time.perf_counter()
vx, vy = fd.kbmod_v1_candidates(32, 5, 40, 32, 0, 1.5)
c = [kb.Trajectory(vx=float(a), vy=float(b)) for a, b in zip(vx, vy)]
for i in range(3):
    t2 = time.perf_counter(); s.search_all(c, True); t3 = time.perf_counter()
    print(f"ctor {t1-t0:.3f}s search_all {t3-t2:.3f}s results {s.get_number_total_results()} kernel_ms {s.last_search_stats()['search_kernel_ms']:.2f}")
s.set_min_lh(10.0)
t2 = time.perf_counter(); s.search_all(c, True); t3 = time.perf_counter()
print(f"min_lh=10: search_all {t3-t2:.3f}s results {s.get_number_total_results()}")
