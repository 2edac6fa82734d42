"""One-off wider sweep of tests/test_gpu_fuzz.py's generator: python tools/fuzz_run.py FIRST LAST"""
import sys

sys.path.insert(0, ".")
import numpy as np

import kbmod_amd.search as kb
from oracle import oracle as orc
from tests import util
from tests.test_gpu_fuzz import DIRECT, ENCODED_STAGING, LDS, _config

bad = []
variants = {}
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    stack, vx, vy, cfg, nb = _config(seed)
    a, exp, _ = util.run_both(kb, orc, stack, vx, vy, cfg, num_bytes=nb, flags=DIRECT)
    b, _, s = util.run_both(kb, orc, stack, vx, vy, cfg, num_bytes=nb, flags=LDS)
    v = s.last_search_stats()["kernel_variant"] // 10000
    variants[v] = variants.get(v, 0) + 1
    ok = a.shape == exp.shape and np.array_equal(a, exp) and np.array_equal(b, exp)
    if nb != -1:
        c, _, _ = util.run_both(kb, orc, stack, vx, vy, cfg, num_bytes=nb, flags=LDS | ENCODED_STAGING)
        ok = ok and np.array_equal(c, exp)
    if not ok:
        bad.append(seed)
print("seeds", sys.argv[1], sys.argv[2], "variants", variants, "mismatches", bad)
