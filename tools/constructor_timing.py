"""StackSearch constructors on the headline stack (64 x 512 x 512): the ingest form from contiguous [T][H][W] stacks --
pageable memory through the library's pinned buffers, the same memory page-locked for the build
(register_host_memory=True), memory that already is pinned (a torch pinned tensor) -- against the reference-style list
constructor.  End to end (allocation, upload over PCIe, correlation); GB/s = the two stacks' bytes / time."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (before the device library: one HIP runtime per process)

import kbmod_amd.search as kb  # noqa: E402
from kbmod_amd import fake_data as fd  # noqa: E402

rng = np.random.default_rng(1)
T, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 512, 512)
sci = (rng.standard_normal((T, H, W)) * 2).astype(np.float32)
var = np.full((T, H, W), 4.0, dtype=np.float32)
sci_pin, var_pin = torch.from_numpy(sci).pin_memory(), torch.from_numpy(var).pin_memory()
psf = fd.make_gaussian_kernel(1.0)
times = list(np.arange(T) / T)
gb = 2 * sci.nbytes / 1e9


def timed(fn):
    t0 = time.perf_counter()
    s = fn()
    dt = time.perf_counter() - t0
    del s
    return dt


for rep in range(3):
    rows = [
        ("pageable, staged", timed(lambda: kb.StackSearch.from_image_stacks(sci, var, [psf] * T, times, register_host_memory=False))),
        ("pageable, registered (default)", timed(lambda: kb.StackSearch.from_image_stacks(sci, var, [psf] * T, times))),
        ("already pinned", timed(lambda: kb.StackSearch.from_image_stacks(sci_pin.numpy(), var_pin.numpy(), [psf] * T, times))),
        ("separable", timed(lambda: kb.StackSearch.from_image_stacks(sci, var, [psf] * T, times, separable_psf=True))),
        ("list constructor", timed(lambda: kb.StackSearch([x for x in sci], [x for x in var], [psf] * T, times))),
    ]
    print(" | ".join(f"{name} {1e3 * dt:.1f} ms ({gb / dt:.1f} GB/s)" for name, dt in rows))
