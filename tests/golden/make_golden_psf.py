#!/usr/bin/env python3
"""Generates tests/golden/psf_twin.npz by IMPORTING the reference's Python twin of
the psi/phi builder (runs only in the build container, where /root/reference
exists; the fixture it writes is data and is committed, this script is its
provenance).

Reference code executed (never copied): /root/reference/src/kbmod/core/psf.py
  PSF.make_gaussian_kernel      psf.py:49-74
  convolve_psf_and_image        psf.py:130-199  (masked, renormalised correlation via torch conv2d)
The reference's own parity test pins this twin against the C++ convolution to 4
decimal places (tests/test_python_parity.py:21-69), which is the tolerance
tests/test_golden_psf.py uses.
"""
import importlib.util
import os

import numpy as np

REF = "/root/reference/src/kbmod/core/psf.py"
spec = importlib.util.spec_from_file_location("kbmod_ref_psf", REF)
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)

out = {}
for sigma in (0.5, 0.9, 1.0, 1.2, 2.0):
    out[f"gauss_{sigma}"] = mod.PSF.make_gaussian_kernel(sigma)

rng = np.random.default_rng(20260929)
cases = []
# case 0: the reference parity-test image (tests/test_python_parity.py:22-33)
img = (0.1 * np.arange(40 * 30)).reshape((40, 30)).astype(np.single)
for py, px in [(3, 1), (10, 10), (10, 11), (10, 12), (15, 4)]:
    img[py, px] = np.nan
cases.append((img, mod.PSF.make_gaussian_kernel(1.2)))
# case 1: non-unit kernel (squared Gaussian), as test_convolve_non_unit
cases.append((img.copy(), mod.PSF.make_gaussian_kernel(0.9) ** 2))
# case 2/3: random images with 5 % masked pixels
for sigma, shape in ((1.0, (37, 53)), (2.0, (64, 48))):
    im = rng.normal(0.0, 2.0, shape).astype(np.float32)
    im[rng.random(shape) < 0.05] = np.nan
    cases.append((im, mod.PSF.make_gaussian_kernel(sigma)))
# case 4: asymmetric kernel (orientation matters: correlation, not convolution)
im = rng.normal(10.0, 1.0, (20, 24)).astype(np.float32)
k = np.array([[0.0, 0.0, 0.0], [0.0, 0.5, 0.4], [0.0, 0.1, 0.0]], dtype=np.float32)
cases.append((im, k))

for i, (im, k) in enumerate(cases):
    out[f"img_{i}"] = im
    out[f"psf_{i}"] = k.astype(np.float32)
    out[f"conv_{i}"] = mod.convolve_psf_and_image(im.copy(), k.astype(np.float32), device="cpu")
out["n_cases"] = np.array(len(cases))

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "psf_twin.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, os.path.getsize(dst), "bytes")
