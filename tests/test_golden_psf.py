"""Golden vectors produced by the reference's Python twin of the psi/phi builder
(tests/golden/make_golden_psf.py imports /root/reference/src/kbmod/core/psf.py and
commits its outputs as psf_twin.npz).  The reference itself pins that twin to the
C++ convolution at 4 decimal places (tests/test_python_parity.py:21-69); the same
tolerance is used here for the oracle and for the product's host convolution."""

import os

import numpy as np
import pytest

from kbmod_amd import fake_data as fd

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "psf_twin.npz"))
N = int(GOLD["n_cases"])


@pytest.mark.parametrize("sigma", [0.5, 0.9, 1.0, 1.2, 2.0])
def test_gaussian_kernel_matches_reference(sigma):
    assert np.array_equal(fd.make_gaussian_kernel(sigma), GOLD[f"gauss_{sigma}"])


def _check(got, exp):
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    m = np.isfinite(exp)
    # assertAlmostEqual(places=4) == |a - b| < 5e-5 on values of magnitude <= ~120
    assert np.allclose(got[m], exp[m], rtol=1e-5, atol=5e-5)


@pytest.mark.parametrize("i", range(N))
def test_oracle_convolution_matches_reference_twin(orc, i):
    _check(orc.convolve(GOLD[f"img_{i}"], GOLD[f"psf_{i}"]), GOLD[f"conv_{i}"])


@pytest.mark.parametrize("i", range(N))
def test_host_convolution_matches_reference_twin(kb, i):
    _check(kb.convolve_image_cpu(GOLD[f"img_{i}"], GOLD[f"psf_{i}"]), GOLD[f"conv_{i}"])


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(N))
def test_device_convolution_matches_reference_twin(kb, i):
    _check(kb.convolve_image_gpu(GOLD[f"img_{i}"], GOLD[f"psf_{i}"]), GOLD[f"conv_{i}"])
