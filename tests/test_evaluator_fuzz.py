"""The oracle's (and the product's host instantiation of the) trajectory evaluator and sigma-G index routine against
vectors produced by the REFERENCE's own device evaluator: kernels/kernels.cu:31-242, host-compiled with hipcc in the
build container (tests/golden/make_evaluator_fuzz.py, oracle/ref_kernels_driver.cpp).  10 020 random trajectories over
float, uint16 and uint8 arrays, dyadic and irregular time stamps, sigma-G off and on, NaN pixels, phi == 0, trajectories
that leave the image -- every field bit for bit.  Labelled corroboration: one macro stands in for the CUDA runtime in
that build, so the formal pin stays with the reference's known answers (tests/test_oracle_kat.py)."""
import ctypes as C
import os

import numpy as np
import pytest

from kbmod_amd import capi

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "evaluator_fuzz.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _same_bits(a, b):
    return a.tobytes() == b.tobytes()


def test_oracle_evaluator_equals_reference_kernel_code(orc, gold):
    n_total = n_clipped = 0
    for si, pi in gold["cases"]:
        m = capi.Meta.from_buffer_copy(gold[f"meta_{si}"].tobytes())
        p = capi.Params.from_buffer_copy(gold[f"params_{si}_{pi}"].tobytes())
        pp = orc.PsiPhi.__new__(orc.PsiPhi)
        pp.meta = orc.Meta(m.num_times, m.width, m.height, m.num_bytes, m.psi_min_val, m.psi_max_val, m.psi_scale,
                           m.phi_min_val, m.phi_max_val, m.phi_scale)
        pp.array = gold[f"array_{si}"]
        pp.times = gold[f"times_{si}"]
        pp.T, pp.H, pp.W, pp.nb = int(m.num_times), int(m.height), int(m.width), int(m.num_bytes)
        op = pp.default_params(min_observations=p.min_observations, min_lh=p.min_lh, do_sigmag_filter=int(p.do_sigmag_filter),
                               sgl_L=p.sgl_L, sgl_H=p.sgl_H, sigmag_coeff=p.sigmag_coeff)
        trj, ref = gold[f"trj_{si}_{pi}"], gold[f"ref_{si}_{pi}"]
        got = np.array([pp.evaluate_kernel(int(t["x"]), int(t["y"]), float(t["vx"]), float(t["vy"]), op, 200) for t in trj],
                       dtype=orc.TRJ_DTYPE)
        assert _same_bits(got, ref), (si, pi, np.flatnonzero([g.tobytes() != r.tobytes() for g, r in zip(got, ref)])[:5])
        n_total += len(trj)
        n_clipped += int(p.do_sigmag_filter) * int((ref["obs_count"] >= max(1, p.min_observations)).sum())
    assert n_total >= 10000 and n_clipped > 1000  # the sigma-G branch was exercised, not just reached


def test_product_host_evaluator_equals_reference_kernel_code(gold):
    """kb_evaluate_trajectory_host (the host instantiation of the evaluator the device epilogues run, csrc/search_math.h)."""
    lib = capi.load_lib()
    lib.kb_evaluate_trajectory_host.argtypes = [C.POINTER(capi.Meta), C.c_void_p, C.c_void_p, capi.Params, C.c_void_p]
    for si, pi in gold["cases"]:
        m = capi.Meta.from_buffer_copy(gold[f"meta_{si}"].tobytes())
        p = capi.Params.from_buffer_copy(gold[f"params_{si}_{pi}"].tobytes())
        arr, times = np.ascontiguousarray(gold[f"array_{si}"]), np.ascontiguousarray(gold[f"times_{si}"])
        got = gold[f"trj_{si}_{pi}"].copy()
        for i in range(len(got)):
            capi.check(lib.kb_evaluate_trajectory_host(C.byref(m), arr.ctypes.data, times.ctypes.data, p,
                                                       got[i:i + 1].ctypes.data))
        assert _same_bits(got, gold[f"ref_{si}_{pi}"]), (si, pi)


def test_sigma_g_indices_equal_reference_kernel_code(orc, gold):
    lib = capi.load_lib()
    lib.kb_sigmag_filtered_indices.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                               C.c_void_p, C.c_void_p]
    lib.kb_sigmag_filtered_indices.restype = None
    for vals, (n, lo, hi, coeff), idx, keep in zip(gold["sg_values"], gold["sg_params"], gold["sg_idx"], gold["sg_keep"]):
        n = int(n)
        v = np.ascontiguousarray(vals[:n])
        kept = orc.sigmag_filtered_indices(v, float(lo), float(hi), float(coeff), 2.0)  # indices kept, in sorted order
        assert kept == [int(i) for i in idx[max(0, keep[0]):min(n - 1, keep[1]) + 1]], (n, keep)
        p_idx = np.arange(n, dtype=np.int32)
        a, b = C.c_int(-99), C.c_int(-99)
        vv = v.copy()
        lib.kb_sigmag_filtered_indices(vv.ctypes.data, n, float(lo), float(hi), float(coeff), 2.0, p_idx.ctypes.data,
                                       C.byref(a), C.byref(b))
        assert (a.value, b.value) == tuple(keep) and np.array_equal(p_idx, idx[:n])
