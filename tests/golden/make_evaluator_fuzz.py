#!/usr/bin/env python3
"""tests/golden/evaluator_fuzz.npz: 10^4 random trajectories (+ 2000 sigma-G index cases) evaluated by the REFERENCE's own
device evaluator -- /root/reference/src/kbmod/search/kernels/kernels.cu:31-242, host-compiled as HIP source by
`make -C oracle ref_kernels` (oracle/ref_kernels_driver.cpp; one macro maps cudaDeviceSynchronize onto the HIP runtime, so
this is labelled corroboration, not the oracle's formal pin).  Runs only in the build container (needs /root/reference);
the vectors travel, the reference does not.

    python tests/golden/make_evaluator_fuzz.py

Inputs are stored whole (arrays, meta data, times, parameters, trajectories), outputs as the reference wrote them."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from kbmod_amd.capi import Meta, Params  # noqa: E402  (same layouts as PsiPhiArrayMeta / SearchParameters: refk_sizes checks)
from oracle import oracle as orc  # noqa: E402

TRJ = orc.TRJ_DTYPE


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref_kernels"])
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libkbmod_ref_kernels.so"))
    sizes = (C.c_int * 4)()
    ref.refk_sizes(sizes)
    assert list(sizes)[:3] == [C.sizeof(Meta), C.sizeof(Params), TRJ.itemsize], list(sizes)
    ref.refk_evaluate.argtypes = [C.POINTER(Meta), C.c_void_p, C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint64]
    ref.refk_sigmag.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    orc.build()

    rng = np.random.default_rng(20260930)
    out = {}
    cases = []
    # (T, H, W, num_bytes, times kind, NaN fraction)
    stacks = [(24, 20, 28, 4, "dyadic", 0.03), (37, 18, 30, 4, "irregular", 0.10), (32, 20, 28, 2, "dyadic", 0.03),
              (40, 16, 24, 1, "irregular", 0.05), (64, 12, 20, 1, "dyadic", 0.0), (150, 8, 12, 2, "irregular", 0.02)]
    for si, (T, H, W, nb, kind, nanf) in enumerate(stacks):
        psi = rng.normal(0.0, 1.5, (T, H, W)).astype(np.float32)
        phi = (0.25 + rng.random((T, H, W)) * 0.5).astype(np.float32)
        if nb != 4:  # quantised arrays produce equal ratios all the time; plant some in the float ones too
            pass
        else:
            psi[:, :4, :] = np.round(psi[:, :4, :] * 2) / 2
            phi[:, :4, :] = 0.5
        bright = rng.integers(0, H), rng.integers(0, W)
        psi[:, bright[0], bright[1]] += 12.0  # something that passes likelihood thresholds
        psi[rng.random((T, H, W)) < nanf] = np.nan
        phi[rng.random((T, H, W)) < nanf / 2] = np.nan
        if nb == 4:
            phi[rng.random((T, H, W)) < 0.002] = 0.0  # the phi == 0 branch of the ratio (kernels.cu:208)
        times = (np.arange(T) / 16.0) if kind == "dyadic" else np.cumsum(rng.random(T) * 0.11)
        times = np.ascontiguousarray(times - times[0], dtype=np.float64)
        pp = orc.PsiPhi(list(psi), list(phi), times, nb)
        m = Meta(T, W, H, H * W, 2 * T * H * W, pp.array.itemsize, 2 * T * H * W * pp.array.itemsize, nb,
                 pp.meta.psi_min_val, pp.meta.psi_max_val, pp.meta.psi_scale, pp.meta.phi_min_val, pp.meta.phi_max_val,
                 pp.meta.phi_scale)
        out[f"array_{si}"] = pp.array
        out[f"times_{si}"] = times
        out[f"meta_{si}"] = np.frombuffer(bytes(m), dtype=np.uint8).copy()
        # parameter sets: (min_obs, min_lh, sigma-G on, sgl_L, sgl_H, coeff)
        for pi, (min_obs, min_lh, sg, lo, hi, coeff) in enumerate([(0, 0.0, 0, 0.25, 0.75, -1.0), (T // 2, 0.0, 0, 0.25, 0.75, -1.0),
                                                                   (0, -5.0, 1, 0.25, 0.75, 0.7413), (T // 3, 1.0, 1, 0.1, 0.9, 0.3901),
                                                                   (1, 3.0, 1, 0.4, 0.6, 1.9738)]):
            n = 334
            trj = np.zeros(n, dtype=TRJ)
            trj["x"] = rng.integers(-4, W + 4, n)
            trj["y"] = rng.integers(-4, H + 4, n)
            span = float(times[-1]) if times[-1] > 0 else 1.0
            trj["vx"] = (rng.normal(0, 0.6 * W, n) / span).astype(np.float32)
            trj["vy"] = (rng.normal(0, 0.6 * H, n) / span).astype(np.float32)
            trj["vx"][:40] = np.round(trj["vx"][:40] * 2) / 2   # dyadic velocities: exact half pixels on dyadic times
            trj["vy"][:40] = 0.0
            trj["x"][40:60], trj["y"][40:60] = bright[1], bright[0]
            trj["vx"][40:60] = trj["vy"][40:60] = 0.0
            trj["vx"][45:60] = rng.normal(0, 0.3, 15).astype(np.float32)
            trj["lh"], trj["flux"], trj["obs_count"] = 7.0, 7.0, 7  # must all be overwritten
            p = Params(min_obs, min_lh, sg, lo, hi, coeff, -1 if nb == 4 else nb, 0, W, 0, H, 8, 0)
            res = trj.copy()
            ref.refk_evaluate(C.byref(m), pp.array.ctypes.data, times.ctypes.data, C.byref(p), res.ctypes.data, n)
            cases.append((si, pi))
            out[f"params_{si}_{pi}"] = np.frombuffer(bytes(p), dtype=np.uint8).copy()
            out[f"trj_{si}_{pi}"] = trj
            out[f"ref_{si}_{pi}"] = res
    out["cases"] = np.array(cases, dtype=np.int32)

    # SigmaGFilteredIndicesCU (kernels.cu:77-147) on its own: values with ties, every length 1 .. 64, three settings
    sg_vals, sg_par, sg_idx, sg_keep = [], [], [], []
    for k in range(2000):
        n = int(rng.integers(1, 65))
        v = rng.normal(0, 2, n).astype(np.float32)
        if k % 3 == 0:
            v = np.round(v)  # ties
        if k % 7 == 0:
            v[rng.integers(0, n)] = 100.0
        lo, hi, coeff = [(0.25, 0.75, 0.7413), (0.1, 0.9, 0.3901), (0.4, 0.6, 1.9738)][k % 3]
        idx = np.arange(n, dtype=np.int32)
        a, b = C.c_int(-99), C.c_int(-99)
        vv = v.copy()
        ref.refk_sigmag(vv.ctypes.data, n, lo, hi, coeff, 2.0, idx.ctypes.data, C.byref(a), C.byref(b))
        pad = np.full(64, np.nan, np.float32)
        pad[:n] = v
        ipad = np.full(64, -1, np.int32)
        ipad[:n] = idx
        sg_vals.append(pad)
        sg_par.append((n, lo, hi, coeff))
        sg_idx.append(ipad)
        sg_keep.append((a.value, b.value))
    out["sg_values"] = np.stack(sg_vals)
    out["sg_params"] = np.array(sg_par, dtype=np.float64)
    out["sg_idx"] = np.stack(sg_idx)
    out["sg_keep"] = np.array(sg_keep, dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "evaluator_fuzz.npz"), **out)
    n_trj = sum(len(out[f"trj_{a}_{b}"]) for a, b in cases)
    print("wrote evaluator_fuzz.npz:", n_trj, "trajectories,", len(sg_vals), "sigma-G cases,",
          os.path.getsize(os.path.join(HERE, "evaluator_fuzz.npz")), "bytes")


if __name__ == "__main__":
    main()
