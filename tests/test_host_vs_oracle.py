"""CPU-side parity: the product's explicit host paths (search_all(..., False), the
host psi/phi builder, evaluate_single_trajectory(use_kernel=True) = the host
instantiation of the device function) against the oracle, bit for bit, on
seeded inputs including masks, off-image starts and all three encodings."""

import numpy as np
import pytest

from kbmod_amd import fake_data as fd
from tests import util

OBJ = [(17, 12, 21.0, 16.0, 250.0)]


@pytest.fixture(scope="module")
def stack():
    return util.make_stack(16, 48, 56, seed=42, noise=3.0, objects=OBJ, mask_fraction=0.02)


@pytest.mark.parametrize("num_bytes", [-1, 1, 2])
def test_host_builder_equals_oracle(kb, orc, stack, num_bytes):
    arr = kb.PsiPhiArray()
    kb.fill_psi_phi_array_from_image_arrays(arr, num_bytes, stack.sci, stack.var, stack.psfs,
                                            list(stack.zeroed_times), True)
    pp = orc.PsiPhi.from_images(stack.sci, stack.var, stack.psfs, stack.zeroed_times, num_bytes)
    got = arr.encoded_array()
    assert got.dtype == pp.array.dtype
    assert np.array_equal(got.view(np.uint8), pp.array.view(np.uint8))
    if num_bytes != -1:
        assert (arr.psi_min_val, arr.psi_max_val, arr.psi_scale) == (pp.meta.psi_min_val, pp.meta.psi_max_val, pp.meta.psi_scale)
        assert (arr.phi_min_val, arr.phi_max_val, arr.phi_scale) == (pp.meta.phi_min_val, pp.meta.phi_max_val, pp.meta.phi_scale)


@pytest.mark.parametrize("num_bytes", [-1, 1, 2])
@pytest.mark.parametrize("cfg", [{}, {"K": 3, "min_obs": 6}, {"xb": (-6, 60), "yb": (-4, 50), "min_lh": -1e30}])
def test_cpu_search_equals_oracle(kb, orc, stack, num_bytes, cfg):
    vx, vy = fd.kbmod_v1_candidates(6, 5.0, 40.0, 5, 0.0, 1.5)
    got, exp, _ = util.run_both(kb, orc, stack, vx, vy, cfg, num_bytes=num_bytes, on_gpu=False)
    assert got.shape == exp.shape and np.array_equal(got, exp)


def test_cpu_search_fewer_candidates_than_slots(kb, orc, stack):
    vx, vy = fd.velocity_grid_candidates(2, -5.0, 5.0, 2, -3.0, 3.0)
    got, exp, _ = util.run_both(kb, orc, stack, vx, vy, {"min_lh": -1e30}, on_gpu=False)
    assert len(got) == 4 * 48 * 56 and np.array_equal(got, exp)


def test_host_instantiation_of_device_function(kb, orc, stack):
    """kb_evaluate_trajectory_host through the C ABI (needs no device): with and without sigma-G."""
    import ctypes as C
    import os

    from tests.test_abi import ROOT

    lib = C.CDLL(os.path.join(ROOT, "kbmod_amd", "lib", "libkbmod_hip.so"))

    class Meta(C.Structure):
        _fields_ = [(n, C.c_uint64) for n in ("num_times", "width", "height", "pixels_per_image", "num_entries",
                                              "block_size", "total_array_size")] + [
            ("num_bytes", C.c_int32), ("psi_min_val", C.c_float), ("psi_max_val", C.c_float), ("psi_scale", C.c_float),
            ("phi_min_val", C.c_float), ("phi_max_val", C.c_float), ("phi_scale", C.c_float)]

    class Params(C.Structure):
        _fields_ = [("min_observations", C.c_int32), ("min_lh", C.c_float), ("do_sigmag_filter", C.c_uint8),
                    ("sgl_L", C.c_float), ("sgl_H", C.c_float), ("sigmag_coeff", C.c_float),
                    ("encode_num_bytes", C.c_int32), ("x_start_min", C.c_int32), ("x_start_max", C.c_int32),
                    ("y_start_min", C.c_int32), ("y_start_max", C.c_int32), ("results_per_pixel", C.c_uint32),
                    ("total_results", C.c_ulonglong)]

    lib.kb_evaluate_trajectory_host.argtypes = [C.POINTER(Meta), C.c_void_p, C.c_void_p, Params, C.c_void_p]
    rng = np.random.default_rng(8)
    for num_bytes in (4, 1, 2):
        pp = orc.PsiPhi.from_images(stack.sci, stack.var, stack.psfs, stack.zeroed_times, num_bytes)
        m = pp.meta
        meta = Meta(m.num_times, m.width, m.height, m.width * m.height, 2 * m.width * m.height * m.num_times,
                    num_bytes, 2 * m.width * m.height * m.num_times * num_bytes, num_bytes, m.psi_min_val,
                    m.psi_max_val, m.psi_scale, m.phi_min_val, m.phi_max_val, m.phi_scale)
        for sig in (0, 1):
            for _ in range(200):
                x, y = int(rng.integers(-5, 60)), int(rng.integers(-5, 52))
                vx, vy = float(np.float32(rng.normal(0, 25))), float(np.float32(rng.normal(0, 25)))
                min_lh = float(rng.choice([-10.0, 0.0, 2.0]))
                op = pp.default_params(do_sigmag_filter=sig, sigmag_coeff=0.7413, min_lh=min_lh, min_observations=3)
                exp = pp.evaluate_kernel(x, y, vx, vy, op)
                t = np.zeros(1, dtype=orc.TRJ_DTYPE)
                t["x"], t["y"], t["vx"], t["vy"] = x, y, vx, vy
                p = Params(3, min_lh, sig, 0.25, 0.75, 0.7413, -1, 0, 0, 0, 0, 8, 0)
                assert lib.kb_evaluate_trajectory_host(C.byref(meta), pp.array.ctypes.data, pp.times.ctypes.data, p,
                                                       t.ctypes.data) == 0
                assert t[0].tobytes() == exp.tobytes(), (num_bytes, sig, t[0], exp)


def test_psi_phi_curves_equal_oracle(kb, orc, stack):
    s = kb.StackSearch(stack.sci, stack.var, stack.psfs, stack.zeroed_times)
    pp = orc.PsiPhi.from_images(stack.sci, stack.var, stack.psfs, stack.zeroed_times)
    rng = np.random.default_rng(2)
    trjs, exp = [], []
    for _ in range(64):
        x, y = int(rng.integers(-3, 58)), int(rng.integers(-3, 50))
        vx, vy = float(np.float32(rng.normal(0, 20))), float(np.float32(rng.normal(0, 20)))
        trjs.append(kb.Trajectory(x=x, y=y, vx=vx, vy=vy))
        exp.append(pp.curve(x, y, vx, vy))
    got = s.get_all_psi_phi_curves(trjs)
    assert np.array_equal(got, np.stack(exp))
