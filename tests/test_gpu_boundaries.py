"""GPU parity tests at boundaries of the default path that earlier rounds never reached:

* 65 534 / 65 535 / 65 536 candidates: the headline list form keeps ``candidate | obs_count << 16`` in one register,
  so the host switches forms at 65 535 candidates (csrc/search_kernels.hip, ``packable``); a 256 x 256 velocity grid is
  exactly 65 536;
* the ingest constructor (StackSearch.from_image_stacks) against the ORACLE's array directly, incl. the separable PSF
  build at the north star's 1e-4 and the device builder's empty-footprint convention.
"""

import numpy as np
import pytest

from kbmod_amd import fake_data as fd
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small():
    ds_stack = util.make_stack(8, 24, 70, seed=31, noise=3.0, psf=1.0, objects=[(10, 8, 12.0, 5.0, 240.0)], mask_fraction=0.02)
    d = util.DeviceStack(ds_stack)
    yield ds_stack, d
    d.close()


@pytest.fixture(scope="module")
def velocity_grid():
    # 256 x 256 = 65 536 candidates; the slow ones coincide on the pixel grid (ties, in candidate order)
    return fd.velocity_grid_candidates(256, -30.0, 30.0, 256, -30.0, 30.0)


@pytest.mark.parametrize("n_cands", [65534, 65535, 65536])
@pytest.mark.parametrize("flags", [2, 4 | 64, 4 | 128])  # kb_search_direct, kb_search_lds 64 x 16, kb_search_lds 64 x 8
def test_list_form_switch_at_65535_candidates(orc, small, velocity_grid, n_cands, flags):
    stack, ds = small
    vx, vy = velocity_grid
    vx, vy = vx[:n_cands], vy[:n_cands]
    got, st = ds.search(ds.params(K=8), ds.candidates(vx, vy), flags)
    assert (st.kernel_variant // 10000 == 0) == (flags == 2)
    pp = orc.PsiPhi.from_images(stack.sci, stack.var, stack.psfs, stack.zeroed_times)
    exp = pp.search_kernel_semantics(orc.make_candidates(vx, vy), pp.default_params(results_per_pixel=8))
    g = util.as_records(got)
    for name in util.FIELDS:
        assert np.array_equal(g[name], exp[name]), (name, n_cands, flags)
    assert len(np.unique(g["vx"])) > 50  # winners from all over the list, also past index 65 533


@pytest.mark.parametrize("n_cands", [65534, 65536])
def test_list_form_switch_with_thresholds_and_sigma_g(orc, small, velocity_grid, n_cands):
    stack, ds = small
    vx, vy = velocity_grid
    vx, vy = vx[:n_cands], vy[:n_cands]
    pp = orc.PsiPhi.from_images(stack.sci, stack.var, stack.psfs, stack.zeroed_times)
    for cfg in (dict(K=3, min_obs=5, min_lh=1.5), dict(K=4, min_obs=4, sigmag=(0.25, 0.75, 0.7413, 2.0))):
        got, _ = ds.search(ds.params(**cfg), ds.candidates(vx, vy), 0)
        exp = pp.search_kernel_semantics(orc.make_candidates(vx, vy), util.oracle_params(pp, cfg))
        g = util.as_records(got)
        for name in util.FIELDS:
            assert np.array_equal(g[name], exp[name]), (name, n_cands, cfg)


# ---------------------------------------------------------------------------------------------------------------
# ingest vs the oracle itself
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def masked_stack():
    st = util.make_stack(9, 50, 84, seed=12, noise=3.0, psf=1.0, objects=[(20, 30, 11.0, 7.0, 220.0)], mask_fraction=0.03)
    st.var[4][10, 11] = np.nan
    st.var[7][20, 60] = 0.0
    st.sci[2][0, 0] = np.inf
    st.var[1][30, 30] = -2.0  # a negative variance is not masked (image_utils_cpp.cpp:142-149)
    return st


def _array_of(search):
    """Every (psi, phi) of the search's array, through read_psi_phi (decoded values; NaN = NO_DATA)."""
    arr = search.get_psi_phi_array()
    T, H, W = arr.num_times, arr.height, arr.width
    out = np.empty((T, H, W, 2), dtype=np.float32)
    for t in range(T):
        for r in range(H):
            for c in range(W):
                v = arr.read_psi_phi(t, r, c)
                out[t, r, c] = (v.psi, v.phi)
    return out


@pytest.mark.parametrize("num_bytes", [-1, 1, 2])
def test_ingest_constructor_equals_the_oracle_array(kb, orc, masked_stack, num_bytes):
    st = masked_stack
    s = kb.StackSearch.from_image_stacks(np.stack(st.sci), np.stack(st.var), st.psfs, st.zeroed_times, num_bytes)
    a = s.get_psi_phi_array()
    assert a.device_resident
    pp = orc.PsiPhi.from_images(st.sci, st.var, st.psfs, st.zeroed_times, 4 if num_bytes == -1 else num_bytes)
    m = pp.meta
    if num_bytes != -1:
        for name in ("psi_min_val", "psi_max_val", "psi_scale", "phi_min_val", "phi_max_val", "phi_scale"):
            assert np.float32(getattr(a, name)).tobytes() == np.float32(getattr(m, name)).tobytes(), name
    got = _array_of(s)
    exp = np.array([[[pp.read(t, r, c) for c in range(pp.W)] for r in range(pp.H)] for t in range(pp.T)], dtype=np.float32)
    assert got.shape == exp.shape and np.array_equal(got.view(np.uint32), exp.view(np.uint32))  # bits, NaN included


def test_separable_ingest_against_the_oracle(kb, orc, masked_stack):
    """The separable PSF build (row pass + column pass over masked values and mask) sums in another order than the
    reference's tap loop: it matches the ORACLE's array to the north star's 1e-4 relative with the identical
    validity pattern."""
    st = masked_stack
    s = kb.StackSearch.from_image_stacks(np.stack(st.sci), np.stack(st.var), st.psfs, st.zeroed_times, separable_psf=True)
    got = _array_of(s)
    pp = orc.PsiPhi.from_images(st.sci, st.var, st.psfs, st.zeroed_times)
    exp = np.array([[[pp.read(t, r, c) for c in range(pp.W)] for r in range(pp.H)] for t in range(pp.T)], dtype=np.float32)
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    ok = ~np.isnan(exp)
    scale = np.maximum(np.abs(exp[ok]), 1e-3 * np.abs(exp[ok]).max())
    assert (np.abs(got[ok] - exp[ok]) / scale).max() < 1e-4  # tolerance of north_star for the float32 path


def test_empty_footprint_is_zero_equals_the_oracle_device_flavour(kb, orc):
    """empty_footprint_is_zero=True is the reference's DEVICE builder (image_kernels.cu:61): byte-compare with the
    oracle's restatement of that flavour, on a stack where the footprint sum does vanish."""
    rng = np.random.default_rng(4)
    T, H, W = 3, 20, 26
    sci = rng.normal(0, 2, (T, H, W)).astype(np.float32)
    var = np.full((T, H, W), 2.0, dtype=np.float32)
    k = np.zeros((3, 3), dtype=np.float32)
    k[1, 0], k[0, 2] = 0.75, 0.25  # no weight on the centre
    sci[:, 5, 4] = np.nan
    sci[:, 4, 6] = np.nan  # both weighted neighbours of (5, 5) masked
    sci[1, 10:14, 3:9] = np.nan
    times = [0.0, 1.0, 2.0]
    for flavour in (False, True):
        s = kb.StackSearch.from_image_stacks(sci, var, [k] * T, times, empty_footprint_is_zero=flavour)
        got = _array_of(s)
        pp = orc.PsiPhi.from_images([x for x in sci], [v for v in var], [k] * T, times, gpu_flavour=flavour)
        exp = np.array([[[pp.read(t, r, c) for c in range(W)] for r in range(H)] for t in range(T)], dtype=np.float32)
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), flavour
    assert got[0, 5, 5, 0] == 0.0  # (the device flavour writes 0 there, the CPU flavour NaN)


def test_padded_copy_of_a_library_built_array_is_made_once(kb, orc):
    """An array kb_build_psi_phi_* built is the library's own: the padded canonical copy its first search makes stands for the
    following searches without the caller's flag 256 -- until a library call writes into the array, the array is freed (and
    another takes its address), or flag 2048 asks for a fresh copy.  Results never change, only whether the pad pass ran."""
    import ctypes as C

    import torch

    from kbmod_amd import capi

    st = util.make_stack(16, 70, 130, seed=77, objects=[(20, 30, 15.0, 6.0, 300.0)], mask_fraction=0.01, times=np.arange(16) / 16.0)
    vx, vy = fd.kbmod_v1_candidates(8, 5.0, 20.0, 8, 0.0, 0.5)
    d = util.DeviceStack(st)
    cands = d.candidates(vx, vy)
    p = d.params(K=8)
    first, s1 = d.search(p, cands, 4)
    again, s2 = d.search(p, cands, 4)
    fresh, s3 = d.search(p, cands, 4 | 2048)
    assert (s1.padded_copy_reused, s2.padded_copy_reused, s3.padded_copy_reused) == (0, 1, 0)
    assert torch.equal(first, again) and torch.equal(first, fresh)
    # another geometry (start bounds that move the frame): a new copy, then kept again
    p2 = d.params(K=8, xb=(-12, 100), yb=(3, 60))
    _, s4 = d.search(p2, cands, 4)
    _, s5 = d.search(p2, cands, 4)
    assert (s4.padded_copy_reused, s5.padded_copy_reused) == (0, 1)
    # a library call writes into the array: a bright trajectory-long streak at row 40 -- the next search must see it
    T, H, W = d.T, d.H, d.W
    host = np.empty((T, H, W, 2), dtype=np.float32)
    capi.check(d.lib.kb_copy_block_to_cpu(host.ctypes.data, d.arr, host.nbytes))
    host[:, 40, 50, 0] += 500.0
    _, s6 = d.search(p, cands, 4)
    assert s6.padded_copy_reused == 0      # (the geometry changed back)
    _, s7 = d.search(p, cands, 4)
    assert s7.padded_copy_reused == 1
    capi.check(d.lib.kb_copy_block_to_gpu(host.ctypes.data, d.arr, host.nbytes))
    changed, s8 = d.search(p, cands, 4)
    assert s8.padded_copy_reused == 0 and not torch.equal(changed, first)
    direct, _ = d.search(p, cands, 2)      # kb_search_direct reads the array itself
    assert torch.equal(changed, direct)
    # the caller writes by means the library cannot see (a device-to-device copy of its own) and says so:
    # kb_note_array_written renews the array's generation, the copy made from the old one no longer stands
    _, s8b = d.search(p, cands, 4)
    assert s8b.padded_copy_reused == 1
    host[:, 41, 60, 0] += 700.0
    mine = torch.from_numpy(host).to(f"cuda:{torch.cuda.current_device()}")
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(d.arr, mine.data_ptr(), host.nbytes, 3) == 0   # hipMemcpyDeviceToDevice
    torch.cuda.synchronize()
    capi.check(d.lib.kb_note_array_written(C.c_void_p(d.arr.value + 12345)))   # (any address inside the array)
    noted, s8c = d.search(p, cands, 4)
    direct2, _ = d.search(p, cands, 2)
    assert s8c.padded_copy_reused == 0 and torch.equal(noted, direct2) and not torch.equal(noted, changed)
    # freed and rebuilt (most likely at the same address): nothing of the old array's copy is taken over
    d.close()
    st2 = util.make_stack(16, 70, 130, seed=78, objects=[(60, 20, -9.0, 12.0, 300.0)], times=np.arange(16) / 16.0)
    d2 = util.DeviceStack(st2)
    one, s9 = d2.search(p, cands, 4)
    two, _ = d2.search(p, cands, 2)
    assert s9.padded_copy_reused == 0 and torch.equal(one, two)
    d2.close()
