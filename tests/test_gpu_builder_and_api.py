"""GPU tests of the psi/phi builder (HIP, fused prep + correlation + encode) and of the
device-state parts of the API, mirroring the GPU-gated reference tests:
tests/test_image_utils_cpp.py (gpu variants), test_psi_phi_array.py (move_to_gpu),
test_trajectory_list.py (GPU state machine), test_stack_search_results.py:31-37
(preload / unload), test_search.py:127-166 (recovery), test_search_encode.py:67-88,
test_gpu_helpers.py."""

import numpy as np
import pytest

from kbmod_amd import fake_data as fd
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def stack():
    st = util.make_stack(9, 70, 101, seed=31, noise=3.0, psf=1.2, objects=[(20, 20, 12.0, 7.0, 200.0)],
                         mask_fraction=0.03)
    st.var[2][10, 10] = 0.0  # zero variance is masked (image_utils_cpp.cpp:144)
    st.sci[4][30, 31] = np.inf
    st.var[5][40, 41] = -1.0  # negative variance is NOT masked in the C++ reference
    return st


@pytest.mark.parametrize("num_bytes", [-1, 1, 2])
def test_device_builder_is_bit_identical_to_oracle(kb, orc, stack, num_bytes):
    s = kb.StackSearch(stack.sci, stack.var, stack.psfs, stack.zeroed_times, num_bytes)
    arr = s.get_psi_phi_array()
    assert arr.device_resident and not arr.on_gpu  # born in HBM, logical flag as in the reference
    pp = orc.PsiPhi.from_images(stack.sci, stack.var, stack.psfs, stack.zeroed_times, num_bytes)
    got = arr.encoded_array()
    assert got.dtype == pp.array.dtype and np.array_equal(got.view(np.uint8), pp.array.view(np.uint8))
    if num_bytes != -1:
        for a, b in ((arr.psi_min_val, pp.meta.psi_min_val), (arr.psi_max_val, pp.meta.psi_max_val),
                     (arr.psi_scale, pp.meta.psi_scale), (arr.phi_min_val, pp.meta.phi_min_val),
                     (arr.phi_max_val, pp.meta.phi_max_val), (arr.phi_scale, pp.meta.phi_scale)):
            assert np.float32(a).tobytes() == np.float32(b).tobytes()


def test_device_builder_per_epoch_psfs(kb, orc):
    st = util.make_stack(4, 33, 47, seed=2, mask_fraction=0.02)
    st.psfs = [fd.make_gaussian_kernel(s) for s in (0.5, 1.0, 1.5, 2.0)]  # 3x3, 7x7, 9x9, 13x13
    s = kb.StackSearch(st.sci, st.var, st.psfs, st.zeroed_times)
    pp = orc.PsiPhi.from_images(st.sci, st.var, st.psfs, st.zeroed_times)
    assert np.array_equal(s.get_psi_phi_array().encoded_array().view(np.uint32), pp.array.view(np.uint32))


def _device_build(sci, var, psf, num_bytes, build_flags):
    """kb_build_psi_phi_from_device_ex on [T][H][W] stacks -> (meta, host copy of the array bytes).  psf: one kernel for
    every epoch, or a list of T kernels."""
    import ctypes as C

    import torch

    from kbmod_amd import capi

    lib = capi.load_lib()
    T, H, W = sci.shape
    d_sci, d_var = torch.from_numpy(sci).cuda(), torch.from_numpy(var).cuda()
    psfs = list(psf) if isinstance(psf, (list, tuple)) else [psf] * T
    psf_all = np.ascontiguousarray(np.concatenate([np.asarray(k, np.float32).ravel() for k in psfs]))
    dims = np.array([k.shape[0] for k in psfs], dtype=np.int32)
    meta, arr = capi.Meta(), C.c_void_p()
    capi.check(lib.kb_build_psi_phi_from_device_ex(d_sci.data_ptr(), d_var.data_ptr(), psf_all.ctypes.data, dims.ctypes.data, T, H,
                                                   W, num_bytes, build_flags, C.byref(meta), C.byref(arr),
                                                   torch.cuda.current_stream().cuda_stream))
    host = np.empty(int(meta.total_array_size), dtype=np.uint8)
    capi.check(lib.kb_copy_block_to_cpu(host.ctypes.data, arr, meta.total_array_size))
    lib.kb_free_gpu_block(arr)
    return meta, host


@pytest.mark.parametrize("sigma", [0.5, 1.0, 1.4])  # 3 x 3, 7 x 7, 9 x 9 kernels: the strip kernel's sizes
@pytest.mark.parametrize("num_bytes", [-1, 2])
def test_strip_builder_variances_over_many_decades(orc, sigma, num_bytes):
    """The strip kernel (64 x 32 tiles, default for one kernel size 3 .. 9) against the oracle AND against the general
    tile kernel, byte for byte, on a stack whose variances span 24 decades -- denormal reciprocals, overflowing
    sci / var quotients, negative and zero variances, NaN and infinities included: phi0 = (float)(1.0 / (double)var)
    of the reference is computed there as the correctly rounded float quotient (innocuous double rounding), and
    masked taps add +-0 instead of being skipped.  Sizes that are no multiple of the tile, tiles with and without
    NO_DATA."""
    rng = np.random.default_rng(int(sigma * 10) + 7)
    T, H, W = 5, 75, 150  # (clean interior tiles exist: 64 x 32 blocks with their halo inside and unmasked)
    sci = rng.normal(0, 3, (T, H, W)).astype(np.float32)
    var = (10.0 ** rng.uniform(-12, 12, (T, H, W))).astype(np.float32)
    var[0, :40, :140] = 4.0                       # a clean region (plain survey values)
    var[1] = (10.0 ** rng.uniform(36, 38.5, (H, W))).astype(np.float32)   # reciprocals in the denormal range
    var[2, 50:, :] = (10.0 ** rng.uniform(-44, -38, (H - 50, W))).astype(np.float32)  # denormal variances: infinite quotients
    sci[3, 5:9, 70:80] = np.nan
    sci[3, 60, 100] = np.inf
    var[3, 20, 20], var[3, 21, 20], var[3, 22, 22] = 0.0, -3.0, np.inf
    var[4, ::7, ::5] = np.nan
    psf = fd.make_gaussian_kernel(sigma)
    times = np.arange(T, dtype=np.float64)
    pp = orc.PsiPhi.from_images([s for s in sci], [v for v in var], [psf] * T, times, 4 if num_bytes == -1 else num_bytes)
    meta_s, strip = _device_build(sci, var, psf, num_bytes, 0)
    meta_g, general = _device_build(sci, var, psf, num_bytes, 4)  # KB_BUILD_GENERAL_TILES
    assert np.array_equal(strip, general)
    assert np.array_equal(strip, pp.array.view(np.uint8).ravel())
    for name in ("psi_min_val", "psi_max_val", "psi_scale", "phi_min_val", "phi_max_val", "phi_scale"):
        assert np.float32(getattr(meta_s, name)).tobytes() == np.float32(getattr(pp.meta, name)).tobytes(), name
        assert np.float32(getattr(meta_s, name)).tobytes() == np.float32(getattr(meta_g, name)).tobytes(), name


@pytest.mark.parametrize("kind", ["mirrored_rows_only", "no_symmetry", "one_epoch_breaks_it"])
@pytest.mark.parametrize("dim", [3, 5, 7])
def test_strip_builder_shares_products_only_between_mirrored_kernel_rows(orc, kind, dim):
    """The strip kernel multiplies a sample once for two kernel rows that hold the same bits (strip_pass, VSYM) -- when EVERY
    epoch's kernel reads the same top to bottom as bottom to top.  Kernels that mirror top-bottom but not left-right, kernels
    without any symmetry, and a stack in which one epoch's kernel differs from its mirror image in a single bit: the array
    equals the oracle's and the general tile kernel's byte for byte either way."""
    rng = np.random.default_rng(100 * dim + len(kind))
    T, H, W = 4, 70, 140
    sci = rng.normal(0, 3, (T, H, W)).astype(np.float32)
    var = rng.uniform(0.5, 8.0, (T, H, W)).astype(np.float32)
    sci[1, 10:14, 60:75] = np.nan     # tiles with and without NO_DATA
    var[2, 40, ::9] = 0.0
    kernels = []
    for t in range(T):
        k = rng.uniform(0.0, 1.0, (dim, dim)).astype(np.float32)
        if kind != "no_symmetry":
            k[dim // 2 + 1:] = k[:dim // 2][::-1]   # row j == row dim - 1 - j, columns unrelated
        kernels.append(k / k.sum(dtype=np.float32))
    if kind != "no_symmetry":
        for k in kernels:
            k[dim // 2 + 1:] = k[:dim // 2][::-1]   # (the normalisation kept the rows equal; make sure of the bits)
    if kind == "one_epoch_breaks_it":
        kernels[2][0, 1] = np.nextafter(kernels[2][0, 1], np.float32(2.0))
    times = np.arange(T, dtype=np.float64)
    pp = orc.PsiPhi.from_images([x for x in sci], [v for v in var], kernels, times, 4)
    _, strip = _device_build(sci, var, kernels, -1, 0)
    _, general = _device_build(sci, var, kernels, -1, 4)  # KB_BUILD_GENERAL_TILES
    assert np.array_equal(strip, pp.array.view(np.uint8).ravel())
    assert np.array_equal(strip, general)


def test_all_nan_stack_is_an_error_when_encoding(kb):
    st = util.make_stack(3, 8, 9, seed=1)
    for im in st.sci:
        im[:, :] = np.nan
    with pytest.raises(RuntimeError):
        kb.StackSearch(st.sci, st.var, st.psfs, st.zeroed_times, 2)


def test_convolve_image_gpu_known_answers(kb, orc):
    a = np.arange(0, 120, dtype=np.single).reshape(12, 10)
    ident = np.zeros((3, 3), dtype=np.single)
    ident[1, 1] = 1.0
    assert np.allclose(kb.convolve_image_gpu(a, ident), a, 0.0001)
    for y, x in [(0, 3), (5, 6), (5, 7)]:
        a[y, x] = np.nan
    p = fd.make_gaussian_kernel(1.0)
    r = kb.convolve_image_gpu(a, p)
    assert np.array_equal(np.isfinite(r), np.isfinite(a))
    # same bits as the oracle's device flavour and as the host loop
    assert np.array_equal(r.view(np.uint32), orc.convolve(a, p, True).view(np.uint32))
    assert np.array_equal(np.nan_to_num(r), np.nan_to_num(kb.convolve_image_cpu(a, p)))
    # empty PSF footprint: 0.0 on the device path (image_kernels.cu:61)
    ones = np.ones((3, 3), dtype=np.float32)
    assert np.all(kb.convolve_image_gpu(ones, np.zeros((3, 3), dtype=np.float32)) == 0.0)
    asym = np.array([[0.0, 0.0, 0.0], [0.0, 0.5, 0.4], [0.0, 0.1, 0.0]], dtype=np.float32)
    b = np.arange(0, 120, dtype=np.single).reshape(12, 10)
    assert np.array_equal(kb.convolve_image_gpu(b, asym), orc.convolve(b, asym, True))
    # convolve_image / generate_psi / generate_phi dispatch to the device when one is present
    assert np.array_equal(kb.convolve_image(b, asym), kb.convolve_image_gpu(b, asym))
    sci, var = stack_pair()
    assert np.array_equal(np.nan_to_num(kb.generate_psi(sci, var, p)), np.nan_to_num(orc.generate_psi(sci, var, p, True)))
    assert np.array_equal(np.nan_to_num(kb.generate_phi(var, p)), np.nan_to_num(orc.generate_phi(var, p, True)))


def stack_pair():
    rng = np.random.default_rng(12)
    sci = rng.normal(0, 2, (20, 30)).astype(np.float32)
    var = np.full((20, 30), 4.0, dtype=np.float32)
    sci[3, 4] = np.nan
    var[7, 7] = np.nan
    return sci, var


def test_psi_phi_array_gpu_state(kb):
    w, h = 4, 5
    psi = [np.arange(0, w * h, dtype=np.single).reshape(h, w), np.arange(w * h, 2 * w * h, dtype=np.single).reshape(h, w)]
    phi = [np.full((h, w), 0.1, dtype=np.single), np.full((h, w), 0.2, dtype=np.single)]
    for nb in (2, 4):
        arr = kb.PsiPhiArray()
        kb.fill_psi_phi_array(arr, nb, psi, phi, [0.0, 1.0])
        assert arr.cpu_array_allocated and not arr.on_gpu and not arr.gpu_array_allocated
        arr.move_to_gpu()
        assert arr.on_gpu and arr.gpu_array_allocated
        arr.clear_from_gpu()
        assert not arr.on_gpu and not arr.gpu_array_allocated
        assert arr.read_psi_phi(1, 2, 3).phi == pytest.approx(0.2, abs=1e-4)
        arr.clear()
        assert not arr.cpu_array_allocated


def test_trajectory_list_gpu_state_machine(kb):
    lst = kb.TrajectoryList([kb.Trajectory(x=i, lh=float(i)) for i in range(5)])
    assert not lst.on_gpu
    lst.move_to_gpu()
    assert lst.on_gpu
    for call in (lambda: lst.get_trajectory(0), lambda: lst.set_trajectory(0, kb.Trajectory()), lst.get_list,
                 lst.sort_by_likelihood, lambda: lst.resize(3), lambda: lst.get_batch(0, 2), lst.reset_all):
        with pytest.raises(RuntimeError):
            call()
    lst.move_to_gpu()  # no-op
    lst.move_to_cpu()
    assert not lst.on_gpu and [t.x for t in lst.get_list()] == [0, 1, 2, 3, 4]
    lst.move_to_cpu()  # no-op


def test_preload_unload(kb, stack):
    s = kb.StackSearch(stack.sci, stack.var, stack.psfs, stack.zeroed_times)
    assert not s.psi_phi_array_on_gpu()
    s.preload_psi_phi_array()
    assert s.psi_phi_array_on_gpu()
    s.search_all([kb.Trajectory(vx=12.0, vy=7.0)], True)
    assert s.psi_phi_array_on_gpu()  # stays while preloaded
    s.unload_psi_phi_array()
    assert not s.psi_phi_array_on_gpu()
    s.search_all([kb.Trajectory(vx=12.0, vy=7.0)], True)  # re-uploads from the host copy
    assert not s.psi_phi_array_on_gpu() and len(s.get_results(0, 10)) > 0


def test_gpu_helpers(kb):
    assert kb.validate_gpu(0) is True and kb.validate_gpu(2**60) is False
    assert 0 < kb.get_gpu_free_memory() <= kb.get_gpu_total_memory()
    kb.print_cuda_stats()


def test_recovery_full_grid(kb):
    """tests/test_search.py:147-166: 150 x 150 velocity/angle grid on 20 x 80 x 60, single mover."""
    st = util.make_stack(20, 80, 60, seed=100, noise=4.0, psf=1.0, objects=[(17, 12, 21.0, 16.0, 250.0)])
    for i in range(0, 20, 2):
        st.sci[i][5, 6] = np.nan
        st.var[i][5, 6] = np.nan
    s = kb.StackSearch(st.sci, st.var, st.psfs, st.zeroed_times)
    vx, vy = fd.kbmod_v1_candidates(150, 5.0, 40.0, 150, 0.0, 1.5)
    s.search_all(util.trajectories(kb, vx, vy), True)
    res = s.get_results(0, 10 * 8 * 60 * 80)
    assert 0 < len(res) <= 8 * 60 * 80
    best = res[0]
    assert abs(best.x - 17) <= 1 and abs(best.y - 12) <= 1
    assert best.vx / 21.0 == pytest.approx(1, abs=0.1) and best.vy / 16.0 == pytest.approx(1, abs=0.1)
    assert best.flux / 250.0 == pytest.approx(1, abs=0.15)
    t = kb.Trajectory(x=17, y=12, vx=21.0, vy=16.0)
    u = kb.Trajectory(x=17, y=12, vx=21.0, vy=16.0)
    s.evaluate_single_trajectory(t, False)
    s.evaluate_single_trajectory(u, True)  # host instantiation of the device function
    assert (t.lh, t.flux, t.obs_count) == (u.lh, u.flux, u.obs_count)


@pytest.mark.parametrize("num_bytes", [-1, 1, 2])
def test_recovery_different_encodings(kb, num_bytes):
    """tests/test_search_encode.py:67-88."""
    st = util.make_stack(20, 110, 100, seed=101, noise=2.0, psf=1.0, objects=[(33, 5, 12.0, 19.0, 250.0)])
    s = kb.StackSearch(st.sci, st.var, st.psfs, st.zeroed_times, num_bytes)
    s.set_min_obs(10)
    vx, vy = fd.kbmod_v1_candidates(150, 5.0, 40.0, 150, 0.0, 1.5)
    s.search_all(util.trajectories(kb, vx, vy), True)
    best = s.get_results(0, 10)[0]
    assert abs(best.x - 33) <= 1 and abs(best.y - 5) <= 1
    assert best.vx / 12.0 == pytest.approx(1, abs=0.1) and best.vy / 19.0 == pytest.approx(1, abs=0.1)
    assert best.flux / 250.0 == pytest.approx(1, abs=0.25)


def test_off_chip_start(kb):
    """tests/test_search.py:233-269: object entering from x = -3."""
    st = util.make_stack(20, 80, 60, seed=100, noise=4.0, psf=1.0, objects=[(-3, 12, 25.0, 10.0, 250.0)])
    s = kb.StackSearch(st.sci, st.var, st.psfs, st.zeroed_times)
    s.set_start_bounds_x(-10, 70)
    s.set_start_bounds_y(-10, 90)
    vx, vy = fd.kbmod_v1_candidates(150, 5.0, 40.0, 150, 0.0, 1.5)
    s.search_all(util.trajectories(kb, vx, vy), True)
    best = s.get_results(0, 10)[0]
    assert abs(best.x + 3) <= 1 and abs(best.y - 12) <= 1
    assert best.vx / 25.0 == pytest.approx(1, abs=0.1) and best.vy / 10.0 == pytest.approx(1, abs=0.1)


@pytest.mark.parametrize("num_bytes", [-1, 1, 2])
def test_psi_phi_curves_on_device_equal_oracle(kb, orc, stack, num_bytes):
    s = kb.StackSearch(stack.sci, stack.var, stack.psfs, stack.zeroed_times, num_bytes)
    assert s.get_psi_phi_array().device_resident
    pp = orc.PsiPhi.from_images(stack.sci, stack.var, stack.psfs, stack.zeroed_times, num_bytes)
    rng = np.random.default_rng(4)
    trjs, exp = [], []
    for _ in range(300):
        x, y = int(rng.integers(-4, 105)), int(rng.integers(-4, 74))
        vx, vy = float(np.float32(rng.normal(0, 25))), float(np.float32(rng.normal(0, 25)))
        trjs.append(kb.Trajectory(x=x, y=y, vx=vx, vy=vy))
        exp.append(pp.curve(x, y, vx, vy))
    got = s.get_all_psi_phi_curves(trjs)
    assert got.shape == (300, 18) and np.array_equal(got, np.stack(exp))
    assert s.get_all_psi_phi_curves([]).shape == (0, 18)
