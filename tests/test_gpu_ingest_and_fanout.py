"""GPU tests of the constructor paths and of the fan-out behind search_all:

* StackSearch.from_image_stacks (contiguous [T][H][W] stacks, chunked pinned upload overlapped with the
  correlation) builds the same array, byte for byte, as the reference-style list constructor;
* the separable PSF build (opt-in) agrees with the default 2-D build to 1e-4 relative -- the tolerance of the
  reference twin's vectors -- and is identical where a kernel does not factor;
* the reference's device-builder convention for an empty PSF footprint (0.0 instead of NaN) is selectable;
* search_all over several search devices (slices run on the same GPU here) returns what one device returns.
"""

import numpy as np
import pytest

from kbmod_amd import fake_data as fd
from tests import util

pytestmark = pytest.mark.gpu



@pytest.fixture(scope="module")
def stack():
    st = util.make_stack(37, 150, 210, seed=77, noise=3.0, psf=1.0, objects=[(20, 30, 11.0, 7.0, 220.0)], mask_fraction=0.02)
    st.var[4][10, 11] = np.nan
    st.var[9][50, 60] = 0.0
    st.sci[2][0, 0] = np.inf
    return st


@pytest.mark.parametrize("num_bytes", [-1, 1, 2])
def test_stack_constructor_builds_the_same_array(kb, stack, num_bytes):
    ref = kb.StackSearch(stack.sci, stack.var, stack.psfs, stack.zeroed_times, num_bytes)
    new = kb.StackSearch.from_image_stacks(np.stack(stack.sci), np.stack(stack.var), stack.psfs, stack.zeroed_times, num_bytes)
    assert new.get_psi_phi_array().device_resident
    a, b = ref.get_psi_phi_array(), new.get_psi_phi_array()
    for name in ("num_times", "width", "height", "num_bytes", "psi_min_val", "psi_max_val", "psi_scale", "phi_min_val",
                 "phi_max_val", "phi_scale"):
        assert getattr(a, name) == getattr(b, name), name
    vx, vy = fd.kbmod_v1_candidates(8, 2.0, 20.0, 5, 0.0, 1.2)
    cands = util.trajectories(kb, vx, vy)
    ref.search_all(cands, True)
    new.search_all(cands, True)
    assert np.array_equal(ref.results_to_numpy(), new.results_to_numpy())
    assert len(new.results_to_numpy()) > 0
    # the arrays themselves, through the psi/phi curves of a dense set of trajectories (every epoch's samples)
    probe = [kb.Trajectory(x=int(x), y=int(y), vx=0.0, vy=0.0) for y in range(0, 150, 7) for x in range(0, 210, 5)]
    ca, cb = ref.get_all_psi_phi_curves(probe), new.get_all_psi_phi_curves(probe)
    assert np.array_equal(np.asarray(ca), np.asarray(cb))


def test_page_locked_ingest_paths_build_the_same_array(kb, stack):
    """One DMA per chunk out of the caller's memory -- registered for the build, or already pinned -- gives the bytes of
    the staged upload."""
    import torch

    sci, var = np.stack(stack.sci), np.stack(stack.var)
    probe = [kb.Trajectory(x=int(x), y=int(y), vx=0.0, vy=0.0) for y in range(0, 150, 7) for x in range(0, 210, 5)]
    ref = np.asarray(kb.StackSearch.from_image_stacks(sci, var, stack.psfs, stack.zeroed_times, register_host_memory=False)
                     .get_all_psi_phi_curves(probe))
    reg = kb.StackSearch.from_image_stacks(sci, var, stack.psfs, stack.zeroed_times, register_host_memory=True)
    assert np.array_equal(np.asarray(reg.get_all_psi_phi_curves(probe)), ref)
    sci_pin, var_pin = torch.from_numpy(sci).pin_memory(), torch.from_numpy(var).pin_memory()
    pin = kb.StackSearch.from_image_stacks(sci_pin.numpy(), var_pin.numpy(), stack.psfs, stack.zeroed_times, 2)
    enc = kb.StackSearch.from_image_stacks(sci, var, stack.psfs, stack.zeroed_times, 2)
    assert np.array_equal(np.asarray(pin.get_all_psi_phi_curves(probe)), np.asarray(enc.get_all_psi_phi_curves(probe)))
    # registering twice in a row works (the build unregisters what it registered)
    again = kb.StackSearch.from_image_stacks(sci, var, stack.psfs, stack.zeroed_times, register_host_memory=True)
    assert np.array_equal(np.asarray(again.get_all_psi_phi_curves(probe)), ref)


def test_large_stack_constructor_chunks(kb):
    # several upload chunks (16 MiB each): 40 epochs of 512 x 512
    rng = np.random.default_rng(5)
    T, H, W = 40, 512, 512
    sci = (rng.standard_normal((T, H, W)) * 2.0).astype(np.float32)
    var = np.full((T, H, W), 4.0, dtype=np.float32)
    var[3, 100:110, 200:210] = np.nan
    psf = fd.make_gaussian_kernel(1.0)
    times = np.arange(T) / float(T)
    new = kb.StackSearch.from_image_stacks(sci, var, [psf] * T, list(times))
    ref = kb.StackSearch([s for s in sci], [v for v in var], [psf] * T, list(times))
    probe = [kb.Trajectory(x=int(x), y=int(y), vx=0.0, vy=0.0) for y in range(0, H, 37) for x in range(0, W, 41)]
    assert np.array_equal(np.asarray(ref.get_all_psi_phi_curves(probe)), np.asarray(new.get_all_psi_phi_curves(probe)))


def test_separable_build_matches_two_dimensional_build(kb, stack):
    sci, var = np.stack(stack.sci), np.stack(stack.var)
    full = kb.StackSearch.from_image_stacks(sci, var, stack.psfs, stack.zeroed_times)
    sep = kb.StackSearch.from_image_stacks(sci, var, stack.psfs, stack.zeroed_times, separable_psf=True)
    probe = [kb.Trajectory(x=int(x), y=int(y), vx=0.0, vy=0.0) for y in range(150) for x in range(0, 210, 3)]
    a, b = np.asarray(full.get_all_psi_phi_curves(probe)), np.asarray(sep.get_all_psi_phi_curves(probe))
    assert a.shape == b.shape and np.isfinite(a).all()
    # same validity pattern (invalid samples read as 0 in the curves) and values to the twin's tolerance
    assert np.array_equal(a == 0.0, b == 0.0)
    scale = np.maximum(np.abs(a), 1e-3 * np.abs(a).max())
    assert (np.abs(a - b) / scale).max() < 1e-4
    assert not np.array_equal(a, b)  # it is a different summation order, not the same kernel
    # a kernel that does not factor: the separable request falls back to the 2-D kernel, bit for bit
    odd = np.array([[0.0, 0.1, 0.0], [0.1, 0.5, 0.2], [0.0, 0.1, 0.0]], dtype=np.float32)
    psfs = [odd] * len(stack.psfs)
    f2 = kb.StackSearch.from_image_stacks(sci, var, psfs, stack.zeroed_times)
    s2 = kb.StackSearch.from_image_stacks(sci, var, psfs, stack.zeroed_times, separable_psf=True)
    assert np.array_equal(np.asarray(f2.get_all_psi_phi_curves(probe)), np.asarray(s2.get_all_psi_phi_curves(probe)))


def test_empty_footprint_convention(kb):
    # a valid centre none of whose PSF taps is valid cannot exist (the centre is a tap); the footprint sum can
    # still be zero: a kernel whose only non-zero taps fall on masked pixels
    T, H, W = 2, 12, 12
    sci = np.ones((T, H, W), dtype=np.float32)
    var = np.ones((T, H, W), dtype=np.float32)
    k = np.zeros((3, 3), dtype=np.float32)
    k[1, 0] = 1.0  # all weight on the left neighbour
    sci[:, 5, 4] = np.nan  # ... which is masked for pixel (5, 5)
    nan_way = kb.StackSearch.from_image_stacks(sci, var, [k] * T, [0.0, 1.0])
    zero_way = kb.StackSearch.from_image_stacks(sci, var, [k] * T, [0.0, 1.0], empty_footprint_is_zero=True)
    t = kb.Trajectory(x=5, y=5, vx=0.0, vy=0.0)
    a = nan_way.search_linear_trajectory(5, 5, 0.0, 0.0, False)
    b = zero_way.search_linear_trajectory(5, 5, 0.0, 0.0, False)
    assert a.obs_count == 0      # NaN: the samples are NO_DATA (image_utils_cpp.cpp:60-61)
    assert b.obs_count == T      # 0.0: they count (image_kernels.cu:61)


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0, 0, 0]])
@pytest.mark.parametrize("cfg", [{"K": 8}, {"K": 4, "min_obs": 10, "sigmag": (0.25, 0.75, 0.7413, 4.0)}])
def test_search_all_over_several_search_devices(kb, orc, devices, cfg):
    st = util.make_stack(20, 70, 200, seed=8, noise=3.0, objects=[(30, 20, 14.0, 6.0, 260.0), (100, 40, 5.0, 9.0, 200.0)])
    # 70 candidates (uneven slices), fast enough that no two of them visit the same pixels at every epoch, searched
    # from start pixels whose every trajectory stays inside the image: no likelihood ties, a unique result
    vx, vy = fd.kbmod_v1_candidates(10, 8.0, 30.0, 7, 0.05, 1.1)
    cands = util.trajectories(kb, vx, vy)
    full_cfg = dict(cfg, xb=(0, 165), yb=(0, 38))
    one = kb.StackSearch(st.sci, st.var, st.psfs, st.zeroed_times)
    util.configure(one, full_cfg)
    one.search_all(cands, True)
    many = kb.StackSearch(st.sci, st.var, st.psfs, st.zeroed_times)
    util.configure(many, full_cfg)
    many.set_search_devices(devices)
    assert many.get_search_devices() == devices
    many.search_all(cands, True)
    a, b = one.results_to_numpy(), many.results_to_numpy()
    assert a.shape == b.shape and len(a) > 100
    assert np.array_equal(a[:, 4], b[:, 4])  # the likelihoods, slot for slot
    assert np.array_equal(a, b)
    # and the oracle agrees
    pp = orc.PsiPhi.from_images(st.sci, st.var, st.psfs, st.zeroed_times)
    params = util.oracle_params(pp, full_cfg)
    exp = util.as_table(orc.filter_sort(pp.search_kernel_semantics(orc.make_candidates(vx, vy), params), params.min_lh,
                                        params.min_observations))
    assert np.array_equal(b, exp)


def _slots(search):
    """Per-pixel result slots of the last search (before the global sort), as the set of result rows: the final sort
    is by likelihood only, so equal likelihoods may come out in another order -- compare as sorted row lists."""
    rows = search.results_to_numpy()
    return rows[np.lexsort(rows.T[::-1])]


@pytest.mark.parametrize("K", [8, 16])
def test_fan_out_is_tie_exact_over_the_whole_image(kb, orc, K):
    """Start pixels at the image border, slow candidates that share samples, masked pixels: likelihood ties among the
    best of a pixel.  With up to 16 results per pixel the fan-out uses 2 K stable lists + kb_merge_compact_exact, so
    the rows equal the single-device rows (and the oracle's), ties included."""
    st = util.make_stack(12, 40, 90, seed=3, noise=3.0, objects=[(30, 20, 9.0, 4.0, 260.0)], mask_fraction=0.03)
    vx, vy = fd.kbmod_v1_candidates(9, 1.0, 25.0, 8, 0.0, 1.5)  # slow ones included: duplicates of each other on the grid
    cands = util.trajectories(kb, vx, vy)
    cfg = {"K": K}
    one = kb.StackSearch(st.sci, st.var, st.psfs, st.zeroed_times)
    util.configure(one, cfg)
    one.search_all(cands, True)
    many = kb.StackSearch(st.sci, st.var, st.psfs, st.zeroed_times)
    util.configure(many, cfg)
    many.set_search_devices([0, 0, 0])
    many.search_all(cands, True)
    a, b = _slots(one), _slots(many)
    assert a.shape == b.shape and np.array_equal(a, b)
    pp = orc.PsiPhi.from_images(st.sci, st.var, st.psfs, st.zeroed_times)
    params = util.oracle_params(pp, cfg)
    every = pp.search_kernel_semantics(orc.make_candidates(vx, vy), util.oracle_params(pp, {"K": K + 1}))
    lh = every["lh"].reshape(-1, K + 1)
    assert (lh[:, :-1] == lh[:, 1:]).any()  # ties among the K + 1 best of some pixel
    exp = util.as_table(orc.filter_sort(pp.search_kernel_semantics(orc.make_candidates(vx, vy), params), params.min_lh,
                                        params.min_observations))
    assert np.array_equal(b, exp[np.lexsort(exp.T[::-1])])


def test_fan_out_falls_back_loudly_for_long_lists(kb):
    """results_per_pixel > 32 is outside the exchange format: the search runs on one device and SAYS so."""
    import logging

    st = util.make_stack(6, 20, 70, seed=4, noise=3.0)
    vx, vy = fd.kbmod_v1_candidates(6, 2.0, 20.0, 8, 0.0, 1.2)
    cands = util.trajectories(kb, vx, vy)
    messages = []

    class Sink(logging.Handler):
        def emit(self, record):
            messages.append((record.levelname, record.getMessage()))

    lg = logging.getLogger("kbmod.search.run_search")
    handler = Sink()
    lg.addHandler(handler)
    lg.setLevel(logging.DEBUG)
    try:
        kb.Logging.registerLogger(lg)
        one = kb.StackSearch(st.sci, st.var, st.psfs, st.zeroed_times)
        one.set_results_per_pixel(40)
        one.search_all(cands, True)
        many = kb.StackSearch(st.sci, st.var, st.psfs, st.zeroed_times)
        many.set_results_per_pixel(40)
        many.set_search_devices([0, 0])
        many.search_all(cands, True)
    finally:
        lg.removeHandler(handler)
    assert np.array_equal(one.results_to_numpy(), many.results_to_numpy())
    assert any(level == "WARNING" and "one device" in text for level, text in messages), messages
