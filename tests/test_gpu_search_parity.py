"""GPU parity: the HIP search (through the C ABI behind StackSearch.search_all(..., True))
against the oracle's kernel-semantics search on the same seeded inputs.

Bar: bit-exact on every field (x, y, obs_count are integers; vx, vy, lh, flux are
compared bit for bit as well, which is stricter than the 1e-4 the north star asks).
"""

import numpy as np
import pytest

from kbmod_amd import fake_data as fd
from tests import util

pytestmark = pytest.mark.gpu

OBJ = [(17, 12, 21.0, 16.0, 250.0), (60, 40, -8.0, 11.0, 180.0)]


def _check(got, exp):
    assert got.shape == exp.shape, (got.shape, exp.shape)
    bad = np.nonzero(np.any(got != exp, axis=1))[0]
    assert len(bad) == 0, f"{len(bad)} rows differ, first {bad[:3]}: {got[bad[:3]]} vs {exp[bad[:3]]}"


@pytest.fixture(scope="module")
def stack():
    return util.make_stack(20, 80, 100, seed=100, noise=4.0, psf=1.0, objects=OBJ, mask_fraction=0.01)


@pytest.fixture(scope="module")
def cands():
    return fd.kbmod_v1_candidates(12, 5.0, 40.0, 11, 0.0, 1.5)  # 132: not a multiple of the chunk


@pytest.fixture(scope="module")
def lds_cands():
    # No candidate lands on an exact half-pixel, so every (chunk, epoch) is staged through LDS.
    return fd.kbmod_v1_candidates(16, 5.0, 20.0, 9, 0.05, 1.45)  # 16 = 2 chunks per angle row, <= 8 px spread


# kb_device_search_filter flags (include/kbmod_hip.h)
EXACT, DIRECT, LDS, DOUBLE_DECODE, ENCODED_STAGING, TALL_TILES, WIDE_TILES = 1, 2, 4, 8, 16, 64, 128
# "lds": 64 x 8 start-pixel tiles (what the small search areas of these tests get by default), "lds_tall": 64 x 16
KERNELS = {"direct": DIRECT, "lds": LDS | WIDE_TILES, "lds_tall": LDS | TALL_TILES,
           "lds_encoded": LDS | ENCODED_STAGING}


@pytest.fixture(params=["direct", "lds", "lds_tall"])
def kern(request):
    return request.param


def _variant(search):
    return search.last_search_stats()["kernel_variant"] // 10000  # 0 direct, 1 LDS/encoded copy, 2 LDS/floats


def _check_kernel(search, kern, num_bytes=-1):
    want = {"direct": 0, "lds": 2, "lds_tall": 2, "lds_encoded": 2 if num_bytes in (-1, 4) else 1}[kern]
    assert _variant(search) == want, search.last_search_stats()


def test_requires_gpu(kb):
    assert kb.kb_has_gpu(), "-m gpu tests need a device; the product has no CPU fallback"


def test_float_default(kb, orc, stack, cands, kern):
    got, exp, s = util.run_both(kb, orc, stack, *cands, {}, flags=KERNELS[kern])
    _check_kernel(s, kern)
    _check(got, exp)
    assert s.last_search_stats()["num_evals"] == 20 * 80 * 100 * 132


def test_default_kernel_choice(kb, orc, stack, cands):
    # flags = 0: the LDS kernel from one full chunk of 8 candidates on, the direct kernel for fewer.
    got, exp, s = util.run_both(kb, orc, stack, *cands, {})
    assert _variant(s) == 2
    _check(got, exp)
    got, exp, s = util.run_both(kb, orc, stack, cands[0][:8], cands[1][:8], {})
    assert _variant(s) == 2
    _check(got, exp)
    got, exp, s = util.run_both(kb, orc, stack, cands[0][:7], cands[1][:7], {})
    assert _variant(s) == 0
    _check(got, exp)


@pytest.mark.parametrize("num_bytes", [-1, 1, 2])
@pytest.mark.parametrize("staging", ["lds", "lds_encoded"])
@pytest.mark.parametrize("cfg", [{}, {"K": 10, "min_obs": 5}, {"xb": (-10, 110), "yb": (-10, 90), "min_lh": -1e30},
                                 {"sigmag": (0.25, 0.75, 0.7413, 5.0), "min_obs": 8}])
def test_lds_kernel(kb, orc, stack, lds_cands, cfg, num_bytes, staging):
    """The LDS-staged kernel: masked pixels and off-image starts come out of the padded copy."""
    if staging == "lds_encoded" and num_bytes == -1:
        pytest.skip("float arrays are always staged as canonical floats")
    got, exp, s = util.run_both(kb, orc, stack, *lds_cands, cfg, num_bytes=num_bytes, flags=KERNELS[staging])
    if staging == "lds_encoded" and cfg.get("K", 8) > 8 and "sigmag" not in cfg:
        assert _variant(s) == 0  # encoded staging is built for lists up to 8 only: longer ones read the array itself
    else:
        _check_kernel(s, staging, num_bytes)
    _check(got, exp)


def test_lds_kernel_clean_stack(kb, orc, lds_cands):
    """No masked pixel anywhere: interior tiles take the count-free specialisation."""
    st = util.make_stack(21, 96, 200, seed=77, objects=OBJ)
    got, exp, s = util.run_both(kb, orc, st, *lds_cands, {"min_obs": 3}, flags=LDS)
    _check_kernel(s, "lds")
    _check(got, exp)
    a, _, s2 = util.run_both(kb, orc, st, *lds_cands, {"min_obs": 3}, flags=DIRECT)
    _check_kernel(s2, "direct")
    _check(a, exp)


def test_lds_kernel_unstaged_epochs(kb, orc):
    """Epochs the LDS kernel cannot stage -- shifts exactly on a rounding boundary, chunks whose
    candidates spread further than a slab -- are evaluated per lane inside the same kernel."""
    times = np.arange(24) * 0.25  # vx * t + 0.5 is an integer for odd vx at every other odd epoch
    st = util.make_stack(24, 40, 150, seed=21, objects=[(30, 10, 9.0, 2.0, 200.0)], mask_fraction=0.02, times=times)
    vx, vy = fd.kbmod_v1_candidates(16, 2.0, 9.0, 16, 0.02, 0.9)
    vx, vy = vx.copy(), vy.copy()
    vx[3], vy[3] = 3.0, 1.0     # half-integer products at odd epochs: per-lane pixels inside the slab
    vx[40], vy[40] = -5.0, 7.0
    vx[77], vy[77] = 60.0, -35.0  # one outlier: its chunk no longer fits a slab once it has moved away
    vx[130], vy[130] = 3.0e7, 1.0  # beyond the proven range of the shift table
    got, exp, s = util.run_both(kb, orc, st, vx, vy, {"min_lh": -1e30, "xb": (-5, 155), "yb": (-3, 44)}, flags=LDS)
    _check_kernel(s, "lds")
    _check(got, exp)


def test_lds_request_falls_back_when_little_stages(kb, orc):
    # half of the candidates move further than the shift table's proven range: every (chunk, epoch)
    # with t > 0 is unstaged -> more than 10 % -> the direct kernel runs instead
    st = util.make_stack(6, 32, 70, seed=11)
    vx = np.tile(np.array([1.0, 3.0e7, -1.0, 5.0], dtype=np.float32), 10)
    vy = np.tile(np.array([3.0, -1.0, 1.0e8, 7.0], dtype=np.float32), 10)
    got, exp, s = util.run_both(kb, orc, st, vx, vy, {"min_lh": -1e30}, flags=LDS)
    assert _variant(s) == 0
    _check(got, exp)


def test_lds_kernel_boundary_shifts_everywhere(kb, orc):
    # every epoch of every chunk has a shift on a rounding boundary: all of them are staged with one
    # pixel of slack and summed per lane out of LDS
    times = np.array([0.0, 0.5, 1.0, 1.5, 2.0, 2.5])
    st = util.make_stack(6, 32, 70, seed=11, times=times, mask_fraction=0.03)
    vx = np.tile(np.array([1.0, 3.0, -1.0, 5.0], dtype=np.float32), 10)
    vy = np.tile(np.array([3.0, -1.0, 1.0, 7.0], dtype=np.float32), 10)
    for nb in (-1, 1):
        got, exp, s = util.run_both(kb, orc, st, vx, vy, {"min_lh": -1e30, "xb": (-4, 74), "yb": (-4, 36)},
                                    num_bytes=nb, flags=LDS)
        assert _variant(s) == 2
        _check(got, exp)


def test_table_path_equals_exact_path(kb, orc, stack, cands):
    a, _, _ = util.run_both(kb, orc, stack, *cands, {}, flags=DIRECT)
    b, _, _ = util.run_both(kb, orc, stack, *cands, {}, flags=EXACT)
    c, _, _ = util.run_both(kb, orc, stack, *cands, {}, flags=LDS)
    _check(a, b)
    _check(c, b)


@pytest.mark.parametrize("K", [1, 5, 8, 10, 16, 20, 32])
def test_results_per_pixel(kb, orc, stack, cands, K, kern):
    got, exp, s = util.run_both(kb, orc, stack, *cands, {"K": K, "min_lh": -1e30}, flags=KERNELS[kern])
    _check_kernel(s, kern)
    _check(got, exp)
    assert len(got) == K * 80 * 100  # every slot survives min_lh = -1e30, placeholders are -FLT_MAX


@pytest.mark.parametrize("K", [12, 32])
def test_lists_of_more_than_eight(kb, orc, stack, cands, K, kern):
    # kb_search_lds keeps lists of more than 8 results in its HBM store between chunks of candidates:
    # (a) no candidate ever enters (start pixels so far off the image that no sample is ever valid): the store is
    #     never written, every slot a placeholder
    got, exp, _ = util.run_both(kb, orc, stack, *cands, {"K": K, "min_obs": 1, "min_lh": -1e30, "xb": (-400, -300)},
                                flags=KERNELS[kern])
    _check(got, exp)
    assert len(got) == 0
    # (b) a quantised array: equal likelihoods in every list, the order among them is the swap-down's
    got, exp, _ = util.run_both(kb, orc, stack, *cands, {"K": K, "min_lh": -1e30}, num_bytes=1, flags=KERNELS[kern])
    _check(got, exp)
    # (c) a threshold that only few candidates pass: most waves never touch the store after the first chunks
    got, exp, _ = util.run_both(kb, orc, stack, *cands, {"K": K, "min_obs": 10, "min_lh": 4.0}, flags=KERNELS[kern])
    _check(got, exp)


@pytest.mark.parametrize("K,mode", [(3, 0), (3, 2), (3, 3), (8, 0), (8, 2), (8, 3), (12, 1), (12, 2), (16, 1), (16, 2)])
def test_list_modes_agree_with_the_oracle(kb, orc, stack, cands, K, mode, kern, monkeypatch):
    # kb_search_lds keeps its lists in registers (0), as (likelihood, candidate) pairs in its HBM store (1), as
    # whole result records there (2) or as packed records in registers (3; 2 and 3 need no re-evaluation of the
    # winners): the host chooses by the shape of the search, KBMOD_LIST_MODE pins the choice.  Flux and observation counts of mode 2 come from the search's own
    # sums, those of modes 0 / 1 from the exact re-evaluation: all must equal the oracle's bit for bit.
    # (kb_search_direct: mode 2 = records in registers, modes 0 / 1 = pairs in registers + re-evaluation)
    monkeypatch.setenv("KBMOD_LIST_MODE", str(mode))
    for cfg, nb in (({"K": K, "min_lh": -1e30}, -1), ({"K": K, "min_obs": 9, "min_lh": 1.0}, 1),
                    ({"K": K, "xb": (-20, 130), "yb": (-15, 95), "min_lh": -1e30}, 2)):
        got, exp, s = util.run_both(kb, orc, stack, *cands, cfg, num_bytes=nb, flags=KERNELS[kern])
        _check_kernel(s, kern)
        _check(got, exp)
        assert len(got) > 100


@pytest.mark.parametrize("cfg", [{"K": 4, "min_lh": -1e30}, {"K": 8, "min_obs": 9, "min_lh": 1.0}, {"K": 12, "min_lh": -1e30},
                                 {"K": 32, "min_obs": 5}, {"sigmag": (0.25, 0.75, 0.7413, 5.0), "min_obs": 8},
                                 {"K": 6, "xb": (-20, 130), "yb": (-15, 95), "min_lh": -1e30}])
@pytest.mark.parametrize("num_bytes", [-1, 1])
@pytest.mark.parametrize("tiles", ["lds", "lds_tall"])
def test_narrow_chunks_pinned(kb, orc, stack, lds_cands, cfg, num_bytes, tiles, monkeypatch):
    # Lists of up to 8 results on a candidate list without per-lane epochs run the instance for chunks of 16 candidates;
    # the instances for chunks of 8 (which also serve every list the wide one cannot take) are pinned here on the same
    # inputs with KBMOD_CHUNK.
    monkeypatch.setenv("KBMOD_CHUNK", "8")
    got, exp, s = util.run_both(kb, orc, stack, *lds_cands, cfg, num_bytes=num_bytes, flags=KERNELS[tiles])
    assert _variant(s) == 2
    assert "kb::kb_search_lds<" in s.last_search_stats()["kernel_name"] and ", 16, " not in s.last_search_stats()["kernel_name"][:24]
    _check(got, exp)
    assert len(got) > 50


def test_repeated_searches_reuse_the_padded_copy(kb, orc, stack, cands):
    # A StackSearch with its array resident in HBM tells the library, from its second search on, that the array is
    # unchanged (flag 256): the padded copy of kb_search_lds is reused while the frame geometry stays the same,
    # rebuilt when the bounds move it, and never taken over by another object's array of the same shape.
    def expect(st, cfg):
        pp = orc.PsiPhi.from_images(st.sci, st.var, st.psfs, st.zeroed_times, -1)
        params = util.oracle_params(pp, cfg)
        raw = pp.search_kernel_semantics(orc.make_candidates(*cands), params)
        return util.as_table(orc.filter_sort(raw, params.min_lh, params.min_observations))

    other = util.make_stack(len(stack.sci), *stack.sci[0].shape, seed=4242, noise=3.0, objects=[(30, 20, 6.0, -3.0, 200.0)])
    a = kb.StackSearch(stack.sci, stack.var, stack.psfs, stack.zeroed_times, -1)
    b = kb.StackSearch(other.sci, other.var, other.psfs, other.zeroed_times, -1)
    for s in (a, b):
        s.set_search_flags(KERNELS["lds"])
        s.preload_psi_phi_array()
    cfg = {"min_lh": -1e30}
    for search, st, c in ((a, stack, cfg), (a, stack, cfg), (b, other, cfg), (a, stack, cfg), (a, stack, {"min_lh": -1e30, "xb": (-9, 70)}),
                          (a, stack, {"min_lh": -1e30, "xb": (-9, 70)}), (b, other, cfg), (b, other, cfg)):
        util.configure(search, c)
        search.search_all(util.trajectories(kb, *cands), True)
        assert _variant(search) == 2
        _check(search.results_to_numpy(), expect(st, c))
    a.unload_psi_phi_array()
    a.preload_psi_phi_array()
    a.search_all(util.trajectories(kb, *cands), True)
    _check(a.results_to_numpy(), expect(stack, {"min_lh": -1e30, "xb": (-9, 70)}))
    # (KBMOD_DEBUG=1 prints "copy reused" for the second, sixth and eighth of the searches above)


def test_large_k_keeps_every_candidate(kb, orc):
    # TrajectoryExplorer-style: K >= number of candidates, all of them come back per pixel
    # (reference: tests/test_trajectory_explorer.py:106-124).
    st = util.make_stack(12, 20, 24, seed=8, objects=[(6, 5, 9.0, 4.0, 150.0)], mask_fraction=0.02)
    vx, vy = fd.velocity_grid_candidates(9, -10.0, 10.0, 7, -6.0, 6.0)  # 63 candidates
    cfg = {"K": 100, "min_lh": -1e30, "xb": (2, 10), "yb": (1, 9)}
    got, exp, _ = util.run_both(kb, orc, st, vx, vy, cfg)
    _check(got, exp)
    assert len(got) == 63 * 8 * 8  # the 37 placeholder slots per pixel carry lh = -FLT_MAX and are filtered
    cfg = {"K": 40, "sigmag": (0.25, 0.75, 0.7413, -5.0), "xb": (2, 10), "yb": (1, 9)}
    got, exp, _ = util.run_both(kb, orc, st, vx, vy, cfg)
    _check(got, exp)


def test_min_obs_and_min_lh(kb, orc, stack, cands, kern):
    got, exp, _ = util.run_both(kb, orc, stack, *cands, {"min_obs": 12, "min_lh": 2.5}, flags=KERNELS[kern])
    _check(got, exp)


def test_extended_bounds(kb, orc, stack, cands, kern):
    cfg = {"xb": (-10, 110), "yb": (-10, 90), "K": 5}
    got, exp, _ = util.run_both(kb, orc, stack, *cands, cfg, flags=KERNELS[kern])
    _check(got, exp)


def test_reduced_bounds(kb, orc, stack, cands, kern):
    cfg = {"xb": (5, 95), "yb": (5, 75), "K": 10, "min_lh": -1e30}
    got, exp, _ = util.run_both(kb, orc, stack, *cands, cfg, flags=KERNELS[kern])
    _check(got, exp)
    assert len(got) == 10 * 90 * 70


def test_fewer_candidates_than_slots(kb, orc, stack, kern):
    vx, vy = fd.velocity_grid_candidates(2, -5.0, 5.0, 2, -3.0, 3.0)
    got, exp, _ = util.run_both(kb, orc, stack, vx, vy, {"min_lh": -1e30}, flags=KERNELS[kern])
    _check(got, exp)


@pytest.mark.parametrize("decode", [0, DOUBLE_DECODE])  # 0: verified fp32-FMA decode
@pytest.mark.parametrize("num_bytes", [1, 2])
@pytest.mark.parametrize("kernel", ["direct", "lds", "lds_encoded"])
def test_encoded(kb, orc, stack, cands, num_bytes, decode, kernel):
    got, exp, s = util.run_both(kb, orc, stack, *cands, {"min_obs": 10}, num_bytes=num_bytes,
                                flags=KERNELS[kernel] | decode)
    _check_kernel(s, kernel, num_bytes)
    _check(got, exp)


@pytest.mark.parametrize("num_bytes", [-1, 1])
def test_sigma_g(kb, orc, stack, cands, num_bytes, kern):
    cfg = {"sigmag": (0.25, 0.75, 0.7413, 5.0), "min_obs": 8}
    got, exp, _ = util.run_both(kb, orc, stack, *cands, cfg, num_bytes=num_bytes, flags=KERNELS[kern])
    _check(got, exp)
    assert len(got) > 0


def test_sigma_g_low_threshold(kb, orc, kern):
    # min_lh below every likelihood: every trajectory with data is clipped.
    st = util.make_stack(9, 24, 40, seed=5, objects=[(8, 8, 10.0, 4.0, 120.0)], mask_fraction=0.05)
    vx, vy = fd.kbmod_v1_candidates(4, 2.0, 20.0, 5, 0.0, 1.2)
    cfg = {"sigmag": (0.15, 0.85, 0.4824, -50.0), "K": 4}
    got, exp, _ = util.run_both(kb, orc, st, vx, vy, cfg, flags=KERNELS[kern])
    _check(got, exp)


def test_sigma_g_more_than_64_epochs(kb, orc, kern):
    # beyond one epoch per lane the clip runs per lane (the literal exchange sort)
    st = util.make_stack(70, 24, 70, seed=6, objects=[(10, 8, 6.0, 2.0, 120.0)], mask_fraction=0.03)
    vx, vy = fd.kbmod_v1_candidates(8, 2.0, 10.0, 5, 0.0, 0.8)
    cfg = {"sigmag": (0.25, 0.75, 0.7413, 3.0), "min_obs": 20, "K": 4}
    got, exp, _ = util.run_both(kb, orc, st, vx, vy, cfg, flags=KERNELS[kern])
    _check(got, exp)
    assert len(got) > 0


def test_sigma_g_equal_ratios(kb, orc, kern):
    # piecewise-constant images: many psi/phi ratios are exactly equal, and the order the reference's
    # exchange sort leaves among them decides the summation order of the clipped sums
    rng = np.random.default_rng(12)
    st = util.make_stack(16, 30, 70, seed=12, noise=1.0, objects=[(12, 9, 5.0, 3.0, 90.0)])
    for t in range(16):
        st.sci[t][:, :] = np.round(st.sci[t] * 0.5) * 2.0  # a handful of distinct values
        st.var[t][:, :] = np.float32(rng.choice([1.0, 2.0, 4.0]))
    vx, vy = fd.kbmod_v1_candidates(8, 1.0, 9.0, 5, 0.0, 0.9)
    cfg = {"sigmag": (0.25, 0.75, 0.7413, -100.0), "min_obs": 4, "K": 6}
    got, exp, _ = util.run_both(kb, orc, st, vx, vy, cfg, flags=KERNELS[kern])
    _check(got, exp)
    got, exp, _ = util.run_both(kb, orc, st, vx, vy, cfg, num_bytes=1, flags=KERNELS[kern])
    _check(got, exp)


def test_half_integer_shifts_take_exact_path(kb, orc, kern):
    # vx * t + 0.5 lands exactly on integers: the shift table must refuse these.
    times = np.array([0.0, 0.25, 0.5, 0.75, 1.0, 1.5])
    st = util.make_stack(6, 32, 70, seed=11, times=times)
    vx = np.array([2.0, 6.0, -2.0, 1.0], dtype=np.float32)
    vy = np.array([2.0, -6.0, 1.0, 0.0], dtype=np.float32)
    got, exp, _ = util.run_both(kb, orc, st, vx, vy, {"min_lh": -1e30, "xb": (-3, 73), "yb": (-3, 35)},
                                flags=KERNELS[kern])
    _check(got, exp)


def test_empty_candidate_list(kb, stack):
    # The reference cannot place an empty list on the device: GPUArray::allocate_gpu_memory
    # (gpu_array.h:113-119) / allocate_gpu_block (kernel_memory.cu:93-103) throw for 0 bytes.
    s = kb.StackSearch(stack.sci, stack.var, stack.psfs, stack.zeroed_times)
    with pytest.raises(RuntimeError):
        s.search_all([], True)


def test_too_many_images(kb):
    st = util.make_stack(1000, 10, 12, seed=1, noise=0.5)
    s = kb.StackSearch(st.sci, st.var, st.psfs, st.zeroed_times)
    t = kb.Trajectory(x=0, y=0, vx=0.0, vy=0.0)
    with pytest.raises(RuntimeError):
        s.search_all([t], True)
    with pytest.raises(RuntimeError):
        s.evaluate_single_trajectory(t, True)


def test_deep_stack_512(kb, orc, kern):
    # beyond the reference's 200-epoch device limit
    st = util.make_stack(512, 16, 70, seed=3, objects=[(5, 5, 30.0, 6.0, 80.0)])
    vx, vy = fd.kbmod_v1_candidates(4, 10.0, 40.0, 3, 0.0, 0.6)
    got, exp, _ = util.run_both(kb, orc, st, vx, vy, {"min_obs": 100}, num_bytes=2, flags=KERNELS[kern])
    _check(got, exp)
