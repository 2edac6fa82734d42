"""GPU tests of the in-search sigma-G resolve (kbmod_amd/csrc/sigmag_kernels.hip): the wavefront
primitives it is built from, the batching of the candidate list, and the hand-over to the literal
exchange-sort code when equal psi/phi ratios come from different (psi, phi) pairs."""

import ctypes as C
import os

import numpy as np
import pytest

from kbmod_amd import fake_data as fd
from tests import util

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIRECT, LDS = 2, 4


def _check(got, exp):
    assert got.shape == exp.shape, (got.shape, exp.shape)
    bad = np.nonzero(np.any(got != exp, axis=1))[0]
    assert len(bad) == 0, f"{len(bad)} rows differ, first {bad[:3]}: {got[bad[:3]]} vs {exp[bad[:3]]}"


def test_wave_sort_and_chain_sum(kb):
    import torch

    lib = C.CDLL(os.path.join(ROOT, "kbmod_amd", "lib", "libkbmod_hip.so"))
    lib.kb_last_error.restype = C.c_char_p
    lib.kb_debug_wave_ops.argtypes = [C.c_void_p] * 6 + [C.c_uint64, C.c_void_p]
    rng = np.random.default_rng(3)
    n = 512
    keys = rng.integers(0, 2**32, size=(n, 64), dtype=np.uint64).astype(np.uint32)
    keys[: n // 4] = rng.integers(0, 6, size=(n // 4, 64)).astype(np.uint32)  # heavy ties
    keys[n // 4] = np.arange(64, dtype=np.uint32)[::-1]
    keys[n // 4 + 1] = 7
    values = (rng.standard_normal((n, 64)) * 10.0 ** rng.integers(-3, 6, size=(n, 64))).astype(np.float32)
    lo = rng.integers(0, 64, size=n)
    hi = np.array([rng.integers(l, 64) for l in lo])
    lo[0], hi[0] = 0, 63
    lo[1], hi[1] = 63, 63
    bounds = np.stack([lo, hi], 1).astype(np.int32)

    dev = torch.device("cuda")
    d_keys = torch.from_numpy(keys.view(np.int32)).to(dev)
    d_out = torch.empty_like(d_keys)
    d_src = torch.empty_like(d_keys)
    d_val = torch.from_numpy(values).to(dev)
    d_bnd = torch.from_numpy(bounds).to(dev)
    d_sum = torch.empty(n, dtype=torch.float32, device=dev)
    rc = lib.kb_debug_wave_ops(d_keys.data_ptr(), d_out.data_ptr(), d_src.data_ptr(), d_val.data_ptr(), d_bnd.data_ptr(),
                               d_sum.data_ptr(), n, None)
    assert rc == 0, lib.kb_last_error()
    out = d_out.cpu().numpy().view(np.uint32)
    src = d_src.cpu().numpy().view(np.uint32)
    assert np.array_equal(out, np.sort(keys, axis=1))
    # the payload is a permutation that carries every key to its place
    assert np.array_equal(np.sort(src, axis=1), np.tile(np.arange(64, dtype=np.uint32), (n, 1)))
    assert np.array_equal(np.take_along_axis(keys, src.astype(np.int64), axis=1), out)
    # strictly sequential float32 sums
    exp = np.empty(n, dtype=np.float32)
    for w in range(n):
        acc = np.float32(0.0)
        for i in range(lo[w], hi[w] + 1):
            acc = np.float32(acc + values[w, i])
        exp[w] = acc
    assert np.array_equal(d_sum.cpu().numpy().view(np.uint32), exp.view(np.uint32))


@pytest.mark.parametrize("kern", [DIRECT, LDS])
@pytest.mark.parametrize("cap", [1, 3000, 20000])
def test_sigma_g_candidate_batches(kb, orc, kern, cap, monkeypatch):
    # A small work-item store cuts the candidate list into batches of whole chunks; the per-pixel lists
    # are carried from batch to batch.
    st = util.make_stack(20, 40, 150, seed=31, noise=4.0, objects=[(17, 12, 21.0, 16.0, 250.0), (60, 20, -8.0, 11.0, 180.0)],
                         mask_fraction=0.01)
    vx, vy = fd.kbmod_v1_candidates(16, 5.0, 20.0, 9, 0.05, 1.45)
    cfg = {"sigmag": (0.25, 0.75, 0.7413, 2.0), "min_obs": 6, "K": 5}
    monkeypatch.setenv("KBMOD_SIGMAG_CAP", str(cap))
    got, exp, s = util.run_both(kb, orc, st, vx, vy, cfg, flags=kern)
    _check(got, exp)
    assert len(got) > 100
    n_rows = 3 * 40
    # candidates per chunk of the instance that ran: 16 for the float-staged kernel on a list without per-lane epochs, else 8
    stats = s.last_search_stats()
    chunk = 16 if stats["kernel_name"].startswith("kb::kb_search_lds<8, 16,") else 8
    n_chunks = -(-len(vx) // chunk)
    want_batches = -(-n_chunks // max(1, min(n_chunks, cap // (n_rows * chunk))))
    assert stats["num_search_launches"] == 3 * want_batches


@pytest.mark.parametrize("kern", [DIRECT, LDS])
@pytest.mark.parametrize("num_bytes", [-1, 2])
def test_sigma_g_equal_ratios_from_different_pairs(kb, orc, kern, num_bytes):
    # Epochs t and t + 6 hold the same science image under variances that differ by a factor of four: for a
    # trajectory that does not move, psi/phi of the two epochs is EQUAL while (psi, phi) differ, so the
    # permutation the reference's exchange sort leaves among the equal ratios decides the summation order.
    # The cooperative clip must hand these to the literal code.
    rng = np.random.default_rng(41)
    T, H, W = 12, 24, 70
    st = util.make_stack(T, H, W, seed=41, noise=1.0)
    for t in range(6):
        st.sci[t + 6][:, :] = st.sci[t]  # psi and phi of epoch t + 6 are those of epoch t divided by 4, exactly
        st.var[t][:, :] = np.float32(1.0)
        st.var[t + 6][:, :] = np.float32(4.0)
    fd.add_fake_object(st, 30, 10, 0.0, 0.0, flux=40.0)
    vx = np.array([0.0, 0.0, 1.5, 0.0, -2.0, 0.3, 0.0, 4.0, 0.0], dtype=np.float32)
    vy = np.array([0.0, 1.0, 0.0, -0.7, 0.5, 0.0, 2.5, 1.0, 0.0], dtype=np.float32)
    cfg = {"sigmag": (0.25, 0.75, 0.7413, -100.0), "min_obs": 4, "K": 4}
    got, exp, s = util.run_both(kb, orc, st, vx, vy, cfg, num_bytes=num_bytes, flags=kern)
    _check(got, exp)
    assert len(got) > 0
    stats = s.last_search_stats()
    assert stats["sigmag_literal"] <= stats["sigmag_trajectories"]
    if num_bytes == -1:
        # (quantised to uint16 the two epochs' values no longer stand in an exact ratio of four: no ties there)
        assert stats["sigmag_literal"] > 0


@pytest.mark.parametrize("kern", [DIRECT, LDS])
@pytest.mark.parametrize("T", [65, 100, 128, 129, 200, 256, 257])
def test_sigma_g_deep_stacks(kb, orc, kern, T):
    # 65 .. 256 epochs: two or four epochs per lane in the cooperative clip (64-lane sorting network on 128 / 256
    # keys, sums carried from slot to slot); beyond 256 the literal per-lane code
    st = util.make_stack(T, 20, 66, seed=1000 + T, noise=2.0, objects=[(9, 7, 4.0, 1.5, 60.0)], mask_fraction=0.04,
                         times=np.arange(T) / 40.0)
    vx, vy = fd.kbmod_v1_candidates(8, 1.0, 6.0, 5, 0.0, 0.9)
    cfg = {"sigmag": (0.25, 0.75, 0.7413, 1.5), "min_obs": T // 3, "K": 4}
    got, exp, s = util.run_both(kb, orc, st, vx, vy, cfg, flags=kern)
    _check(got, exp)
    assert len(got) > 50
    stats = s.last_search_stats()
    assert stats["sigmag_trajectories"] > 100
    # float data without equal ratios: nothing takes the literal code up to 256 epochs, everything beyond
    assert stats["sigmag_literal"] == (0 if T <= 256 else stats["sigmag_trajectories"])


def test_sigma_g_deep_stack_mixed_ties(kb, orc):
    # the equal-ratio / different-pair construction of the shallow test on 140 epochs (the slot-boundary neighbours
    # included): the cooperative clip must hand these trajectories to the literal code
    T, H, W = 140, 16, 66
    st = util.make_stack(T, H, W, seed=77, noise=1.0)
    for t in range(70):
        st.sci[t + 70][:, :] = st.sci[t]
        st.var[t][:, :] = np.float32(1.0)
        st.var[t + 70][:, :] = np.float32(4.0)
    vx = np.array([0.0, 0.0, 0.4, 0.0, -0.3, 0.1, 0.0, 0.6], dtype=np.float32)
    vy = np.array([0.0, 0.2, 0.0, -0.1, 0.1, 0.0, 0.3, 0.2], dtype=np.float32)
    cfg = {"sigmag": (0.25, 0.75, 0.7413, -100.0), "min_obs": 4, "K": 4}
    got, exp, s = util.run_both(kb, orc, st, vx, vy, cfg, flags=LDS)
    _check(got, exp)
    assert len(got) > 0
    assert s.last_search_stats()["sigmag_literal"] > 0
