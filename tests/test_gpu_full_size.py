"""BASELINE.json's configurations at their full sizes, through size-independent properties.

The oracle cannot finish these sizes in seconds, so each run of the hot path (bench.py, C ABI) is
checked with ``--verify``: (1) kb_search_lds and kb_search_direct -- two different data paths --
produce the same result buffer bit for bit; (2) every per-pixel list is in descending likelihood
order; (3) a start window re-done with exact per-lane double positions (no shift table, no staging:
the arithmetic the oracle parity tests pin at small sizes) reproduces the same slots.
"""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONFIGS = {
    # configs[1]: 64 x 512 x 512 float32, 1024 candidates, no sigma-G
    "cfg2_f32": [],
    # configs[2]: same stack uint8-encoded + in-kernel sigma-G
    "cfg3_u8_sigmag": ["--num-bytes", "1", "--sigmag"],
    # configs[3], one GPU's share: 128 x 4096 x 4096 float32 (17.2 GB), 1.07e9 trajectories
    "cfg4_shard": ["--frames", "128", "--size", "4096", "--vel-steps", "32", "--ang-steps", "2"],
    # configs[4]: 512 x 2048 x 2048 uint16-encoded, 4096 candidates per pixel (8.8e12 evals)
    "cfg5_deep_u16": ["--frames", "512", "--size", "2048", "--num-bytes", "2", "--vel-steps", "64", "--ang-steps", "64"],
    # configs[1] with a likelihood threshold: the lists' floor (flag 1024, what StackSearch.search_all passes) at full size --
    # every kernel compared runs with it, the exact-position window included
    "cfg2_f32_min_lh_10": ["--min-lh", "10"],
    # ... and one GPU's share of configs[3] with it (64 candidates: two 64 x 8 tiles per CU)
    "cfg4_shard_min_lh_10": ["--frames", "128", "--size", "4096", "--vel-steps", "32", "--ang-steps", "2", "--min-lh", "10"],
}


@pytest.mark.parametrize("name", list(CONFIGS))
def test_full_size_properties(name):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
           "--no-live-traffic", "--no-masked", "--verify"] + CONFIGS[name]
    proc = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert lines, proc.stderr[-2000:]
    out = json.loads(lines[-1])
    v = out["verify"]
    assert v["kernels_agree_ok"] and v["lists_sorted_ok"] and v["exact_window_ok"], v
    assert proc.returncode == 0
    assert out["roofline"]["kernel"].startswith("kb::kb_search_lds<") and v["other_kernel"] == "kb_search_direct"
    assert out["value"] > 1e9  # north star floor, evals/s


@pytest.mark.parametrize("extra", [[], ["--min-lh", "10"]])
def test_headline_size_through_the_rccl_branch(extra):
    """BASELINE configs[1] at full size through bench.py's N > 1 branch on the real RCCL backend (world size 1,
    KBMOD_FORCE_DIST): 16 stable records per pixel, one gather (dense, finished behind the next search) or the sparse exchange
    (with a likelihood threshold), the tie-exact merge -- the merged lists equal ONE search over the same candidates."""
    env = dict(os.environ, KBMOD_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--verify"] + extra
    proc = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert lines and proc.returncode == 0, proc.stderr[-2000:]
    out = json.loads(lines[-1])
    assert out["verify"]["merged_equals_single_device_ok"] and out["verify"]["backend"] == "nccl"
    assert out["exchange"]["form"] == ("sparse" if extra else "dense")
    if extra:
        assert out["exchange"]["wire_bytes_per_rank"] < out["exchange"]["dense_bytes_per_rank"] // 16 and out["verify_survivors"] > 0


def _oracle_view(orc, lib, meta, arr, times):
    """The device array copied to the host, wrapped for the oracle (the array itself was produced by the HIP
    builder, whose bytes test_gpu_builder_and_api pins against the oracle's)."""
    import numpy as np

    host = np.empty(int(meta.num_entries), dtype={4: np.float32, 2: np.uint16, 1: np.uint8}[meta.num_bytes])
    rc = lib.kb_copy_block_to_cpu(host.ctypes.data, arr, meta.total_array_size)
    assert rc == 0
    pp = orc.PsiPhi.__new__(orc.PsiPhi)
    pp.meta = orc.Meta(meta.num_times, meta.width, meta.height, meta.num_bytes, meta.psi_min_val, meta.psi_max_val,
                       meta.psi_scale, meta.phi_min_val, meta.phi_max_val, meta.phi_scale)
    pp.array = host
    pp.times = np.ascontiguousarray(times, dtype=np.float64)
    pp.T, pp.H, pp.W, pp.nb = int(meta.num_times), int(meta.height), int(meta.width), int(meta.num_bytes)
    return pp


@pytest.mark.parametrize("num_bytes,sigmag,mask_fraction", [(-1, False, 0.0), (1, True, 0.0), (-1, False, 0.01)])
def test_headline_search_windows_against_the_oracle(orc, num_bytes, sigmag, mask_fraction):
    """configs[1] / configs[2] at full size (64 x 512 x 512, 1024 candidates, the bench's own stack; and configs[1] on the
    stack with 1 % of its science pixels masked, the bench line's `masked` entry): windows of
    the product's result buffer against the ORACLE on the same array -- the kernel-semantics search bit for bit
    and, for configs[1] (min_obs 0, no sigma-G: the regime where the reference's CPU and GPU searches select the
    same trajectories), the reference CPU StackSearch semantics as per-pixel likelihood lists."""
    import ctypes as C

    import numpy as np
    import torch

    import bench
    from kbmod_amd import capi
    from kbmod_amd import fake_data as fd

    lib = capi.load_lib()
    dev = torch.device("cuda")
    T, H, W, K = 64, 512, 512, 8
    sci, var, times, psf = bench.synthetic_stack(torch, dev, T, H, W, mask_fraction)
    psf_all = np.ascontiguousarray(np.tile(psf.ravel(), T), dtype=np.float32)
    psf_dims = np.full(T, psf.shape[0], dtype=np.int32)
    meta, arr = capi.Meta(), C.c_void_p()
    stream = torch.cuda.current_stream().cuda_stream
    capi.check(lib.kb_build_psi_phi_from_device(sci.data_ptr(), var.data_ptr(), psf_all.ctypes.data, psf_dims.ctypes.data,
                                                T, H, W, num_bytes, C.byref(meta), C.byref(arr), stream))
    torch.cuda.synchronize()
    vx, vy = fd.kbmod_v1_candidates(32, 5.0, 40.0, 32, 0.0, 1.5)
    cands_np = np.zeros((len(vx), 7), dtype=np.float32)
    cands_np[:, 0], cands_np[:, 1] = vx, vy
    cands = torch.from_numpy(cands_np).to(dev)
    nb = -1 if num_bytes in (-1, 4) else num_bytes
    if sigmag:
        params = capi.Params(T // 2, 10.0, 1, 0.25, 0.75, 0.7413, nb, 0, W, 0, H, K, 0)
    else:
        params = capi.Params(0, 0.0, 0, 0.25, 0.75, -1.0, nb, 0, W, 0, H, K, 0)
    results = torch.empty((H * W * K, 7), dtype=torch.float32, device=dev)
    st = capi.Stats()
    capi.check(lib.kb_device_search_filter(C.byref(meta), arr, times.data_ptr(), params, cands.data_ptr(), len(vx),
                                           results.data_ptr(), H * W * K, 0, stream, C.byref(st)))
    torch.cuda.synchronize()
    assert st.kernel_variant // 10000 == 2  # kb_search_lds on the float copy
    got = results.cpu().numpy().reshape(H, W, K, 7)

    pp = _oracle_view(orc, lib, meta, arr, times.cpu().numpy())
    ocands = orc.make_candidates(vx, vy)
    # windows: an image corner (trajectories leave the image), the interior, and a mover's neighbourhood
    obj_rng = np.random.default_rng(99)
    mx, my = int(obj_rng.integers(20, W - 60)), int(obj_rng.integers(20, H - 60))
    for (x0, y0, w, h) in [(0, 0, 48, 6), (230, 250, 40, 4), (max(0, mx - 8), max(0, my - 2), 24, 6), (W - 40, H - 5, 40, 5)]:
        kw = dict(x_start_min=x0, x_start_max=x0 + w, y_start_min=y0, y_start_max=y0 + h, results_per_pixel=K)
        if sigmag:
            kw.update(do_sigmag_filter=1, sgl_L=0.25, sgl_H=0.75, sigmag_coeff=0.7413, min_lh=10.0, min_observations=T // 2)
        p = pp.default_params(**kw)
        exp = pp.search_kernel_semantics(ocands, p).reshape(h, w, K)
        win = got[y0:y0 + h, x0:x0 + w]
        for i, name in enumerate(("vx", "vy", "lh", "flux", "x", "y", "obs_count")):
            col = win[..., i]
            if name in ("x", "y", "obs_count"):
                col = col.view(np.int32)
            assert np.array_equal(col, exp[name]), (name, (x0, y0))
        if not sigmag:
            cpu = pp.search_cpu(ocands, p).reshape(h, w, K)
            assert np.array_equal(win[..., 2], cpu["lh"]), ("cpu semantics lh", (x0, y0))
            # pixels whose K + 1 best likelihoods are distinct: the selection is unique there
            kw1 = dict(kw, results_per_pixel=K + 1)
            lh = pp.search_cpu(ocands, pp.default_params(**kw1)).reshape(h, w, K + 1)["lh"]
            untied = ~(lh[..., :-1] == lh[..., 1:]).any(axis=-1)
            assert untied.any()
            for i, name in ((0, "vx"), (1, "vy"), (3, "flux")):
                assert np.array_equal(win[..., i][untied], cpu[name][untied]), ("cpu semantics", name, (x0, y0))
    lib.kb_free_gpu_block(arr)


def _row_band_view(orc, lib, meta, arr, times, y_lo, y_hi):
    """Rows [y_lo, y_hi) of every epoch of the device array, full width, wrapped for the oracle as an array of y_hi - y_lo
    rows (start row y of the band = image row y + y_lo).  The arrays of configs[3] / configs[4] do not fit a host copy in
    seconds; a window's trajectories only ever read the rows between the window and its farthest shift, and with epochs at
    i / T days (T a power of two) every predicted position x + v t + 0.5 is exact in double, so the translation changes no
    rounding: the band search equals the full-array search of the same start pixels as long as the band holds every row they
    reach -- rows beyond the band's end are beyond the image's end too (the caller clips y_hi to H), or unreachable."""
    import numpy as np

    T, H, W = int(meta.num_times), int(meta.height), int(meta.width)
    rows = y_hi - y_lo
    per_px = 2 * int(meta.block_size)
    dtype = {4: np.float32, 2: np.uint16, 1: np.uint8}[meta.num_bytes]
    host = np.empty((T, rows, W, 2), dtype=dtype)
    for t in range(T):   # one contiguous block of rows per epoch
        src = arr.value + (t * H + y_lo) * W * per_px
        assert lib.kb_copy_block_to_cpu(host[t].ctypes.data, src, rows * W * per_px) == 0
    pp = orc.PsiPhi.__new__(orc.PsiPhi)
    pp.meta = orc.Meta(T, W, rows, meta.num_bytes, meta.psi_min_val, meta.psi_max_val, meta.psi_scale, meta.phi_min_val,
                       meta.phi_max_val, meta.phi_scale)
    pp.array = host.reshape(-1)
    pp.times = np.ascontiguousarray(times, dtype=np.float64)
    pp.T, pp.H, pp.W, pp.nb = T, rows, W, int(meta.num_bytes)
    return pp


BIG = {
    # configs[3], one GPU's share: 128 x 4096 x 4096 float32, 64 candidates (bench.py --frames 128 --size 4096 --vel-steps 32 --ang-steps 2)
    "cfg4_shard": dict(T=128, N=4096, vel=32, ang=2, num_bytes=-1, win=(96, 8)),
    # configs[4]: 512 x 2048 x 2048 uint16, 4096 candidates (3e8 oracle evaluations per 24 x 6 window)
    "cfg5_deep_u16": dict(T=512, N=2048, vel=64, ang=64, num_bytes=2, win=(24, 6)),
}


@pytest.mark.parametrize("name", list(BIG))
def test_hbm_resident_search_windows_against_the_oracle(orc, name):
    """configs[3] (one GPU's share) and configs[4] AT FULL SIZE against the oracle itself: the bench's own stack and candidate
    list, one whole-grid search through the C ABI, then four windows of the result buffer -- an image corner (trajectories
    start on the border), the interior, a mover's neighbourhood, the far edge (trajectories leave the image) -- compared on all
    seven fields with the oracle's kernel-semantics search over the band of rows those start pixels can reach."""
    import ctypes as C

    import numpy as np
    import torch

    import bench
    from kbmod_amd import capi
    from kbmod_amd import fake_data as fd

    cfg = BIG[name]
    lib = capi.load_lib()
    dev = torch.device("cuda")
    T, H, W, K = cfg["T"], cfg["N"], cfg["N"], 8
    sci, var, times, psf = bench.synthetic_stack(torch, dev, T, H, W)
    psf_all = np.ascontiguousarray(np.tile(psf.ravel(), T), dtype=np.float32)
    psf_dims = np.full(T, psf.shape[0], dtype=np.int32)
    meta, arr = capi.Meta(), C.c_void_p()
    stream = torch.cuda.current_stream().cuda_stream
    capi.check(lib.kb_build_psi_phi_from_device(sci.data_ptr(), var.data_ptr(), psf_all.ctypes.data, psf_dims.ctypes.data,
                                                T, H, W, cfg["num_bytes"], C.byref(meta), C.byref(arr), stream))
    torch.cuda.synchronize()
    del sci, var
    vx, vy = fd.kbmod_v1_candidates(cfg["vel"], 5.0, 40.0, cfg["ang"], 0.0, 1.5)
    cands_np = np.zeros((len(vx), 7), dtype=np.float32)
    cands_np[:, 0], cands_np[:, 1] = vx, vy
    cands = torch.from_numpy(cands_np).to(dev)
    nb = -1 if cfg["num_bytes"] in (-1, 4) else cfg["num_bytes"]
    params = capi.Params(0, 0.0, 0, 0.25, 0.75, -1.0, nb, 0, W, 0, H, K, 0)
    results = torch.empty((H * W * K, 7), dtype=torch.float32, device=dev)
    st = capi.Stats()
    capi.check(lib.kb_device_search_filter(C.byref(meta), arr, times.data_ptr(), params, cands.data_ptr(), len(vx),
                                           results.data_ptr(), H * W * K, 0, stream, C.byref(st)))
    torch.cuda.synchronize()
    assert st.kernel_variant // 10000 == 2, st.kernel_name   # kb_search_lds on the canonical float copy
    grid = results.view(H, W, K, 7)

    tcpu = times.cpu().numpy()
    assert float(vy.min()) >= 0.0 and float(vx.min()) >= 0.0   # (the grid's angles: shifts towards +x, +y only)
    reach = int(np.ceil(float(max(vx.max(), vy.max())) * float(tcpu[-1]))) + 2
    ocands = orc.make_candidates(vx, vy)
    obj_rng = np.random.default_rng(99)
    mx, my = int(obj_rng.integers(20, W - 60)), int(obj_rng.integers(20, H - 60))
    w, h = cfg["win"]
    for (x0, y0) in [(0, 0), (W // 2 - 19, H // 2 + 7), (max(0, mx - w // 3), max(0, my - 2)), (W - w, H - h)]:
        y_lo, y_hi = y0, min(H, y0 + h + reach)
        pp = _row_band_view(orc, lib, meta, arr, tcpu, y_lo, y_hi)
        p = pp.default_params(x_start_min=x0, x_start_max=x0 + w, y_start_min=0, y_start_max=h, results_per_pixel=K)
        exp = pp.search_kernel_semantics(ocands, p).reshape(h, w, K)
        win = grid[y0:y0 + h, x0:x0 + w].cpu().numpy()
        for i, field in enumerate(("vx", "vy", "lh", "flux", "x", "y", "obs_count")):
            col = win[..., i]
            want = exp[field]
            if field in ("x", "y", "obs_count"):
                col = col.view(np.int32)
            if field == "y":
                want = want + y_lo   # band rows -> image rows
            assert np.array_equal(col, want), (name, field, (x0, y0))
        assert (exp["lh"] > -1e30).all()   # (every slot filled: min_lh 0 on noise)
    lib.kb_free_gpu_block(arr)


def test_readme_configuration_at_full_size(kb, orc):
    """BASELINE configs[0] as stated: 10 x 512 x 512 float32, 25 candidates of KBMODV1Search(5, 0, 4, 5, -0.1, 0.1),
    min_obs 7 -- psi/phi from the HIP builder, then the CPU StackSearch (on_gpu=False) against the oracle's
    restatement of the reference CPU search, and the GPU search against the kernel-semantics oracle."""
    import numpy as np

    from kbmod_amd import fake_data as fd
    from tests import util

    times = fd.create_fake_times(10, 57130.2)
    st = fd.make_fake_image_stack(512, 512, times, noise_level=2.0, psf_val=0.5, rng=np.random.default_rng(1))
    fd.add_fake_object(st, 2, 0, 10.7, 15.3, flux=275.0)
    vx, vy = fd.kbmod_v1_candidates(5, 0, 4, 5, -0.1, 0.1)
    assert len(vx) == 25
    pp = orc.PsiPhi.from_images(st.sci, st.var, st.psfs, st.zeroed_times)
    ocands = orc.make_candidates(vx, vy)
    for on_gpu in (False, True):
        s = kb.StackSearch(st.sci, st.var, st.psfs, st.zeroed_times)
        assert s.get_psi_phi_array().device_resident
        s.set_min_obs(7)
        s.search_all(util.trajectories(kb, vx, vy), on_gpu)
        got = s.results_to_numpy()
        p = pp.default_params(min_observations=7)
        raw = pp.search_kernel_semantics(ocands, p) if on_gpu else pp.search_cpu(ocands, p)
        exp = util.as_table(orc.filter_sort(raw, 0.0, 7))
        assert got.shape == exp.shape and np.array_equal(got, exp), on_gpu
        assert len(got) > 1000
