"""Shared helpers of the parity tests (seeded inputs, oracle comparison)."""

import numpy as np

from kbmod_amd import fake_data as fd

FIELDS = ("x", "y", "vx", "vy", "lh", "flux", "obs_count")


def make_stack(T, H, W, seed, noise=2.0, psf=1.0, objects=(), mask_fraction=0.0, times=None):
    rng = np.random.default_rng(seed)
    if times is None:
        times = np.arange(T) / float(T)
    stack = fd.make_fake_image_stack(H, W, times, noise_level=noise, psf_val=psf, rng=rng)
    if mask_fraction > 0:
        fd.add_random_masks(stack, mask_fraction, rng)
    for (x, y, vx, vy, flux) in objects:
        fd.add_fake_object(stack, x, y, vx, vy, flux=flux)
    return stack


def as_table(res):
    """Oracle structured array -> (N, 7) float64 table in results_to_numpy() column order."""
    return np.stack([res[k] for k in FIELDS], 1).astype(np.float64)


def trajectories(kb, vx, vy):
    return [kb.Trajectory(vx=float(a), vy=float(b)) for a, b in zip(vx, vy)]


def oracle_params(pp, search_cfg):
    """Params for the oracle from a dict of StackSearch-style settings."""
    kw = {}
    if "min_obs" in search_cfg:
        kw["min_observations"] = search_cfg["min_obs"]
    if "min_lh" in search_cfg:
        kw["min_lh"] = search_cfg["min_lh"]
    if "K" in search_cfg:
        kw["results_per_pixel"] = search_cfg["K"]
    if "xb" in search_cfg:
        kw["x_start_min"], kw["x_start_max"] = search_cfg["xb"]
    if "yb" in search_cfg:
        kw["y_start_min"], kw["y_start_max"] = search_cfg["yb"]
    if "sigmag" in search_cfg:
        lo, hi, coeff, min_lh = search_cfg["sigmag"]
        kw.update(do_sigmag_filter=1, sgl_L=lo, sgl_H=hi, sigmag_coeff=coeff, min_lh=min_lh)
    return pp.default_params(**kw)


def configure(search, cfg):
    if "min_obs" in cfg:
        search.set_min_obs(cfg["min_obs"])
    if "min_lh" in cfg:
        search.set_min_lh(cfg["min_lh"])
    if "K" in cfg:
        search.set_results_per_pixel(cfg["K"])
    if "xb" in cfg:
        search.set_start_bounds_x(*cfg["xb"])
    if "yb" in cfg:
        search.set_start_bounds_y(*cfg["yb"])
    if "sigmag" in cfg:
        lo, hi, coeff, min_lh = cfg["sigmag"]
        search.enable_gpu_sigmag_filter([lo, hi], coeff, min_lh)


def run_both(kb, orc, stack, vx, vy, cfg, num_bytes=-1, on_gpu=True, flags=0):
    """(product table, oracle table) for one configuration."""
    search = kb.StackSearch(stack.sci, stack.var, stack.psfs, stack.zeroed_times, num_bytes)
    configure(search, cfg)
    search.set_search_flags(flags)
    search.search_all(trajectories(kb, vx, vy), on_gpu)
    got = search.results_to_numpy()

    pp = orc.PsiPhi.from_images(stack.sci, stack.var, stack.psfs, stack.zeroed_times, num_bytes)
    params = oracle_params(pp, cfg)
    cands = orc.make_candidates(vx, vy)
    raw = pp.search_kernel_semantics(cands, params) if on_gpu else pp.search_cpu(cands, params)
    exp = as_table(orc.filter_sort(raw, params.min_lh, params.min_observations))
    return got, exp, search


class DeviceStack:
    """psi/phi of a fake stack built in HBM through the C ABI (kb_build_psi_phi_from_device), for tests that
    drive kb_device_search_* with raw device pointers the way bench.py and kbmod_amd.distributed do."""

    def __init__(self, stack, num_bytes=-1):
        import ctypes as C

        import torch

        from kbmod_amd import capi

        self.lib = capi.load_lib()
        self.torch = torch
        dev = torch.device("cuda")
        T = len(stack.sci)
        H, W = stack.sci[0].shape
        sci = torch.from_numpy(np.stack(stack.sci).astype(np.float32)).to(dev)
        var = torch.from_numpy(np.stack(stack.var).astype(np.float32)).to(dev)
        psf_all = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.float32).ravel() for p in stack.psfs]))
        psf_dims = np.array([np.asarray(p).shape[0] for p in stack.psfs], dtype=np.int32)
        self.meta = capi.Meta()
        self.arr = C.c_void_p()
        self.stream = torch.cuda.current_stream().cuda_stream
        capi.check(self.lib.kb_build_psi_phi_from_device(sci.data_ptr(), var.data_ptr(), psf_all.ctypes.data,
                                                         psf_dims.ctypes.data, T, H, W, num_bytes, C.byref(self.meta),
                                                         C.byref(self.arr), self.stream))
        torch.cuda.synchronize()
        self.times = torch.from_numpy(np.asarray(stack.zeroed_times, dtype=np.float64)).to(dev)
        self.T, self.H, self.W = T, H, W
        self.num_bytes = num_bytes

    def params(self, K=8, min_obs=0, min_lh=0.0, sigmag=None, xb=None, yb=None):
        from kbmod_amd import capi

        xb = xb or (0, self.W)
        yb = yb or (0, self.H)
        nb = -1 if self.num_bytes in (-1, 4) else self.num_bytes
        if sigmag is not None:
            lo, hi, coeff, mlh = sigmag
            return capi.Params(min_obs, mlh, 1, lo, hi, coeff, nb, xb[0], xb[1], yb[0], yb[1], K, 0)
        return capi.Params(min_obs, min_lh, 0, 0.25, 0.75, -1.0, nb, xb[0], xb[1], yb[0], yb[1], K, 0)

    def candidates(self, vx, vy):
        c = np.zeros((len(vx), 7), dtype=np.float32)
        c[:, 0], c[:, 1] = vx, vy
        return self.torch.from_numpy(c).cuda()

    def search(self, params, cands, flags=0):
        """kb_device_search_filter -> structured numpy records [S*K]."""
        import ctypes as C

        from kbmod_amd import capi

        n = (params.x_start_max - params.x_start_min) * (params.y_start_max - params.y_start_min) * params.results_per_pixel
        out = self.torch.empty((n, 7), dtype=self.torch.float32, device="cuda")
        st = capi.Stats()
        capi.check(self.lib.kb_device_search_filter(C.byref(self.meta), self.arr, self.times.data_ptr(), params,
                                                    cands.data_ptr(), cands.shape[0], out.data_ptr(), n, flags, self.stream,
                                                    C.byref(st)))
        self.torch.cuda.synchronize()
        return out, st

    def search_compact(self, params, cands, cand_base, flags=0):
        import ctypes as C

        from kbmod_amd import capi

        n = (params.x_start_max - params.x_start_min) * (params.y_start_max - params.y_start_min) * params.results_per_pixel
        out = self.torch.empty((n, 4), dtype=self.torch.int32, device="cuda")
        st = capi.Stats()
        capi.check(self.lib.kb_device_search_compact(C.byref(self.meta), self.arr, self.times.data_ptr(), params,
                                                     cands.data_ptr(), cands.shape[0], cand_base, out.data_ptr(), n, flags,
                                                     self.stream, C.byref(st)))
        self.torch.cuda.synchronize()
        return out, st

    def search_counted(self, params, cands, cand_base, flags=0, poison=None):
        """kb_device_search_counted -> (records, counts header, counts_written, stats); ``poison``: the int32 value the record
        buffer is filled with first (records the search skipped keep it)."""
        import ctypes as C

        from kbmod_amd import capi

        S = (params.x_start_max - params.x_start_min) * (params.y_start_max - params.y_start_min)
        n = S * params.results_per_pixel
        out = self.torch.empty((n, 4), dtype=self.torch.int32, device="cuda")
        if poison is not None:
            out.fill_(poison)
        header = self.torch.full((int(self.lib.kb_sparse_header_bytes(S)),), 0xEE, dtype=self.torch.uint8, device="cuda")
        st, written = capi.Stats(), C.c_int32(-1)
        capi.check(self.lib.kb_device_search_counted(C.byref(self.meta), self.arr, self.times.data_ptr(), params,
                                                     cands.data_ptr(), cands.shape[0], cand_base, out.data_ptr(), n,
                                                     header.data_ptr(), flags, self.stream, C.byref(st), C.byref(written)))
        self.torch.cuda.synchronize()
        return out, header, int(written.value), st

    def close(self):
        if self.arr:
            self.lib.kb_free_gpu_block(self.arr)
            self.arr = None


TRJ_DTYPE = np.dtype([("vx", "<f4"), ("vy", "<f4"), ("lh", "<f4"), ("flux", "<f4"), ("x", "<i4"), ("y", "<i4"),
                      ("obs_count", "<i4")])
COMPACT_DTYPE = np.dtype([("lh", "<f4"), ("flux", "<f4"), ("cand", "<i4"), ("obs_count", "<i4")])


def as_records(t, dtype=TRJ_DTYPE):
    return t.cpu().numpy().reshape(-1).view(dtype)
