"""ctypes front-end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product (``kbmod_amd``) never does.

Wraps ``oracle/libkbmod_oracle.so`` (the plain-C restatement in
``kbmod_oracle.c``) and, when it has been built in this container,
``oracle/_ref/libkbmod_ref.so`` (shims around the buildable pieces of the real
reference, see ``ref_driver.cpp``).
"""

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

TRJ_DTYPE = np.dtype(
    [("vx", "<f4"), ("vy", "<f4"), ("lh", "<f4"), ("flux", "<f4"), ("x", "<i4"), ("y", "<i4"), ("obs_count", "<i4")]
)
assert TRJ_DTYPE.itemsize == 28


class Params(C.Structure):
    _fields_ = [
        ("min_observations", C.c_int32),
        ("min_lh", C.c_float),
        ("do_sigmag_filter", C.c_int32),
        ("sgl_L", C.c_float),
        ("sgl_H", C.c_float),
        ("sigmag_coeff", C.c_float),
        ("x_start_min", C.c_int32),
        ("x_start_max", C.c_int32),
        ("y_start_min", C.c_int32),
        ("y_start_max", C.c_int32),
        ("results_per_pixel", C.c_uint32),
    ]


class Meta(C.Structure):
    _fields_ = [
        ("num_times", C.c_uint64),
        ("width", C.c_uint64),
        ("height", C.c_uint64),
        ("num_bytes", C.c_int32),
        ("psi_min_val", C.c_float),
        ("psi_max_val", C.c_float),
        ("psi_scale", C.c_float),
        ("phi_min_val", C.c_float),
        ("phi_max_val", C.c_float),
        ("phi_scale", C.c_float),
    ]


def build(force=False):
    """Compile the C restatement (and the reference shims when /root/reference exists)."""
    so = os.path.join(_HERE, "libkbmod_oracle.so")
    src = os.path.join(_HERE, "kbmod_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libkbmod_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src/kbmod/search"):
        ref = os.path.join(_HERE, "_ref", "libkbmod_ref.so")
        rsrc = os.path.join(_HERE, "ref_driver.cpp")
        if force or not os.path.exists(ref) or os.path.getmtime(ref) < os.path.getmtime(rsrc):
            subprocess.check_call(["make", "-C", _HERE, "-B", "ref"], stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "libkbmod_oracle.so")
        if not os.path.exists(so):
            build()
        _lib = C.CDLL(so)
        fp = C.POINTER(C.c_float)
        _lib.orc_encode_uint_scalar.restype = C.c_float
        _lib.orc_encode_uint_scalar.argtypes = [C.c_float] * 4
        _lib.orc_decode_uint_scalar.restype = C.c_float
        _lib.orc_decode_uint_scalar.argtypes = [C.c_float] * 3
        _lib.orc_filter_sort.restype = C.c_uint64
        _lib.orc_filter_sort.argtypes = [C.c_void_p, C.c_uint64, C.c_float, C.c_int]
        _lib.orc_num_threads.restype = C.c_int
        _lib.orc_convolve.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        _lib.orc_square_psf.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _lib.orc_generate_psi.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        _lib.orc_generate_phi.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        _lib.orc_scale_params.argtypes = [C.c_void_p, C.c_uint64, C.c_int, fp]
        _lib.orc_fill_array.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Meta), C.c_void_p]
        _lib.orc_read_psi_phi.argtypes = [C.POINTER(Meta), C.c_void_p, C.c_uint64, C.c_int, C.c_int, fp, fp]
        _lib.orc_evaluate_trajectory_cpu.argtypes = [C.POINTER(Meta), C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_sigmag_filtered_indices.argtypes = [
            C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p,
            C.POINTER(C.c_int), C.POINTER(C.c_int),
        ]
        _lib.orc_evaluate_trajectory_kernel.argtypes = [
            C.POINTER(Meta), C.c_void_p, C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_int,
        ]
        _lib.orc_search_cpu.argtypes = [
            C.POINTER(Meta), C.c_void_p, C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint64, C.c_void_p,
        ]
        _lib.orc_search_kernel_semantics.argtypes = [
            C.POINTER(Meta), C.c_void_p, C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint64, C.c_void_p, C.c_int,
        ]
        _lib.orc_psi_phi_curve.argtypes = [C.POINTER(Meta), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return _lib


def ref_lib():
    """The real-reference shim library, or None when it has not been built."""
    global _ref
    if _ref is None:
        so = os.path.join(_HERE, "_ref", "libkbmod_ref.so")
        if not os.path.exists(so):
            return None
        _ref = C.CDLL(so)
        _ref.ref_encode_uint_scalar.restype = C.c_float
        _ref.ref_encode_uint_scalar.argtypes = [C.c_float] * 4
        _ref.ref_decode_uint_scalar.restype = C.c_float
        _ref.ref_decode_uint_scalar.argtypes = [C.c_float] * 3
        for n in ("ref_get_x_pos", "ref_get_y_pos"):
            getattr(_ref, n).restype = C.c_float
            getattr(_ref, n).argtypes = [C.c_void_p, C.c_double, C.c_int]
        for n in ("ref_get_x_index", "ref_get_y_index"):
            getattr(_ref, n).restype = C.c_int
            getattr(_ref, n).argtypes = [C.c_void_p, C.c_double]
        _ref.ref_is_valid.argtypes = [C.c_void_p]
        _ref.ref_list_filter_sort.restype = C.c_uint64
        _ref.ref_list_filter_sort.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        _ref.ref_list_get_batch.restype = C.c_int64
        _ref.ref_list_get_batch.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
    return _ref


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# ----------------------------------------------------------------------------
# image functions
# ----------------------------------------------------------------------------
def convolve(img, psf, gpu_flavour=False):
    img = _f32(img)
    psf = _f32(psf)
    out = np.empty_like(img)
    lib().orc_convolve(_ptr(img), img.shape[0], img.shape[1], _ptr(psf), psf.shape[0], _ptr(out), int(gpu_flavour))
    return out


def square_psf(psf):
    psf = _f32(psf)
    out = np.empty_like(psf)
    lib().orc_square_psf(_ptr(psf), psf.size, _ptr(out))
    return out


def generate_psi(sci, var, psf, gpu_flavour=False):
    sci, var, psf = _f32(sci), _f32(var), _f32(psf)
    out = np.empty_like(sci)
    lib().orc_generate_psi(_ptr(sci), _ptr(var), sci.shape[0], sci.shape[1], _ptr(psf), psf.shape[0], _ptr(out), int(gpu_flavour))
    return out


def generate_phi(var, psf, gpu_flavour=False):
    var, psf = _f32(var), _f32(psf)
    out = np.empty_like(var)
    lib().orc_generate_phi(_ptr(var), var.shape[0], var.shape[1], _ptr(psf), psf.shape[0], _ptr(out), int(gpu_flavour))
    return out


def encode_uint_scalar(v, mn, mx, sc):
    return float(lib().orc_encode_uint_scalar(v, mn, mx, sc))


def decode_uint_scalar(v, mn, sc):
    return float(lib().orc_decode_uint_scalar(v, mn, sc))


def scale_params(imgs, num_bytes):
    flat = np.concatenate([_f32(i).ravel() for i in imgs])
    out = (C.c_float * 3)()
    lib().orc_scale_params(_ptr(flat), flat.size, num_bytes, out)
    return [out[0], out[1], out[2]]


class PsiPhi:
    """An encoded, interleaved [t][row][col][psi,phi] array plus its meta data."""

    def __init__(self, psi_imgs, phi_imgs, times, num_bytes=4):
        psi = np.stack([_f32(p) for p in psi_imgs])
        phi = np.stack([_f32(p) for p in phi_imgs])
        T, H, W = psi.shape
        nb = 4 if num_bytes in (-1, 4) else num_bytes
        self.meta = Meta(T, W, H, nb, 3.4028234663852886e38, -3.4028234663852886e38, 1.0,
                         3.4028234663852886e38, -3.4028234663852886e38, 1.0)
        if nb in (1, 2):
            p = scale_params([psi], nb)
            self.meta.psi_min_val, self.meta.psi_max_val, self.meta.psi_scale = p
            p = scale_params([phi], nb)
            self.meta.phi_min_val, self.meta.phi_max_val, self.meta.phi_scale = p
        dt = {4: np.float32, 2: np.uint16, 1: np.uint8}[nb]
        self.array = np.empty(2 * T * H * W, dtype=dt)
        lib().orc_fill_array(_ptr(psi), _ptr(phi), C.byref(self.meta), _ptr(self.array))
        self.times = np.ascontiguousarray(times, dtype=np.float64)
        self.T, self.H, self.W, self.nb = T, H, W, nb

    @classmethod
    def from_images(cls, sci, var, psfs, times, num_bytes=4, gpu_flavour=False):
        psi = [generate_psi(s, v, p, gpu_flavour) for s, v, p in zip(sci, var, psfs)]
        phi = [generate_phi(v, p, gpu_flavour) for v, p in zip(var, psfs)]
        return cls(psi, phi, times, num_bytes)

    def read(self, t, row, col):
        a, b = C.c_float(), C.c_float()
        lib().orc_read_psi_phi(C.byref(self.meta), _ptr(self.array), t, row, col, C.byref(a), C.byref(b))
        return a.value, b.value

    def default_params(self, **kw):
        p = Params(0, 0.0, 0, 0.25, 0.75, -1.0, 0, self.W, 0, self.H, 8)
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    def evaluate_cpu(self, x, y, vx, vy):
        t = np.zeros(1, dtype=TRJ_DTYPE)
        t["x"], t["y"], t["vx"], t["vy"] = x, y, vx, vy
        lib().orc_evaluate_trajectory_cpu(C.byref(self.meta), _ptr(self.array), _ptr(self.times), _ptr(t))
        return t[0]

    def evaluate_kernel(self, x, y, vx, vy, params, max_images=1000):
        t = np.zeros(1, dtype=TRJ_DTYPE)
        t["x"], t["y"], t["vx"], t["vy"] = x, y, vx, vy
        lib().orc_evaluate_trajectory_kernel(
            C.byref(self.meta), _ptr(self.array), _ptr(self.times), C.byref(params), _ptr(t), max_images
        )
        return t[0]

    def search_cpu(self, cands, params):
        cands = np.ascontiguousarray(cands, dtype=TRJ_DTYPE)
        sw = params.x_start_max - params.x_start_min
        sh = params.y_start_max - params.y_start_min
        R = min(len(cands), params.results_per_pixel)
        res = np.zeros(R * sw * sh, dtype=TRJ_DTYPE)
        lib().orc_search_cpu(C.byref(self.meta), _ptr(self.array), _ptr(self.times), C.byref(params),
                             _ptr(cands), len(cands), _ptr(res))
        return res

    def search_kernel_semantics(self, cands, params, max_images=1000):
        cands = np.ascontiguousarray(cands, dtype=TRJ_DTYPE)
        sw = params.x_start_max - params.x_start_min
        sh = params.y_start_max - params.y_start_min
        res = np.zeros(params.results_per_pixel * sw * sh, dtype=TRJ_DTYPE)
        lib().orc_search_kernel_semantics(C.byref(self.meta), _ptr(self.array), _ptr(self.times), C.byref(params),
                                          _ptr(cands), len(cands), _ptr(res), max_images)
        return res

    def curve(self, x, y, vx, vy):
        t = np.zeros(1, dtype=TRJ_DTYPE)
        t["x"], t["y"], t["vx"], t["vy"] = x, y, vx, vy
        out = np.zeros(2 * self.T, dtype=np.float32)
        lib().orc_psi_phi_curve(C.byref(self.meta), _ptr(self.array), _ptr(self.times), _ptr(t), _ptr(out))
        return out


def filter_sort(results, min_lh, min_obs):
    res = np.ascontiguousarray(results, dtype=TRJ_DTYPE).copy()
    n = lib().orc_filter_sort(_ptr(res), len(res), min_lh, min_obs)
    return res[:n]


def sigmag_filtered_indices(values, sgl0, sgl1, coeff, width):
    v = _f32(values)
    n = len(v)
    idx = np.zeros(max(n, 1), dtype=np.int32)
    lo, hi = C.c_int(0), C.c_int(n - 1)
    lib().orc_sigmag_filtered_indices(_ptr(v), n, sgl0, sgl1, coeff, width, _ptr(idx), C.byref(lo), C.byref(hi))
    return [int(idx[i]) for i in range(lo.value, hi.value + 1)]


def num_threads():
    return lib().orc_num_threads()


def make_candidates(vxs, vys):
    c = np.zeros(len(vxs), dtype=TRJ_DTYPE)
    c["vx"] = vxs
    c["vy"] = vys
    return c
