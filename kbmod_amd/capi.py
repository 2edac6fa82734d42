"""ctypes view of the C ABI of ``lib/libkbmod_hip.so`` (include/kbmod_hip.h): the structs and the
prototypes of the entry points that Python callers (bench.py, kbmod_amd.distributed, the GPU tests)
drive directly with raw device pointers.  No compute happens here and nothing falls back: loading
fails when the library is not built."""

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))


class Meta(C.Structure):
    """kb_psi_phi_meta"""
    _fields_ = [(n, C.c_uint64) for n in ("num_times", "width", "height", "pixels_per_image", "num_entries",
                                          "block_size", "total_array_size")] + [
        ("num_bytes", C.c_int32), ("psi_min_val", C.c_float), ("psi_max_val", C.c_float), ("psi_scale", C.c_float),
        ("phi_min_val", C.c_float), ("phi_max_val", C.c_float), ("phi_scale", C.c_float)]


class Params(C.Structure):
    """kb_search_params"""
    _fields_ = [("min_observations", C.c_int32), ("min_lh", C.c_float), ("do_sigmag_filter", C.c_uint8),
                ("sgl_L", C.c_float), ("sgl_H", C.c_float), ("sigmag_coeff", C.c_float),
                ("encode_num_bytes", C.c_int32), ("x_start_min", C.c_int32), ("x_start_max", C.c_int32),
                ("y_start_min", C.c_int32), ("y_start_max", C.c_int32), ("results_per_pixel", C.c_uint32),
                ("total_results", C.c_ulonglong)]


class Stats(C.Structure):
    """kb_search_stats"""
    _fields_ = [("search_kernel_ms", C.c_float), ("table_kernel_ms", C.c_float), ("num_evals", C.c_uint64),
                ("algorithmic_bytes", C.c_uint64), ("kernel_variant", C.c_int32), ("num_search_launches", C.c_int32),
                ("sigmag_work_items", C.c_uint64), ("sigmag_trajectories", C.c_uint64), ("lds_read_bytes", C.c_uint64),
                ("sigmag_literal", C.c_uint64), ("kernel_name", C.c_char * 96), ("padded_copy_reused", C.c_int32),
                ("special_epochs", C.c_int32), ("edge_count_tables", C.c_int32), ("env_overrides", C.c_int32)]


def lib_path():
    return os.environ.get("KBMOD_HIP_LIB", os.path.join(_PKG, "lib", "libkbmod_hip.so"))


_lib = None


def load_lib():
    """The device library with argument types set; raises when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError("libkbmod_hip.so is not built (run __graft_entry__.build()); there is no fallback path")
    lib = C.CDLL(path)
    lib.kb_last_error.restype = C.c_char_p
    lib.kb_last_build_kernel_ms.restype = C.c_float
    lib.kb_last_build_kernel_ms.argtypes = []
    lib.kb_build_psi_phi_from_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                                 C.c_int32, C.c_int32, C.POINTER(Meta), C.POINTER(C.c_void_p), C.c_void_p]
    lib.kb_build_psi_phi_from_device_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                                    C.c_int32, C.c_int32, C.c_uint32, C.POINTER(Meta), C.POINTER(C.c_void_p),
                                                    C.c_void_p]
    lib.kb_device_search_filter.argtypes = [C.POINTER(Meta), C.c_void_p, C.c_void_p, Params, C.c_void_p, C.c_uint64,
                                            C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.POINTER(Stats)]
    lib.kb_device_search_compact.argtypes = [C.POINTER(Meta), C.c_void_p, C.c_void_p, Params, C.c_void_p, C.c_uint64,
                                             C.c_int32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.POINTER(Stats)]
    lib.kb_merge_topk.argtypes = [C.c_void_p, C.c_int32, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p]
    lib.kb_merge_compact.argtypes = [C.c_void_p, C.c_int32, Params, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.kb_merge_compact_exact.argtypes = [C.c_void_p, C.c_int32, C.c_int32, Params, C.c_void_p, C.c_uint64, C.c_void_p,
                                           C.c_void_p]
    lib.kb_merge_compact_repairable.argtypes = [C.c_void_p, C.c_int32, Params, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                                C.POINTER(C.c_uint64), C.c_void_p]
    lib.kb_repair_pixels.argtypes = [C.POINTER(Meta), C.c_void_p, C.c_void_p, Params, C.c_void_p, C.c_uint64, C.c_void_p,
                                     C.c_uint64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.kb_sparse_header_bytes.restype = C.c_uint64
    lib.kb_sparse_header_bytes.argtypes = [C.c_uint64]
    lib.kb_sparsify_compact.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_uint64,
                                        C.POINTER(C.c_uint64), C.c_void_p]
    lib.kb_device_search_counted.argtypes = [C.POINTER(Meta), C.c_void_p, C.c_void_p, Params, C.c_void_p, C.c_uint64, C.c_int32,
                                             C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(Stats),
                                             C.POINTER(C.c_int32)]
    lib.kb_device_search_filter_counted.argtypes = [C.POINTER(Meta), C.c_void_p, C.c_void_p, Params, C.c_void_p, C.c_uint64,
                                                    C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(Stats),
                                                    C.POINTER(C.c_int32)]
    lib.kb_filter_sort_results_counted.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_float, C.c_int32, C.c_void_p,
                                                   C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.c_void_p]
    lib.kb_sparsify_counted.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64,
                                        C.POINTER(C.c_uint64), C.c_void_p]
    lib.kb_merge_sparse_exact.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, Params,
                                          C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.kb_merge_sparse_exact_counted.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, Params,
                                                  C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.kb_free_gpu_block.argtypes = [C.c_void_p]
    lib.kb_note_array_written.argtypes = [C.c_void_p]
    lib.kb_copy_block_to_cpu.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    lib.kb_copy_block_to_gpu.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    lib.kb_measure_copy_bandwidth.argtypes = [C.c_uint64, C.c_int32, C.c_void_p, C.POINTER(C.c_double)]
    lib.kb_measure_read_bandwidth.argtypes = [C.c_uint64, C.c_int32, C.c_void_p, C.POINTER(C.c_double)]
    lib.kb_measure_lds_bandwidth.argtypes = [C.c_int32, C.c_void_p, C.POINTER(C.c_double)]
    lib.kb_debug_wave_ops.argtypes = [C.c_void_p] * 6 + [C.c_uint64, C.c_void_p]
    _lib = lib
    return lib


def check(rc):
    """Raise the library's message for a non-zero status."""
    if rc != 0:
        raise RuntimeError(load_lib().kb_last_error().decode())
