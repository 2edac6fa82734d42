"""The C-ABI library loads without a GPU and exports every entry point that
include/kbmod_hip.h declares (no compute calls here)."""

import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "kbmod_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kb_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_path():
    names = declared_functions()
    for required in ("kb_device_search_filter", "kb_evaluate_trajectory_host", "kb_sigmag_filtered_indices",
                     "kb_device_convolve", "kb_build_psi_phi_from_device", "kb_allocate_gpu_block",
                     "kb_free_gpu_block", "kb_copy_block_to_gpu", "kb_copy_block_to_cpu", "kb_merge_topk"):
        assert required in names


def test_library_exports_every_declared_symbol(kb):
    lib = ctypes.CDLL(os.path.join(ROOT, "kbmod_amd", "lib", "libkbmod_hip.so"))
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing


def test_struct_sizes_match_reference_layout(kb):
    # common.h:55-68 (28 bytes, pinned by the reference's tests/test_trajectory_list.py:28,47)
    assert kb.TrajectoryList.estimate_memory(1) == 28
    lst = kb.TrajectoryList(10)
    assert lst.get_memory() == 280


def test_no_gpu_is_reported_not_hidden(kb):
    lib = ctypes.CDLL(os.path.join(ROOT, "kbmod_amd", "lib", "libkbmod_hip.so"))
    lib.kb_last_error.restype = ctypes.c_char_p
    n = lib.kb_device_count()
    assert n >= 0
    assert kb.kb_has_gpu() == (n > 0)
    if n == 0:
        p = ctypes.c_void_p()
        # device entry points fail loudly instead of falling back
        rc = lib.kb_device_convolve(None, None, 4, 4, None, 1, 0)
        assert rc != 0 and lib.kb_last_error()
