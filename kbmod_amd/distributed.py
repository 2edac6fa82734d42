"""Trajectory-space sharding across the GPUs of one node (one process per GPU).

The reference is single-GPU (no collectives anywhere); this is
new functionality.  Every (start pixel, candidate) pair is independent and only
the per-pixel top-K couples candidates, so:

* the candidate list is split into ``world`` contiguous slices (rank r owns the
  r-th slice: for the usual (theta outer, v inner) grid that is a band of
  angles), psi/phi is replicated in every GPU's HBM;
* each rank runs the single-GPU search over all start pixels on its slice;
* ONE all_gather (RCCL over xGMI on GPUs; gloo in the CPU tests) exchanges the
  per-rank ``[S*K]`` result lists, and a per-pixel K-way merge (HIP kernel
  ``kb_merge_topk`` on device tensors, the host twin on CPU tensors) selects the
  global top-K.  Ties go to the lower rank, i.e. to the lower global candidate
  index.

torch is plumbing only here: tensors as device buffers and torch.distributed
as the RCCL front-end.
"""

import ctypes as C
import os

import numpy as np

TRJ_FLOATS = 7  # 28-byte Trajectory viewed as 7 x 32-bit words


def shard_bounds(n_items, rank, world):
    """Contiguous slice [lo, hi) of ``n_items`` owned by ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_lib = None


def device_lib():
    """libkbmod_hip.so via ctypes; raises when it is not built (no fallback)."""
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libkbmod_hip.so")
        if not os.path.exists(path):
            raise RuntimeError("libkbmod_hip.so is not built; run __graft_entry__.build()")
        _lib = C.CDLL(path)
        _lib.kb_last_error.restype = C.c_char_p
        _lib.kb_merge_topk.argtypes = [C.c_void_p, C.c_int32, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p]
    return _lib


def merge_topk(gathered, n_pixels, K, out=None):
    """Merge ``gathered`` = [world, n_pixels*K, 7] per-rank lists into [n_pixels*K, 7].

    Device tensors go through the HIP kernel (C ABI); CPU tensors through the
    host twin in the pybind11 module (used by the gloo tests)."""
    import torch

    world = gathered.shape[0]
    if out is None:
        out = torch.empty_like(gathered[0])
    if gathered.is_cuda:
        lib = device_lib()
        stream = torch.cuda.current_stream().cuda_stream
        rc = lib.kb_merge_topk(gathered.data_ptr(), world, n_pixels, K, out.data_ptr(), stream)
        if rc != 0:
            raise RuntimeError(lib.kb_last_error().decode())
    else:
        import kbmod_amd.search as kb

        res = kb.merge_topk_host(np.ascontiguousarray(gathered.numpy()).view(np.uint8).reshape(-1), world,
                                 n_pixels, K)
        out.copy_(torch.from_numpy(res.view(np.float32).reshape(out.shape)))
    return out


def gather_and_merge(local_results, n_pixels, K, group=None, gathered=None, out=None):
    """The multi-GPU exchange step: one all_gather of the per-rank top-K lists +
    per-pixel merge.  ``local_results``: [n_pixels*K, 7] float32 tensor (the
    28-byte trajectories of this rank's search, on the rank's device)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if gathered is None:
        gathered = torch.empty((world,) + tuple(local_results.shape), dtype=local_results.dtype,
                               device=local_results.device)
    dist.all_gather_into_tensor(gathered.view(-1), local_results.contiguous().view(-1), group=group)
    return merge_topk(gathered, n_pixels, K, out)
