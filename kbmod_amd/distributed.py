"""Trajectory-space sharding across the GPUs of one node (one process per GPU).

The reference is single-GPU (no collectives anywhere); this is
new functionality.  Every (start pixel, candidate) pair is independent and only
the per-pixel top-K couples candidates, so:

* the candidate list is split into ``world`` contiguous slices (rank r owns the
  r-th slice: for the usual (theta outer, v inner) grid that is a band of
  angles), psi/phi is replicated in every GPU's HBM;
* each rank runs the single-GPU search over all start pixels on its slice and
  leaves 16-byte records (lh, flux, job-wide candidate index, obs_count) per
  slot -- ``kb_device_search_compact``; x / y follow from the slot, vx / vy from
  the candidate, so nothing else has to travel;
* ONE gather to rank 0 (RCCL over xGMI on GPUs: seven point-to-point transfers
  into the root, each on its own link; gloo in the CPU tests) collects the
  per-rank ``[S*K]`` record lists, and a per-pixel K-way merge (HIP kernel
  ``kb_merge_compact_exact`` / ``kb_merge_compact`` on device tensors, the host
  twins on CPU tensors) selects the global top-K and writes full trajectories.

Two forms of the exchange:

* **tie-exact** (``list_len = 2 K``; the default of bench.py and of StackSearch's
  fan-out): every rank searches with flag 512 -- per-pixel lists by STABLE insertion,
  i.e. the top ``2 K`` by (likelihood descending, candidate ascending) -- and the merge
  reproduces the reference's insertion exactly.  The reference's swap-down
  (kernels.cu:323-330) ROTATES a run of equal likelihoods whenever something is
  inserted in front of it and drops the run's first member at the bottom of a full
  list, so which members of a tie survive depends on the order of all candidates; but
  only candidates above the K-th likelihood and the first K equal to it can ever be
  involved, and those lie within the first ``2 K - 1`` entries of the stable order,
  which merges exactly across ranks.  The merged result then EQUALS the single-GPU
  result, ties included (tests/test_tie_exact_merge.py, tests/test_gpu_multi.py).
  K <= 16.
* **plain** (``list_len = K``): per-rank lists as the single-GPU search builds them, ties
  to the lower candidate index: the same likelihoods in the same slots as a single-GPU
  search, possibly another member of a tie.  Half the records on the wire.

torch is plumbing only here: tensors as device buffers and torch.distributed
as the RCCL front-end.
"""

import numpy as np

TRJ_FLOATS = 7      # 28-byte Trajectory viewed as 7 x 32-bit words
COMPACT_WORDS = 4   # 16-byte kb_compact_result viewed as 4 x 32-bit words


def shard_bounds(n_items, rank, world):
    """Contiguous slice [lo, hi) of ``n_items`` owned by ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def device_lib():
    """libkbmod_hip.so via ctypes (kbmod_amd.capi); raises when it is not built (no fallback)."""
    from kbmod_amd import capi

    return capi.load_lib()


def _bounds(x_bounds, y_bounds, K):
    from kbmod_amd import capi

    p = capi.Params()
    p.x_start_min, p.x_start_max = int(x_bounds[0]), int(x_bounds[1])
    p.y_start_min, p.y_start_max = int(y_bounds[0]), int(y_bounds[1])
    p.results_per_pixel = int(K)
    return p


def merge_compact(gathered, x_bounds, y_bounds, K, all_cands, out=None):
    """Merge ``gathered`` = [world, S*K, 4] compact per-rank lists (32-bit words of kb_compact_result) into
    [S*K, 7] full trajectories.  ``all_cands``: [n, 7] float32 tensor, the job-wide candidate list (vx, vy in
    columns 0, 1) on the same device.  Device tensors go through the HIP kernel (C ABI); CPU tensors through
    the host twin in the pybind11 module (used by the gloo tests)."""
    import torch

    _check_exchange_tensors(gathered, all_cands)
    world = gathered.shape[0]
    n_slots = gathered.shape[1]
    if out is None:
        out = torch.empty((n_slots, TRJ_FLOATS), dtype=torch.float32, device=gathered.device)
    if gathered.is_cuda:
        lib = device_lib()
        stream = torch.cuda.current_stream().cuda_stream
        rc = lib.kb_merge_compact(gathered.data_ptr(), world, _bounds(x_bounds, y_bounds, K), all_cands.data_ptr(),
                                  all_cands.shape[0], out.data_ptr(), stream)
        if rc != 0:
            raise RuntimeError(lib.kb_last_error().decode())
    else:
        import kbmod_amd.search as kb

        cands = [kb.Trajectory(vx=float(v[0]), vy=float(v[1])) for v in all_cands.numpy()]
        raw = np.ascontiguousarray(gathered.numpy()).view(np.uint8).reshape(-1)
        res = kb.merge_compact_host(raw, world, K, int(x_bounds[0]), int(x_bounds[1]), int(y_bounds[0]),
                                    int(y_bounds[1]), cands)
        out.copy_(torch.from_numpy(res.view(np.float32).reshape(out.shape)))
    return out


def merge_compact_exact(gathered, x_bounds, y_bounds, K, list_len, all_cands, out=None):
    """Tie-exact merge: ``gathered`` = [world, S*list_len, 4] per-rank lists built by stable insertion (flag 512,
    ``list_len`` = 2 K records per pixel) -> [S*K, 7] trajectories equal to the single-device search on the whole
    candidate list.  Device tensors: kb_merge_compact_exact; CPU tensors: its host twin."""
    import torch

    _check_exchange_tensors(gathered, all_cands)
    world = gathered.shape[0]
    n_pixels = gathered.shape[1] // int(list_len)
    if out is None:
        out = torch.empty((n_pixels * K, TRJ_FLOATS), dtype=torch.float32, device=gathered.device)
    if gathered.is_cuda:
        lib = device_lib()
        stream = torch.cuda.current_stream().cuda_stream
        rc = lib.kb_merge_compact_exact(gathered.data_ptr(), world, int(list_len), _bounds(x_bounds, y_bounds, K),
                                        all_cands.data_ptr(), all_cands.shape[0], out.data_ptr(), stream)
        if rc != 0:
            raise RuntimeError(lib.kb_last_error().decode())
    else:
        import kbmod_amd.search as kb

        cands = [kb.Trajectory(vx=float(v[0]), vy=float(v[1])) for v in all_cands.numpy()]
        raw = np.ascontiguousarray(gathered.numpy()).view(np.uint8).reshape(-1)
        res = kb.merge_compact_exact_host(raw, world, int(list_len), K, int(x_bounds[0]), int(x_bounds[1]),
                                          int(y_bounds[0]), int(y_bounds[1]), cands)
        out.copy_(torch.from_numpy(res.view(np.float32).reshape(out.shape)))
    return out


def _check_exchange_tensors(gathered, all_cands):
    """Raw pointers cross into the C ABI: insist on the layout it reads."""
    import torch

    if not (gathered.dtype == torch.int32 and gathered.dim() == 3 and gathered.shape[2] == COMPACT_WORDS
            and gathered.is_contiguous()):
        raise ValueError("gathered: expected a contiguous int32 tensor [world, slots, 4] of kb_compact_result records")
    if not (all_cands.dtype == torch.float32 and all_cands.dim() == 2 and all_cands.shape[1] == TRJ_FLOATS
            and all_cands.is_contiguous()):
        raise ValueError("all_cands: expected a contiguous float32 tensor [n, 7] (28-byte trajectories)")
    if all_cands.device != gathered.device:
        raise ValueError("gathered and all_cands must live on the same device")


def _is_root(dst, group):
    """``dst`` is a GLOBAL rank (what torch.distributed.gather takes), also when ``group`` is a sub-group."""
    import torch.distributed as dist

    return dist.get_rank() == dst


def gather_and_merge_compact(local_records, x_bounds, y_bounds, K, all_cands, group=None, gathered=None, out=None,
                             dst=0, list_len=None):
    """The multi-GPU exchange step: ONE gather of the per-rank compact record lists to global rank ``dst`` + the
    per-pixel merge there.  ``local_records``: [S*list_len, 4] int32 tensor on the rank's device; ``list_len`` = 2 K
    (lists built with flag 512) selects the tie-exact merge, None / K the plain one.  Returns the merged [S*K, 7]
    float32 trajectories on rank ``dst`` and **None on every other rank**."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    local_records = local_records.contiguous()
    # a host backend (gloo) with device tensors: the records cross through host memory (CI on fewer GPUs than ranks)
    via_host = local_records.is_cuda and dist.get_backend(group) != "nccl"
    send = local_records.cpu() if via_host else local_records
    if not _is_root(dst, group):
        dist.gather(send, None, dst=dst, group=group)
        return None
    if gathered is None:
        gathered = torch.empty((world,) + tuple(local_records.shape), dtype=local_records.dtype,
                               device=local_records.device)
    if via_host:
        staged = torch.empty(gathered.shape, dtype=gathered.dtype)
        dist.gather(send, [staged[r] for r in range(world)], dst=dst, group=group)
        gathered.copy_(staged)
    else:
        dist.gather(send, [gathered[r] for r in range(world)], dst=dst, group=group)
    if list_len is None or int(list_len) == int(K):
        return merge_compact(gathered, x_bounds, y_bounds, K, all_cands, out)
    return merge_compact_exact(gathered, x_bounds, y_bounds, K, list_len, all_cands, out)


class ExchangeInFlight:
    """One gather of per-rank record lists under way (``start_gather_compact``); ``finish()`` waits for it and merges on the
    root.  Lets a caller that searches batch after batch overlap the exchange of batch i with the search of batch i + 1:
    on GPUs the gather runs on RCCL's own stream (it starts when the search that filled ``local_records`` has finished,
    and the calling stream is not held up), the merge is enqueued behind it when ``finish()`` is called.  The caller must
    leave ``local_records`` (and ``gathered``) alone until then."""

    def __init__(self, work, gathered, staged, merge_args, is_root):
        self._work, self._gathered, self._staged, self._merge_args, self._is_root = work, gathered, staged, merge_args, is_root

    def finish(self):
        if self._work is not None:
            self._work.wait()  # device backends: the current stream waits for the collective; host backends: blocks
            self._work = None
        if not self._is_root:
            return None
        if self._staged is not None:
            self._gathered.copy_(self._staged)
        x_bounds, y_bounds, K, list_len, all_cands, out = self._merge_args
        if list_len is None or int(list_len) == int(K):
            return merge_compact(self._gathered, x_bounds, y_bounds, K, all_cands, out)
        return merge_compact_exact(self._gathered, x_bounds, y_bounds, K, list_len, all_cands, out)


def start_gather_compact(local_records, x_bounds, y_bounds, K, all_cands, group=None, gathered=None, out=None, dst=0,
                         list_len=None):
    """The exchange of ``gather_and_merge_compact`` in two halves: starts the ONE gather (asynchronously) and returns an
    ``ExchangeInFlight`` whose ``finish()`` completes it and merges on global rank ``dst`` (None elsewhere)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    local_records = local_records.contiguous()
    via_host = local_records.is_cuda and dist.get_backend(group) != "nccl"
    send = local_records.cpu() if via_host else local_records
    merge_args = (x_bounds, y_bounds, K, list_len, all_cands, out)
    if not _is_root(dst, group):
        work = dist.gather(send, None, dst=dst, group=group, async_op=True)
        return ExchangeInFlight(work, None, None, merge_args, False)
    if gathered is None:
        gathered = torch.empty((world,) + tuple(local_records.shape), dtype=local_records.dtype,
                               device=local_records.device)
    staged = torch.empty(gathered.shape, dtype=gathered.dtype) if via_host else None
    target = staged if via_host else gathered
    work = dist.gather(send, [target[r] for r in range(world)], dst=dst, group=group, async_op=True)
    return ExchangeInFlight(work, gathered, staged, merge_args, True)


def merge_topk(gathered, n_pixels, K, out=None):
    """Merge ``gathered`` = [world, n_pixels*K, 7] per-rank lists of full 28-byte trajectories into
    [n_pixels*K, 7] (the exchange format for results_per_pixel > 32, which the compact search does not
    cover)."""
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


def gather_and_merge(local_results, n_pixels, K, group=None, gathered=None, out=None, dst=0):
    """Full-record variant of the exchange (results_per_pixel > 32): one gather of [n_pixels*K, 7] float32 lists to
    global rank ``dst`` + merge there.  Returns the merged lists on rank ``dst`` and **None on every other rank**
    (an all_gather would move world times the bytes for a result only the root consumes)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    local_results = local_results.contiguous()
    if not _is_root(dst, group):
        dist.gather(local_results, None, dst=dst, group=group)
        return None
    if gathered is None:
        gathered = torch.empty((world,) + tuple(local_results.shape), dtype=local_results.dtype,
                               device=local_results.device)
    dist.gather(local_results, [gathered[r] for r in range(world)], dst=dst, group=group)
    return merge_topk(gathered, n_pixels, K, out)
