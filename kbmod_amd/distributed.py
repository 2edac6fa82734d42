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
* **K records + repair** (``list_len = K`` with ``repair_stack``; round 6): the same wire and the same per-rank search as
  the plain form -- every rank runs the fastest single-GPU instance on its slice, no stable lists of 2 K --, and the result
  of the tie-exact form.  The root's merge (``kb_merge_compact_repairable``) decides every pixel whose lists determine its
  final list and names the others: a FULL list that ends at the pixel's K-th likelihood may have dropped a candidate that
  ties with it.  Those pixels (a fraction of a percent on the BASELINE grids) are searched again over the whole candidate
  list on the root, one wavefront per pixel (``kb_repair_pixels``: psi/phi is replicated, the root has it).

And one way of putting either on the wire:

* **sparse** (``gather_and_merge_sparse``; SURVEY 8(e): "compacting lh >= min_lh first"): a search with a
  likelihood threshold fills almost no slot -- the reference drops everything below ``min_lh`` right after its kernel
  (stack_search.cpp:266-270), and its insertion never lets a smaller likelihood touch the part of a list at or above a
  larger one, so dropping those records before the exchange changes nothing that survives.  Every rank turns its dense
  lists into one count byte per pixel + the surviving records (``kb_sparsify_compact``), ONE gather collects the
  headers, the records follow as one point-to-point message per rank of exactly the size its header announced, and the
  root runs the same tie-exact merge through the counts (``kb_merge_sparse_exact``).  BASELINE configs[3] (128 x 4096 x
  4096, K = 8, min_lh = 10): 16.8 MB + a few MB per rank instead of 4.3 GB.

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


_last_merge_events = None


class _MergeTimer:
    """HIP events around the last device merge on the current stream (read later by last_merge_ms: no sync here)."""

    def __init__(self, on_device):
        self.on = on_device

    def __enter__(self):
        if self.on:
            import torch

            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        global _last_merge_events
        if self.on:
            self.e1.record()
            _last_merge_events = (self.e0, self.e1)
        return False


def last_merge_ms():
    """Device time of this process's last merge launch sequence (None before the first, or for host-twin merges)."""
    if _last_merge_events is None:
        return None
    _last_merge_events[1].synchronize()
    return float(_last_merge_events[0].elapsed_time(_last_merge_events[1]))


def device_lib():
    """libkbmod_hip.so via ctypes (kbmod_amd.capi); raises when it is not built (no fallback)."""
    from kbmod_amd import capi

    return capi.load_lib()


def _n_pixels(x_bounds, y_bounds):
    return (int(x_bounds[1]) - int(x_bounds[0])) * (int(y_bounds[1]) - int(y_bounds[0]))


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
    n_pixels = _n_pixels(x_bounds, y_bounds)
    if n_slots != n_pixels * int(K):
        raise ValueError(f"gathered holds {n_slots} records per list, the bounds say {n_pixels} pixels x {int(K)}")
    if out is None:
        out = torch.empty((n_slots, TRJ_FLOATS), dtype=torch.float32, device=gathered.device)
    _check_out(out, n_pixels, K, gathered.device)
    if gathered.is_cuda:
        lib = device_lib()
        stream = torch.cuda.current_stream().cuda_stream
        with _MergeTimer(True):
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
    n_pixels = _n_pixels(x_bounds, y_bounds)  # (what kb_merge_compact_exact derives its grid from)
    if gathered.shape[1] != n_pixels * int(list_len):
        raise ValueError(f"gathered holds {gathered.shape[1]} records per list, the bounds say {n_pixels} pixels x {int(list_len)}")
    if out is None:
        out = torch.empty((n_pixels * K, TRJ_FLOATS), dtype=torch.float32, device=gathered.device)
    _check_out(out, n_pixels, K, gathered.device)
    if gathered.is_cuda:
        lib = device_lib()
        stream = torch.cuda.current_stream().cuda_stream
        with _MergeTimer(True):
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


_last_repair = None


def last_repair():
    """{"hazards": pixels the last repairable merge of this process could not decide, "pixels": all of them} or None."""
    return _last_repair


def merge_compact_repair(gathered, x_bounds, y_bounds, K, all_cands, stack, out=None, min_obs=0, hazard_buf=None,
                         list_begin=None):
    """The merge of the exchange with K records per rank: ``gathered`` = [world, S*K, 4] per-rank lists, each the
    reference's insertion over that rank's slice (``kb_device_search_compact`` with ``list_len = K``, no flag 512, no list
    floor) -> [S*K, 7] trajectories equal to the single-device search on the whole candidate list.  Pixels the lists do
    not decide are re-made from the stack: ``stack`` = ``(meta, psi_phi_ptr, times_ptr)`` of the root's replica for
    device tensors (capi.Meta, two device addresses: ``kb_repair_pixels``); for CPU tensors (the gloo tests) a callable
    ``stack(x, y, vx, vy) -> (lh, flux, obs_count)``, the evaluation of one trajectory, driven through the reference's
    insertion here.  ``hazard_buf``: an int32/uint32 device tensor of S entries to reuse between calls.  ``list_begin``:
    world + 1 candidate indices, list r = the candidates [list_begin[r], list_begin[r + 1]) of ``all_cands`` (ascending,
    tiling the whole list: ``shard_bounds``) -- the repair then evaluates only the slices of the lists that may have dropped
    something (same result, ``world`` times fewer evaluations); None: every candidate."""
    global _last_repair
    import torch

    _check_exchange_tensors(gathered, all_cands)
    world = gathered.shape[0]
    n_pixels = _n_pixels(x_bounds, y_bounds)
    K = int(K)
    if gathered.shape[1] != n_pixels * K:
        raise ValueError(f"gathered holds {gathered.shape[1]} records per list, the bounds say {n_pixels} pixels x {K}")
    if out is None:
        out = torch.empty((n_pixels * K, TRJ_FLOATS), dtype=torch.float32, device=gathered.device)
    _check_out(out, n_pixels, K, gathered.device)
    params = _bounds(x_bounds, y_bounds, K)
    params.min_observations = int(min_obs)
    if gathered.is_cuda:
        import ctypes as C

        lib = device_lib()
        stream = torch.cuda.current_stream().cuda_stream
        if hazard_buf is None:
            hazard_buf = torch.empty(n_pixels, dtype=torch.int32, device=gathered.device)
        if hazard_buf.numel() < n_pixels or hazard_buf.element_size() != 4 or hazard_buf.device != gathered.device:
            raise ValueError("hazard_buf: expected a 4-byte integer tensor of at least S entries on the lists' device")
        n_hazard = C.c_uint64(0)
        with _MergeTimer(True):
            rc = lib.kb_merge_compact_repairable(gathered.data_ptr(), world, params, all_cands.data_ptr(), all_cands.shape[0],
                                                 out.data_ptr(), hazard_buf.data_ptr(), C.byref(n_hazard), stream)
            if rc == 0 and n_hazard.value:
                if stack is None or callable(stack):
                    raise ValueError("merge_compact_repair: device lists need stack = (meta, psi_phi_ptr, times_ptr) for the repair")
                meta, psi_phi_ptr, times_ptr = stack
                begin = None
                if list_begin is not None:
                    if len(list_begin) != world + 1:
                        raise ValueError("list_begin: expected world + 1 candidate indices")
                    begin = (C.c_int32 * (world + 1))(*[int(b) for b in list_begin])
                rc = lib.kb_repair_pixels(C.byref(meta), C.c_void_p(int(psi_phi_ptr)), C.c_void_p(int(times_ptr)), params,
                                          all_cands.data_ptr(), all_cands.shape[0], hazard_buf.data_ptr(), n_hazard.value,
                                          gathered.data_ptr() if begin is not None else None, world, begin, out.data_ptr(), stream)
        if rc != 0:
            raise RuntimeError(lib.kb_last_error().decode())
        _last_repair = {"hazards": int(n_hazard.value), "pixels": n_pixels}
    else:
        import kbmod_amd.search as kb

        vel = all_cands.numpy()
        cands = [kb.Trajectory(vx=float(v[0]), vy=float(v[1])) for v in vel]
        raw = np.ascontiguousarray(gathered.numpy()).view(np.uint8).reshape(-1)
        res, hazards = kb.merge_compact_repairable_host(raw, world, K, int(x_bounds[0]), int(x_bounds[1]), int(y_bounds[0]),
                                                        int(y_bounds[1]), cands)
        rows = np.array(res.view(np.float32).reshape(n_pixels * K, TRJ_FLOATS))
        if len(hazards):
            if not callable(stack):
                raise ValueError("merge_compact_repair: CPU lists need stack = callable(x, y, vx, vy) -> (lh, flux, obs_count)")
            sw = int(x_bounds[1]) - int(x_bounds[0])
            words = rows.view(np.int32)
            for pix in hazards:
                y_i, x_i = divmod(int(pix), sw)
                x, y = x_i + int(x_bounds[0]), y_i + int(y_bounds[0])
                slots = [(np.float32(-3.4028234663852886e38), 0.0, -1, 0)] * K  # kernels.cu:293-301
                for c in range(len(vel)):
                    lh, flux, obs = stack(x, y, float(vel[c, 0]), float(vel[c, 1]))
                    if obs < int(min_obs):
                        continue
                    cur = (np.float32(lh), np.float32(flux), c, int(obs))
                    for s in range(K):  # kernels.cu:323-330
                        if cur[0] > slots[s][0]:
                            cur, slots[s] = slots[s], cur
                for s, (lh, flux, c, obs) in enumerate(slots):
                    row = int(pix) * K + s
                    rows[row] = 0.0
                    words[row, 4], words[row, 5] = x, y
                    rows[row, 2] = lh
                    if c >= 0:
                        rows[row, 0], rows[row, 1], rows[row, 3] = vel[c, 0], vel[c, 1], flux
                        words[row, 6] = obs
        out.copy_(torch.from_numpy(rows))
        _last_repair = {"hazards": int(len(hazards)), "pixels": n_pixels}
    return out


def sparsify_compact(records, n_pixels, list_len, min_lh, header=None, packed=None):
    """Dense per-pixel lists ``records`` = [n_pixels*list_len, 4] int32 (kb_compact_result) -> ``(header, packed, total)``:
    ``header`` uint8 [kb_sparse_header_bytes(n_pixels)] (one count per pixel, then the total), ``packed`` int32 [>= total, 4]
    (the records with cand >= 0 and not lh < min_lh, pixel after pixel), ``total`` their number.  Device tensors:
    kb_sparsify_compact; CPU tensors: its host twin.  ``header`` / ``packed``: preallocated buffers to reuse (``packed`` must
    have room for every record kept; without one a buffer of exactly ``total`` records is made by a second pass)."""
    import torch

    if not (records.dtype == torch.int32 and records.dim() == 2 and records.shape[1] == COMPACT_WORDS
            and records.is_contiguous() and records.shape[0] == int(n_pixels) * int(list_len)):
        raise ValueError("records: expected a contiguous int32 tensor [n_pixels * list_len, 4] of kb_compact_result records")
    min_lh = float("-inf") if min_lh is None else float(min_lh)
    if records.is_cuda:
        import ctypes as C

        lib = device_lib()
        hb = int(lib.kb_sparse_header_bytes(int(n_pixels)))
        if header is None:
            header = torch.empty(hb, dtype=torch.uint8, device=records.device)
        if not (header.dtype == torch.uint8 and header.numel() == hb and header.is_contiguous() and header.device == records.device):
            raise ValueError(f"header: expected a contiguous uint8 tensor of {hb} bytes on the records' device")
        stream = torch.cuda.current_stream().cuda_stream
        total = C.c_uint64(0)
        cap = 0 if packed is None else int(packed.shape[0])
        if packed is not None and not (packed.dtype == torch.int32 and packed.dim() == 2 and packed.shape[1] == COMPACT_WORDS
                                       and packed.is_contiguous() and packed.device == records.device):
            raise ValueError("packed: expected a contiguous int32 tensor [capacity, 4] on the records' device")
        rc = lib.kb_sparsify_compact(records.data_ptr(), int(n_pixels), int(list_len), min_lh, header.data_ptr(),
                                     0 if packed is None else packed.data_ptr(), cap, C.byref(total), stream)
        if rc != 0 and packed is None and total.value > 0:
            # no buffer was offered: now that the count is known, make one of exactly that size
            packed = torch.empty((int(total.value), COMPACT_WORDS), dtype=torch.int32, device=records.device)
            rc = lib.kb_sparsify_compact(records.data_ptr(), int(n_pixels), int(list_len), min_lh, header.data_ptr(),
                                         packed.data_ptr(), int(total.value), C.byref(total), stream)
        if rc != 0:
            raise RuntimeError(lib.kb_last_error().decode())
        if packed is None:
            packed = torch.empty((0, COMPACT_WORDS), dtype=torch.int32, device=records.device)
        return header, packed, int(total.value)
    import kbmod_amd.search as kb

    raw = np.ascontiguousarray(records.numpy()).view(np.uint8).reshape(-1)
    h, p = kb.sparsify_compact_host(raw, int(n_pixels), int(list_len), min_lh)
    return torch.from_numpy(h), torch.from_numpy(p.view(np.int32).reshape(-1, COMPACT_WORDS)), int(p.size // 16)


def sparsify_counted(records, n_pixels, list_len, header, packed=None):
    """The rest of :func:`sparsify_compact` when the SEARCH wrote the count bytes (kb_device_search_counted with
    ``counts_dev = header``): block sums, scan, total, and a scatter that reads only the counted records of ``records`` -- the
    runs the search skipped are never touched.  Device tensors only; ``packed`` as in :func:`sparsify_compact`.
    -> ``(header, packed, total)``."""
    import ctypes as C

    import torch

    if not (records.is_cuda and records.dtype == torch.int32 and records.dim() == 2 and records.shape[1] == COMPACT_WORDS
            and records.is_contiguous() and records.shape[0] == int(n_pixels) * int(list_len)):
        raise ValueError("records: expected a contiguous int32 device tensor [n_pixels * list_len, 4] of kb_compact_result records")
    lib = device_lib()
    hb = int(lib.kb_sparse_header_bytes(int(n_pixels)))
    if not (header.dtype == torch.uint8 and header.numel() == hb and header.is_contiguous() and header.device == records.device):
        raise ValueError(f"header: expected a contiguous uint8 tensor of {hb} bytes on the records' device")
    if packed is not None and not (packed.dtype == torch.int32 and packed.dim() == 2 and packed.shape[1] == COMPACT_WORDS
                                   and packed.is_contiguous() and packed.device == records.device):
        raise ValueError("packed: expected a contiguous int32 tensor [capacity, 4] on the records' device")
    total = C.c_uint64(0)
    stream = torch.cuda.current_stream().cuda_stream
    rc = lib.kb_sparsify_counted(records.data_ptr(), int(n_pixels), int(list_len), header.data_ptr(),
                                 0 if packed is None else packed.data_ptr(), 0 if packed is None else int(packed.shape[0]),
                                 C.byref(total), stream)
    if rc != 0 and packed is None and total.value > 0:
        packed = torch.empty((int(total.value), COMPACT_WORDS), dtype=torch.int32, device=records.device)
        rc = lib.kb_sparsify_counted(records.data_ptr(), int(n_pixels), int(list_len), header.data_ptr(), packed.data_ptr(),
                                     int(total.value), C.byref(total), stream)
    if rc != 0:
        raise RuntimeError(lib.kb_last_error().decode())
    if packed is None:
        packed = torch.empty((0, COMPACT_WORDS), dtype=torch.int32, device=records.device)
    return header, packed, int(total.value)


def sparse_totals(headers, n_pixels):
    """The record totals the headers [n_lists, header_bytes] announce (int64 on the host)."""
    at = (int(n_pixels) + 15) // 16 * 16
    return headers[:, at:at + 8].contiguous().cpu().view(headers.shape[0], 8).numpy().view(np.int64).reshape(-1)


def merge_sparse_exact(headers, packed_list, x_bounds, y_bounds, K, list_len, all_cands, out=None, counts_out=None):
    """Tie-exact merge over sparse lists: ``headers`` uint8 [n_lists, header_bytes] (one gather of the ranks' headers),
    ``packed_list[r]`` int32 [total_r, 4] -> [S*K, 7] trajectories: wherever a record survives the likelihood filter they
    equal ``merge_compact_exact`` on the dense lists (hence the single-device search), every other slot is the empty-slot
    placeholder.  Device tensors: kb_merge_sparse_exact; CPU tensors: its host twin.  ``counts_out`` (device only): a uint8
    tensor [n_pixels] that receives the number of merged records per start pixel -- a prefix of its K slots --, and then NO
    slot is written for a wave of 64 start pixels nothing reaches (kb_merge_sparse_exact_counted: what
    kb_filter_sort_results_counted reads through)."""
    import torch

    n_lists = int(headers.shape[0])
    n_pixels = (int(x_bounds[1]) - int(x_bounds[0])) * (int(y_bounds[1]) - int(y_bounds[0]))
    if not (headers.dtype == torch.uint8 and headers.dim() == 2 and headers.is_contiguous() and len(packed_list) == n_lists):
        raise ValueError("headers: expected a contiguous uint8 tensor [n_lists, header_bytes] and one record tensor per list")
    totals = sparse_totals(headers, n_pixels)
    for r, p in enumerate(packed_list):
        if not (p.dtype == torch.int32 and p.dim() == 2 and p.shape[1] == COMPACT_WORDS and p.is_contiguous()
                and p.device == headers.device and p.shape[0] >= totals[r]):
            raise ValueError(f"packed_list[{r}]: expected a contiguous int32 tensor [>= {totals[r]}, 4] on the headers' device")
    if not (all_cands.dtype == torch.float32 and all_cands.dim() == 2 and all_cands.shape[1] == TRJ_FLOATS
            and all_cands.is_contiguous() and all_cands.device == headers.device):
        raise ValueError("all_cands: expected a contiguous float32 tensor [n, 7] on the headers' device")
    if out is None:
        out = torch.empty((n_pixels * int(K), TRJ_FLOATS), dtype=torch.float32, device=headers.device)
    _check_out(out, n_pixels, K, headers.device)
    if headers.is_cuda:
        import ctypes as C

        lib = device_lib()
        ptrs = (C.c_void_p * n_lists)(*[(p.data_ptr() if p.shape[0] else None) for p in packed_list])
        if counts_out is not None and not (counts_out.dtype == torch.uint8 and counts_out.numel() == n_pixels
                                           and counts_out.is_contiguous() and counts_out.device == headers.device):
            raise ValueError(f"counts_out: expected a contiguous uint8 tensor of {n_pixels} bytes on the headers' device")
        with _MergeTimer(True):
            rc = lib.kb_merge_sparse_exact_counted(headers.data_ptr(), int(headers.shape[1]), ptrs, n_lists, int(list_len),
                                                   _bounds(x_bounds, y_bounds, K), all_cands.data_ptr(), all_cands.shape[0],
                                                   out.data_ptr(), None if counts_out is None else counts_out.data_ptr(),
                                                   torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError(lib.kb_last_error().decode())
    else:
        import kbmod_amd.search as kb

        if counts_out is not None:
            raise ValueError("counts_out: the host twin of the merge writes every slot")
        cands = [kb.Trajectory(vx=float(v[0]), vy=float(v[1])) for v in all_cands.numpy()]
        res = kb.merge_sparse_exact_host(np.ascontiguousarray(headers.numpy()).reshape(-1), int(headers.shape[1]),
                                         [np.ascontiguousarray(p.numpy()).view(np.uint8).reshape(-1) for p in packed_list],
                                         int(list_len), int(K), int(x_bounds[0]), int(x_bounds[1]), int(y_bounds[0]),
                                         int(y_bounds[1]), cands)
        out.copy_(torch.from_numpy(res.view(np.float32).reshape(out.shape)))
    return out


def gather_and_merge_sparse(local_records, x_bounds, y_bounds, K, list_len, min_lh, all_cands, group=None, out=None, dst=0,
                            header=None, packed=None, stats=None, counted=False, counts_out=None):
    """The exchange step in its sparse form: sparsify on every rank, ONE gather of the headers to global rank ``dst``, one
    point-to-point message per rank with exactly the records its header announced (all of them in flight together: seven
    transfers into the root, each on its own xGMI link), and the tie-exact merge there.  Returns the merged [S*K, 7]
    trajectories on rank ``dst`` and **None on every other rank**.  ``stats``: a dict that receives ``wire_bytes`` (what this
    rank sent: header + records) and, on the root, ``totals``.  ``counted``: this rank's search wrote the count bytes into
    ``header`` itself (kb_device_search_counted said so) -- :func:`sparsify_counted` finishes the header and packs.
    ``counts_out``: see :func:`merge_sparse_exact` (the root's merge then skips the slots of waves nothing reaches)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    # `dst` is a GLOBAL rank (what gather / send take); the lists are ordered by GROUP rank, and with a sub-group, or a
    # group whose ranks do not start at 0, the two differ: peers of the point-to-point messages are named globally
    me_global = dist.get_rank()
    me = dist.get_rank(group)
    peer = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
    n_pixels = (int(x_bounds[1]) - int(x_bounds[0])) * (int(y_bounds[1]) - int(y_bounds[0]))
    if counted:
        header, packed, total = sparsify_counted(local_records, n_pixels, list_len, header, packed)
    else:
        header, packed, total = sparsify_compact(local_records.contiguous(), n_pixels, list_len, min_lh, header, packed)
    via_host = header.is_cuda and dist.get_backend(group) != "nccl"
    h_send = header.cpu() if via_host else header
    p_send = packed[:total].cpu() if via_host else packed[:total]
    if stats is not None:
        stats["wire_bytes"] = int(header.numel()) + 16 * total
    if me_global != dst:
        dist.gather(h_send, None, dst=dst, group=group)
        if total:
            dist.send(p_send, dst=dst, group=group)
        return None
    headers = torch.empty((world, h_send.numel()), dtype=torch.uint8, device=h_send.device)
    dist.gather(h_send, [headers[r] for r in range(world)], dst=dst, group=group)
    totals = sparse_totals(headers, n_pixels)
    if stats is not None:
        stats["totals"] = [int(t) for t in totals]
    bufs, ops = [], []
    for r in range(world):
        if r == me:
            bufs.append(p_send)
            continue
        bufs.append(torch.empty((int(totals[r]), COMPACT_WORDS), dtype=torch.int32, device=h_send.device))
        if totals[r]:
            ops.append(dist.P2POp(dist.irecv, bufs[r], peer(r), group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if via_host:
        headers = headers.to(local_records.device)
        bufs = [b.to(local_records.device) for b in bufs]
    return merge_sparse_exact(headers, bufs, x_bounds, y_bounds, K, list_len, all_cands, out, counts_out=counts_out)


def _check_out(out, n_pixels, K, device):
    """Raw pointers cross into the C ABI: the preallocated result buffer must be what the merge kernels write."""
    import torch

    if not (out.dtype == torch.float32 and out.dim() == 2 and tuple(out.shape) == (int(n_pixels) * int(K), TRJ_FLOATS)
            and out.is_contiguous() and out.device == device):
        raise ValueError(f"out: expected a contiguous float32 tensor [{int(n_pixels) * int(K)}, 7] on {device}")


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


def _merge_gathered(gathered, x_bounds, y_bounds, K, list_len, all_cands, out, repair_stack, min_obs, list_begin=None):
    if list_len is None or int(list_len) == int(K):
        if repair_stack is not None:
            return merge_compact_repair(gathered, x_bounds, y_bounds, K, all_cands, repair_stack, out, min_obs,
                                        list_begin=list_begin)
        return merge_compact(gathered, x_bounds, y_bounds, K, all_cands, out)
    return merge_compact_exact(gathered, x_bounds, y_bounds, K, list_len, all_cands, out)


def gather_and_merge_compact(local_records, x_bounds, y_bounds, K, all_cands, group=None, gathered=None, out=None,
                             dst=0, list_len=None, repair_stack=None, min_obs=0, list_begin=None):
    """The multi-GPU exchange step: ONE gather of the per-rank compact record lists to global rank ``dst`` + the
    per-pixel merge there.  ``local_records``: [S*list_len, 4] int32 tensor on the rank's device; ``list_len`` = 2 K
    (lists built with flag 512) selects the tie-exact merge, None / K the plain one -- or, with ``repair_stack`` (see
    ``merge_compact_repair``), the merge that re-makes the pixels K records do not decide.  Returns the merged [S*K, 7]
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
    return _merge_gathered(gathered, x_bounds, y_bounds, K, list_len, all_cands, out, repair_stack, min_obs, list_begin)


class ExchangeInFlight:
    """One gather of per-rank record lists under way (``start_gather_compact``); ``finish()`` waits for it and merges on the
    root.  Lets a caller that searches batch after batch overlap the exchange of batch i with the search of batch i + 1:
    on GPUs the gather runs on RCCL's own stream (it starts when the search that filled ``local_records`` has finished,
    and the calling stream is not held up), the merge is enqueued behind it when ``finish()`` is called.  The caller must
    leave ``local_records`` (and ``gathered``) alone until then."""

    def __init__(self, work, gathered, staged, merge_args, is_root, send=None):
        self._work, self._gathered, self._staged, self._merge_args, self._is_root = work, gathered, staged, merge_args, is_root
        self._send = send  # the tensor on the wire (possibly a host copy made here): alive until finish()

    def finish(self):
        if self._work is not None:
            self._work.wait()  # device backends: the current stream waits for the collective; host backends: blocks
            self._work = None
        self._send = None
        if not self._is_root:
            return None
        if self._staged is not None:
            self._gathered.copy_(self._staged)
        x_bounds, y_bounds, K, list_len, all_cands, out, repair_stack, min_obs, list_begin = self._merge_args
        return _merge_gathered(self._gathered, x_bounds, y_bounds, K, list_len, all_cands, out, repair_stack, min_obs, list_begin)


def start_gather_compact(local_records, x_bounds, y_bounds, K, all_cands, group=None, gathered=None, out=None, dst=0,
                         list_len=None, repair_stack=None, min_obs=0, list_begin=None):
    """The exchange of ``gather_and_merge_compact`` in two halves: starts the ONE gather (asynchronously) and returns an
    ``ExchangeInFlight`` whose ``finish()`` completes it and merges on global rank ``dst`` (None elsewhere)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    local_records = local_records.contiguous()
    via_host = local_records.is_cuda and dist.get_backend(group) != "nccl"
    send = local_records.cpu() if via_host else local_records
    merge_args = (x_bounds, y_bounds, K, list_len, all_cands, out, repair_stack, min_obs, list_begin)
    if not _is_root(dst, group):
        work = dist.gather(send, None, dst=dst, group=group, async_op=True)
        return ExchangeInFlight(work, None, None, merge_args, False, send)
    if gathered is None:
        gathered = torch.empty((world,) + tuple(local_records.shape), dtype=local_records.dtype,
                               device=local_records.device)
    staged = torch.empty(gathered.shape, dtype=gathered.dtype) if via_host else None
    target = staged if via_host else gathered
    work = dist.gather(send, [target[r] for r in range(world)], dst=dst, group=group, async_op=True)
    return ExchangeInFlight(work, gathered, staged, merge_args, True, send)


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
