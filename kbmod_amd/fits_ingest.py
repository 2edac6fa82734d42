"""WorkUnit FITS files -> device-resident science / variance stacks.

The reference reads its WorkUnit files through astropy (``WorkUnit.from_fits`` work_unit.py:489-608,
``from_sharded_fits`` :782-897, ``read_image_data_from_hdul`` :1149-1200): every layer is decompressed on the host into a
float32 array, masked there, appended to an ``ImageStackPy``, converted again for ``StackSearch`` and only then uploaded.
Here the file's bytes go to HBM as they lie on disk -- one DMA out of a page-locked buffer -- and are decoded THERE
(``csrc/fits_kernels.hip`` through the C ABI: RICE_1 tiles, big-endian image HDUs, the mask pass) straight into the
``[T][H][W]`` stacks ``kb_build_psi_phi_from_device_ex`` reads.  The host parses header cards and table rows only
(kilobytes); what crosses PCIe is the compressed file.

Same layer semantics as the reference: ``SCI_i`` / ``VAR_i`` as float32, ``sci[mask > 0] = var[mask > 0] = nan`` when
``MSK_i`` exists, ``PSF_i`` (identity when absent), ``SCI_i.header["MJD"]`` as the epoch; ``NUMIMG`` of the primary header
counts the images.  Tiled-compressed HDUs: ``ZCMPTYPE = RICE_1`` with one tile per image row and ``ZQUANTIZ = NO_DITHER``
(what ``CompImageHDU(compression_type="RICE_1", quantize_level=-0.01)`` of work_unit.py:1108-1122 writes) or integer
pixels; tiles the writer fell back to ``GZIP_COMPRESSED_DATA`` for are inflated on the host and patched in.  Anything else
(other compression types, dithered quantisation, tiles that are not rows) raises ``ValueError`` naming it.

There is no CPU decode path: without the HIP library and a GPU ``load_workunit`` raises.
"""

import ctypes as C
import os
import struct
import zlib

import numpy as np

BLOCK = 2880

# kb_fits_tile (include/kbmod_hip.h)
TILE_DTYPE = np.dtype([("offset", "<u8"), ("out_index", "<u8"), ("zscale", "<f8"), ("zzero", "<f8"), ("nbytes", "<u4"),
                       ("mode", "<i4")])
assert TILE_DTYPE.itemsize == 40
TILE_SKIP, TILE_RICE = 0, 1


class Hdu:
    """One header + the place of its data unit in the file."""

    __slots__ = ("header", "data_offset", "data_size")

    def __init__(self, header, data_offset, data_size):
        self.header, self.data_offset, self.data_size = header, data_offset, data_size

    @property
    def name(self):
        return str(self.header.get("EXTNAME", "")).strip().upper()

    @property
    def is_compressed_image(self):
        return self.header.get("ZIMAGE") is True

    @property
    def image_shape(self):
        """(height, width) of a 2-D image HDU, compressed or not."""
        pre = "ZNAXIS" if self.is_compressed_image else "NAXIS"
        if int(self.header.get(pre, 0)) != 2:
            raise ValueError(f"HDU {self.name or '?'}: expected a 2-D image, found {pre} = {self.header.get(pre)}")
        return int(self.header[pre + "2"]), int(self.header[pre + "1"])


def _value(field):
    """The value of a header card's value field (columns 11-80)."""
    s = field.lstrip()
    if s.startswith("'"):
        chars, i = [], 1
        while i < len(s):
            if s[i] == "'":
                if s[i + 1:i + 2] == "'":
                    chars.append("'")
                    i += 2
                    continue
                break
            chars.append(s[i])
            i += 1
        return "".join(chars).rstrip()
    s = s.split("/", 1)[0].strip()
    if s in ("T", "F"):
        return s == "T"
    if not s:
        return None
    try:
        return int(s)
    except ValueError:
        try:
            return float(s.replace("D", "E").replace("d", "e"))
        except ValueError:
            return s


def parse_fits(buf):
    """Walk the HDUs of a FITS file held in a bytes-like ``buf``: header cards into a dict (first occurrence wins), the data
    unit's offset and size (NAXISn product x |BITPIX| / 8 + PCOUNT, padded to 2880-byte blocks on disk)."""
    view = memoryview(buf)
    total = len(view)
    hdus, off = [], 0
    if total < BLOCK or bytes(view[0:9]) != b"SIMPLE  =":
        raise ValueError("not a FITS file: the first card is not SIMPLE")
    while off + BLOCK <= total:
        header, ended = {}, False
        while not ended:
            if off + BLOCK > total:
                raise ValueError("FITS header runs past the end of the file (no END card)")
            block = bytes(view[off:off + BLOCK]).decode("ascii", "replace")
            off += BLOCK
            for c in range(0, BLOCK, 80):
                card = block[c:c + 80]
                key = card[:8].rstrip()
                if key == "END":
                    ended = True
                    break
                if card[8:10] == "= " and key and key not in header:
                    header[key] = _value(card[10:])
        size = 0
        naxis = int(header.get("NAXIS", 0) or 0)
        if naxis > 0:
            size = abs(int(header["BITPIX"])) // 8
            for a in range(1, naxis + 1):
                size *= int(header[f"NAXIS{a}"])
            size = size * int(header.get("GCOUNT", 1) or 1) + int(header.get("PCOUNT", 0) or 0)
        if off + size > total:
            raise ValueError(f"HDU {header.get('EXTNAME', len(hdus))}: data unit runs past the end of the file")
        hdus.append(Hdu(header, off, size))
        off += -(-size // BLOCK) * BLOCK
    return hdus


def find_hdu(hdus, name):
    name = name.upper()
    for h in hdus:
        if h.name == name:
            return h
    return None


_TFORM_BYTES = {"L": 1, "X": 1, "B": 1, "I": 2, "J": 4, "K": 8, "A": 1, "E": 4, "D": 8, "C": 8, "M": 16, "P": 8, "Q": 16}


def table_columns(header):
    """{column name: (byte offset within a row, type code, repeat, heap element type of a P / Q column)} of a BINTABLE header."""
    cols, off = {}, 0
    for c in range(1, int(header["TFIELDS"]) + 1):
        form = str(header[f"TFORM{c}"]).strip()
        k = 0
        while k < len(form) and form[k].isdigit():
            k += 1
        repeat = int(form[:k]) if k else 1
        code = form[k]
        if code not in _TFORM_BYTES:
            raise ValueError(f"unknown TFORM{c} = {form!r}")
        # (variable-length columns, rPt / rQt: the letter behind the descriptor code names the heap elements)
        elem = form[k + 1] if code in "PQ" and k + 1 < len(form) else ""
        cols[str(header.get(f"TTYPE{c}", f"COL{c}")).strip().upper()] = (off, code, repeat, elem)
        off += (-(-repeat // 8) if code == "X" else repeat * _TFORM_BYTES[code])
    if off != int(header["NAXIS1"]):
        raise ValueError(f"table columns add up to {off} bytes per row, NAXIS1 says {header['NAXIS1']}")
    return cols


class CompressedLayout:
    """What the tile decoder needs to know about one tiled-compressed image HDU, from its header."""

    def __init__(self, hdu):
        h = hdu.header
        name = hdu.name or "?"
        ctype = str(h.get("ZCMPTYPE", "")).strip()
        if ctype != "RICE_1":
            raise ValueError(f"HDU {name}: ZCMPTYPE {ctype!r} is not read here (RICE_1, what the reference writes, is)")
        self.height, self.width = hdu.image_shape
        if int(h.get("ZTILE1", self.width)) != self.width or int(h.get("ZTILE2", 1)) != 1:
            raise ValueError(f"HDU {name}: tiles of {h.get('ZTILE1')} x {h.get('ZTILE2')} pixels; only whole-row tiles are read")
        self.zbitpix = int(h["ZBITPIX"])
        self.blocksize, self.bytepix = 32, 4
        for k in range(1, 100):
            key = h.get(f"ZNAME{k}")
            if key is None:
                break
            if str(key).strip() == "BLOCKSIZE":
                self.blocksize = int(h[f"ZVAL{k}"])
            elif str(key).strip() == "BYTEPIX":
                self.bytepix = int(h[f"ZVAL{k}"])
        if self.bytepix not in (1, 2, 4) or self.blocksize <= 0:
            raise ValueError(f"HDU {name}: RICE parameters BLOCKSIZE {self.blocksize} / BYTEPIX {self.bytepix}")
        self.quantized = self.zbitpix < 0
        if self.quantized:
            method = str(h.get("ZQUANTIZ", "NO_DITHER")).strip()
            if method != "NO_DITHER":
                raise ValueError(f"HDU {name}: ZQUANTIZ {method!r}; only NO_DITHER (astropy's default, what the reference "
                                 "writes) is read")
            if self.bytepix != 4:
                raise ValueError(f"HDU {name}: quantised floats with BYTEPIX {self.bytepix}")
        elif self.zbitpix not in (8, 16, 32):
            raise ValueError(f"HDU {name}: ZBITPIX {self.zbitpix}")
        self.row_bytes, self.n_rows = int(h["NAXIS1"]), int(h["NAXIS2"])
        if self.n_rows != self.height:
            raise ValueError(f"HDU {name}: {self.n_rows} table rows for {self.height} image rows")
        self.theap = int(h.get("THEAP", self.row_bytes * self.n_rows))
        self.columns = table_columns(h)
        if "COMPRESSED_DATA" not in self.columns or self.columns["COMPRESSED_DATA"][1] != "P":
            raise ValueError(f"HDU {name}: no COMPRESSED_DATA column of 32-bit array descriptors")
        # a descriptor counts heap ELEMENTS: bytes for the 1PB cfitsio / astropy write, 2 or 4 bytes each for 1PI / 1PJ
        self.heap_elem_bytes = {}
        for col in ("COMPRESSED_DATA", "GZIP_COMPRESSED_DATA"):
            if col in self.columns:
                elem = self.columns[col][3]
                if elem not in ("B", "I", "J"):
                    raise ValueError(f"HDU {name}: {col} is a column of {elem!r} elements (expected 1PB, 1PI or 1PJ)")
                self.heap_elem_bytes[col] = {"B": 1, "I": 2, "J": 4}[elem]
        self.bscale, self.bzero = float(h.get("BSCALE", 1.0)), float(h.get("BZERO", 0.0))
        self.zscale_key, self.zzero_key = h.get("ZSCALE"), h.get("ZZERO")
        self.blank = h.get("ZBLANK", h.get("BLANK") if not self.quantized else None)
        if "ZBLANK" in self.columns:
            raise ValueError(f"HDU {name}: per-tile ZBLANK column (not written by astropy / cfitsio for these files)")

    def tiles(self, buf, hdu, out_index0):
        """The HDU's tile table as a TILE_DTYPE array (offsets relative to the start of ``buf``) and the rows stored in
        GZIP_COMPRESSED_DATA as [(row, float32 values)]."""
        base = hdu.data_offset
        rows = np.frombuffer(buf, dtype=np.uint8, count=self.row_bytes * self.n_rows, offset=base).reshape(self.n_rows, self.row_bytes)

        def column(name, dtype, width):
            off = self.columns[name][0]
            return np.ascontiguousarray(rows[:, off:off + width]).view(dtype).reshape(self.n_rows, -1)

        desc = column("COMPRESSED_DATA", ">i4", 8)
        t = np.zeros(self.n_rows, dtype=TILE_DTYPE)
        t["nbytes"] = desc[:, 0] * self.heap_elem_bytes["COMPRESSED_DATA"]  # (the descriptor counts elements)
        t["offset"] = base + self.theap + desc[:, 1].astype(np.int64)
        t["out_index"] = out_index0 + np.arange(self.n_rows, dtype=np.uint64) * np.uint64(self.width)
        t["mode"] = TILE_RICE
        if self.quantized:
            t["zscale"] = column("ZSCALE", ">f8", 8)[:, 0] if "ZSCALE" in self.columns else float(self.zscale_key)
            t["zzero"] = column("ZZERO", ">f8", 8)[:, 0] if "ZZERO" in self.columns else float(self.zzero_key)
        else:
            t["zscale"], t["zzero"] = self.bscale, self.bzero
        heap_end = hdu.data_offset + hdu.data_size
        if np.any(t["offset"] + t["nbytes"] > heap_end):
            raise ValueError(f"HDU {hdu.name}: a tile's stream lies outside the heap")
        patches = []
        empty = np.nonzero(desc[:, 0] == 0)[0]
        if len(empty):
            if "GZIP_COMPRESSED_DATA" not in self.columns:
                raise ValueError(f"HDU {hdu.name}: empty tiles and no GZIP_COMPRESSED_DATA column")
            gz = column("GZIP_COMPRESSED_DATA", ">i4", 8)
            for r in empty:
                n, ho = int(gz[r, 0]) * self.heap_elem_bytes["GZIP_COMPRESSED_DATA"], int(gz[r, 1])
                if n == 0:
                    raise ValueError(f"HDU {hdu.name}: tile {int(r)} holds no data")
                start = base + self.theap + ho
                raw = zlib.decompress(bytes(memoryview(buf)[start:start + n]), 47)
                dt = {-32: ">f4", -64: ">f8", 8: ">u1", 16: ">i2", 32: ">i4"}[self.zbitpix]
                vals = np.frombuffer(raw, dtype=dt, count=self.width)
                vals = vals.astype(np.float32) if self.quantized else (vals.astype(np.float64) * self.bscale + self.bzero).astype(np.float32)
                patches.append((int(r), vals))
                t["mode"][r] = TILE_SKIP
        return t, patches


def host_image(buf, hdu):
    """A small plain IMAGE HDU (PSF kernels) decoded on the host as float32."""
    h = hdu.header
    bitpix = int(h["BITPIX"])
    dt = {8: ">u1", 16: ">i2", 32: ">i4", 64: ">i8", -32: ">f4", -64: ">f8"}[bitpix]
    shape = tuple(int(h[f"NAXIS{a}"]) for a in range(int(h["NAXIS"]), 0, -1))
    raw = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape)), offset=hdu.data_offset)
    bscale, bzero = float(h.get("BSCALE", 1.0)), float(h.get("BZERO", 0.0))
    if bscale == 1.0 and bzero == 0.0:
        return raw.astype(np.float32).reshape(shape)
    return (raw.astype(np.float64) * bscale + bzero).astype(np.float32).reshape(shape)


def workunit_plan(buf, first_image=0, num_images=None, hdus=None):
    """Host half of the ingest, no device needed: which HDUs hold the layers of images ``first_image`` ... and how they
    are stored.  Returns a dict with ``times`` (MJD per image), ``shape`` (H, W), ``psfs`` and per image the SCI / VAR / MSK
    HDUs.  ``num_images`` defaults to the primary header's NUMIMG (work_unit.py:529)."""
    hdus = parse_fits(buf) if hdus is None else hdus
    if num_images is None:
        if "NUMIMG" not in hdus[0].header:
            raise ValueError("the primary header has no NUMIMG: not a WorkUnit file")
        num_images = int(hdus[0].header["NUMIMG"])
    images, times, psfs, shape = [], [], [], None
    for i in range(first_image, first_image + num_images):
        sci, var = find_hdu(hdus, f"SCI_{i}"), find_hdu(hdus, f"VAR_{i}")
        if sci is None or var is None:
            raise ValueError(f"WorkUnit file has no SCI_{i} / VAR_{i} extension")
        msk, psf = find_hdu(hdus, f"MSK_{i}"), find_hdu(hdus, f"PSF_{i}")
        for layer in (sci, var, msk):
            if layer is None:
                continue
            if shape is None:
                shape = layer.image_shape
            elif layer.image_shape != shape:
                raise ValueError(f"{layer.name}: {layer.image_shape} pixels, the stack has {shape}")
        if "MJD" not in sci.header:
            raise ValueError(f"SCI_{i} has no MJD card")
        times.append(float(sci.header["MJD"]))
        psfs.append(host_image(buf, psf) if psf is not None else np.ones((1, 1), dtype=np.float32))
        images.append({"sci": sci, "var": var, "msk": msk})
    return {"times": np.asarray(times, dtype=np.float64), "shape": shape, "psfs": psfs, "images": images, "hdus": hdus}


# ---- the device half ------------------------------------------------------------------------------------------------
_bound = False


def _lib():
    from kbmod_amd import capi

    lib = capi.load_lib()
    global _bound
    if not _bound:
        lib.kb_fits_decode_rice.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                            C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.kb_fits_decode_image.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_uint64, C.c_void_p, C.c_void_p]
        lib.kb_fits_apply_mask.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        _bound = True
    return lib


def _check(lib, rc):
    if rc != 0:
        raise RuntimeError(lib.kb_last_error().decode())


class DeviceWorkUnit:
    """The image layers of a WorkUnit, resident on the device: ``sci`` / ``var`` = [T][H][W] float32 (torch tensors, masked
    pixels NaN), ``times`` the MJDs, ``psfs`` the per-image kernels (host, small)."""

    def __init__(self, sci, var, times, psfs, stats):
        self.sci, self.var, self.times, self.psfs, self.stats = sci, var, np.asarray(times, dtype=np.float64), psfs, stats

    @property
    def zeroed_times(self):
        return self.times - self.times[0]  # image_stack_py.py:388

    def build_psi_phi(self, num_bytes=-1, build_flags=0):
        """psi/phi on the device (kb_build_psi_phi_from_device_ex): returns (capi.Meta, device pointer as int); the caller
        releases the array with kb_free_gpu_block."""
        import torch
        from kbmod_amd import capi

        lib = _lib()
        T, H, W = self.sci.shape
        psf_all = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.float32).ravel() for p in self.psfs]))
        dims = np.asarray([p.shape[0] for p in self.psfs], dtype=np.int32)
        meta, arr = capi.Meta(), C.c_void_p()
        with torch.cuda.device(self.sci.device):
            stream = torch.cuda.current_stream().cuda_stream
            _check(lib, lib.kb_build_psi_phi_from_device_ex(self.sci.data_ptr(), self.var.data_ptr(), psf_all.ctypes.data,
                                                            dims.ctypes.data, T, H, W, num_bytes, build_flags, C.byref(meta),
                                                            C.byref(arr), stream))
            torch.cuda.synchronize()
        return meta, arr.value

    def device_stack(self):
        """The layers as the ``kbmod_amd.stamp_utils.DeviceStack`` the stamp / coadd stages read (append_coadds,
        append_all_stamps) -- the same memory, no copy."""
        from kbmod_amd.stamp_utils import DeviceStack

        return DeviceStack.from_device(self.sci, self.var, zeroed_times=self.zeroed_times, times=self.times)

    def stack_search(self, num_bytes=-1, separable_psf=False, empty_footprint_is_zero=False):
        """A ``StackSearch`` over these layers -- built from the device-resident stacks, nothing returns to the host."""
        import torch
        import kbmod_amd.search as kb

        with torch.cuda.device(self.sci.device):
            torch.cuda.synchronize()
            T, H, W = self.sci.shape
            return kb.StackSearch.from_device_stacks(self.sci.data_ptr(), self.var.data_ptr(), T, H, W,
                                                     [np.asarray(p, dtype=np.float32) for p in self.psfs],
                                                     [float(t) for t in self.zeroed_times], num_bytes, separable_psf,
                                                     empty_footprint_is_zero)


_pinned_pool = {}


def release_pinned_buffer():
    """Give the page-locked file buffer back (it is kept between loads)."""
    _pinned_pool.clear()


def _pinned_file(paths, device=None):
    """The files' bytes back to back (each padded to 16 bytes) in ONE page-locked buffer; returns (tensor, [start offsets])."""
    import torch

    sizes = [os.path.getsize(p) for p in paths]
    starts, total = [], 0
    for s in sizes:
        starts.append(total)
        total += -(-s // 16) * 16
    # one page-locked buffer per process, kept between calls and grown when a larger file arrives: page-locking hundreds
    # of megabytes costs more than reading them (the loaders synchronise before they return, so it is free again by then)
    host = _pinned_pool.get("buffer")
    if host is None or host.numel() < total + 16:
        _pinned_pool["buffer"] = host = None
        try:
            host = torch.empty(total + 16, dtype=torch.uint8, pin_memory=True)
            _pinned_pool["buffer"] = host
        except RuntimeError:  # no page-locked memory of that size: the upload then goes through the runtime's staging buffers
            host = torch.empty(total + 16, dtype=torch.uint8)
    host = host[:total + 16]
    view = host.numpy()
    # pieces of 32 MiB read by a few threads straight into the buffer (the read system call releases the interpreter lock;
    # one thread copies out of the page cache at a fraction of what the DMA behind it moves)
    piece = 32 << 20
    jobs = [(p, o, off, min(piece, s - off)) for p, s, o in zip(paths, sizes, starts) for off in range(0, s, piece)]

    # with a device: every piece goes out on a copy stream as soon as it has been read (the DMA of one piece under the
    # reads of the next); the caller's stream is made to wait for that stream before anything reads the bytes
    dev = copy_stream = None
    if device is not None and total >= (64 << 20):  # (a small file goes out in one DMA behind its read: no stream to set up)
        dev = torch.empty(total + 16, dtype=torch.uint8, device=device)
        dev[total:].zero_()
        copy_stream = torch.cuda.Stream(device=device)
        copy_stream.wait_stream(torch.cuda.current_stream(device))

    def read_piece(job):
        p, o, off, n = job
        fd = os.open(p, os.O_RDONLY)
        try:
            done = 0
            while done < n:
                got = os.preadv(fd, [memoryview(view[o + off + done:o + off + n])], off + done)
                if got <= 0:
                    raise IOError(f"short read of {p}")
                done += got
        finally:
            os.close(fd)
        if dev is not None:
            with torch.cuda.device(device), torch.cuda.stream(copy_stream):
                dev[o + off:o + off + n].copy_(host[o + off:o + off + n], non_blocking=True)

    if len(jobs) > 1:
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as pool:
            list(pool.map(read_piece, jobs))
    else:
        for job in jobs:
            read_piece(job)
    if dev is not None:
        torch.cuda.current_stream(device).wait_stream(copy_stream)
        _pinned_pool["in_flight"] = copy_stream  # (kept until the loader has synchronised)
    if device is not None:
        return host, starts, sizes, dev
    return host, starts, sizes


def _decode_layers(lib, torch, dev_bytes, host_view, file_base, layer_hdus, out, stream):
    """Decode the T HDUs ``layer_hdus`` (all SCI, all VAR or all MSK; same shape) from the uploaded file bytes into
    ``out`` = [T][H][W] float32 on the device.  ``file_base[t]``: where the file holding HDU t starts in the buffer."""
    T, H, W = out.shape
    device = out.device
    status = torch.zeros(2, dtype=torch.int32, device=device)
    comp = [(t, h) for t, h in enumerate(layer_hdus) if h.is_compressed_image]
    plain = [(t, h) for t, h in enumerate(layer_hdus) if not h.is_compressed_image]
    patches = []
    groups = {}
    for t, h in comp:
        lay = CompressedLayout(h)
        sub = host_view[file_base[t]:]
        tiles, gz = lay.tiles(sub, h, t * H * W)
        tiles["offset"] += np.uint64(file_base[t])
        key = (lay.blocksize, lay.bytepix, lay.quantized, lay.blank)
        groups.setdefault(key, []).append(tiles)
        patches += [(t, r, v) for r, v in gz]
    n_streams = 0
    for (blocksize, bytepix, quantized, blank), parts in groups.items():
        table = np.concatenate(parts)
        n_streams += len(table)
        tiles_dev = torch.from_numpy(table.view(np.uint8).reshape(-1)).to(device)
        _check(lib, lib.kb_fits_decode_rice(dev_bytes.data_ptr(), dev_bytes.numel(), tiles_dev.data_ptr(), len(table), W,
                                            blocksize, bytepix, 1 if quantized else 0, 0 if blank is None else 1,
                                            0 if blank is None else int(blank), out.data_ptr(), status.data_ptr(), stream))
        bad = status.cpu().numpy()  # (synchronises: the tile table may be released afterwards)
        if bad[0] != 0:
            raise ValueError(f"{int(bad[0])} compressed tile(s) end before their pixels do (first: table row {int(bad[1]) - 1})")
    for t, h in plain:
        hd = h.header
        bitpix = int(hd["BITPIX"])
        _check(lib, lib.kb_fits_decode_image(dev_bytes.data_ptr() + file_base[t] + h.data_offset, bitpix,
                                             float(hd.get("BSCALE", 1.0)), float(hd.get("BZERO", 0.0)), H * W,
                                             out[t].data_ptr(), stream))
    for t, r, vals in patches:
        out[t, r].copy_(torch.from_numpy(np.ascontiguousarray(vals)))
    return n_streams, len(patches)


def _resolve_device(device):
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError("load_workunit needs a GPU: the FITS layers are decoded on the device, there is no CPU path")
    _lib()  # (raises when the device library is not built)
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)


def _load(paths, plans, file_of_image, device):
    import time

    import torch

    lib = _lib()
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    t_start = time.perf_counter()
    with torch.cuda.device(device):
        host, starts, sizes = plans["host"], plans["starts"], plans["sizes"]
        host_view = host.numpy()
        stream = torch.cuda.current_stream().cuda_stream
        dev_bytes = plans.get("dev_bytes")
        if dev_bytes is None:
            dev_bytes = host.to(device, non_blocking=True)
        images = plans["images"]
        T = len(images)
        H, W = plans["shape"]
        base = [starts[file_of_image[t]] for t in range(T)]
        sci = torch.empty((T, H, W), dtype=torch.float32, device=device)
        var = torch.empty((T, H, W), dtype=torch.float32, device=device)
        n_sci = _decode_layers(lib, torch, dev_bytes, host_view, base, [im["sci"] for im in images], sci, stream)
        n_var = _decode_layers(lib, torch, dev_bytes, host_view, base, [im["var"] for im in images], var, stream)
        with_mask = [t for t in range(T) if images[t]["msk"] is not None]
        if with_mask:
            mask = torch.empty((len(with_mask), H, W), dtype=torch.float32, device=device)
            _decode_layers(lib, torch, dev_bytes, host_view, [base[t] for t in with_mask],
                           [images[t]["msk"] for t in with_mask], mask, stream)
            if len(with_mask) == T:
                _check(lib, lib.kb_fits_apply_mask(sci.data_ptr(), var.data_ptr(), mask.data_ptr(), T * H * W, stream))
            else:
                for k, t in enumerate(with_mask):
                    _check(lib, lib.kb_fits_apply_mask(sci[t].data_ptr(), var[t].data_ptr(), mask[k].data_ptr(), H * W, stream))
        torch.cuda.synchronize()
    stats = {"file_bytes": int(sum(sizes)), "decoded_bytes": int(2 * T * H * W * 4), "rice_tiles": n_sci[0] + n_var[0],
             "gzip_tiles": n_sci[1] + n_var[1], "seconds": time.perf_counter() - t_start}
    return DeviceWorkUnit(sci, var, plans["times"], plans["psfs"], stats)


def load_workunit(filename, device=None):
    """``WorkUnit.from_fits`` for the image layers (work_unit.py:489-608): the single-file form.  Returns a
    ``DeviceWorkUnit`` whose stacks were decoded on the device."""
    if not os.path.isfile(filename):
        raise ValueError(f"WorkUnit file {filename} not found.")  # work_unit.py:511-512
    device = _resolve_device(device)
    host, starts, sizes, dev_bytes = _pinned_file([filename], device)
    plan = workunit_plan(host.numpy()[:sizes[0]])
    plan.update(host=host, starts=starts, sizes=sizes, dev_bytes=dev_bytes)
    return _load([filename], plan, [0] * len(plan["images"]), device)


def load_sharded_workunit(filename, directory, device=None):
    """``WorkUnit.from_sharded_fits`` for the image layers (work_unit.py:782-897): the primary file names NUMIMG, image i
    lives in ``{i}_{filename}`` of the same directory."""
    primary = os.path.join(directory, filename)
    if not os.path.isfile(primary):
        raise ValueError(f"WorkUnit file {filename} not found.")  # work_unit.py:813-814
    with open(primary, "rb") as fh:
        head = parse_fits(fh.read())
    if "NUMIMG" not in head[0].header:
        raise ValueError("the primary header has no NUMIMG: not a WorkUnit file")
    n = int(head[0].header["NUMIMG"])
    shards = [os.path.join(directory, f"{i}_{filename}") for i in range(n)]
    for i, p in enumerate(shards):
        if not os.path.isfile(p):
            raise ValueError(f"No shard provided for index {i} for {filename}")  # work_unit.py:862-863
    device = _resolve_device(device)
    host, starts, sizes, dev_bytes = _pinned_file(shards, device)
    view = host.numpy()
    merged = {"times": [], "psfs": [], "images": [], "shape": None}
    for i in range(n):
        part = workunit_plan(view[starts[i]:starts[i] + sizes[i]], first_image=i, num_images=1)
        if merged["shape"] is None:
            merged["shape"] = part["shape"]
        elif part["shape"] != merged["shape"]:
            raise ValueError(f"shard {i}: {part['shape']} pixels, the stack has {merged['shape']}")
        merged["times"] += list(part["times"])
        merged["psfs"] += part["psfs"]
        merged["images"] += part["images"]
    merged["times"] = np.asarray(merged["times"], dtype=np.float64)
    merged.update(host=host, starts=starts, sizes=sizes, dev_bytes=dev_bytes)
    return _load(shards, merged, list(range(n)), device)
