"""CPU oracle of the FITS ingest path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/`` may import this module; the product (``kbmod_amd.fits_ingest`` + ``csrc/fits_kernels.hip``) never does.

What the reference does on this path is ``WorkUnit.from_fits`` / ``read_image_data_from_hdul``
(/root/reference/src/kbmod/work_unit.py:489-608, 1149-1200): ``hdul[f"SCI_{i}"].data.astype(np.single)``, the same for
``VAR_i``, ``sci[mask > 0] = var[mask > 0] = nan`` for ``MSK_i``, ``PSF_i`` as the kernel, ``SCI_i.header["MJD"]`` as the
epoch.  The decoding itself lives in a third-party dependency that is absent from /root/reference: **astropy**
(``astropy.io.fits``; /root/reference/pyproject.toml lists ``astropy>=5.3``, no upper pin) and the **cfitsio** it vendors.
The files the reference writes (work_unit.py:1066-1147) hold SCI_i / VAR_i as ``CompImageHDU(compression_type="RICE_1",
quantize_level=-0.01)`` -- astropy's default quantize method, ``NO_DITHER`` -- and MSK_i / PSF_i as plain image HDUs.
This file restates the published algorithms those call into:

* FITS 4.0 standard, sections 3-4 (2880-byte blocks, 80-character cards), 5.2-5.3 (big-endian two's-complement integers
  and IEEE floats), 4.4.2.5 (physical = BZERO + BSCALE * array), 7.3.5 (``P`` array descriptors: count, heap offset);
* the tiled-image convention (FITS 4.0 section 10): one table row per tile, ``COMPRESSED_DATA`` / ``GZIP_COMPRESSED_DATA`` /
  ``ZSCALE`` / ``ZZERO`` columns, ``ZBLANK``, ``ZQUANTIZ = NO_DITHER``: value = ZSCALE * integer + ZZERO;
* ``RICE_1`` (section 10.4.1; cfitsio ``ricecomp.c`` ``fits_rdecomp`` / ``fits_rcomp``, White & Percival 1994): the first
  pixel verbatim (big-endian, BYTEPIX bytes), then per block of BLOCKSIZE pixels an FS code of FSBITS bits -- 0: every
  difference is zero; FSMAX + 1: the mapped differences verbatim in BBITS bits each; otherwise FS = code - 1 split bits,
  each mapped difference as (value >> FS) zero bits, a one bit, and the low FS bits -- with differences to the previous
  pixel mapped to non-negative integers (d >= 0 -> 2 d, d < 0 -> -2 d - 1), all arithmetic modulo 2^32.

Pinned on the reference's own data files (copied as fixtures to tests/golden/, see tests/test_oracle_fits.py):
``tests/data/shifted_wcs_diff_dimms_tiled.fits`` -- every one of its 4 x 2 x 50 RICE tiles ends exactly on its last byte,
the variance layers decode to the 4.0 the reference's fake-data generator wrote (noise level 2), the science layers to
N(0, 2^2) noise on the 0.01 grid that ``quantize_level=-0.01`` prescribes, the brightest source of each to flux x the file's
own PSF_i kernel within that noise -- and ``data/small/*.fits`` (plain BITPIX -32 HDUs).  astropy is not installed in this image, so no vector could be produced BY the reference for this path: **parity of
the RICE leg is pinned on those properties and on the published algorithm, not on reference-produced pixel values**.
"""

import struct
import zlib

import numpy as np

BLOCK = 2880


# ---- headers ------------------------------------------------------------------------------------------------------
def _card_value(card):
    """Value of one 80-character card (FITS 4.0 section 4.2): strings without their quotes, T / F as bool, numbers."""
    text = card[10:]
    s = text.lstrip()
    if s.startswith("'"):
        out, i = [], 1
        while i < len(s):
            if s[i] == "'":
                if i + 1 < len(s) and s[i + 1] == "'":
                    out.append("'")
                    i += 2
                    continue
                break
            out.append(s[i])
            i += 1
        return "".join(out).rstrip()
    s = s.split("/")[0].strip()
    if s == "T":
        return True
    if s == "F":
        return False
    if s == "":
        return None
    try:
        return int(s)
    except ValueError:
        return float(s.replace("D", "E"))


def parse(buf):
    """All HDUs of a FITS file held in ``buf``: [(header dict, data offset, data size in bytes)]."""
    hdus, off = [], 0
    while off + BLOCK <= len(buf):
        header, done = {}, False
        while not done:
            blk = buf[off:off + BLOCK]
            off += BLOCK
            for i in range(0, BLOCK, 80):
                card = blk[i:i + 80].decode("ascii", "replace")
                key = card[:8].rstrip()
                if key == "END":
                    done = True
                    break
                if card[8:10] == "= " and key not in header:
                    header[key] = _card_value(card)
        naxis = int(header.get("NAXIS", 0))
        size = 0
        if naxis > 0:
            size = abs(int(header["BITPIX"])) // 8
            for a in range(naxis):
                size *= int(header[f"NAXIS{a + 1}"])
            size += int(header.get("PCOUNT", 0))
        hdus.append((header, off, size))
        off += (size + BLOCK - 1) // BLOCK * BLOCK
    return hdus


def find(hdus, name):
    for h in hdus:
        if str(h[0].get("EXTNAME", "")).upper() == name.upper():
            return h
    return None


# ---- plain image HDUs ---------------------------------------------------------------------------------------------
_BITPIX = {8: ">u1", 16: ">i2", 32: ">i4", 64: ">i8", -32: ">f4", -64: ">f8"}


def image_data(buf, hdu):
    """A plain IMAGE HDU as float32 (``.data.astype(np.single)``): physical = BZERO + BSCALE * array, in double."""
    header, off, _ = hdu
    shape = tuple(int(header[f"NAXIS{a}"]) for a in range(int(header["NAXIS"]), 0, -1))
    n = int(np.prod(shape))
    bitpix = int(header["BITPIX"])
    raw = np.frombuffer(buf, dtype=_BITPIX[bitpix], count=n, offset=off)
    bscale, bzero = float(header.get("BSCALE", 1.0)), float(header.get("BZERO", 0.0))
    if bitpix < 0 and bscale == 1.0 and bzero == 0.0:
        return raw.astype(np.float32).reshape(shape)
    return (raw.astype(np.float64) * bscale + bzero).astype(np.float32).reshape(shape)


# ---- RICE_1 -------------------------------------------------------------------------------------------------------
_RICE = {1: (3, 6, 8), 2: (4, 14, 16), 4: (5, 25, 32)}  # BYTEPIX -> (FSBITS, FSMAX, BBITS)
_M = 0xFFFFFFFF


def rice_decode(data, nx, blocksize=32, bytepix=4):
    """``nx`` integers of one tile; returns (int64 array of the BYTEPIX-wide values -- two's complement for 2 and 4 bytes,
    unsigned for 1 --, bytes consumed)."""
    fsbits, fsmax, bbits = _RICE[bytepix]
    wrap = (1 << bbits) - 1
    lastpix = int.from_bytes(data[0:bytepix], "big")
    p = bytepix
    b, nbits = data[p], 8
    p += 1
    out = np.zeros(nx, dtype=np.int64)
    i = 0
    while i < nx:
        nbits -= fsbits
        while nbits < 0:
            b = (b << 8) | data[p]
            p += 1
            nbits += 8
        fs = (b >> nbits) - 1
        b &= (1 << nbits) - 1
        imax = min(i + blocksize, nx)
        if fs < 0:  # a block of zero differences
            out[i:imax] = lastpix
            i = imax
        elif fs == fsmax:  # a block of verbatim differences
            while i < imax:
                k = bbits - nbits
                diff = (b << k) & _M
                k -= 8
                while k >= 0:
                    b = data[p]
                    p += 1
                    diff |= (b << k) & _M
                    k -= 8
                if nbits > 0:
                    b = data[p]
                    p += 1
                    diff |= b >> (-k)
                    b &= (1 << nbits) - 1
                else:
                    b = 0
                diff &= wrap
                diff = (~(diff >> 1)) & _M if diff & 1 else diff >> 1
                lastpix = (lastpix + diff) & wrap
                out[i] = lastpix
                i += 1
        else:
            while i < imax:
                while b == 0:
                    nbits += 8
                    b = data[p]
                    p += 1
                nzero = nbits - b.bit_length()
                nbits -= nzero + 1
                b ^= 1 << nbits
                nbits -= fs
                while nbits < 0:
                    b = (b << 8) | data[p]
                    p += 1
                    nbits += 8
                diff = ((nzero << fs) | (b >> nbits)) & _M
                b &= (1 << nbits) - 1
                diff = (~(diff >> 1)) & _M if diff & 1 else diff >> 1
                lastpix = (lastpix + diff) & wrap
                out[i] = lastpix
                i += 1
    if bytepix == 1:  # 8-bit FITS pixels are unsigned (cfitsio fits_rdecomp_byte fills an unsigned char array)
        return out, p
    sign = 1 << (bbits - 1)
    return np.where(out >= sign, out - (1 << bbits), out), p


class _BitWriter:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value, bits):
        self.acc = (self.acc << bits) | (value & ((1 << bits) - 1))
        self.n += bits
        while self.n >= 8:
            self.n -= 8
            self.out.append((self.acc >> self.n) & 0xFF)
        self.acc &= (1 << self.n) - 1

    def done(self):
        if self.n:
            self.out.append((self.acc << (8 - self.n)) & 0xFF)
            self.acc = self.n = 0
        return bytes(self.out)


def rice_encode(values, blocksize=32, bytepix=4, force_fs=None):
    """A RICE_1 stream that ``rice_decode`` (and any conforming decoder) turns back into ``values``.  The split FS of a
    block follows cfitsio's estimate from the block's mean mapped difference (ricecomp.c fits_rcomp); ``force_fs`` pins
    it for tests (-1 is refused unless the block is constant; ``"raw"`` writes every block verbatim)."""
    fsbits, fsmax, bbits = _RICE[bytepix]
    wrap = (1 << bbits) - 1
    v = [int(x) & wrap for x in values]
    w = _BitWriter()
    w.put(v[0], 8 * bytepix)
    last = v[0]
    for i in range(0, len(v), blocksize):
        blk = v[i:i + blocksize]
        mapped = []
        for x in blk:
            d = (x - last) & wrap
            if d >= (1 << (bbits - 1)):
                d -= 1 << bbits
            mapped.append((2 * d if d >= 0 else -2 * d - 1) & wrap)
            last = x
        total = sum(mapped)
        dpsum = (total - len(blk) // 2 - 1) / len(blk)
        psum = int(max(dpsum, 0.0)) >> 1
        fs = 0
        while psum > 0:
            fs += 1
            psum >>= 1
        if force_fs == "raw":
            fs = fsmax
        elif force_fs is not None:
            fs = int(force_fs)
        if fs >= fsmax:
            w.put(fsmax + 1, fsbits)
            for m in mapped:
                w.put(m, bbits)
        elif fs == 0 and total == 0:
            w.put(0, fsbits)
        else:
            w.put(fs + 1, fsbits)
            for m in mapped:
                top = m >> fs
                while top >= 24:  # long zero runs in pieces the writer's accumulator takes
                    w.put(0, 24)
                    top -= 24
                w.put(1, top + 1)
                if fs:
                    w.put(m, fs)
    return w.done()


# ---- tiled-compressed image HDUs ----------------------------------------------------------------------------------
_TFORM_BYTES = {"L": 1, "B": 1, "I": 2, "J": 4, "K": 8, "E": 4, "D": 8, "A": 1}


def _columns(header):
    """{TTYPE: (byte offset in a row, TFORM)} of a binary table."""
    cols, off = {}, 0
    for c in range(1, int(header["TFIELDS"]) + 1):
        form = str(header[f"TFORM{c}"]).strip()
        digits = ""
        while form and form[0].isdigit():
            digits, form = digits + form[0], form[1:]
        rep = int(digits) if digits else 1
        code = form[0]
        size = 8 if code == "P" else 16 if code == "Q" else _TFORM_BYTES[code]
        cols[str(header.get(f"TTYPE{c}", f"COL{c}")).strip()] = (off, code + form[1:])
        off += rep * size
    return cols


def compressed_image_data(buf, hdu):
    """A tiled-compressed image HDU (ZIMAGE = T, RICE_1, row tiles) as float32."""
    header, off, _ = hdu
    assert header.get("ZIMAGE") is True and str(header["ZCMPTYPE"]).strip() == "RICE_1"
    width, height = int(header["ZNAXIS1"]), int(header["ZNAXIS2"])
    assert int(header.get("ZTILE1", width)) == width and int(header.get("ZTILE2", 1)) == 1
    zbitpix = int(header["ZBITPIX"])
    blocksize, bytepix = 32, 4
    for k in range(1, 10):
        name = header.get(f"ZNAME{k}")
        if name is None:
            break
        if str(name).strip() == "BLOCKSIZE":
            blocksize = int(header[f"ZVAL{k}"])
        if str(name).strip() == "BYTEPIX":
            bytepix = int(header[f"ZVAL{k}"])
    row_bytes, n_rows = int(header["NAXIS1"]), int(header["NAXIS2"])
    assert n_rows == height
    theap = int(header.get("THEAP", row_bytes * n_rows))
    cols = _columns(header)
    quantized = zbitpix < 0
    if quantized:
        assert str(header.get("ZQUANTIZ", "NO_DITHER")).strip() == "NO_DITHER"
    out = np.zeros((height, width), dtype=np.float32)
    for r in range(n_rows):
        row = buf[off + r * row_bytes:off + (r + 1) * row_bytes]
        n, ho = struct.unpack(">ii", row[cols["COMPRESSED_DATA"][0]:cols["COMPRESSED_DATA"][0] + 8])
        if n == 0 and "GZIP_COMPRESSED_DATA" in cols:  # a tile the writer could not quantise: gzip of the big-endian floats
            gn, gho = struct.unpack(">ii", row[cols["GZIP_COMPRESSED_DATA"][0]:cols["GZIP_COMPRESSED_DATA"][0] + 8])
            raw = zlib.decompress(bytes(buf[off + theap + gho:off + theap + gho + gn]), 47)
            out[r] = np.frombuffer(raw, dtype=">f4" if zbitpix == -32 else ">f8", count=width).astype(np.float32)
            continue
        ints, used = rice_decode(buf[off + theap + ho:off + theap + ho + n], width, blocksize, bytepix)
        assert used == n, (r, used, n)
        if not quantized:
            bscale, bzero = float(header.get("BSCALE", 1.0)), float(header.get("BZERO", 0.0))
            out[r] = (ints.astype(np.float64) * bscale + bzero).astype(np.float32)
            continue
        zscale = struct.unpack(">d", row[cols["ZSCALE"][0]:cols["ZSCALE"][0] + 8])[0] if "ZSCALE" in cols else float(header["ZSCALE"])
        zzero = struct.unpack(">d", row[cols["ZZERO"][0]:cols["ZZERO"][0] + 8])[0] if "ZZERO" in cols else float(header["ZZERO"])
        vals = (ints.astype(np.float64) * zscale + zzero).astype(np.float32)
        if "ZBLANK" in cols:
            blank = struct.unpack(">i", row[cols["ZBLANK"][0]:cols["ZBLANK"][0] + 4])[0]
            vals[ints == blank] = np.nan
        elif "ZBLANK" in header:
            vals[ints == int(header["ZBLANK"])] = np.nan
        out[r] = vals
    return out


def hdu_data(buf, hdu):
    if hdu[0].get("ZIMAGE") is True:
        return compressed_image_data(buf, hdu)
    return image_data(buf, hdu)


def read_workunit_layers(buf):
    """What ``WorkUnit.from_fits`` appends to its image stack (work_unit.py:581-592, 1149-1200): per image
    (obstime, sci, var, mask, psf) with the mask applied to sci and var."""
    hdus = parse(buf)
    n = int(hdus[0][0]["NUMIMG"])
    layers = []
    for i in range(n):
        sci_hdu = find(hdus, f"SCI_{i}")
        sci = hdu_data(buf, sci_hdu).copy()
        var = hdu_data(buf, find(hdus, f"VAR_{i}")).copy()
        msk = find(hdus, f"MSK_{i}")
        if msk is not None:
            mask = hdu_data(buf, msk)
            sci[mask > 0] = np.nan
            var[mask > 0] = np.nan
        else:
            mask = np.zeros_like(sci)
        psf_hdu = find(hdus, f"PSF_{i}")
        psf = hdu_data(buf, psf_hdu) if psf_hdu is not None else np.ones((1, 1), dtype=np.float32)
        layers.append((float(sci_hdu[0]["MJD"]), sci, var, mask, psf))
    return layers


# ---- a writer for test files (own layout choices; what matters is that a conforming reader decodes it) -----------------
def _card(key, value, comment=""):
    if isinstance(value, bool):
        v = f"{'T' if value else 'F':>20}"
    elif isinstance(value, str):
        v = f"'{value:<8}'"
        v = f"{v:<20}"
    elif isinstance(value, float):
        v = f"{value!r:>20}".replace("e", "E")
    else:
        v = f"{value:>20}"
    return f"{key:<8}= {v}{' / ' + comment if comment else ''}"[:80].ljust(80)


def _header_bytes(cards):
    text = "".join(cards) + "END".ljust(80)
    text += " " * (-len(text) % BLOCK)
    return text.encode("ascii")


def _pad(data):
    return data + b"\0" * (-len(data) % BLOCK)


def write_image_hdu(name, array, extra=()):
    """A plain IMAGE extension (float32 / float64 / int8 with BZERO = -128 like astropy / int16 / int32)."""
    a = np.asarray(array)
    cards = [_card("XTENSION", "IMAGE")]
    if a.dtype == np.int8:
        bitpix, raw, scale = 8, (a.astype(np.int16) + 128).astype(">u1"), [_card("BSCALE", 1), _card("BZERO", -128)]
    else:
        bitpix = {"f4": -32, "f8": -64, "i2": 16, "i4": 32, "u1": 8}[a.dtype.str[1:]]
        raw, scale = a.astype(_BITPIX[bitpix]), []
    cards += [_card("BITPIX", bitpix), _card("NAXIS", a.ndim)]
    cards += [_card(f"NAXIS{k + 1}", int(s)) for k, s in enumerate(a.shape[::-1])]
    cards += [_card("PCOUNT", 0), _card("GCOUNT", 1)] + scale + [_card("EXTNAME", name)]
    cards += [_card(k, v) for k, v in extra]
    return _header_bytes(cards) + _pad(raw.tobytes())


def write_compressed_hdu(name, array, quantum=0.01, extra=(), blank=None, force_fs=None, gzip_rows=()):
    """A tiled-compressed float32 image the way the reference's writer lays it out (row tiles, RICE_1, BLOCKSIZE 32,
    BYTEPIX 4, NO_DITHER, per-row ZSCALE / ZZERO columns with ZZERO = the row's minimum); NaN pixels become ``blank``
    (ZBLANK keyword).  ``gzip_rows``: rows stored losslessly in GZIP_COMPRESSED_DATA instead.  Returns (bytes, the float32
    image a reader must produce)."""
    a = np.asarray(array, dtype=np.float32)
    height, width = a.shape
    table, heap, expect = bytearray(), bytearray(), np.zeros_like(a)
    for r in range(height):
        rowv = a[r].astype(np.float64)
        if r in gzip_rows:
            comp = zlib.compress(a[r].astype(">f4").tobytes())
            table += struct.pack(">iiiidd", 0, 0, len(comp), len(heap), 1.0, 0.0)
            heap += comp
            expect[r] = a[r]
            continue
        good = np.isfinite(rowv)
        zzero = float(rowv[good].min()) if good.any() else 0.0
        ints = np.zeros(width, dtype=np.int64)
        ints[good] = np.floor((rowv[good] - zzero) / quantum + 0.5).astype(np.int64)
        if blank is not None:
            ints[~good] = blank
        comp = rice_encode(ints, 32, 4, force_fs)
        table += struct.pack(">iiiidd", len(comp), len(heap), 0, 0, float(quantum), zzero)
        heap += comp
        vals = (ints.astype(np.float64) * float(quantum) + zzero).astype(np.float32)
        if blank is not None:
            vals[~good] = np.nan
        expect[r] = vals
    cards = [_card("XTENSION", "BINTABLE"), _card("BITPIX", 8), _card("NAXIS", 2), _card("NAXIS1", 32),
             _card("NAXIS2", height), _card("PCOUNT", len(heap)), _card("GCOUNT", 1), _card("TFIELDS", 4),
             _card("TTYPE1", "COMPRESSED_DATA"), _card("TFORM1", "1PB"), _card("TTYPE2", "GZIP_COMPRESSED_DATA"),
             _card("TFORM2", "1PB"), _card("TTYPE3", "ZSCALE"), _card("TFORM3", "1D"), _card("TTYPE4", "ZZERO"),
             _card("TFORM4", "1D"), _card("ZIMAGE", True), _card("ZTENSION", "IMAGE"), _card("ZBITPIX", -32),
             _card("ZNAXIS", 2), _card("ZNAXIS1", width), _card("ZNAXIS2", height), _card("ZTILE1", width),
             _card("ZTILE2", 1), _card("ZCMPTYPE", "RICE_1"), _card("ZNAME1", "BLOCKSIZE"), _card("ZVAL1", 32),
             _card("ZNAME2", "BYTEPIX"), _card("ZVAL2", 4), _card("ZQUANTIZ", "NO_DITHER"), _card("EXTNAME", name)]
    if blank is not None:
        cards.append(_card("ZBLANK", int(blank)))
    cards += [_card(k, v) for k, v in extra]
    return _header_bytes(cards) + _pad(bytes(table) + bytes(heap)), expect


def write_workunit(layers, compressed=True, quantum=0.01, blank=-2147483647, force_fs=None):
    """A single-file WorkUnit with the extensions ``from_fits`` reads (work_unit.py:489-608): PRIMARY with NUMIMG, then
    SCI_i / VAR_i / MSK_i / PSF_i per image.  ``layers``: [(mjd, sci, var, mask int8, psf)].  Returns (bytes, [(sci, var)
    as a reader must produce them BEFORE the mask is applied])."""
    out = bytearray(_header_bytes([_card("SIMPLE", True), _card("BITPIX", 8), _card("NAXIS", 0), _card("EXTEND", True),
                                   _card("NUMIMG", len(layers)), _card("REPRJCTD", False)]))
    expect = []
    for i, (mjd, sci, var, mask, psf) in enumerate(layers):
        extra = [("MJD", float(mjd)), ("NIND", 1), ("IND_0", i)]
        if compressed:
            s_bytes, s_exp = write_compressed_hdu(f"SCI_{i}", sci, quantum, extra, blank, force_fs)
            v_bytes, v_exp = write_compressed_hdu(f"VAR_{i}", var, quantum, [("MJD", float(mjd))], blank, force_fs)
        else:
            s_bytes, s_exp = write_image_hdu(f"SCI_{i}", np.asarray(sci, np.float32), extra), np.asarray(sci, np.float32)
            v_bytes, v_exp = write_image_hdu(f"VAR_{i}", np.asarray(var, np.float32), [("MJD", float(mjd))]), np.asarray(var, np.float32)
        out += s_bytes + v_bytes
        if mask is not None:
            out += write_image_hdu(f"MSK_{i}", np.asarray(mask, np.int8), [("MJD", float(mjd))])
        if psf is not None:
            out += write_image_hdu(f"PSF_{i}", np.asarray(psf, np.float32))
        expect.append((s_exp, v_exp))
    return bytes(out), expect
