"""FITS ingest on the device: kbmod_amd.fits_ingest + csrc/fits_kernels.hip through the C ABI against
oracle/fits_decode.py -- bit for bit: the decoded integers are exact and the float32 value is one double multiply-add
rounded once on both sides.  Inputs: the reference's own data files (tests/golden/) and files from the oracle's writer."""

import ctypes as C
import os

import numpy as np
import pytest

from oracle import fits_decode as fd

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TILED = os.path.join(GOLDEN, "shifted_wcs_diff_dimms_tiled.fits")
SMALL = os.path.join(GOLDEN, "small_000000.fits")


@pytest.fixture(scope="module")
def fi(kb):
    from kbmod_amd import fits_ingest

    return fits_ingest


def same(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)) or np.array_equal(a, b, equal_nan=True)


def check_against_oracle(wu, data):
    layers = fd.read_workunit_layers(data)
    sci, var = wu.sci.cpu().numpy(), wu.var.cpu().numpy()
    assert sci.shape[0] == len(layers)
    for t, (mjd, s, v, m, psf) in enumerate(layers):
        assert wu.times[t] == mjd
        assert np.array_equal(sci[t], s, equal_nan=True), f"science layer {t}"
        assert np.array_equal(var[t], v, equal_nan=True), f"variance layer {t}"
        assert np.array_equal(np.isnan(sci[t]), np.isnan(s))
        assert np.array_equal(wu.psfs[t], psf)


def test_reference_workunit_file(fi):
    wu = fi.load_workunit(TILED)
    check_against_oracle(wu, open(TILED, "rb").read())
    assert wu.stats["rice_tiles"] == 4 * 2 * 50 and wu.stats["gzip_tiles"] == 0
    assert float(wu.var.min()) == 4.0 == float(wu.var.max())
    assert list(wu.zeroed_times) == [0.0, 1.0, 2.0, 2.0]


def _layers(rng, T, H, W, nan_fraction=0.01, mask_fraction=0.02, bright=True):
    out = []
    for t in range(T):
        sci = rng.normal(0, 2, (H, W)).astype(np.float32)
        if bright:
            sci[H // 2, W // 3] = 1.0e6  # a difference too wide for any split: the verbatim block code
        sci[rng.random((H, W)) < nan_fraction] = np.nan
        var = (4.0 + rng.random((H, W))).astype(np.float32)
        mask = (rng.random((H, W)) < mask_fraction).astype(np.int8)
        out.append((59000.25 + 0.5 * t, sci, var, mask, np.full((5, 5), 0.04, np.float32)))
    return out


@pytest.mark.parametrize("shape", [(1, 1, 1), (2, 3, 31), (3, 64, 32), (2, 65, 33), (1, 130, 100), (2, 70, 257)])
@pytest.mark.parametrize("force", [None, "raw"])
def test_written_workunits(fi, tmp_path, shape, force):
    T, H, W = shape
    layers = _layers(np.random.default_rng(T * 1000 + H + W), T, H, W)
    data, _ = fd.write_workunit(layers, force_fs=force)
    path = tmp_path / "wu.fits"
    path.write_bytes(data)
    wu = fi.load_workunit(str(path))
    check_against_oracle(wu, data)
    assert tuple(wu.sci.shape) == shape


def test_every_split_code(fi, tmp_path):
    # rows whose differences call for FS = 0 ... 24, constant rows, and rows of alternating extremes
    W = 96
    rows = []
    for fs in range(0, 25):
        rng = np.random.default_rng(fs)
        rows.append(np.cumsum(rng.integers(-(1 << fs), 1 << fs, size=W, endpoint=True)).astype(np.float64) * 0.01)
    rows.append(np.full(W, 3.25))
    rows.append(np.where(np.arange(W) % 2 == 0, -1.0e7, 1.0e7))
    img = np.asarray(rows, dtype=np.float32)
    layers = [(1.0, img, np.abs(img) + 1, None, None)]
    data, _ = fd.write_workunit(layers)
    path = tmp_path / "fs.fits"
    path.write_bytes(data)
    wu = fi.load_workunit(str(path))
    check_against_oracle(wu, data)
    assert wu.psfs[0].shape == (1, 1)  # identity PSF when PSF_i is absent (work_unit.py:1197-1198)


def test_uncompressed_workunit_and_reference_plain_file(fi, tmp_path):
    layers = _layers(np.random.default_rng(3), 2, 40, 50, bright=False)
    data, _ = fd.write_workunit(layers, compressed=False)
    path = tmp_path / "plain.fits"
    path.write_bytes(data)
    check_against_oracle(fi.load_workunit(str(path)), data)


@pytest.mark.parametrize("bitpix", [8, 16, 32, -32, -64])
def test_image_decode_every_bitpix(fi, bitpix):
    import torch

    from kbmod_amd import fits_ingest

    lib = fits_ingest._lib()
    rng = np.random.default_rng(abs(bitpix))
    n = 5000
    dt = {8: ">u1", 16: ">i2", 32: ">i4", -32: ">f4", -64: ">f8"}[bitpix]
    if bitpix > 0:
        info = np.iinfo(np.dtype(dt).newbyteorder("="))
        vals = rng.integers(info.min, info.max, size=n, endpoint=True).astype(dt)
    else:
        vals = (rng.normal(0, 1e3, n)).astype(dt)
        vals[7] = np.nan
    for bscale, bzero in ((1.0, 0.0), (1.0, -128.0), (0.25, 1.0e3)):
        raw = torch.from_numpy(np.frombuffer(vals.tobytes(), dtype=np.uint8).copy()).cuda()
        out = torch.empty(n, dtype=torch.float32, device="cuda")
        rc = lib.kb_fits_decode_image(raw.data_ptr(), bitpix, bscale, bzero, n, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.kb_last_error()
        hdr = {"BITPIX": bitpix, "NAXIS": 1, "NAXIS1": n, "BSCALE": bscale, "BZERO": bzero}
        exp = fd.image_data(vals.tobytes(), (hdr, 0, 0))
        assert np.array_equal(out.cpu().numpy(), exp, equal_nan=True)
    # the reference's plain data file: the three layers as they lie in it
    buf = open(SMALL, "rb").read()
    hdus = fd.parse(buf)
    raw = torch.from_numpy(np.frombuffer(buf, dtype=np.uint8).copy()).cuda()
    for h in hdus[1:]:
        out = torch.empty(64 * 64, dtype=torch.float32, device="cuda")
        assert lib.kb_fits_decode_image(raw.data_ptr() + h[1], -32, 1.0, 0.0, 64 * 64, out.data_ptr(), None) == 0
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().reshape(64, 64), fd.image_data(buf, h))


@pytest.mark.parametrize("bytepix", [1, 2])
def test_integer_tiles_through_the_c_abi(fi, bytepix):
    import torch

    lib = fi._lib()
    rng = np.random.default_rng(bytepix)
    H, W = 70, 75
    lo, hi = (0, 255) if bytepix == 1 else (-32768, 32767)  # 8-bit FITS pixels are unsigned
    img = rng.integers(lo, hi, size=(H, W), endpoint=True)
    img[3] = 5
    heap, tiles = bytearray(), np.zeros(H, dtype=fi.TILE_DTYPE)
    for r in range(H):
        enc = fd.rice_encode(img[r], 32, bytepix, "raw" if r % 3 == 0 else None)
        tiles[r] = (len(heap), r * W, 2.0, -1.0, len(enc), fi.TILE_RICE)
        heap += enc
    heap_dev = torch.from_numpy(np.frombuffer(bytes(heap), dtype=np.uint8).copy()).cuda()
    tiles_dev = torch.from_numpy(tiles.view(np.uint8).reshape(-1).copy()).cuda()
    out = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    status = torch.zeros(2, dtype=torch.int32, device="cuda")
    rc = lib.kb_fits_decode_rice(heap_dev.data_ptr(), len(heap), tiles_dev.data_ptr(), H, W, 32, bytepix, 0, 1, 5, out.data_ptr(),
                                 status.data_ptr(), None)
    assert rc == 0, lib.kb_last_error()
    torch.cuda.synchronize()
    assert status.cpu().tolist() == [0, 0]
    exp = (img.astype(np.float64) * 2.0 - 1.0).astype(np.float32)
    exp[img == 5] = np.nan  # BLANK
    assert np.array_equal(out.cpu().numpy().reshape(H, W), exp, equal_nan=True)


def test_gzip_fallback_tiles_and_truncated_streams(fi, tmp_path):
    rng = np.random.default_rng(8)
    img = rng.normal(0, 2, (70, 40)).astype(np.float32)
    sci_bytes, expect = fd.write_compressed_hdu("SCI_0", img, gzip_rows=(2, 65), extra=[("MJD", 1.0)])
    var_bytes, _ = fd.write_compressed_hdu("VAR_0", np.full((70, 40), 4.0, np.float32))
    head = fd._header_bytes([fd._card("SIMPLE", True), fd._card("BITPIX", 8), fd._card("NAXIS", 0), fd._card("NUMIMG", 1)])
    data = head + sci_bytes + var_bytes
    path = tmp_path / "gz.fits"
    path.write_bytes(data)
    wu = fi.load_workunit(str(path))
    assert wu.stats["gzip_tiles"] == 2
    assert np.array_equal(wu.sci.cpu().numpy()[0], expect) and np.array_equal(expect[2], img[2])
    # a stream cut short: the table says fewer bytes than the pixels need
    hdus = fd.parse(data)
    sci = fd.find(hdus, "SCI_0")
    bad = bytearray(data)
    row0 = sci[1] + 10 * 32
    n = int.from_bytes(bad[row0:row0 + 4], "big")
    bad[row0:row0 + 4] = (n // 2).to_bytes(4, "big")
    path.write_bytes(bytes(bad))
    with pytest.raises(ValueError, match="end before their pixels"):
        fi.load_workunit(str(path))


def test_sharded_workunit(fi, tmp_path):
    layers = _layers(np.random.default_rng(11), 3, 33, 70)
    whole, _ = fd.write_workunit(layers)
    primary = fd._header_bytes([fd._card("SIMPLE", True), fd._card("BITPIX", 8), fd._card("NAXIS", 0), fd._card("NUMIMG", 3)])
    (tmp_path / "wu.fits").write_bytes(primary)
    for i, lay in enumerate(layers):
        mjd, sci, var, mask, psf = lay
        s, _ = fd.write_compressed_hdu(f"SCI_{i}", sci, extra=[("MJD", float(mjd))], blank=-2147483647)
        v, _ = fd.write_compressed_hdu(f"VAR_{i}", var, blank=-2147483647)
        shard = primary + s + v + fd.write_image_hdu(f"MSK_{i}", mask) + fd.write_image_hdu(f"PSF_{i}", psf)
        (tmp_path / f"{i}_wu.fits").write_bytes(shard)
    wu = fi.load_sharded_workunit("wu.fits", str(tmp_path))
    check_against_oracle(wu, whole)
    os.remove(tmp_path / "1_wu.fits")
    with pytest.raises(ValueError, match="No shard provided for index 1"):
        fi.load_sharded_workunit("wu.fits", str(tmp_path))


def test_search_from_a_fits_file_equals_search_from_the_decoded_layers(fi, kb, tmp_path):
    from kbmod_amd import fake_data as fdata

    from .util import make_stack

    T, H, W = 12, 48, 80
    stack = make_stack(T, H, W, seed=21, objects=[(20, 15, 12.0, 6.0, 400.0)], mask_fraction=0.01)
    layers = [(58000.0 + stack.times[t], stack.sci[t], stack.var[t], None, stack.psfs[t]) for t in range(T)]
    data, _ = fd.write_workunit(layers)
    path = tmp_path / "search.fits"
    path.write_bytes(data)
    wu = fi.load_workunit(str(path))
    search = wu.stack_search()
    # the same layers as the oracle decodes them, through the host constructor
    dec = fd.read_workunit_layers(data)
    times = np.asarray([d[0] for d in dec])
    ref = kb.StackSearch([d[1] for d in dec], [d[2] for d in dec], [d[4] for d in dec], list(times - times[0]), -1)
    a = np.asarray(search.get_psi_phi_array().encoded_array())
    b = np.asarray(ref.get_psi_phi_array().encoded_array())
    assert a.tobytes() == b.tobytes()
    vx, vy = fdata.kbmod_v1_candidates(6, 5.0, 20.0, 6, 0.0, 1.0)
    cands = [kb.Trajectory(vx=float(x), vy=float(y)) for x, y in zip(vx, vy)]
    for s in (search, ref):
        s.set_min_obs(6)
        s.search_all(cands, True)
    r1, r2 = search.results_to_numpy(), ref.results_to_numpy()
    assert r1.tobytes() == r2.tobytes() and len(r1) > 0
    best = search.get_results(0, 1)[0]
    assert abs(best.x - 20) <= 2 and abs(best.y - 15) <= 2 and best.lh > 20  # the injected mover


def test_many_tiles_property(fi):
    """16 384 tiles of 4096 pixels (a 4096-wide image stack's worth of rows) decode to the rows they were made from: the
    table names 48 distinct streams over and over."""
    import torch

    lib = fi._lib()
    rng = np.random.default_rng(77)
    W, distinct, n_tiles = 4096, 48, 16384
    ints = np.floor(rng.normal(0, 200, (distinct, W)) + 0.5).astype(np.int64)
    ints[5, 100:200] = -2147483647
    heap, place = bytearray(), []
    for r in range(distinct):
        enc = fd.rice_encode(ints[r])
        place.append((len(heap), len(enc)))
        heap += enc
    which = rng.integers(0, distinct, size=n_tiles)
    tiles = np.zeros(n_tiles, dtype=fi.TILE_DTYPE)
    tiles["offset"] = [place[w][0] for w in which]
    tiles["nbytes"] = [place[w][1] for w in which]
    tiles["out_index"] = np.arange(n_tiles, dtype=np.uint64) * W
    tiles["zscale"], tiles["zzero"], tiles["mode"] = 0.01, -3.5, fi.TILE_RICE
    heap_dev = torch.from_numpy(np.frombuffer(bytes(heap), dtype=np.uint8).copy()).cuda()
    tiles_dev = torch.from_numpy(tiles.view(np.uint8).reshape(-1).copy()).cuda()
    out = torch.empty(n_tiles * W, dtype=torch.float32, device="cuda")
    status = torch.zeros(2, dtype=torch.int32, device="cuda")
    rc = lib.kb_fits_decode_rice(heap_dev.data_ptr(), len(heap), tiles_dev.data_ptr(), n_tiles, W, 32, 4, 1, 1, -2147483647,
                                 out.data_ptr(), status.data_ptr(), None)
    assert rc == 0, lib.kb_last_error()
    torch.cuda.synchronize()
    assert status.cpu().tolist() == [0, 0]
    exp = (ints.astype(np.float64) * 0.01 - 3.5).astype(np.float32)
    exp[ints == -2147483647] = np.nan
    got = out.cpu().numpy().reshape(n_tiles, W)
    assert np.array_equal(got, exp[which], equal_nan=True)


def test_stamps_from_the_ingested_layers_without_a_copy(fi, kb, tmp_path):
    """file -> HBM -> coadds / stamps: the DeviceStack over the decoded tensors gives what a DeviceStack uploaded from the
    oracle-decoded host layers gives (kb_coadd_stamps / kb_extract_stamps read the same pixels)."""
    from kbmod_amd import stamp_utils as su

    rng = np.random.default_rng(31)
    layers = _layers(rng, 6, 40, 60, bright=False)
    data, _ = fd.write_workunit(layers)
    path = tmp_path / "stamps.fits"
    path.write_bytes(data)
    wu = fi.load_workunit(str(path))
    on_dev = wu.device_stack()
    dec = fd.read_workunit_layers(data)
    times = np.asarray([d[0] for d in dec])
    ref = su.DeviceStack(np.stack([d[1] for d in dec]), np.stack([d[2] for d in dec]), zeroed_times=times - times[0], times=times)
    assert (on_dev.num_times, on_dev.height, on_dev.width) == (6, 40, 60)
    n = 25
    x0, y0 = rng.integers(0, 60, n), rng.integers(0, 40, n)
    vx, vy = rng.uniform(-8, 8, n), rng.uniform(-6, 6, n)
    xs = su.predict_pixel_locations(on_dev.zeroed_times, x0, vx)
    ys = su.predict_pixel_locations(on_dev.zeroed_times, y0, vy)
    a = on_dev.coadds(xs, ys, 3, ["sum", "mean", "median", "weighted"])
    b = ref.coadds(xs, ys, 3, ["sum", "mean", "median", "weighted"])
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    assert np.array_equal(on_dev.all_stamps(xs, ys, 2), ref.all_stamps(xs, ys, 2), equal_nan=True)
    del wu  # the stack keeps the tensors alive
    assert np.array_equal(on_dev.all_stamps(xs, ys, 2), ref.all_stamps(xs, ys, 2), equal_nan=True)


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_written_workunits(fi, tmp_path, seed):
    """Random shapes, noise scales from 0.05 to 10^4 quanta per pixel step (every split code up to verbatim blocks), NaN and
    mask fractions, quantisation steps, GZIP fallback rows, compressed or plain layers -- product == oracle, bit for bit."""
    rng = np.random.default_rng(1000 + seed)
    T, H, W = int(rng.integers(1, 4)), int(rng.integers(1, 71)), int(rng.integers(1, 201))
    quantum = float(rng.choice([0.01, 0.5, 1.0e-3]))
    sigma = float(rng.choice([0.0005, 0.05, 2.0, 300.0, 1.0e4]))
    compressed = bool(rng.random() < 0.8)
    layers = []
    for t in range(T):
        sci = rng.normal(0, sigma, (H, W)).astype(np.float32)
        sci[rng.random((H, W)) < rng.choice([0.0, 0.02, 0.3])] = np.nan
        var = np.abs(rng.normal(4, sigma, (H, W))).astype(np.float32)
        mask = (rng.random((H, W)) < rng.choice([0.0, 0.05])).astype(np.int8) if rng.random() < 0.7 else None
        psf = np.full((3, 3), 1 / 9, np.float32) if rng.random() < 0.7 else None
        layers.append((50000.0 + 3 * t + rng.random(), sci, var, mask, psf))
    if compressed and H > 2 and rng.random() < 0.5:
        # a file with GZIP fallback rows in its science layers
        out = bytearray(fd._header_bytes([fd._card("SIMPLE", True), fd._card("BITPIX", 8), fd._card("NAXIS", 0), fd._card("NUMIMG", T)]))
        for i, (mjd, sci, var, mask, psf) in enumerate(layers):
            rows = tuple(int(r) for r in rng.choice(H, size=min(3, H), replace=False))
            s, _ = fd.write_compressed_hdu(f"SCI_{i}", sci, quantum, [("MJD", float(mjd))], -2147483647, None, rows)
            v, _ = fd.write_compressed_hdu(f"VAR_{i}", var, quantum, (), -2147483647)
            out += s + v
            if mask is not None:
                out += fd.write_image_hdu(f"MSK_{i}", mask)
            if psf is not None:
                out += fd.write_image_hdu(f"PSF_{i}", psf)
        data = bytes(out)
    else:
        data, _ = fd.write_workunit(layers, compressed=compressed, quantum=quantum)
    path = tmp_path / "fuzz.fits"
    path.write_bytes(data)
    wu = fi.load_workunit(str(path))
    check_against_oracle(wu, data)
    assert tuple(wu.sci.shape) == (T, H, W)


def test_corrupt_split_code_is_reported(fi):
    """A RICE block whose 5-bit split code exceeds FSMAX + 1 = 26 does not exist in a valid stream: the tile is flagged through
    status_dev (the host raises), not decoded into garbage."""
    import torch

    lib = fi._lib()
    W = 32
    good = fd.rice_encode(np.arange(W), 32, 4, None)
    bad = bytearray(good)
    bad[4] |= 0xF8  # the five bits behind the first pixel: split code 31
    heap = bytes(good) + bytes(bad) + bytes(8)
    tiles = np.zeros(2, dtype=fi.TILE_DTYPE)
    tiles[0] = (0, 0, 1.0, 0.0, len(good), fi.TILE_RICE)
    tiles[1] = (len(good), W, 1.0, 0.0, len(bad), fi.TILE_RICE)
    heap_dev = torch.from_numpy(np.frombuffer(heap, dtype=np.uint8).copy()).cuda()
    tiles_dev = torch.from_numpy(tiles.view(np.uint8).reshape(-1).copy()).cuda()
    out = torch.zeros(2 * W, dtype=torch.float32, device="cuda")
    status = torch.zeros(2, dtype=torch.int32, device="cuda")
    rc = lib.kb_fits_decode_rice(heap_dev.data_ptr(), len(heap), tiles_dev.data_ptr(), 2, W, 32, 4, 0, 0, 0, out.data_ptr(),
                                 status.data_ptr(), None)
    assert rc == 0, lib.kb_last_error()
    torch.cuda.synchronize()
    assert status.cpu().tolist() == [1, 2]
    assert np.array_equal(out.cpu().numpy()[:W], np.arange(W, dtype=np.float32))


def test_heap_element_type_of_the_descriptor_column(fi, tmp_path):
    """COMPRESSED_DATA descriptors count heap ELEMENTS: 1PB (what cfitsio / astropy write) is bytes; anything but B / I / J
    is refused by name instead of being read as bytes."""
    img = np.random.default_rng(3).normal(0, 2, (20, 40)).astype(np.float32)
    sci, _ = fd.write_compressed_hdu("SCI_0", img, extra=[("MJD", 1.0)])
    var, _ = fd.write_compressed_hdu("VAR_0", np.full((20, 40), 4.0, np.float32))
    head = fd._header_bytes([fd._card("SIMPLE", True), fd._card("BITPIX", 8), fd._card("NAXIS", 0), fd._card("NUMIMG", 1)])
    data = head + sci + var
    assert data.count(b"'1PB") >= 2
    path = tmp_path / "pe.fits"
    path.write_bytes(data.replace(b"'1PB", b"'1PE", 1))
    with pytest.raises(ValueError, match="column of 'E' elements"):
        fi.load_workunit(str(path))
