"""The FITS ingest oracle (oracle/fits_decode.py) against the reference's own data files, and the product's host-side parse
(kbmod_amd.fits_ingest: headers, table rows -> tile table) against the oracle.  No device work here."""

import os
import struct

import numpy as np
import pytest

from oracle import fits_decode as fd

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TILED = os.path.join(GOLDEN, "shifted_wcs_diff_dimms_tiled.fits")  # /root/reference/tests/data/ (test_reprojection.py:18)
SMALL = os.path.join(GOLDEN, "small_000000.fits")                  # /root/reference/data/small/000000.fits


def test_reference_workunit_file_decodes_to_what_its_writer_put_in():
    buf = open(TILED, "rb").read()
    hdus = fd.parse(buf)
    assert int(hdus[0][0]["NUMIMG"]) == 4
    # every RICE tile ends exactly on its last byte (compressed_image_data asserts `used == n` per tile)
    layers = fd.read_workunit_layers(buf)
    assert [l[0] for l in layers] == [60414.0, 60415.0, 60416.0, 60416.0]
    for mjd, sci, var, mask, psf in layers:
        assert sci.shape == var.shape == mask.shape == (50, 60) and psf.shape == (3, 3)
        # the reference's fake data: variance = noise level squared = 4 everywhere (fake_data_creator.py), exactly
        assert np.all(var == np.float32(4.0))
        assert not mask.any() and np.isfinite(sci).all()
        # N(0, 2^2) noise plus a few bright sources, on the 0.01 grid of quantize_level=-0.01 (work_unit.py:1111)
        assert 1.8 < np.std(sci[sci < 10]) < 2.3 and abs(np.median(sci)) < 0.2
        assert abs(psf.sum() - 1.0) < 1e-2 and psf[1, 1] == psf.max()  # a Gaussian kernel
        # the brightest source is an inserted object: flux x the file's own PSF_i kernel (a plain HDU) + that noise -- the
        # RICE-decoded science pixels and the trivially decoded kernel agree on it
        y, x = np.unravel_index(np.argmax(sci), sci.shape)
        if 1 <= y < 49 and 1 <= x < 59:
            patch = sci[y - 1:y + 2, x - 1:x + 2].astype(np.float64)
            flux = float((patch * psf).sum() / (psf.astype(np.float64) ** 2).sum())
            assert flux > 100 and np.abs(patch - flux * psf).max() < 8.0, (flux, patch - flux * psf)
    # What the reference's own test knows about this file (tests/test_reprojection.py:118-131: the WorkUnit loaded through
    # astropy, image 0 reprojected onto its own WCS -- the "no-op case" --, "make sure the PSF for the object hasn't been warped":
    # an object sits at [5][53] of the first science layer; the variance is 4.0).  The reprojection rescales the peak (115.5
    # there), so only WHERE the object is carries over: the decoded layer has its brightest pixel exactly there.
    sci0 = layers[0][1]
    assert np.unravel_index(np.argmax(sci0), sci0.shape) == (5, 53) and sci0[5, 53] > 100.0
    assert layers[2][2][25, 9] == np.float32(4.0)
    sci_hdu = fd.find(hdus, "SCI_0")
    cols = fd._columns(sci_hdu[0])
    for r in range(50):
        row = buf[sci_hdu[1] + 32 * r:sci_hdu[1] + 32 * (r + 1)]
        zscale = struct.unpack(">d", row[cols["ZSCALE"][0]:cols["ZSCALE"][0] + 8])[0]
        assert abs(zscale - 0.01) < 1e-9


def test_reference_plain_image_file():
    buf = open(SMALL, "rb").read()
    hdus = fd.parse(buf)
    assert [h[0]["BITPIX"] for h in hdus] == [16, -32, -32, -32]
    sci, msk, var = (fd.image_data(buf, h) for h in hdus[1:])
    assert sci.shape == (64, 64) and np.all(var == np.float32(4.0)) and not msk.any()
    assert np.array_equal(sci, np.frombuffer(buf, ">f4", 64 * 64, hdus[1][1]).astype(np.float32).reshape(64, 64))
    assert 1.9 < sci.std() < 2.3


@pytest.mark.parametrize("bytepix,lo,hi", [(4, -2**31, 2**31 - 1), (4, -5000, 5000), (2, -2**15, 2**15 - 1), (1, 0, 255)])
@pytest.mark.parametrize("force", [None, "raw", 0, 1, 7])
def test_rice_round_trip(bytepix, lo, hi, force):
    rng = np.random.default_rng(bytepix * 100 + (hash(str(force)) & 15))
    for n in (1, 31, 32, 33, 100):
        v = rng.integers(lo, hi, size=n, endpoint=True)
        if isinstance(force, int):  # differences of the size that split suits (a forced small split of large ones is all unary)
            if force == 7 and bytepix == 1:
                continue  # beyond FSMAX = 6 of 8-bit pixels
            step = 1 << force
            v = np.clip(np.cumsum(rng.integers(-step, step, size=n, endpoint=True)) + (lo + hi) // 2, lo, hi)
        enc = fd.rice_encode(v, 32, bytepix, force)
        dec, used = fd.rice_decode(enc, n, 32, bytepix)
        assert used == len(enc) and np.array_equal(dec, v)


def test_constant_tile_is_six_bytes_like_the_reference_file():
    # VAR_0 of the reference file: 60 pixels of 4.0 -> 4 bytes + two 5-bit zero codes = 6 bytes (PCOUNT 300 for 50 rows)
    assert len(fd.rice_encode(np.zeros(60, dtype=np.int64))) == 6


def _layers(rng, T, H, W):
    out = []
    for t in range(T):
        sci = rng.normal(0, 2, (H, W)).astype(np.float32)
        sci[rng.random((H, W)) < 0.01] = np.nan
        var = np.full((H, W), 4.0, np.float32)
        mask = (rng.random((H, W)) < 0.02).astype(np.int8)
        out.append((59000.5 + t, sci, var, mask, np.full((3, 3), 1 / 9, np.float32)))
    return out


@pytest.mark.parametrize("compressed", [True, False])
def test_writer_reader_round_trip(compressed):
    layers = _layers(np.random.default_rng(5), 3, 20, 45)
    data, expect = fd.write_workunit(layers, compressed=compressed)
    got = fd.read_workunit_layers(data)
    for (mjd, sci, var, mask, psf), (es, ev), lay in zip(got, expect, layers):
        es, ev = es.copy(), ev.copy()
        es[lay[3] > 0] = np.nan
        ev[lay[3] > 0] = np.nan
        assert mjd == lay[0] and np.array_equal(sci, es, equal_nan=True) and np.array_equal(var, ev, equal_nan=True)
        assert np.array_equal(mask, lay[3].astype(np.float32)) and np.array_equal(psf, lay[4])
        assert np.array_equal(np.isnan(sci), np.isnan(lay[1]) | (lay[3] > 0))
        if compressed:
            assert np.nanmax(np.abs(sci - lay[1])) <= 0.005001


# ---- the product's host half against the oracle -------------------------------------------------------------------
def test_product_header_parse_matches_oracle():
    from kbmod_amd import fits_ingest as fi

    for path in (TILED, SMALL):
        buf = open(path, "rb").read()
        mine, ref = fi.parse_fits(buf), fd.parse(buf)
        assert len(mine) == len(ref)
        for m, r in zip(mine, ref):
            assert m.data_offset == r[1] and m.data_size == r[2]
            assert {k: v for k, v in m.header.items()} == r[0]


def test_product_tile_table_matches_the_file():
    from kbmod_amd import fits_ingest as fi

    buf = open(TILED, "rb").read()
    plan = fi.workunit_plan(buf)
    assert plan["shape"] == (50, 60) and list(plan["times"]) == [60414.0, 60415.0, 60416.0, 60416.0]
    for p, (_, _, _, _, psf) in zip(plan["psfs"], fd.read_workunit_layers(buf)):
        assert np.array_equal(p, psf)
    hdu = plan["images"][1]["sci"]
    lay = fi.CompressedLayout(hdu)
    assert (lay.blocksize, lay.bytepix, lay.quantized, lay.width, lay.height) == (32, 4, True, 60, 50)
    tiles, patches = lay.tiles(buf, hdu, 3000)
    assert not patches and len(tiles) == 50 and np.all(tiles["mode"] == fi.TILE_RICE)
    assert list(tiles["out_index"][:3]) == [3000, 3060, 3120]
    # every stream decodes with the oracle's decoder from exactly the bytes the table names
    for r in (0, 17, 49):
        t = tiles[r]
        ints, used = fd.rice_decode(buf[int(t["offset"]):int(t["offset"]) + int(t["nbytes"])], 60)
        assert used == int(t["nbytes"])
        vals = (ints.astype(np.float64) * t["zscale"] + t["zzero"]).astype(np.float32)
        assert np.array_equal(vals, fd.hdu_data(buf, fd.find(fd.parse(buf), "SCI_1"))[r])


def test_product_refuses_what_it_does_not_read():
    from kbmod_amd import fits_ingest as fi

    layers = _layers(np.random.default_rng(6), 1, 8, 40)
    data, _ = fd.write_workunit(layers)
    for old, new, what in ((b"'RICE_1  '", b"'GZIP_1  '", "ZCMPTYPE"), (b"'NO_DITHER'", b"'SUBTRACTIVE_DITHER_1'"[:11], "ZQUANTIZ")):
        bad = data.replace(old, new.ljust(len(old))[:len(old)])
        plan = fi.workunit_plan(bad)
        with pytest.raises(ValueError, match=what):
            fi.CompressedLayout(plan["images"][0]["sci"])
    with pytest.raises(ValueError, match="NUMIMG"):
        fi.workunit_plan(open(SMALL, "rb").read())
    with pytest.raises(ValueError, match="not a FITS file"):
        fi.parse_fits(b"x" * 5760)
    with pytest.raises(ValueError, match="not found"):
        fi.load_workunit("/nonexistent/file.fits")


def test_gzip_fallback_rows_are_found_by_the_product():
    from kbmod_amd import fits_ingest as fi

    rng = np.random.default_rng(8)
    img = rng.normal(0, 2, (6, 40)).astype(np.float32)
    hdu_bytes, expect = fd.write_compressed_hdu("SCI_0", img, gzip_rows=(2, 5), extra=[("MJD", 1.0)])
    head = fd._header_bytes([fd._card("SIMPLE", True), fd._card("BITPIX", 8), fd._card("NAXIS", 0), fd._card("NUMIMG", 1)])
    var_bytes, _ = fd.write_compressed_hdu("VAR_0", np.full((6, 40), 4.0, np.float32))
    data = head + hdu_bytes + var_bytes
    assert np.array_equal(fd.read_workunit_layers(data)[0][1], expect)
    plan = fi.workunit_plan(data)
    hdu = plan["images"][0]["sci"]
    tiles, patches = fi.CompressedLayout(hdu).tiles(data, hdu, 0)
    assert [p[0] for p in patches] == [2, 5] and list(tiles["mode"]) == [1, 1, 0, 1, 1, 0]
    assert np.array_equal(patches[0][1], img[2]) and np.array_equal(patches[1][1], img[5])
