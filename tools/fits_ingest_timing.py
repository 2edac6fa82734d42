#!/usr/bin/env python3
"""Timing of the FITS ingest on one GPU.

1. kb_fits_decode_rice alone, HIP events: T x H x W x 2 layers' worth of row tiles whose streams are copies (at distinct
   addresses) of a few dozen encoded rows -- N(0, 2^2) noise on the reference's 0.01 grid, 1.2 bytes per pixel.
2. kb_fits_decode_image alone (BITPIX -32).
3. load_workunit end to end on a file written by the oracle's writer (file read + one DMA + decode + mask pass).
"""
import argparse
import functools
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--file-frames", type=int, default=16)
    ap.add_argument("--file-size", type=int, default=1024)
    args = ap.parse_args()
    import torch

    from kbmod_amd import fits_ingest as fi
    from oracle import fits_decode as fd  # the writer only: test infrastructure, nothing of it is timed

    lib = fi._lib()
    rng = np.random.default_rng(1)
    T, H, W = args.frames, args.size, args.size
    distinct = 64
    ints = np.floor(rng.normal(0, 200, (distinct, W)) + 0.5).astype(np.int64)
    streams = [fd.rice_encode(ints[r]) for r in range(distinct)]
    pitch = -(-max(len(s) for s in streams) // 16) * 16
    block = np.zeros((distinct, pitch), dtype=np.uint8)
    for r, s in enumerate(streams):
        block[r, :len(s)] = np.frombuffer(s, dtype=np.uint8)
    n_tiles = 2 * T * H
    reps = -(-n_tiles // distinct)
    heap_dev = torch.from_numpy(block).cuda().repeat(reps, 1).contiguous()
    tiles = np.zeros(n_tiles, dtype=fi.TILE_DTYPE)
    idx = np.arange(n_tiles)
    tiles["offset"] = idx.astype(np.uint64) * pitch
    tiles["nbytes"] = np.asarray([len(s) for s in streams], dtype=np.uint32)[idx % distinct]
    tiles["out_index"] = idx.astype(np.uint64) * W
    tiles["zscale"], tiles["zzero"], tiles["mode"] = 0.01, -4.0, fi.TILE_RICE
    tiles_dev = torch.from_numpy(tiles.view(np.uint8).reshape(-1).copy()).cuda()
    out = torch.empty(n_tiles * W, dtype=torch.float32, device="cuda")
    status = torch.zeros(2, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    comp_bytes = int(tiles["nbytes"].astype(np.int64).sum())

    def run():
        rc = lib.kb_fits_decode_rice(heap_dev.data_ptr(), heap_dev.numel(), tiles_dev.data_ptr(), n_tiles, W, 32, 4, 1, 1,
                                     -2147483647, out.data_ptr(), status.data_ptr(), stream)
        assert rc == 0, lib.kb_last_error()

    run()
    torch.cuda.synchronize()
    assert status.cpu().tolist() == [0, 0]
    exp = (ints.astype(np.float64) * 0.01 - 4.0).astype(np.float32)
    got = out[:distinct * W].cpu().numpy().reshape(distinct, W)
    assert np.array_equal(got, exp)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    out_bytes = n_tiles * W * 4
    print(f"kb_fits_decode_rice  {T} x {H} x {W} x 2 layers: {n_tiles} tiles, {comp_bytes / 1e9:.2f} GB compressed "
          f"({comp_bytes / (n_tiles * W):.2f} B/pixel) -> {out_bytes / 1e9:.2f} GB float32: {ms:.2f} ms, "
          f"{(comp_bytes + out_bytes) / ms / 1e6:.0f} GB/s in + out, {n_tiles * W / ms / 1e6:.1f} Gpixel/s")

    raw = torch.randint(0, 255, (min(out_bytes, 4 << 30),), dtype=torch.uint8, device="cuda")
    n = raw.numel() // 4
    e0.record()
    for _ in range(5):
        assert lib.kb_fits_decode_image(raw.data_ptr(), -32, 1.0, 0.0, n, out.data_ptr(), stream) == 0
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"kb_fits_decode_image BITPIX -32, {n * 4 / 1e9:.2f} GB: {ms:.3f} ms, {2 * n * 4 / ms / 1e6:.0f} GB/s in + out")
    del raw, out, heap_dev

    # ---- a file end to end ----
    T, H, W = args.file_frames, args.file_size, args.file_size
    pool = [rng.normal(0, 2, W).astype(np.float32) for _ in range(96)]
    enc = functools.lru_cache(maxsize=None)(lambda key, bs, bp, fs: fd_rice(np.frombuffer(key, dtype=np.int64), bs, bp, fs))
    fd_rice = fd.rice_encode
    fd.rice_encode = lambda v, bs=32, bp=4, fs=None: enc(np.asarray(v, dtype=np.int64).tobytes(), bs, bp, fs)
    layers = []
    for t in range(T):
        sci = np.stack([pool[k] for k in rng.integers(0, len(pool), size=H)])
        layers.append((60000.0 + t, sci, np.full((H, W), 4.0, np.float32), (rng.random((H, W)) < 0.01).astype(np.int8),
                       np.full((5, 5), 0.04, np.float32)))
    data, _ = fd.write_workunit(layers)
    fd.rice_encode = fd_rice
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "wu.fits")
        with open(path, "wb") as fh:
            fh.write(data)
        fi.load_workunit(path)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            wu = fi.load_workunit(path)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        raw_bytes = 2 * T * H * W * 4
        print(f"load_workunit {T} x {H} x {W}: file {len(data) / 1e6:.1f} MB -> {raw_bytes / 1e6:.1f} MB of float32 layers on the "
              f"device in {best * 1e3:.1f} ms ({raw_bytes / best / 1e9:.2f} GB/s of decoded layers; the same layers as "
              f"float32 over PCIe at 50 GB/s: {raw_bytes / 50e9 * 1e3:.1f} ms + host decompression)")
        t0 = time.perf_counter()
        search = wu.stack_search()
        print(f"stack_search() from the device-resident layers: {(time.perf_counter() - t0) * 1e3:.1f} ms")


if __name__ == "__main__":
    main()
