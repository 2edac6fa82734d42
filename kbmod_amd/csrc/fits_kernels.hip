// FITS ingest on the device: the bytes of a WorkUnit file -- SCI_i / VAR_i as tiled-compressed RICE_1
// tables, MSK_i / PSF_i as plain image HDUs (work_unit.py:1066-1147 writes them, work_unit.py:489-608 / 1149-1200 reads
// them back through astropy) -- go to HBM AS THEY LIE IN THE FILE and are decoded there into the [T][H][W] float32
// science / variance stacks the psi/phi builder reads.  What crosses PCIe is the compressed file (a quarter of the
// float32 layers at the reference's quantisation), and no decoded layer ever exists in host memory.
//
//   kb_fits_rice_decode_kernel   one lane per tile (= image row), one RICE block (32 pixels) per lane and round into the
//                                wave's LDS patch, then the wave writes the 64 x 32 patch out as 128-byte row segments:
//                                the bit-serial part runs 64 tiles wide per wave, the HBM side stays coalesced;
//   kb_fits_image_decode_kernel  big-endian BITPIX 8 / 16 / 32 / -32 / -64 -> float32 with BSCALE / BZERO;
//   kb_fits_apply_mask_kernel    sci[mask > 0] = var[mask > 0] = NaN (work_unit.py:1187-1190).
//
// Bit-exact against oracle/fits_decode.py (integers exactly; value = (float)((double) integer * ZSCALE + ZZERO), one
// multiply and one add, separately rounded -- the build keeps -ffp-contract=off).
#include <algorithm>
#include <cstdint>
#include <string>

#include "kb_common.h"

namespace kb {
namespace {

struct RiceArgs {
    const uint8_t* heap;
    uint64_t heap_bytes;
    const kb_fits_tile* tiles;
    float* out;
    int32_t* status;  // [0] tiles whose stream ran past its end, [1] first such tile + 1
    int32_t n_tiles;
    int32_t tile_len;
    int32_t blocksize;
    int32_t quantized;
    int32_t has_blank;
    int32_t blank;
};

// MSB-first bit reader over a tile's stream: a 64-bit window refilled 32 bits at a time out of a two-word ready queue; the
// eight bytes behind the queue are requested from memory the moment the queue is filled, so that a load has the decoding
// of 64 bits (about six pixels) to land in instead of stalling the lane's serial chain.  Loads may run past the tile's
// last byte into its neighbour's (never past the heap: those bytes read as zero); a well-formed stream does not consume
// them, and a stream that does is flagged by its bit count.
struct BitReader {
    const uint8_t* base;      // where offsets count from: the tile's first byte, or the heap's last eight when the tile starts in them
    uint32_t pos;             // offset of the next eight bytes to request
    uint32_t lim;             // largest offset at which eight bytes may be read (the heap ends eight bytes behind it)
    uint64_t buf;             // the window, valid bits at the top
    uint64_t ready;           // up to two big-endian words, the next one in the high half
    uint64_t pending;         // the eight bytes behind them, as loaded
    uint32_t pending_over;    // how many of those lie past the heap's end (the load was moved back by as many)
    int avail;                // valid bits in the window
    int n_ready;
    uint32_t taken;           // bytes moved into the window so far
    __device__ __forceinline__ void request8() {
        // Branch-free, 32-bit arithmetic, and the loaded value is not touched here (any arithmetic on it -- or a branch
        // that makes the compiler merge it into another register -- is a wait for the load this queue exists to hide):
        // within eight bytes of the heap's end the load is moved back to end there, and the shift that makes bytes past
        // the end read as zero is applied when the word is promoted.  Global loads need no alignment on gfx950.
        const uint32_t at = min(pos, lim);
        pending_over = min(pos - at, 8u);
        pending = *reinterpret_cast<const uint64_t*>(base + at);
        pos += 8u;
    }
    __device__ __forceinline__ void start(const uint8_t* first, const uint8_t* end_of_heap) {
        // (the heap holds at least eight bytes: checked by the launcher)
        const uint8_t* last8 = end_of_heap - 8;
        base = first < last8 ? first : last8;
        pos = (uint32_t)(first - base);
        const uint64_t room = (uint64_t)(last8 - base);
        lim = room < 0xffffff00ull ? (uint32_t)room : 0xffffff00u;
        buf = 0;
        ready = 0;
        avail = 0;
        n_ready = 0;
        taken = 0;
        request8();
    }
    __device__ __forceinline__ void refill() {  // afterwards at least 33 bits are available
        if (avail <= 32) {
            if (n_ready == 0) {
                ready = __builtin_bswap64(pending_over >= 8u ? 0ull : (pending >> (8u * pending_over)));
                n_ready = 2;
                request8();
            }
            buf |= (ready >> 32) << (32 - avail);
            ready <<= 32;
            --n_ready;
            avail += 32;
            taken += 4;
        }
    }
    __device__ __forceinline__ uint32_t take(int n) {  // 0 <= n <= 32, after refill()
        const uint32_t v = n ? (uint32_t)(buf >> (64 - n)) : 0u;
        buf = n ? (buf << n) : buf;
        avail -= n;
        return v;
    }
    __device__ __forceinline__ uint64_t bits_used() const { return (uint64_t)taken * 8u - (uint64_t)avail; }
};

#ifndef KB_FITS_PATCH_COLS
#define KB_FITS_PATCH_COLS 32
#endif
constexpr int PATCH_COLS = KB_FITS_PATCH_COLS;  // pixels per lane and round (32 = one RICE block of the reference's files; 16 -- 19 KB of LDS per block, eight waves per SIMD -- measured slower: 3.86 vs 3.37 ms)
constexpr int PATCH_PITCH = PATCH_COLS + 1;  // words per lane in the wave's LDS patch (odd: lanes hit distinct banks)

template <int BYTEPIX>
__global__ __launch_bounds__(256) void kb_fits_rice_decode_kernel(RiceArgs a) {
    constexpr int FSBITS = BYTEPIX == 4 ? 5 : (BYTEPIX == 2 ? 4 : 3);
    constexpr int FSMAX = BYTEPIX == 4 ? 25 : (BYTEPIX == 2 ? 14 : 6);
    constexpr int BBITS = 8 * BYTEPIX;
    constexpr uint32_t WRAP = BYTEPIX == 4 ? 0xffffffffu : ((1u << (BBITS & 31)) - 1u);
    __shared__ float patch_all[4][WAVE * PATCH_PITCH];
    __shared__ uint64_t first_all[4][WAVE];  // a row's first output pixel; ~0 for rows that are not decoded here
    const int lane = (int)threadIdx.x & (WAVE - 1), wv = (int)threadIdx.x / WAVE;
    float* patch = patch_all[wv];
    const int64_t tile0 = ((int64_t)blockIdx.x * 4 + wv) * WAVE;
    if (tile0 >= a.n_tiles) return;  // (whole waves leave: nothing below synchronises across waves)
    const int64_t tile = tile0 + lane;
    const bool live = tile < a.n_tiles;
    kb_fits_tile td{};
    if (live) td = a.tiles[tile];
    const bool decode = live && td.mode == KB_FITS_TILE_RICE && td.nbytes > (uint32_t)BYTEPIX;
    first_all[wv][lane] = (live && td.mode == KB_FITS_TILE_RICE) ? td.out_index : ~0ull;
    // integer -> pixel value, by the lane that decoded it (it holds its row's ZSCALE / ZZERO): two double operations, each
    // rounded, then one rounding to float -- what numpy / cfitsio compute
    const double zscale = td.zscale, zzero = td.zzero;
    const int has_blank = a.has_blank, blank = a.blank;
    auto value_of = [&](uint32_t pix) -> float {
        int32_t iv = (int32_t)pix;
        if (BYTEPIX == 2) iv = (int32_t)(int16_t)pix;
        if (BYTEPIX == 1) iv = (int32_t)(pix & 0xffu);  // (8-bit FITS pixels are unsigned: cfitsio fits_rdecomp_byte)
        const float v = (float)((double)iv * zscale + zzero);
        return (has_blank && iv == blank) ? __builtin_nanf("") : v;
    };

    BitReader br{};
    uint32_t lastpix = 0;
    if (decode) {
        br.start(a.heap + td.offset, a.heap + a.heap_bytes);
        br.refill();
        lastpix = br.take(BBITS);  // the first pixel verbatim
    }
    bool bad_code = false;
    const int nblk = (a.tile_len + a.blocksize - 1) / a.blocksize;
    const int rows_here = (int)min((int64_t)WAVE, (int64_t)a.n_tiles - tile0);
    float* my_row = patch + lane * PATCH_PITCH;
    for (int blk = 0; blk < nblk; ++blk) {
        const int i0 = blk * a.blocksize;
        const int n = min(a.blocksize, a.tile_len - i0);
        // a block (BLOCKSIZE is a file parameter, 32 in the reference's files) goes out in pieces of PATCH_COLS pixels
        int done = 0;
        uint32_t fs_code = 0;
        if (decode) {
            br.refill();
            fs_code = br.take(FSBITS);
            if (fs_code > (uint32_t)FSMAX + 1u) {  // no such split code: a corrupt stream (reported like one that ends early)
                bad_code = true;
                fs_code = 0;
            }
        }
        while (done < n) {
            const int m = min(PATCH_COLS, n - done);
            if (decode) {
                if (fs_code == 0) {  // every difference is zero
                    const float v = value_of(lastpix);
                    for (int k = 0; k < m; ++k) my_row[k] = v;
                } else if (fs_code == (uint32_t)FSMAX + 1u) {  // verbatim differences
                    for (int k = 0; k < m; ++k) {
                        br.refill();
                        uint32_t d = br.take(BBITS);
                        d = (d & 1u) ? ~(d >> 1) : (d >> 1);
                        lastpix = (lastpix + d) & WRAP;
                        my_row[k] = value_of(lastpix);
                    }
                } else {
                    const int fs = (int)fs_code - 1;
                    for (int k = 0; k < m; ++k) {
                        br.refill();  // at least 33 valid bits: the top 32 of the window are all stream bits
                        const uint32_t hi = (uint32_t)(br.buf >> 32);
                        const int z32 = hi ? __builtin_clz(hi) : 32;
                        const int len = z32 + 1 + fs;
                        uint32_t d;
                        if (len <= 32) {
                            // the whole code -- zeros, the one bit, FS low bits -- lies in those 32 bits: 32-bit arithmetic
                            // and ONE shift of the window (64-bit shifts are the slow instructions of this loop)
                            const uint32_t low = fs ? ((hi << (z32 + 1)) >> (32 - fs)) : 0u;  // (fs > 0: z32 + 1 <= 31)
                            d = ((uint32_t)z32 << fs) | low;
                            br.buf <<= len;
                            br.avail -= len;
                        } else {
                            uint32_t zeros = 0;
                            for (;;) {  // a long unary part: zero bits up to the next one bit, window by window
                                br.refill();
                                const int z = br.buf ? __builtin_clzll(br.buf) : 64;
                                if (z < br.avail) {
                                    zeros += (uint32_t)z;
                                    br.buf <<= z;  // z <= 63
                                    br.buf <<= 1;
                                    br.avail -= z + 1;
                                    break;
                                }
                                zeros += (uint32_t)br.avail;
                                br.buf = 0;
                                br.avail = 0;
                                if (br.taken > td.nbytes + 16u) break;  // a stream without its end: flagged below
                            }
                            br.refill();
                            d = (zeros << fs) | br.take(fs);
                        }
                        d = (d & 1u) ? ~(d >> 1) : (d >> 1);
                        lastpix = (lastpix + d) & WRAP;
                        my_row[k] = value_of(lastpix);
                    }
                }
            }
            // (one wave = one patch: the wave's own LDS writes are visible to it once they have landed)
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
            __builtin_amdgcn_wave_barrier();
            // the 64 x PATCH_COLS patch out: four pixels per lane, PATCH_COLS / 4 lanes per row
            constexpr int LANES_PER_ROW = PATCH_COLS / 4, ROWS_PER_TRIP = WAVE / LANES_PER_ROW;
            const int c0 = 4 * (lane % LANES_PER_ROW), sub = lane / LANES_PER_ROW;
            for (int r8 = 0; r8 < rows_here; r8 += ROWS_PER_TRIP) {
                const int r = r8 + sub;
                if (r < rows_here && c0 < m) {
                    const uint64_t first = first_all[wv][r];
                    if (first != ~0ull) {
                        const float* src = patch + r * PATCH_PITCH + c0;
                        float* dst = a.out + first + (uint64_t)(i0 + done + c0);
                        const float v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
                        if (c0 + 3 < m && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
                            *reinterpret_cast<float4*>(dst) = make_float4(v0, v1, v2, v3);
                        } else {
                            dst[0] = v0;
                            if (c0 + 1 < m) dst[1] = v1;
                            if (c0 + 2 < m) dst[2] = v2;
                            if (c0 + 3 < m) dst[3] = v3;
                        }
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
            done += m;
        }
    }
    if (decode) {
        const uint64_t used = (br.bits_used() + 7u) / 8u;
        if (used > (uint64_t)td.nbytes || bad_code) {
            atomicAdd(&a.status[0], 1);
            atomicCAS(&a.status[1], 0, (int32_t)(tile + 1));
        }
    } else if (live && td.mode == KB_FITS_TILE_RICE) {  // too short to hold its first pixel
        atomicAdd(&a.status[0], 1);
        atomicCAS(&a.status[1], 0, (int32_t)(tile + 1));
    }
}

struct ImageArgs {
    const uint8_t* raw;
    float* out;
    uint64_t n;
    double bscale, bzero;
    int32_t plain;  // BSCALE = 1, BZERO = 0
};

template <int BITPIX>
__device__ __forceinline__ float decode_pixel(const uint8_t* raw, uint64_t i, const ImageArgs& a) {
    if constexpr (BITPIX == -32) {
        const uint32_t w = __builtin_bswap32(reinterpret_cast<const uint32_t*>(raw)[i]);
        const float f = __uint_as_float(w);
        return a.plain ? f : (float)((double)f * a.bscale + a.bzero);
    } else if constexpr (BITPIX == -64) {
        const uint64_t w = __builtin_bswap64(reinterpret_cast<const uint64_t*>(raw)[i]);
        const double d = __longlong_as_double((long long)w);
        return a.plain ? (float)d : (float)(d * a.bscale + a.bzero);
    } else if constexpr (BITPIX == 8) {
        return (float)((double)raw[i] * a.bscale + a.bzero);
    } else if constexpr (BITPIX == 16) {
        const uint16_t w = reinterpret_cast<const uint16_t*>(raw)[i];
        const int16_t v = (int16_t)((w >> 8) | (w << 8));
        return (float)((double)v * a.bscale + a.bzero);
    } else {
        const int32_t v = (int32_t)__builtin_bswap32(reinterpret_cast<const uint32_t*>(raw)[i]);
        return (float)((double)v * a.bscale + a.bzero);
    }
}

// One pixel per thread and trip, four trips per thread a block apart (coalesced on both sides).
template <int BITPIX>
__global__ __launch_bounds__(256) void kb_fits_image_decode_kernel(ImageArgs a) {
    const uint64_t base = (uint64_t)blockIdx.x * 1024u + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint64_t i = base + (uint64_t)k * 256u;
        if (i < a.n) a.out[i] = decode_pixel<BITPIX>(a.raw, i, a);
    }
}

__global__ __launch_bounds__(256) void kb_fits_apply_mask_kernel(float* sci, float* var, const float* mask, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n && mask[i] > 0.0f) {
        sci[i] = __builtin_nanf("");
        if (var != nullptr) var[i] = __builtin_nanf("");
    }
}

}  // namespace
}  // namespace kb

extern "C" int kb_fits_decode_rice(const uint8_t* heap_dev, uint64_t heap_bytes, const kb_fits_tile* tiles_dev,
                                   int32_t n_tiles, int32_t tile_len, int32_t blocksize, int32_t bytepix, int32_t quantized,
                                   int32_t has_blank, int32_t blank, float* out_dev, int32_t* status_dev, void* stream_v) {
    using namespace kb;
    if (n_tiles == 0) return 0;
    KB_REQUIRE_DEVICE("the FITS tile decoder.");
    if (heap_dev == nullptr || tiles_dev == nullptr || out_dev == nullptr || status_dev == nullptr) {
        return fail("fits_decode_rice: null pointer");
    }
    if (n_tiles < 0 || tile_len <= 0 || blocksize <= 0) return fail("fits_decode_rice: invalid tile geometry");
    if (bytepix != 1 && bytepix != 2 && bytepix != 4) return fail("fits_decode_rice: BYTEPIX must be 1, 2 or 4");
    if (heap_bytes < 8) return fail("fits_decode_rice: the heap must hold at least eight bytes");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    RiceArgs a;
    a.heap = heap_dev;
    a.heap_bytes = heap_bytes;
    a.tiles = tiles_dev;
    a.out = out_dev;
    a.status = status_dev;
    a.n_tiles = n_tiles;
    a.tile_len = tile_len;
    a.blocksize = blocksize;
    a.quantized = quantized;
    a.has_blank = has_blank;
    a.blank = blank;
    KB_HIP_TRY(hipMemsetAsync(status_dev, 0, 2 * sizeof(int32_t), stream));
    const unsigned blocks = (unsigned)(((int64_t)n_tiles + 4 * WAVE - 1) / (4 * WAVE));
    if (bytepix == 4) {
        hipLaunchKernelGGL(kb_fits_rice_decode_kernel<4>, dim3(blocks), dim3(256), 0, stream, a);
    } else if (bytepix == 2) {
        hipLaunchKernelGGL(kb_fits_rice_decode_kernel<2>, dim3(blocks), dim3(256), 0, stream, a);
    } else {
        hipLaunchKernelGGL(kb_fits_rice_decode_kernel<1>, dim3(blocks), dim3(256), 0, stream, a);
    }
    KB_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int kb_fits_decode_image(const uint8_t* raw_dev, int32_t bitpix, double bscale, double bzero, uint64_t n_pixels,
                                    float* out_dev, void* stream_v) {
    using namespace kb;
    if (n_pixels == 0) return 0;
    KB_REQUIRE_DEVICE("the FITS image decoder.");
    if (raw_dev == nullptr || out_dev == nullptr) return fail("fits_decode_image: null pointer");
    const int width = bitpix < 0 ? -bitpix / 8 : bitpix / 8;
    if (bitpix != 8 && bitpix != 16 && bitpix != 32 && bitpix != -32 && bitpix != -64) {
        return fail("fits_decode_image: BITPIX " + std::to_string(bitpix) + " is not an image type this decoder reads");
    }
    if (reinterpret_cast<uintptr_t>(raw_dev) % (uintptr_t)width != 0) {
        return fail("fits_decode_image: the data unit must be aligned to its element size");
    }
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    ImageArgs a;
    a.raw = raw_dev;
    a.out = out_dev;
    a.n = n_pixels;
    a.bscale = bscale;
    a.bzero = bzero;
    a.plain = (bscale == 1.0 && bzero == 0.0) ? 1 : 0;
    const dim3 grid((unsigned)((n_pixels + 1023u) / 1024u)), block(256);
    switch (bitpix) {
        case -32: hipLaunchKernelGGL(kb_fits_image_decode_kernel<-32>, grid, block, 0, stream, a); break;
        case -64: hipLaunchKernelGGL(kb_fits_image_decode_kernel<-64>, grid, block, 0, stream, a); break;
        case 8: hipLaunchKernelGGL(kb_fits_image_decode_kernel<8>, grid, block, 0, stream, a); break;
        case 16: hipLaunchKernelGGL(kb_fits_image_decode_kernel<16>, grid, block, 0, stream, a); break;
        default: hipLaunchKernelGGL(kb_fits_image_decode_kernel<32>, grid, block, 0, stream, a); break;
    }
    KB_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int kb_fits_apply_mask(float* sci_dev, float* var_dev, const float* mask_dev, uint64_t n_pixels, void* stream_v) {
    using namespace kb;
    if (n_pixels == 0) return 0;
    KB_REQUIRE_DEVICE("the FITS mask pass.");
    if (sci_dev == nullptr || mask_dev == nullptr) return fail("fits_apply_mask: null pointer");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    hipLaunchKernelGGL(kb_fits_apply_mask_kernel, dim3((unsigned)((n_pixels + 255u) / 256u)), dim3(256), 0, stream, sci_dev,
                       var_dev, mask_dev, n_pixels);
    KB_HIP_TRY(hipGetLastError());
    return 0;
}
