// psi/phi builder for MI355X (gfx950): fused pixel preparation + masked PSF
// correlation + (optional) uint8/uint16 encoding, leaving the interleaved
// PsiPhiArray resident in HBM.
//
// Replaces, for the device path, the reference's
//   generate_psi / generate_phi / square_psf_values   image_utils_cpp.cpp:110-177
//   convolve_psf kernel + deviceConvolve               kernels/image_kernels.cu:29-108
//   compute_scale_params_from_image_vect, set_*_cpu_psi_phi_array, fill_psi_phi_array
//                                                      psi_phi_array.cpp:219-372
// (2*T single-image launches each with malloc + H2D + D2H there; here: one
// batched launch over all epochs, tiles staged through LDS, results written
// straight into the [t][row][col][psi,phi] array).
//
// Numerics follow the reference CPU loop exactly: taps visited row-major
// (j outer, i inner), separate multiply and add (no FMA), invalid taps skipped,
// result = (sum * psf_total) / psf_portion, invalid centre passed through.
#include <cfloat>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "kb_common.h"

#pragma clang fp contract(off)

namespace kb {

constexpr int CONV_BX = 32;  // output tile: 32 x 8 pixels per 256-thread workgroup
constexpr int CONV_BY = 8;

struct ConvArgs {
    const float* in0;  // MODE 0: image ; MODE 1: sci [T][H][W]
    const float* in1;  // MODE 1: var [T][H][W]
    float* out_plain;  // MODE 0: convolved image
    float* out_pairs;  // MODE 1: interleaved [T][H][W][2] float32 (final array or staging)
    const float* psf;      // all kernels, concatenated (MODE 1: followed by their squares)
    const float* psf_pairs;  // MODE 1: the same kernels as (k, k^2) pairs, [2 * psf_off[t] ...] (the strip kernel's packed weights)
    const int* psf_off;    // [T] offset of kernel t in psf
    const int* psf_dim;    // [T]
    const float* psf_tot;  // [T] sum of kernel t ; MODE 1: [T..2T) sums of the squared kernels
    int sq_base;           // MODE 1: offset of the squared kernels inside psf
    int W, H, T;
    int t0;                // first epoch of this launch (blockIdx.z counts from it): the host-stack build
                           // launches chunk by chunk behind the uploads
    int max_radius;
    int empty_is_nan;
    unsigned* minmax;  // MODE 1, encoded output: {psi_min, psi_max, phi_min, phi_max} as ordered keys, or null
    const float* sep;  // separable build: per epoch the column factor u and the row factor w of the kernel
                       // (K[j][i] = u[j] * w[i]), each padded to 2 * max_radius + 1 values: [T][2][dim_max]
};

// Order-preserving map float -> unsigned (finite values only are ever inserted).
__device__ __forceinline__ unsigned float_key(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
static inline float key_float(unsigned k) {
    unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}

// psi_phi_array.cpp:223-232: min/max over the finite values of every image, as ordered keys (per-thread extrema in,
// wave reduction, one atomic per wave and bound that can still move the global value).
__device__ __forceinline__ void reduce_key_range(unsigned* minmax, unsigned kmin0, unsigned kmax0, unsigned kmin1, unsigned kmax1) {
    for (int o = 32; o > 0; o >>= 1) {
        kmin0 = min(kmin0, (unsigned)__shfl_xor((int)kmin0, o));
        kmax0 = max(kmax0, (unsigned)__shfl_xor((int)kmax0, o));
        kmin1 = min(kmin1, (unsigned)__shfl_xor((int)kmin1, o));
        kmax1 = max(kmax1, (unsigned)__shfl_xor((int)kmax1, o));
    }
    // One lane per wave, and only when the wave's extremum can still move the
    // global value (plain pre-read: the atomic itself stays the arbiter).  Without
    // the pre-read ~10^6 same-address atomics serialise into milliseconds.
    if ((threadIdx.x & 63) == 0) {
        volatile unsigned* mm = minmax;
        if (kmin0 != 0xffffffffu) {
            if (kmin0 < mm[0]) atomicMin(&minmax[0], kmin0);
            if (kmax0 > mm[1]) atomicMax(&minmax[1], kmax0);
        }
        if (kmin1 != 0xffffffffu) {
            if (kmin1 < mm[2]) atomicMin(&minmax[2], kmin1);
            if (kmax1 > mm[3]) atomicMax(&minmax[3], kmax1);
        }
    }
}
__device__ __forceinline__ void reduce_value_range(unsigned* minmax, bool inside, float psi, float phi) {
    unsigned kmin0 = 0xffffffffu, kmax0 = 0u, kmin1 = 0xffffffffu, kmax1 = 0u;
    if (inside && __builtin_isfinite(psi)) kmin0 = kmax0 = float_key(psi);
    if (inside && __builtin_isfinite(phi)) kmin1 = kmax1 = float_key(phi);
    reduce_key_range(minmax, kmin0, kmax0, kmin1, kmax1);
}

// One masked correlation at LDS tile position (lx, ly) (tile pitch = pitch).
__device__ __forceinline__ float masked_correlate(const float* __restrict__ tile, int pitch, int lx, int ly,
                                                  const float* __restrict__ k, int dim, int rad, int tile_rad,
                                                  float ktot, int empty_is_nan) {
    const float centre = tile[(ly + tile_rad) * pitch + lx + tile_rad];
    if (!__builtin_isfinite(centre)) return centre;  // image_utils_cpp.cpp:41-44
    float sum = 0.0f, part = 0.0f;
    for (int j = -rad; j <= rad; ++j) {
        const float* row = tile + (ly + tile_rad + j) * pitch + lx + tile_rad;
        const float* krow = k + (j + rad) * dim + rad;
        for (int i = -rad; i <= rad; ++i) {
            const float v = row[i];
            if (__builtin_isfinite(v)) {  // out-of-image taps were staged as NaN
                const float kk = krow[i];
                part += kk;
                sum += v * kk;
            }
        }
    }
    if (part == 0.0f) return empty_is_nan ? NAN : 0.0f;
    return (sum * ktot) / part;
}

// The same correlation where every tap is known to be valid (a tile that lies inside the image with its halo and
// holds no NaN): the validity tests and the running sum of the taps' weights disappear -- that sum, taken in
// the same row-major order from 0, IS ktot (image_utils_cpp.cpp:27-33 sums the kernel the same way), so the
// result (sum * ktot) / ktot has the same bits.  DIM is a compile-time constant: the loops unroll, the samples are
// read at immediate offsets, the weights sit in scalar registers.
typedef const __attribute__((address_space(4))) float* ConstWeights;  // the kernel's weights in global memory, read by scalar loads
template <int DIM>
__device__ __forceinline__ float clean_correlate(const float* __restrict__ tile, int pitch, int lx, int ly,
                                                 ConstWeights k, int tile_rad, float ktot, int empty_is_nan) {
    constexpr int RAD = (DIM - 1) / 2;
    const float* centre = tile + (ly + tile_rad) * pitch + lx + tile_rad;
    float sum = 0.0f;
#pragma unroll
    for (int j = -RAD; j <= RAD; ++j) {
#pragma unroll
        for (int i = -RAD; i <= RAD; ++i) {
            const float kk = k[(j + RAD) * DIM + i + RAD];  // uniform address in the constant address space: a scalar load
            sum += centre[j * pitch + i] * kk;
        }
    }
    if (ktot == 0.0f) return empty_is_nan ? NAN : 0.0f;
    return (sum * ktot) / ktot;
}

// Dispatch on the epoch's kernel size (uniform); sizes without an unrolled instance take the general loop, which
// on a clean tile gives the same bits.
__device__ __forceinline__ float clean_correlate_dim(const float* __restrict__ tile, int pitch, int lx, int ly,
                                                     const float* __restrict__ k, const float* k_global, int dim, int rad,
                                                     int tile_rad, float ktot, int empty_is_nan) {
    const ConstWeights kg = (ConstWeights)(uintptr_t)k_global;
    switch (dim) {
        case 3:
            return clean_correlate<3>(tile, pitch, lx, ly, kg, tile_rad, ktot, empty_is_nan);
        case 5:
            return clean_correlate<5>(tile, pitch, lx, ly, kg, tile_rad, ktot, empty_is_nan);
        case 7:
            return clean_correlate<7>(tile, pitch, lx, ly, kg, tile_rad, ktot, empty_is_nan);
        case 9:
            return clean_correlate<9>(tile, pitch, lx, ly, kg, tile_rad, ktot, empty_is_nan);
        case 11:
            return clean_correlate<11>(tile, pitch, lx, ly, kg, tile_rad, ktot, empty_is_nan);
        case 13:
            return clean_correlate<13>(tile, pitch, lx, ly, kg, tile_rad, ktot, empty_is_nan);
        default:
            return masked_correlate(tile, pitch, lx, ly, k, dim, rad, tile_rad, ktot, empty_is_nan);
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void kb_conv_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = blockIdx.z + a.t0;
    const int dim = a.psf_dim[t];
    const int rad = (dim - 1) / 2;
    const int R = a.max_radius;
    const int pitch = CONV_BX + 2 * R;
    const int rows = CONV_BY + 2 * R;
    float* tile0 = smem;                                  // psi0 (or the image)
    float* tile1 = smem + pitch * rows;                   // phi0
    float* k0 = smem + (MODE == 1 ? 2 : 1) * pitch * rows;  // kernel
    float* k1 = k0 + (2 * R + 1) * (2 * R + 1);           // squared kernel

    const int x0 = blockIdx.x * CONV_BX, y0 = blockIdx.y * CONV_BY;
    const size_t img = (size_t)t * a.W * a.H;

    for (int e = threadIdx.x; e < dim * dim; e += 256) {
        k0[e] = a.psf[a.psf_off[t] + e];
        if (MODE == 1) k1[e] = a.psf[a.sq_base + a.psf_off[t] + e];
    }
    bool all_finite = true;  // every value this thread staged (out-of-image taps are staged as NaN)
    for (int e = threadIdx.x; e < pitch * rows; e += 256) {
        const int ly = e / pitch, lx = e - ly * pitch;
        const int gx = x0 + lx - R, gy = y0 + ly - R;
        float v0 = NAN, v1 = NAN;
        if (gx >= 0 && gx < a.W && gy >= 0 && gy < a.H) {
            const size_t p = img + (size_t)gy * a.W + gx;
            if (MODE == 0) {
                v0 = a.in0[p];
            } else {
                const float sci = a.in0[p];
                const float var = a.in1[p];
                // image_utils_cpp.cpp:142-149 and :165-172
                const bool var_ok = __builtin_isfinite(var) && var != 0.0f;
                v0 = (var_ok && __builtin_isfinite(sci)) ? (sci / var) : NAN;
                v1 = var_ok ? (float)__ddiv_rn(1.0, (double)var) : NAN;
            }
        }
        tile0[e] = v0;
        if (MODE == 1) tile1[e] = v1;
        all_finite = all_finite && __builtin_isfinite(v0) && (MODE == 0 || __builtin_isfinite(v1));
    }
    // (also the barrier between staging and correlation) a clean tile: inside the image with its halo, no NaN
    const bool clean = __syncthreads_and(all_finite ? 1 : 0) != 0;

    const int lx = threadIdx.x % CONV_BX, ly = threadIdx.x / CONV_BX;
    const int gx = x0 + lx, gy = y0 + ly;
    const bool inside = gx < a.W && gy < a.H;
    float psi = NAN, phi = NAN;
    if (inside && clean) {  // clean is uniform, and a clean tile has every pixel inside
        psi = clean_correlate_dim(tile0, pitch, lx, ly, k0, a.psf + a.psf_off[t], dim, rad, R, a.psf_tot[t], a.empty_is_nan);
        if (MODE == 1) {
            phi = clean_correlate_dim(tile1, pitch, lx, ly, k1, a.psf + a.sq_base + a.psf_off[t], dim, rad, R, a.psf_tot[a.T + t],
                                      a.empty_is_nan);
        }
    } else if (inside) {
        psi = masked_correlate(tile0, pitch, lx, ly, k0, dim, rad, R, a.psf_tot[t], a.empty_is_nan);
        if (MODE == 1) phi = masked_correlate(tile1, pitch, lx, ly, k1, dim, rad, R, a.psf_tot[a.T + t], a.empty_is_nan);
    }
    if (inside) {
        const size_t p = img + (size_t)gy * a.W + gx;
        if (MODE == 0) {
            a.out_plain[p] = psi;
        } else {
            reinterpret_cast<float2*>(a.out_pairs)[p] = make_float2(psi, phi);
        }
    }

    if (MODE == 1 && a.minmax != nullptr) reduce_value_range(a.minmax, inside, psi, phi);
}

// ---------------------------------------------------------------------------------------------------------------
// The psi/phi build proper (all epochs share one kernel size 3 .. 9: the PSFs of a survey's stack): 64 x 32 outputs per
// 256-thread workgroup, each thread a COLUMN STRIP of eight outputs.
//  * halo over-fetch (64 + 2r) x (32 + 2r) / (64 x 32) = 1.30 for 7 x 7 kernels, against 2.08 of the 32 x 8 tiles of
//    kb_conv_kernel (which stays for single images and mixed kernel sizes);
//  * an input row of the strip is read from LDS once (2r + 1 values) and feeds every output of the strip it is a tap
//    row of: 12 LDS reads per output and quantity instead of 49; the weights sit in scalar registers for the whole pass;
//  * each output still receives its products in the reference's order -- kernel rows top to bottom (= input rows in
//    the order the strip walks them), columns left to right, separate multiply and add (image_utils_cpp.cpp:45-58) --,
//    so the bits are those of kb_conv_kernel and of the oracle;
//  * a tile with NO_DATA inside or a halo off the image takes the masked form: every sample read becomes the pair
//    (value or 0, 1 or 0) once and every tap adds value x weight and mask x weight unconditionally -- a masked tap adds
//    +-0 to sums that started at +0, which changes no bit, so there is no branch per tap (2 x the arithmetic of a
//    clean tile instead of a compare-and-skip chain);
//  * phi0 = (float)(1.0 / (double)var) of the reference is computed as the correctly rounded float quotient 1.0f / var:
//    the double quotient is correctly rounded to 53 >= 2 x 24 + 2 bits, so rounding it again to 24 bits cannot differ
//    from rounding the exact quotient once (innocuous double rounding); pinned bit for bit against the oracle's
//    double division over 24 decades of variances, denormal quotients included (tests/test_gpu_builder_and_api.py).
// ---------------------------------------------------------------------------------------------------------------
constexpr int STRIP_BX = 64, STRIP_BY = 32, STRIP_OUT = 8;

__device__ __forceinline__ void prepare_pixel(float sci, float var, float* psi0, float* phi0) {
    // image_utils_cpp.cpp:142-149 and :165-172
    const bool var_ok = __builtin_isfinite(var) && var != 0.0f;
    *psi0 = (var_ok && __builtin_isfinite(sci)) ? (sci / var) : NAN;
    *phi0 = var_ok ? (1.0f / var) : NAN;
}

// Both quantities of one strip, as pairs: (psi0, phi0) samples x (k, k^2) weights are packed multiplies and adds
// (v_pk_mul_f32 / v_pk_add_f32 -- a plain VALU instruction occupies the SIMD for a quad-cycle, so the packed form is
// what halves the arithmetic time; each half is the IEEE operation of the scalar form, there is no FMA).
// tile: this thread's top-left tap in the staged tile of pairs (pitch pairs per row; NO_DATA is staged as what it is,
// NaN or an infinity).  MASKED: every sample read becomes (value or 0, 1 or 0) once, for all the outputs it is a tap of.
typedef float Pair2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) Pair2* ConstWeightPairs;  // (k, k^2) pairs in global memory: scalar loads
// VSYM: kernel rows j and DIM - 1 - j hold the same bits (every Gaussian PSF; the host checks each epoch's kernel).  The
// weights of row j are then read from row min(j, DIM - 1 - j): an input sample's product with a weight is the same expression
// for the two outputs of the strip that meet it in mirrored kernel rows, and is multiplied once -- 44 instead of 56 packed
// multiplies per column of a strip of eight (7 x 7), the adds and their order untouched -- and 28 weight pairs instead of 49
// sit in scalar registers.
template <int DIM, bool MASKED, bool VSYM>
__device__ __forceinline__ void strip_pass(const Pair2* tile, int pitch,  // (no __restrict__: the row fence below must order the reads)
                                           ConstWeightPairs k, Pair2 (&acc)[STRIP_OUT], Pair2 (&part)[STRIP_OUT]) {
#pragma unroll
    for (int r = 0; r < STRIP_OUT; ++r) {
        acc[r] = Pair2{0.0f, 0.0f};
        part[r] = Pair2{0.0f, 0.0f};
    }
#pragma unroll
    for (int yy = 0; yy < STRIP_OUT + DIM - 1; ++yy) {
        Pair2 v[DIM], m[DIM];
#pragma unroll
        for (int i = 0; i < DIM; ++i) {
            v[i] = tile[yy * pitch + i];
            if (MASKED) {
                const bool d0 = __builtin_isfinite(v[i].x), d1 = __builtin_isfinite(v[i].y);
                m[i] = Pair2{d0 ? 1.0f : 0.0f, d1 ? 1.0f : 0.0f};
                v[i] = Pair2{d0 ? v[i].x : 0.0f, d1 ? v[i].y : 0.0f};
            }
        }
        if constexpr (!VSYM) {
#pragma unroll
            for (int r = 0; r < STRIP_OUT; ++r) {
                const int j = yy - r;  // kernel row of input row yy for output r
                if (j >= 0 && j < DIM) {
#pragma unroll
                    for (int i = 0; i < DIM; ++i) {
                        const Pair2 kk = k[j * DIM + i];  // compile-time index into the constant address space: a scalar register pair
                        if (MASKED) part[r] += m[i] * kk;
                        acc[r] += v[i] * kk;
                    }
                }
            }
        } else {
            // column by column (an output still receives this row's products left to right): the products of ONE sample -- at
            // most (DIM + 1) / 2 different ones -- are all that is alive between two columns
#pragma unroll
            for (int i = 0; i < DIM; ++i) {
#pragma unroll
                for (int r = 0; r < STRIP_OUT; ++r) {
                    const int j = yy - r;
                    if (j >= 0 && j < DIM) {
                        const Pair2 kk = k[(j < DIM - 1 - j ? j : DIM - 1 - j) * DIM + i];
                        if (MASKED) part[r] += m[i] * kk;
                        acc[r] += v[i] * kk;
                    }
                }
#pragma unroll
                for (int r = 0; r < STRIP_OUT; ++r) {
                    asm volatile("" : "+v"(acc[r]));
                    if (MASKED) asm volatile("" : "+v"(part[r]));
                }
            }
        }
        // one input row at a time: left alone the compiler issues the LDS reads of every row up front and the strip's
        // (8 + 2r) x (2r + 1) samples cost the kernel its occupancy
#pragma unroll
        for (int r = 0; r < STRIP_OUT; ++r) {
            asm volatile("" : "+v"(acc[r])::"memory");  // (the sums of this row are due here, in front of the next row's reads)
            if (MASKED) asm volatile("" : "+v"(part[r]));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Tile bookkeeping and staging shared by the two strip kernels: which tile this workgroup owns, and the staged frame of
// (psi0, phi0) pairs in LDS.  Returns false for the workgroups past the last tile; *clean = the frame lies inside the
// image and holds no NO_DATA.
struct StripTile {
    int t, x0, y0;
    size_t img;
};
template <int DIM>
struct StripGeometry {
    static constexpr int R = (DIM - 1) / 2;
    static_assert(R <= 4, "the staged rows start 4 columns left of the tile (16-byte aligned quads)");
    static constexpr int APRON = 4;                     // columns staged left and right of the tile
    static constexpr int PITCH = STRIP_BX + 2 * APRON;  // 72 pairs per staged row
    static constexpr int ROWS = STRIP_BY + 2 * R;
    static constexpr int QUADS_PER_ROW = PITCH / 4, N_QUADS = QUADS_PER_ROW * ROWS;
};
template <int DIM>
__device__ __forceinline__ bool stage_strip_tile(const ConvArgs& a, int tiles_x, int tiles_y, int n_tiles, Pair2* tile,
                                                 StripTile* out, bool* clean_out) {
    using G = StripGeometry<DIM>;
    constexpr int R = G::R, APRON = G::APRON, QUADS_PER_ROW = G::QUADS_PER_ROW, N_QUADS = G::N_QUADS;

    // XCD-aware tile order: workgroup b runs on XCD b % 8; every XCD walks a contiguous run of the (epoch, tile row,
    // tile column) order, so that the halo rows and columns a tile shares with its neighbours are hits in that XCD's L2
    // (dealt round-robin, neighbouring tiles sit on eight different L2s and every halo line is fetched from the fabric again)
    const int b = blockIdx.x;
    const int per_xcd = (n_tiles + 7) >> 3;
    const int linear = (b & 7) * per_xcd + (b >> 3);
    if (linear >= n_tiles) return false;  // whole workgroup
    const int per_epoch = tiles_x * tiles_y;
    const int tz = linear / per_epoch, in_epoch = linear - tz * per_epoch;
    const int tyi = in_epoch / tiles_x, txi = in_epoch - tyi * tiles_x;
    const int t = tz + a.t0;
    const int x0 = txi * STRIP_BX, y0 = tyi * STRIP_BY;
    const size_t img = (size_t)t * a.W * a.H;

    // Staging.  Rows are fetched as 16-byte quads from 4 columns left of the tile on (tile origins are multiples of 64
    // columns: aligned whenever the image width is a multiple of 4); a quad that straddles the image edge, or any quad
    // of an image whose rows are not 16-byte aligned, is fetched element by element.  Every load of the thread is issued
    // before the first quotient (a rolled loop of load -> divide -> store runs one memory latency per element).
    constexpr int N_STAGE = (N_QUADS + 255) / 256;
    typedef float Quad4 __attribute__((ext_vector_type(4)));
    Quad4 s_in[N_STAGE], v_in[N_STAGE];
    const bool rows_aligned = (a.W & 3) == 0;
#pragma unroll
    for (int it = 0; it < N_STAGE; ++it) {
        const int q = (int)threadIdx.x + 256 * it;
        const int ly = q / QUADS_PER_ROW, lq = q - ly * QUADS_PER_ROW;
        const int gx = x0 - APRON + 4 * lq, gy = y0 + ly - R;
        const bool row_in = q < N_QUADS && gy >= 0 && gy < a.H;
        const size_t row = img + (size_t)(row_in ? gy : 0) * a.W;
        Quad4 sv = Quad4{NAN, NAN, NAN, NAN}, vv = sv;
        if (row_in && rows_aligned && gx >= 0 && gx + 3 < a.W) {
            sv = *reinterpret_cast<const Quad4*>(a.in0 + row + gx);
            vv = *reinterpret_cast<const Quad4*>(a.in1 + row + gx);
        } else if (row_in) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (gx + c >= 0 && gx + c < a.W) {
                    sv[c] = a.in0[row + gx + c];
                    vv[c] = a.in1[row + gx + c];
                }
            }
        }
        s_in[it] = sv;
        v_in[it] = vv;
    }
    bool all_data = true;
#pragma unroll
    for (int it = 0; it < N_STAGE; ++it) {
        const int q = (int)threadIdx.x + 256 * it;
        if (q < N_QUADS) {
            Pair2 out[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float p0, p1;
                prepare_pixel(s_in[it][c], v_in[it][c], &p0, &p1);  // (NaN variance -> NO_DATA: what an off-image tap is)
                out[c] = Pair2{p0, p1};
                all_data = all_data && __builtin_isfinite(p0) && __builtin_isfinite(p1);
            }
            typedef float Oct8 __attribute__((ext_vector_type(8)));
            *reinterpret_cast<Oct8*>(tile + 4 * q) = Oct8{out[0].x, out[0].y, out[1].x, out[1].y, out[2].x, out[2].y, out[3].x, out[3].y};
        }
    }
    // (a tile counts as clean only with its whole staged frame in the image and free of NO_DATA: the unused apron
    // columns beyond the kernel's reach included -- a slightly stricter test than needed, same results)
    *clean_out = __syncthreads_and(all_data ? 1 : 0) != 0;  // (also the barrier behind the staging)
    out->t = t;
    out->x0 = x0;
    out->y0 = y0;
    out->img = img;
    return true;
}


template <int DIM, bool VSYM>
// (second launch bound = waves per SIMD: four 4-wave workgroups per CU, 128 registers a lane)
__global__ __launch_bounds__(256, 4) void kb_psi_phi_strip_kernel(const ConvArgs a, int tiles_x, int tiles_y, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using G = StripGeometry<DIM>;
    constexpr int R = G::R, APRON = G::APRON, PITCH = G::PITCH;
    Pair2* tile = reinterpret_cast<Pair2*>(smem);  // [ROWS][PITCH] (psi0, phi0)
    StripTile where;
    bool clean;
    if (!stage_strip_tile<DIM>(a, tiles_x, tiles_y, n_tiles, tile, &where, &clean)) return;
    const int t = where.t, x0 = where.x0, y0 = where.y0;
    const size_t img = where.img;

    const int tx = threadIdx.x & (STRIP_BX - 1), ty = threadIdx.x / STRIP_BX;
    const int corner = (ty * STRIP_OUT) * PITCH + tx + (APRON - R);  // the strip's top-left tap
    const ConstWeightPairs k = (ConstWeightPairs)(uintptr_t)(a.psf_pairs + 2 * (size_t)a.psf_off[t]);
    const float tot0 = a.psf_tot[t], tot1 = a.psf_tot[a.T + t];
    const float empty = a.empty_is_nan ? NAN : 0.0f;
    float psi[STRIP_OUT], phi[STRIP_OUT];
    Pair2 acc[STRIP_OUT], part[STRIP_OUT];
    if (clean) {
        // every tap counts: the weights seen add up, in this very order, to the kernel total (kb_conv_kernel's clean path)
        strip_pass<DIM, false, VSYM>(tile + corner, PITCH, k, acc, part);
#pragma unroll
        for (int r = 0; r < STRIP_OUT; ++r) {
            psi[r] = (tot0 == 0.0f) ? empty : (acc[r].x * tot0) / tot0;
            phi[r] = (tot1 == 0.0f) ? empty : (acc[r].y * tot1) / tot1;
        }
    } else {
        strip_pass<DIM, true, VSYM>(tile + corner, PITCH, k, acc, part);
#pragma unroll
        for (int r = 0; r < STRIP_OUT; ++r) {
            psi[r] = (part[r].x == 0.0f) ? empty : (acc[r].x * tot0) / part[r].x;
            phi[r] = (part[r].y == 0.0f) ? empty : (acc[r].y * tot1) / part[r].y;
        }
    }

    const int gx = x0 + tx;
    unsigned kmin0 = 0xffffffffu, kmax0 = 0u, kmin1 = 0xffffffffu, kmax1 = 0u;
#pragma unroll
    for (int r = 0; r < STRIP_OUT; ++r) {
        const int gy = y0 + ty * STRIP_OUT + r;
        if (gx < a.W && gy < a.H) {
            if (!clean) {
                // an invalid centre passes through unchanged (image_utils_cpp.cpp:41-44): NaN, or the infinity it was
                const Pair2 centre = tile[(ty * STRIP_OUT + r + R) * PITCH + tx + APRON];
                if (!__builtin_isfinite(centre.x)) psi[r] = centre.x;
                if (!__builtin_isfinite(centre.y)) phi[r] = centre.y;
            }
            reinterpret_cast<float2*>(a.out_pairs)[img + (size_t)gy * a.W + gx] = make_float2(psi[r], phi[r]);
            if (a.minmax != nullptr) {  // uniform: the value range is only wanted for encoded arrays
                if (__builtin_isfinite(psi[r])) {
                    kmin0 = min(kmin0, float_key(psi[r]));
                    kmax0 = max(kmax0, float_key(psi[r]));
                }
                if (__builtin_isfinite(phi[r])) {
                    kmin1 = min(kmin1, float_key(phi[r]));
                    kmax1 = max(kmax1, float_key(phi[r]));
                }
            }
        }
    }
    if (a.minmax != nullptr) reduce_key_range(a.minmax, kmin0, kmax0, kmin1, kmax1);
}

// Separable form of the strip kernel (KB_BUILD_SEPARABLE; rank-1 kernels K[j][i] = u[j] * w[i]: every Gaussian PSF,
// core/psf.py:49-74).  Each thread first reduces an input row of its strip to one pair of row sums (2r + 1 packed
// multiply-adds, mask sums beside them in a masked tile) and then feeds that pair to every output it is a tap row of:
// (2r + 1) + (2r + 1) products per output instead of (2r + 1)^2 -- no second LDS pass, no barrier.  The summation order
// is not the reference's tap loop, so the result agrees with it to rounding (<= 1e-4 relative, same NO_DATA pattern:
// tests/test_gpu_boundaries.py against the oracle), not bit for bit: opt-in.  sep_pairs: per epoch (u_j, u_j^2)[DIM]
// then (w_i, w_i^2)[DIM] -- the squared kernel of phi factors as u^2 x w^2 (image_utils_cpp.cpp:110-120).
template <int DIM, bool MASKED>
__device__ __forceinline__ void strip_pass_separable(const Pair2* tile, int pitch, ConstWeightPairs u, ConstWeightPairs w,
                                                     Pair2 (&acc)[STRIP_OUT], Pair2 (&part)[STRIP_OUT]) {
#pragma unroll
    for (int r = 0; r < STRIP_OUT; ++r) {
        acc[r] = Pair2{0.0f, 0.0f};
        part[r] = Pair2{0.0f, 0.0f};
    }
#pragma unroll
    for (int yy = 0; yy < STRIP_OUT + DIM - 1; ++yy) {
        Pair2 h = Pair2{0.0f, 0.0f}, hm = Pair2{0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < DIM; ++i) {
            Pair2 v = tile[yy * pitch + i];
            if (MASKED) {
                const bool d0 = __builtin_isfinite(v.x), d1 = __builtin_isfinite(v.y);
                hm += Pair2{d0 ? 1.0f : 0.0f, d1 ? 1.0f : 0.0f} * w[i];
                v = Pair2{d0 ? v.x : 0.0f, d1 ? v.y : 0.0f};
            }
            h += v * w[i];
        }
#pragma unroll
        for (int r = 0; r < STRIP_OUT; ++r) {
            const int j = yy - r;
            if (j >= 0 && j < DIM) {
                acc[r] += h * u[j];
                if (MASKED) part[r] += hm * u[j];
            }
        }
#pragma unroll
        for (int r = 0; r < STRIP_OUT; ++r) {
            asm volatile("" : "+v"(acc[r])::"memory");
            if (MASKED) asm volatile("" : "+v"(part[r]));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int DIM>
__global__ __launch_bounds__(256, 4) void kb_psi_phi_strip_sep_kernel(const ConvArgs a, int tiles_x, int tiles_y, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using G = StripGeometry<DIM>;
    constexpr int R = G::R, APRON = G::APRON, PITCH = G::PITCH;
    Pair2* tile = reinterpret_cast<Pair2*>(smem);
    StripTile where;
    bool clean;
    if (!stage_strip_tile<DIM>(a, tiles_x, tiles_y, n_tiles, tile, &where, &clean)) return;
    const int t = where.t, x0 = where.x0, y0 = where.y0;
    const size_t img = where.img;

    const int tx = threadIdx.x & (STRIP_BX - 1), ty = threadIdx.x / STRIP_BX;
    const int corner = (ty * STRIP_OUT) * PITCH + tx + (APRON - R);
    const ConstWeightPairs u = (ConstWeightPairs)(uintptr_t)(a.sep + (size_t)t * 4 * DIM);
    const ConstWeightPairs w = u + DIM;
    const float tot0 = a.psf_tot[t], tot1 = a.psf_tot[a.T + t];
    const float empty = a.empty_is_nan ? NAN : 0.0f;
    float psi[STRIP_OUT], phi[STRIP_OUT];
    Pair2 acc[STRIP_OUT], part[STRIP_OUT];
    if (clean) {
        strip_pass_separable<DIM, false>(tile + corner, PITCH, u, w, acc, part);
        // the weight every output has seen: the sums the masked form takes with every mask at 1, in its order
        Pair2 hm = Pair2{0.0f, 0.0f}, seen = Pair2{0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < DIM; ++i) hm += Pair2{1.0f, 1.0f} * w[i];
#pragma unroll
        for (int j = 0; j < DIM; ++j) seen += hm * u[j];
#pragma unroll
        for (int r = 0; r < STRIP_OUT; ++r) part[r] = seen;
    } else {
        strip_pass_separable<DIM, true>(tile + corner, PITCH, u, w, acc, part);
    }
#pragma unroll
    for (int r = 0; r < STRIP_OUT; ++r) {
        psi[r] = (part[r].x == 0.0f) ? empty : (acc[r].x * tot0) / part[r].x;
        phi[r] = (part[r].y == 0.0f) ? empty : (acc[r].y * tot1) / part[r].y;
    }
    const int gx = x0 + tx;
    unsigned kmin0 = 0xffffffffu, kmax0 = 0u, kmin1 = 0xffffffffu, kmax1 = 0u;
#pragma unroll
    for (int r = 0; r < STRIP_OUT; ++r) {
        const int gy = y0 + ty * STRIP_OUT + r;
        if (gx < a.W && gy < a.H) {
            if (!clean) {  // an invalid centre passes through unchanged (image_utils_cpp.cpp:41-44)
                const Pair2 centre = tile[(ty * STRIP_OUT + r + R) * PITCH + tx + APRON];
                if (!__builtin_isfinite(centre.x)) psi[r] = centre.x;
                if (!__builtin_isfinite(centre.y)) phi[r] = centre.y;
            }
            reinterpret_cast<float2*>(a.out_pairs)[img + (size_t)gy * a.W + gx] = make_float2(psi[r], phi[r]);
            if (a.minmax != nullptr) {
                if (__builtin_isfinite(psi[r])) {
                    kmin0 = min(kmin0, float_key(psi[r]));
                    kmax0 = max(kmax0, float_key(psi[r]));
                }
                if (__builtin_isfinite(phi[r])) {
                    kmin1 = min(kmin1, float_key(phi[r]));
                    kmax1 = max(kmax1, float_key(phi[r]));
                }
            }
        }
    }
    if (a.minmax != nullptr) reduce_key_range(a.minmax, kmin0, kmax0, kmin1, kmax1);
}

// Separable variant of the MODE 1 build for rank-1 kernels K[j][i] = u[j] * w[i] (every Gaussian PSF,
// the reference default: core/psf.py:49-74).  The masked correlation (sum over valid taps of v * K) * sum(K)
// / (sum over valid taps of K) splits into a row pass and a column pass over two planes per quantity,
// the masked values m * v and the mask m itself: 4 * (2r + 1) multiply-adds per pixel and quantity
// instead of 2 * (2r + 1)^2.  The summation order differs from the reference's row-major tap loop, so
// the result agrees to rounding (tested to 1e-4 relative against the reference twin's vectors), not
// bit for bit: opt-in (KB_BUILD_SEPARABLE), the 2-D kernel stays the default.  This 32 x 8 tile form serves mixed
// kernel sizes and sizes beyond 9 x 9; stacks with one kernel size 3 .. 9 take kb_psi_phi_strip_sep_kernel.
__global__ __launch_bounds__(256) void kb_conv_sep_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = blockIdx.z + a.t0;
    const int dim = a.psf_dim[t];
    const int rad = (dim - 1) / 2;
    const int R = a.max_radius;
    const int DM = 2 * R + 1;
    const int pitch = CONV_BX + 2 * R;
    const int rows = CONV_BY + 2 * R;
    // staged planes: masked psi0, its mask, masked phi0, its mask; then the four row-pass planes; then factors
    float* in_pl = smem;                          // [4][rows][pitch]
    float* row_pl = smem + 4 * pitch * rows;      // [4][rows][CONV_BX]
    float* fac = row_pl + 4 * rows * CONV_BX;     // u, w, u^2, w^2: [4][DM]

    const int x0 = blockIdx.x * CONV_BX, y0 = blockIdx.y * CONV_BY;
    const size_t img = (size_t)t * a.W * a.H;
    const float* sep_t = a.sep + (size_t)t * 2 * DM;
    for (int e = threadIdx.x; e < 2 * DM; e += 256) {
        const float f = sep_t[e];
        fac[e] = f;
        fac[2 * DM + e] = f * f;  // (u_j w_i)^2 = u_j^2 w_i^2: the squared kernel of phi (image_utils_cpp.cpp:110-120)
    }
    for (int e = threadIdx.x; e < pitch * rows; e += 256) {
        const int ly = e / pitch, lx = e - ly * pitch;
        const int gx = x0 + lx - R, gy = y0 + ly - R;
        float v0 = NAN, v1 = NAN;
        if (gx >= 0 && gx < a.W && gy >= 0 && gy < a.H) {
            const size_t p = img + (size_t)gy * a.W + gx;
            const float sci = a.in0[p];
            const float var = a.in1[p];
            const bool var_ok = __builtin_isfinite(var) && var != 0.0f;  // image_utils_cpp.cpp:142-149, :165-172
            v0 = (var_ok && __builtin_isfinite(sci)) ? (sci / var) : NAN;
            v1 = var_ok ? (float)__ddiv_rn(1.0, (double)var) : NAN;
        }
        const bool m0 = __builtin_isfinite(v0), m1 = __builtin_isfinite(v1);
        in_pl[e] = m0 ? v0 : 0.0f;
        in_pl[pitch * rows + e] = m0 ? 1.0f : 0.0f;
        in_pl[2 * pitch * rows + e] = m1 ? v1 : 0.0f;
        in_pl[3 * pitch * rows + e] = m1 ? 1.0f : 0.0f;
    }
    __syncthreads();
    // row pass: every row of the tile and its halo, the tile's own columns
    const float* w = fac + DM;
    const float* w2 = fac + 3 * DM;
    for (int e = threadIdx.x; e < rows * CONV_BX; e += 256) {
        const int ly = e / CONV_BX, lx = e - ly * CONV_BX;
        const float* r0 = in_pl + ly * pitch + lx + R;
        float s0 = 0.0f, p0 = 0.0f, s1 = 0.0f, p1 = 0.0f;
        for (int i = -rad; i <= rad; ++i) {
            const float wi = w[i + rad], wi2 = w2[i + rad];
            s0 += r0[i] * wi;
            p0 += r0[pitch * rows + i] * wi;
            s1 += r0[2 * pitch * rows + i] * wi2;
            p1 += r0[3 * pitch * rows + i] * wi2;
        }
        row_pl[e] = s0;
        row_pl[rows * CONV_BX + e] = p0;
        row_pl[2 * rows * CONV_BX + e] = s1;
        row_pl[3 * rows * CONV_BX + e] = p1;
    }
    __syncthreads();

    const int lx = threadIdx.x % CONV_BX, ly = threadIdx.x / CONV_BX;
    const int gx = x0 + lx, gy = y0 + ly;
    const bool inside = gx < a.W && gy < a.H;
    float psi = NAN, phi = NAN;
    if (inside) {
        const float* u = fac;
        const float* u2 = fac + 2 * DM;
        const float* c0 = row_pl + (ly + R) * CONV_BX + lx;
        float s0 = 0.0f, p0 = 0.0f, s1 = 0.0f, p1 = 0.0f;
        for (int j = -rad; j <= rad; ++j) {
            const float uj = u[j + rad], uj2 = u2[j + rad];
            s0 += c0[j * CONV_BX] * uj;
            p0 += c0[rows * CONV_BX + j * CONV_BX] * uj;
            s1 += c0[2 * rows * CONV_BX + j * CONV_BX] * uj2;
            p1 += c0[3 * rows * CONV_BX + j * CONV_BX] * uj2;
        }
        const int ce = (ly + R) * pitch + lx + R;
        const bool c_psi = in_pl[pitch * rows + ce] != 0.0f, c_phi = in_pl[3 * pitch * rows + ce] != 0.0f;
        const float empty = a.empty_is_nan ? NAN : 0.0f;
        psi = (p0 == 0.0f) ? empty : (s0 * a.psf_tot[t]) / p0;
        phi = (p1 == 0.0f) ? empty : (s1 * a.psf_tot[a.T + t]) / p1;
        if (!c_psi || !c_phi) {  // an invalid centre passes through unchanged (image_utils_cpp.cpp:41-44)
            const size_t p = img + (size_t)gy * a.W + gx;
            const float sci = a.in0[p];
            const float var = a.in1[p];
            const bool var_ok = __builtin_isfinite(var) && var != 0.0f;
            if (!c_psi) psi = (var_ok && __builtin_isfinite(sci)) ? (sci / var) : NAN;
            if (!c_phi) phi = var_ok ? (float)__ddiv_rn(1.0, (double)var) : NAN;
        }
        reinterpret_cast<float2*>(a.out_pairs)[img + (size_t)gy * a.W + gx] = make_float2(psi, phi);
    }
    if (a.minmax != nullptr) reduce_value_range(a.minmax, inside, psi, phi);
}

// psi_phi_array_ds.h:40-43 + psi_phi_array.cpp:284-285 (truncating cast).
__device__ __forceinline__ unsigned encode_value(float v, float mn, float safe_max, float scale) {
    if (!__builtin_isfinite(v)) return 0u;
    const float lo = (safe_max < v) ? safe_max : v;
    const float cl = (lo < mn) ? mn : lo;
    const float q = (cl - mn) / scale;
    const float e = (float)__dadd_rn((double)q, 1.0);
    return (unsigned)e;
}

template <typename OUT>
__global__ __launch_bounds__(256) void kb_encode_kernel(const float2* __restrict__ pairs, OUT* __restrict__ out,
                                                        size_t n, float psi_min, float psi_safe_max,
                                                        float psi_scale, float phi_min, float phi_safe_max,
                                                        float phi_scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float2 v = pairs[i];
        OUT o;
        o.x = encode_value(v.x, psi_min, psi_safe_max, psi_scale);
        o.y = encode_value(v.y, phi_min, phi_safe_max, phi_scale);
        out[i] = o;
    }
}

struct DeviceBuffer {  // frees on scope exit
    void* p = nullptr;
    ~DeviceBuffer() {
        if (p) (void)hipFree(p);
    }
};

// psi_phi_array.cpp:234-242
static float scale_from_range(float min_val, float max_val, int num_bytes) {
    float width = (max_val - min_val);
    if (width < 1e-6) width = 1e-6;
    const uint64_t num_values = (1 << (8 * num_bytes)) - 1;
    return (float)(width / (double)num_values);
}

static void fill_meta(kb_psi_phi_meta* m, int num_bytes, uint64_t T, uint64_t H, uint64_t W) {
    // psi_phi_array.cpp:113-148
    m->num_bytes = (num_bytes == 1 || num_bytes == 2) ? num_bytes : 4;
    m->block_size = (uint64_t)m->num_bytes;
    m->num_times = T;
    m->width = W;
    m->height = H;
    m->pixels_per_image = W * H;
    m->num_entries = 2 * m->pixels_per_image * T;
    m->total_array_size = m->block_size * m->num_entries;
    m->psi_min_val = FLT_MAX;
    m->psi_max_val = -FLT_MAX;
    m->psi_scale = 1.0f;
    m->phi_min_val = FLT_MAX;
    m->phi_max_val = -FLT_MAX;
    m->phi_scale = 1.0f;
}

// Packs kernels (+ squares when with_squares) and their float sums.
static int pack_psfs(const float* psf_host, const int32_t* psf_dims, int T, bool with_squares,
                     std::vector<float>* packed, std::vector<int>* offs, std::vector<float>* totals,
                     int* max_radius, int* sq_base) {
    int total = 0;
    *max_radius = 0;
    offs->resize(T);
    for (int t = 0; t < T; ++t) {
        if (psf_dims[t] <= 0 || (psf_dims[t] % 2) == 0) return fail("PSF kernels must be square with an odd width.");
        (*offs)[t] = total;
        total += psf_dims[t] * psf_dims[t];
        *max_radius = std::max(*max_radius, (psf_dims[t] - 1) / 2);
    }
    *sq_base = total;
    packed->assign(psf_host, psf_host + total);
    totals->assign(with_squares ? 2 * T : T, 0.0f);
    if (with_squares) packed->resize(2 * (size_t)total);
    for (int t = 0; t < T; ++t) {
        const int n = psf_dims[t] * psf_dims[t];
        float tot = 0.0f, tot_sq = 0.0f;
        for (int e = 0; e < n; ++e) {  // image_utils_cpp.cpp:30-35 row-major float sum
            const float k = psf_host[(*offs)[t] + e];
            tot += k;
            if (with_squares) {
                const float k2 = k * k;  // image_utils_cpp.cpp:110-120
                (*packed)[(size_t)total + (*offs)[t] + e] = k2;
                tot_sq += k2;
            }
        }
        (*totals)[t] = tot;
        if (with_squares) (*totals)[T + t] = tot_sq;
    }
    return 0;
}

static size_t conv_lds_bytes(int max_radius, bool two_tiles) {
    const int pitch = CONV_BX + 2 * max_radius, rows = CONV_BY + 2 * max_radius;
    const int kd = 2 * max_radius + 1;
    return sizeof(float) * ((size_t)(two_tiles ? 2 : 1) * pitch * rows + (size_t)(two_tiles ? 2 : 1) * kd * kd);
}

}  // namespace kb

extern "C" {

}  // extern "C"

namespace kb {

// Rank-1 factors of kernel t: K[j][i] = u[j] * w[i] to 1e-6 of the largest tap, or false.
static bool factor_kernel(const float* k, int dim, float* u, float* w) {
    int j0 = 0, i0 = 0;
    double big = 0.0;
    for (int j = 0; j < dim; ++j) {
        for (int i = 0; i < dim; ++i) {
            if (std::fabs((double)k[j * dim + i]) > big) {
                big = std::fabs((double)k[j * dim + i]);
                j0 = j;
                i0 = i;
            }
        }
    }
    if (!(big > 0.0)) return false;
    const double pivot = k[j0 * dim + i0];
    for (int j = 0; j < dim; ++j) u[j] = k[j * dim + i0];
    for (int i = 0; i < dim; ++i) w[i] = (float)((double)k[j0 * dim + i] / pivot);
    for (int j = 0; j < dim; ++j) {
        for (int i = 0; i < dim; ++i) {
            if (std::fabs((double)u[j] * (double)w[i] - (double)k[j * dim + i]) > 1e-6 * big) return false;
        }
    }
    return true;
}

// HIP-event time of the correlation launch of this thread's last build from device stacks (kb_last_build_kernel_ms).
static thread_local float g_last_build_kernel_ms = 0.0f;

// Pinned staging of the host-stack build: two buffers, kept between calls.
struct PinnedStage {
    void* p[2] = {nullptr, nullptr};
    size_t bytes = 0;
};
static PinnedStage g_stage;
static std::mutex g_stage_mutex;

// The psi/phi build.  The image stacks are either resident (sci_dev / var_dev) or contiguous host
// stacks (sci_host / var_host: [T][H][W]) that are uploaded chunk by chunk through pinned staging
// buffers on a copy stream while the correlation kernel of the chunk before runs.
static int build_psi_phi(const float* sci_dev, const float* var_dev, const float* sci_host, const float* var_host,
                         const float* psf_host, const int32_t* psf_dims, int32_t num_times, int32_t height,
                         int32_t width, int32_t num_bytes, uint32_t build_flags, kb_psi_phi_meta* meta_out,
                         void** psi_phi_dev_out, hipStream_t stream) {
    if (meta_out == nullptr || psi_phi_dev_out == nullptr) return fail("build_psi_phi: null output pointer");
    *psi_phi_dev_out = nullptr;
    if (num_times <= 0) return fail("Trying to fill PsiPhi from empty vectors.");
    if (width <= 0 || height <= 0) return fail("Invalid image dimensions for PsiPhi build.");
    if (num_bytes != -1 && num_bytes != 1 && num_bytes != 2 && num_bytes != 4) {
        return fail("Invalid setting of num_bytes. Must be (-1 [use default], 1, 2, or 4). Got " +
                    std::to_string(num_bytes));
    }
    const bool from_host = sci_host != nullptr;
    if ((!from_host && (sci_dev == nullptr || var_dev == nullptr)) || (from_host && var_host == nullptr) ||
        psf_host == nullptr || psf_dims == nullptr) {
        return fail("build_psi_phi: null input pointer");
    }
    if (kb_device_count() == 0) return fail("GPU is not available for the psi/phi build.");
    (void)hipGetLastError();

    fill_meta(meta_out, num_bytes, (uint64_t)num_times, (uint64_t)height, (uint64_t)width);
    const bool encoded = meta_out->num_bytes != 4;
    const size_t img = (size_t)height * width;
    const size_t n_pix = (size_t)num_times * img;

    std::vector<float> packed, totals;
    std::vector<int> offs;
    int max_radius = 0, sq_base = 0;
    if (pack_psfs(psf_host, psf_dims, num_times, true, &packed, &offs, &totals, &max_radius, &sq_base)) return 1;

    // separable build: every kernel must factor
    const int DM = 2 * max_radius + 1;
    std::vector<float> factors;
    bool separable = (build_flags & KB_BUILD_SEPARABLE) != 0;
    if (separable) {
        factors.assign((size_t)num_times * 2 * DM, 0.0f);
        for (int t = 0; t < num_times && separable; ++t) {
            separable = factor_kernel(psf_host + offs[t], psf_dims[t], &factors[(size_t)t * 2 * DM],
                                      &factors[(size_t)t * 2 * DM + DM]);
        }
    }
    // one kernel size 3 .. 9 for every epoch (the rule): the strip kernels; else the general 32 x 8 tile kernels
    int strip_dim = psf_dims[0];
    for (int t = 1; t < num_times; ++t) {
        if (psf_dims[t] != strip_dim) strip_dim = 0;
    }
    if (strip_dim < 3 || strip_dim > 9 || (build_flags & KB_BUILD_GENERAL_TILES) != 0) strip_dim = 0;
    // every epoch's kernel reads the same top to bottom as bottom to top, bit for bit: the strip kernel multiplies a sample once
    // for the two kernel rows that mirror each other (strip_pass, VSYM)
    bool rows_mirror = strip_dim != 0;
    for (int t = 0; t < num_times && rows_mirror; ++t) {
        const float* kt = packed.data() + offs[t];
        for (int j = 0; j < strip_dim / 2 && rows_mirror; ++j) {
            rows_mirror = std::memcmp(kt + (size_t)j * strip_dim, kt + (size_t)(strip_dim - 1 - j) * strip_dim,
                                      sizeof(float) * (size_t)strip_dim) == 0;
        }
    }
    if (std::getenv("KBMOD_BUILD_NO_MIRROR") != nullptr) rows_mirror = false;  // (timing comparisons, tests)
    const int pitch = CONV_BX + 2 * max_radius, rows = CONV_BY + 2 * max_radius;
    const size_t lds = separable ? sizeof(float) * ((size_t)4 * pitch * rows + (size_t)4 * rows * CONV_BX + 4 * DM)
                                 : conv_lds_bytes(max_radius, true);
    if (lds > 160 * 1024) return fail("PSF radius too large for the LDS-tiled convolution.");

    DeviceBuffer d_psf, d_off, d_dim, d_tot, d_stage, d_minmax, d_sep, d_sci, d_var, d_pairs;
    KB_HIP_TRY(hipMalloc(&d_psf.p, packed.size() * sizeof(float)));
    KB_HIP_TRY(hipMalloc(&d_off.p, offs.size() * sizeof(int)));
    KB_HIP_TRY(hipMalloc(&d_dim.p, (size_t)num_times * sizeof(int)));
    KB_HIP_TRY(hipMalloc(&d_tot.p, totals.size() * sizeof(float)));
    KB_HIP_TRY(hipMemcpyAsync(d_psf.p, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice, stream));
    KB_HIP_TRY(hipMemcpyAsync(d_off.p, offs.data(), offs.size() * sizeof(int), hipMemcpyHostToDevice, stream));
    KB_HIP_TRY(hipMemcpyAsync(d_dim.p, psf_dims, (size_t)num_times * sizeof(int), hipMemcpyHostToDevice, stream));
    KB_HIP_TRY(hipMemcpyAsync(d_tot.p, totals.data(), totals.size() * sizeof(float), hipMemcpyHostToDevice, stream));
    {   // (k, k^2) pairs for the strip kernel's packed multiplies
        std::vector<float> pairs_host(2 * (size_t)sq_base);
        for (int e = 0; e < sq_base; ++e) {
            pairs_host[2 * (size_t)e] = packed[e];
            pairs_host[2 * (size_t)e + 1] = packed[(size_t)sq_base + e];
        }
        KB_HIP_TRY(hipMalloc(&d_pairs.p, pairs_host.size() * sizeof(float)));
        KB_HIP_TRY(hipMemcpy(d_pairs.p, pairs_host.data(), pairs_host.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (separable && strip_dim != 0) {
        // the strip kernel's layout: per epoch (u_j, u_j^2)[dim] then (w_i, w_i^2)[dim]  (DM == dim here)
        std::vector<float> fp((size_t)num_times * 4 * DM);
        for (int t = 0; t < num_times; ++t) {
            for (int e = 0; e < 2 * DM; ++e) {
                const float f = factors[(size_t)t * 2 * DM + e];
                fp[(size_t)t * 4 * DM + 2 * (size_t)e] = f;
                fp[(size_t)t * 4 * DM + 2 * (size_t)e + 1] = f * f;
            }
        }
        factors.swap(fp);
    }
    if (separable) {
        KB_HIP_TRY(hipMalloc(&d_sep.p, factors.size() * sizeof(float)));
        KB_HIP_TRY(hipMemcpyAsync(d_sep.p, factors.data(), factors.size() * sizeof(float), hipMemcpyHostToDevice, stream));
    }

    void* final_arr = nullptr;
    KB_HIP_TRY(hipMalloc(&final_arr, meta_out->total_array_size));
    DeviceBuffer final_guard;
    final_guard.p = final_arr;  // released on any early return

    float* pairs = reinterpret_cast<float*>(final_arr);
    if (encoded) {
        KB_HIP_TRY(hipMalloc(&d_stage.p, n_pix * 2 * sizeof(float)));
        pairs = reinterpret_cast<float*>(d_stage.p);
        KB_HIP_TRY(hipMalloc(&d_minmax.p, 4 * sizeof(unsigned)));
        const unsigned init[4] = {0xffffffffu, 0u, 0xffffffffu, 0u};
        KB_HIP_TRY(hipMemcpyAsync(d_minmax.p, init, sizeof(init), hipMemcpyHostToDevice, stream));
    }
    if (from_host) {
        KB_HIP_TRY(hipMalloc(&d_sci.p, n_pix * sizeof(float)));
        KB_HIP_TRY(hipMalloc(&d_var.p, n_pix * sizeof(float)));
        sci_dev = reinterpret_cast<const float*>(d_sci.p);
        var_dev = reinterpret_cast<const float*>(d_var.p);
    }

    ConvArgs a;
    a.in0 = sci_dev;
    a.in1 = var_dev;
    a.out_plain = nullptr;
    a.out_pairs = pairs;
    a.psf = reinterpret_cast<const float*>(d_psf.p);
    a.psf_pairs = reinterpret_cast<const float*>(d_pairs.p);
    a.psf_off = reinterpret_cast<const int*>(d_off.p);
    a.psf_dim = reinterpret_cast<const int*>(d_dim.p);
    a.psf_tot = reinterpret_cast<const float*>(d_tot.p);
    a.sq_base = sq_base;
    a.W = width;
    a.H = height;
    a.T = num_times;
    a.t0 = 0;
    a.max_radius = max_radius;
    // The array is the CPU StackSearch's array: an empty PSF footprint gives NaN (image_utils_cpp.cpp:60-61).
    // KB_BUILD_EMPTY_IS_ZERO reproduces the reference's own device builder instead (0.0, image_kernels.cu:61).
    a.empty_is_nan = (build_flags & KB_BUILD_EMPTY_IS_ZERO) ? 0 : 1;
    a.minmax = reinterpret_cast<unsigned*>(d_minmax.p);
    a.sep = reinterpret_cast<const float*>(d_sep.p);

    if (lds > 64 * 1024) {
        KB_HIP_TRY(hipFuncSetAttribute(separable ? reinterpret_cast<const void*>(&kb_conv_sep_kernel)
                                                 : reinterpret_cast<const void*>(&kb_conv_kernel<1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const size_t strip_lds = sizeof(float) * 2 * (size_t)(STRIP_BX + 8) * (STRIP_BY + strip_dim - 1);
    auto launch_strip = [&](int nt, hipStream_t s, const ConvArgs& c) {
        const int tiles_x = (width + STRIP_BX - 1) / STRIP_BX, tiles_y = (height + STRIP_BY - 1) / STRIP_BY;
        const int n_tiles = tiles_x * tiles_y * nt;
        const dim3 grid((unsigned)(((n_tiles + 7) / 8) * 8));  // (whole rounds of the eight XCDs)
        auto go = [&](auto dim_tag) {
            constexpr int D = decltype(dim_tag)::value;
            if (separable) {
                hipLaunchKernelGGL(kb_psi_phi_strip_sep_kernel<D>, grid, dim3(256), strip_lds, s, c, tiles_x, tiles_y, n_tiles);
            } else if (rows_mirror && D <= 7) {  // (9 x 9: the allocator puts the mirrored form's sums into scratch memory)
                if constexpr (D <= 7) {
                    hipLaunchKernelGGL((kb_psi_phi_strip_kernel<D, true>), grid, dim3(256), strip_lds, s, c, tiles_x, tiles_y, n_tiles);
                }
            } else {
                hipLaunchKernelGGL((kb_psi_phi_strip_kernel<D, false>), grid, dim3(256), strip_lds, s, c, tiles_x, tiles_y, n_tiles);
            }
        };
        switch (strip_dim) {
            case 3:
                go(std::integral_constant<int, 3>{});
                break;
            case 5:
                go(std::integral_constant<int, 5>{});
                break;
            case 7:
                go(std::integral_constant<int, 7>{});
                break;
            default:
                go(std::integral_constant<int, 9>{});
                break;
        }
    };
    auto launch_epochs = [&](int t0, int nt, hipStream_t s) {
        ConvArgs c = a;
        c.t0 = t0;
        if (strip_dim != 0) {
            launch_strip(nt, s, c);
            return;
        }
        const dim3 grid((width + CONV_BX - 1) / CONV_BX, (height + CONV_BY - 1) / CONV_BY, nt);
        if (separable) {
            hipLaunchKernelGGL(kb_conv_sep_kernel, grid, dim3(256), lds, s, c);
        } else {
            hipLaunchKernelGGL((kb_conv_kernel<1>), grid, dim3(256), lds, s, c);
        }
    };

    if (!from_host) {
        EventTimer timer(stream, true);  // (the build synchronises the stream at its end anyway)
        timer.begin();
        launch_epochs(0, num_times, stream);
        KB_HIP_TRY(hipGetLastError());
        g_last_build_kernel_ms = timer.end();
    } else {
        // chunks of whole epochs, at most ~16 MiB per stack and chunk.  Where the caller's stacks are page-locked
        // -- allocated pinned (hipHostMalloc, a torch pinned tensor) or registered, which KB_BUILD_REGISTER_HOST does
        // here for the duration of the build -- every chunk is ONE DMA out of the caller's memory; otherwise chunk c is
        // copied into the pinned buffer c % 2 by this thread, sent by the copy engine, and correlated behind its arrival.
        std::lock_guard<std::mutex> lock(g_stage_mutex);
        const int chunk_epochs = (int)std::max<size_t>(1, std::min<size_t>((size_t)num_times, (16u << 20) / (img * sizeof(float))));
        const size_t chunk_bytes = (size_t)chunk_epochs * img * sizeof(float);
        auto page_locked = [](const void* p) {
            hipPointerAttribute_t attr;
            if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
                (void)hipGetLastError();  // plain pageable memory is reported as an error
                return false;
            }
            return attr.type == hipMemoryTypeHost;
        };
        bool direct = page_locked(sci_host) && page_locked(var_host);
        bool registered_here[2] = {false, false};
        if (!direct && (build_flags & KB_BUILD_REGISTER_HOST) != 0) {
            const size_t whole = n_pix * sizeof(float);
            registered_here[0] = hipHostRegister(const_cast<float*>(sci_host), whole, hipHostRegisterDefault) == hipSuccess;
            registered_here[1] = registered_here[0] &&
                                 hipHostRegister(const_cast<float*>(var_host), whole, hipHostRegisterDefault) == hipSuccess;
            direct = registered_here[0] && registered_here[1];
            if (!direct) {  // (could not be locked: the staged path still works)
                (void)hipGetLastError();
                if (registered_here[0]) (void)hipHostUnregister(const_cast<float*>(sci_host));
                registered_here[0] = registered_here[1] = false;
            }
        }
        struct Unregister {
            const float *a, *b;
            bool* on;
            ~Unregister() {
                if (on[0]) (void)hipHostUnregister(const_cast<float*>(a));
                if (on[1]) (void)hipHostUnregister(const_cast<float*>(b));
            }
        } unregister{sci_host, var_host, registered_here};
        if (!direct && g_stage.bytes < 2 * chunk_bytes) {
            for (void*& p : g_stage.p) {
                if (p != nullptr) (void)hipHostFree(p);
                p = nullptr;
            }
            g_stage.bytes = 0;
            KB_HIP_TRY(hipHostMalloc(&g_stage.p[0], 2 * chunk_bytes, hipHostMallocDefault));
            KB_HIP_TRY(hipHostMalloc(&g_stage.p[1], 2 * chunk_bytes, hipHostMallocDefault));
            g_stage.bytes = 2 * chunk_bytes;
        }
        hipStream_t copy_stream = nullptr;
        KB_HIP_TRY(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
        hipEvent_t arrived[2] = {nullptr, nullptr}, consumed[2] = {nullptr, nullptr};
        for (int i = 0; i < 2; ++i) {
            (void)hipEventCreateWithFlags(&arrived[i], hipEventDisableTiming);
            (void)hipEventCreateWithFlags(&consumed[i], hipEventDisableTiming);
        }
        int rc = 0;
        int c = 0;
        for (int t0 = 0; t0 < num_times && rc == 0; t0 += chunk_epochs, ++c) {
            const int nt = std::min(chunk_epochs, num_times - t0);
            const size_t nb = (size_t)nt * img * sizeof(float);
            const int b = c & 1;
            const char *src_sci = reinterpret_cast<const char*>(sci_host + (size_t)t0 * img),
                       *src_var = reinterpret_cast<const char*>(var_host + (size_t)t0 * img);
            if (!direct) {
                if (c >= 2 && hipEventSynchronize(consumed[b]) != hipSuccess) rc = 1;  // the copy two chunks ago has left the buffer
                char* host_buf = reinterpret_cast<char*>(g_stage.p[b]);
                std::memcpy(host_buf, src_sci, nb);
                std::memcpy(host_buf + chunk_bytes, src_var, nb);
                src_sci = host_buf;
                src_var = host_buf + chunk_bytes;
            }
            if (hipMemcpyAsync(reinterpret_cast<float*>(d_sci.p) + (size_t)t0 * img, src_sci, nb, hipMemcpyHostToDevice,
                               copy_stream) != hipSuccess ||
                hipMemcpyAsync(reinterpret_cast<float*>(d_var.p) + (size_t)t0 * img, src_var, nb, hipMemcpyHostToDevice,
                               copy_stream) != hipSuccess) {
                rc = 1;
                break;
            }
            (void)hipEventRecord(consumed[b], copy_stream);
            (void)hipEventRecord(arrived[b], copy_stream);
            (void)hipStreamWaitEvent(stream, arrived[b], 0);
            launch_epochs(t0, nt, stream);
        }
        const hipError_t launch_err = hipGetLastError();
        (void)hipStreamSynchronize(copy_stream);
        for (int i = 0; i < 2; ++i) {
            (void)hipEventDestroy(arrived[i]);
            (void)hipEventDestroy(consumed[i]);
        }
        (void)hipStreamDestroy(copy_stream);
        if (rc != 0 || launch_err != hipSuccess) {
            (void)hipStreamSynchronize(stream);
            return fail(std::string("build_psi_phi: upload failed: ") + hipGetErrorString(launch_err));
        }
    }

    if (encoded) {
        unsigned keys[4];
        KB_HIP_TRY(hipMemcpyAsync(keys, d_minmax.p, sizeof(keys), hipMemcpyDeviceToHost, stream));
        KB_HIP_TRY(hipStreamSynchronize(stream));
        const bool psi_any = keys[0] != 0xffffffffu, phi_any = keys[2] != 0xffffffffu;
        meta_out->psi_min_val = psi_any ? key_float(keys[0]) : FLT_MAX;
        meta_out->psi_max_val = psi_any ? key_float(keys[1]) : -FLT_MAX;
        meta_out->phi_min_val = phi_any ? key_float(keys[2]) : FLT_MAX;
        meta_out->phi_max_val = phi_any ? key_float(keys[3]) : -FLT_MAX;
        meta_out->psi_scale = scale_from_range(meta_out->psi_min_val, meta_out->psi_max_val, meta_out->num_bytes);
        meta_out->phi_scale = scale_from_range(meta_out->phi_min_val, meta_out->phi_max_val, meta_out->num_bytes);
        // psi_phi_array.cpp:150-168 set_psi_scaling / set_phi_scaling validation
        if (meta_out->psi_min_val > meta_out->psi_max_val) {
            return fail("Min value needs to be < max value. Got " + std::to_string(meta_out->psi_min_val) + " and " +
                        std::to_string(meta_out->psi_max_val));
        }
        if (meta_out->phi_min_val > meta_out->phi_max_val) {
            return fail("Min value needs to be < max value. Got " + std::to_string(meta_out->phi_min_val) + " and " +
                        std::to_string(meta_out->phi_max_val));
        }
        if (meta_out->psi_scale <= 0 || meta_out->phi_scale <= 0) {
            return fail("Scale value must be greater than zero.");
        }
        // psi_phi_array.cpp:264-265
        const float psi_safe = (float)((double)meta_out->psi_max_val - (double)meta_out->psi_scale / 100.0);
        const float phi_safe = (float)((double)meta_out->phi_max_val - (double)meta_out->phi_scale / 100.0);
        const int blocks = (int)std::min<size_t>((n_pix + 255) / 256, 256 * 32);
        if (meta_out->num_bytes == 1) {
            hipLaunchKernelGGL((kb_encode_kernel<uchar2>), dim3(blocks), dim3(256), 0, stream,
                               reinterpret_cast<const float2*>(pairs), reinterpret_cast<uchar2*>(final_arr), n_pix,
                               meta_out->psi_min_val, psi_safe, meta_out->psi_scale, meta_out->phi_min_val, phi_safe,
                               meta_out->phi_scale);
        } else {
            hipLaunchKernelGGL((kb_encode_kernel<ushort2>), dim3(blocks), dim3(256), 0, stream,
                               reinterpret_cast<const float2*>(pairs), reinterpret_cast<ushort2*>(final_arr), n_pix,
                               meta_out->psi_min_val, psi_safe, meta_out->psi_scale, meta_out->phi_min_val, phi_safe,
                               meta_out->phi_scale);
        }
        KB_HIP_TRY(hipGetLastError());
    }
    KB_HIP_TRY(hipStreamSynchronize(stream));
    final_guard.p = nullptr;  // ownership passes to the caller
    note_array_built(final_arr, meta_out->total_array_size);  // (the library's own array: kb_common.h)
    *psi_phi_dev_out = final_arr;
    return 0;
}

}  // namespace kb

extern "C" {

float kb_last_build_kernel_ms(void) { return kb::g_last_build_kernel_ms; }

int kb_build_psi_phi_from_device(const float* sci_dev, const float* var_dev, const float* psf_host,
                                 const int32_t* psf_dims, int32_t num_times, int32_t height, int32_t width,
                                 int32_t num_bytes, kb_psi_phi_meta* meta_out, void** psi_phi_dev_out,
                                 void* stream_v) {
    return kb::build_psi_phi(sci_dev, var_dev, nullptr, nullptr, psf_host, psf_dims, num_times, height, width, num_bytes,
                             0, meta_out, psi_phi_dev_out, reinterpret_cast<hipStream_t>(stream_v));
}

int kb_build_psi_phi_from_device_ex(const float* sci_dev, const float* var_dev, const float* psf_host,
                                    const int32_t* psf_dims, int32_t num_times, int32_t height, int32_t width,
                                    int32_t num_bytes, uint32_t build_flags, kb_psi_phi_meta* meta_out,
                                    void** psi_phi_dev_out, void* stream_v) {
    return kb::build_psi_phi(sci_dev, var_dev, nullptr, nullptr, psf_host, psf_dims, num_times, height, width, num_bytes,
                             build_flags, meta_out, psi_phi_dev_out, reinterpret_cast<hipStream_t>(stream_v));
}

int kb_build_psi_phi_from_host_stack(const float* sci_host, const float* var_host, const float* psf_host,
                                     const int32_t* psf_dims, int32_t num_times, int32_t height, int32_t width,
                                     int32_t num_bytes, uint32_t build_flags, kb_psi_phi_meta* meta_out,
                                     void** psi_phi_dev_out) {
    if (sci_host == nullptr || var_host == nullptr) return kb::fail("build_psi_phi: null input pointer");
    return kb::build_psi_phi(nullptr, nullptr, sci_host, var_host, psf_host, psf_dims, num_times, height, width, num_bytes,
                             build_flags, meta_out, psi_phi_dev_out, nullptr);
}

int kb_build_psi_phi_from_host(const float* const* sci_host, const float* const* var_host,
                               const float* psf_host, const int32_t* psf_dims, int32_t num_times,
                               int32_t height, int32_t width, int32_t num_bytes, kb_psi_phi_meta* meta_out,
                               void** psi_phi_dev_out) {
    using namespace kb;
    if (num_times <= 0) return fail("Trying to fill PsiPhi from empty vectors.");
    if (sci_host == nullptr || var_host == nullptr) return fail("build_psi_phi: null input pointer");
    if (kb_device_count() == 0) return fail("GPU is not available for the psi/phi build.");
    const size_t img = (size_t)height * width;
    DeviceBuffer d_sci, d_var;
    KB_HIP_TRY(hipMalloc(&d_sci.p, img * num_times * sizeof(float)));
    KB_HIP_TRY(hipMalloc(&d_var.p, img * num_times * sizeof(float)));
    for (int t = 0; t < num_times; ++t) {
        KB_HIP_TRY(hipMemcpy(reinterpret_cast<float*>(d_sci.p) + img * t, sci_host[t], img * sizeof(float),
                             hipMemcpyHostToDevice));
        KB_HIP_TRY(hipMemcpy(reinterpret_cast<float*>(d_var.p) + img * t, var_host[t], img * sizeof(float),
                             hipMemcpyHostToDevice));
    }
    return kb_build_psi_phi_from_device(reinterpret_cast<const float*>(d_sci.p),
                                        reinterpret_cast<const float*>(d_var.p), psf_host, psf_dims, num_times,
                                        height, width, num_bytes, meta_out, psi_phi_dev_out, nullptr);
}

int kb_device_convolve(const float* src_host, float* dst_host, int width, int height, const float* psf_host,
                       int psf_radius, int empty_is_nan) {
    using namespace kb;
    // image_kernels.cu:70-72
    if (width <= 0) return fail("Invalid width = " + std::to_string(width));
    if (height <= 0) return fail("Invalid height = " + std::to_string(height));
    if (psf_radius < 0) return fail("Invalid PSF radius = " + std::to_string(psf_radius));
    if (src_host == nullptr || dst_host == nullptr || psf_host == nullptr) return fail("deviceConvolve: null pointer");
    if (kb_device_count() == 0) return fail("Unable to perform convolve_image_gpu() without GPU.");

    const int32_t dim = 2 * psf_radius + 1;
    std::vector<float> packed, totals;
    std::vector<int> offs;
    int max_radius = 0, sq_base = 0;
    if (pack_psfs(psf_host, &dim, 1, false, &packed, &offs, &totals, &max_radius, &sq_base)) return 1;
    const size_t lds = conv_lds_bytes(max_radius, false);
    if (lds > 160 * 1024) return fail("PSF radius too large for the LDS-tiled convolution.");

    const size_t n = (size_t)width * height;
    DeviceBuffer d_src, d_dst, d_psf, d_meta;
    KB_HIP_TRY(hipMalloc(&d_src.p, n * sizeof(float)));
    KB_HIP_TRY(hipMalloc(&d_dst.p, n * sizeof(float)));
    KB_HIP_TRY(hipMalloc(&d_psf.p, packed.size() * sizeof(float)));
    KB_HIP_TRY(hipMalloc(&d_meta.p, 16));
    KB_HIP_TRY(hipMemcpy(d_src.p, src_host, n * sizeof(float), hipMemcpyHostToDevice));
    KB_HIP_TRY(hipMemcpy(d_psf.p, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
    const int meta_i[2] = {0, dim};
    KB_HIP_TRY(hipMemcpy(d_meta.p, meta_i, sizeof(meta_i), hipMemcpyHostToDevice));
    KB_HIP_TRY(hipMemcpy(reinterpret_cast<char*>(d_meta.p) + 8, totals.data(), sizeof(float), hipMemcpyHostToDevice));

    ConvArgs a;
    a.in0 = reinterpret_cast<const float*>(d_src.p);
    a.in1 = nullptr;
    a.out_plain = reinterpret_cast<float*>(d_dst.p);
    a.out_pairs = nullptr;
    a.psf = reinterpret_cast<const float*>(d_psf.p);
    a.psf_pairs = nullptr;
    a.psf_off = reinterpret_cast<const int*>(d_meta.p);
    a.psf_dim = reinterpret_cast<const int*>(d_meta.p) + 1;
    a.psf_tot = reinterpret_cast<const float*>(reinterpret_cast<char*>(d_meta.p) + 8);
    a.sq_base = 0;
    a.W = width;
    a.H = height;
    a.T = 1;
    a.t0 = 0;
    a.max_radius = max_radius;
    a.empty_is_nan = empty_is_nan;
    a.minmax = nullptr;
    a.sep = nullptr;
    if (lds > 64 * 1024) {
        KB_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&kb_conv_kernel<0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const dim3 grid((width + CONV_BX - 1) / CONV_BX, (height + CONV_BY - 1) / CONV_BY, 1);
    hipLaunchKernelGGL((kb_conv_kernel<0>), grid, dim3(256), lds, nullptr, a);
    KB_HIP_TRY(hipGetLastError());
    KB_HIP_TRY(hipMemcpy(dst_host, d_dst.p, n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int kb_generate_psi_phi_host(const float* sci_host, const float* var_host, int width, int height,
                             const float* psf_host, int psf_dim, float* psi_out_host, float* phi_out_host) {
    using namespace kb;
    if (sci_host == nullptr || var_host == nullptr || psf_host == nullptr) return fail("generate_psi_phi: null input");
    kb_psi_phi_meta meta;
    void* arr = nullptr;
    const float* sci_list[1] = {sci_host};
    const float* var_list[1] = {var_host};
    const int32_t dims[1] = {psf_dim};
    if (kb_build_psi_phi_from_host(sci_list, var_list, psf_host, dims, 1, height, width, 4, &meta, &arr)) return 1;
    const size_t n = (size_t)width * height;
    std::vector<float> pairs(2 * n);
    const int rc = kb_copy_block_to_cpu(pairs.data(), arr, 2 * n * sizeof(float));
    (void)hipFree(arr);
    if (rc) return rc;
    for (size_t i = 0; i < n; ++i) {
        if (psi_out_host) psi_out_host[i] = pairs[2 * i];
        if (phi_out_host) phi_out_host[i] = pairs[2 * i + 1];
    }
    return 0;
}

}  // extern "C"
