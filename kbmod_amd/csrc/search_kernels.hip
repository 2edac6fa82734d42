// Shift-and-stack trajectory search for MI355X (gfx950).
//
// Replaces kernels/kernels.cu:252-397 of the reference (searchFilterImages +
// deviceSearchFilter) with a design built around one observation: for a fixed
// candidate velocity and epoch, floor(x + vx*t + 0.5) = x + d for every start
// pixel x, i.e. the "gather" is an integer translation of the start tile.
//
//   * kb_shift_table_kernel : per (candidate, epoch) integer shift (dx, dy) in the
//     reference's double arithmetic, PROVEN valid for every start pixel (guard
//     band around the rounding boundary; unprovable entries are flagged and take
//     the exact path), plus per (chunk, epoch) footprint boxes / LDS offsets.
//   * kb_search_lds         : the fast path.  A 256-thread workgroup owns a 64 x 4
//     tile of start pixels (one wavefront per row).  Per chunk of C candidates
//     and per epoch it stages the union footprint of the C translated tiles ONCE
//     from HBM/L2 into LDS -- sanitised: NO_DATA becomes (+0,+0) plus a validity
//     plane -- double-buffered against the compute on the previous epoch, and each
//     wave then reads its C shifted 512-byte rows from LDS (ds_read_b64, address =
//     lane base + scalar offset) and accumulates fp32 sums in strict epoch order.
//   * kb_search_direct      : same tile mapping with direct coalesced global loads
//     (interior / edge / exact-position loop bodies); used when a chunk's
//     footprint does not fit the LDS stage (scattered candidate lists) or a shift
//     could not be proven.
//   Both keep the per-pixel top-K in registers with the reference's swap-down
//   insertion reproduced slot by slot, and re-evaluate the K winners with exact
//   per-lane positions for flux / obs_count (and the sigma-G clipped values).
//
// Numerics: fp32 sums in epoch order, correctly rounded sqrt/divide, double
// position arithmetic without FMA -- results equal the reference's host
// instantiation bit for bit (oracle: oracle/kbmod_oracle.c).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "kb_common.h"
#include "search_math.h"

#pragma clang fp contract(off)

namespace kb {

constexpr int SHIFT_UNSAFE = INT32_MIN;  // dx marker: no uniform shift proven for this (candidate, epoch)
constexpr int TILE_ROWS = 4;             // waves (rows) per 256-thread workgroup
constexpr int CHUNK = 8;                 // candidates accumulated together per wave

// LDS staging (kb_search_lds): per (chunk, epoch) the workgroup stages a slab of
// rows_max(chunk) x LDS_COLS raw pairs -- the union footprint of its 64 x 4 tile
// under the chunk's shifts -- straight from a padded HBM copy of the array into
// LDS with LDS-DMA (global_load_lds_dwordx4, no VGPR round trip), several epochs
// per barrier, double-buffered.
constexpr int LDS_COLS = 88;          // slab pitch in pixels: 64 start columns + up to 24 of dx spread
                                      // (88 * {8,4,2} bytes are multiples of the 16-byte DMA granule)
constexpr int LDS_GROUP_BYTES = 20480;  // one group buffer; two per workgroup = 40 KiB -> 4 workgroups per CU
constexpr int LDS_ALIGN_PX = 8;         // slab origins are multiples of 8 columns of the padded frame
constexpr int LDS_SLOTS = 3;            // 16-byte pieces a thread holds in registers at once; slabs beyond
                                        // 3 x 4 KiB are copied in further, non-overlapped rounds

struct ChunkInfo {
    int dx_min, dx_max, dy_min, dy_max;  // bounding box of the chunk's shifts over all epochs
    int unsafe;                          // any entry flagged SHIFT_UNSAFE
    int lds_ok;                          // every epoch's footprint fits one LDS slab
    int rows_max;                        // TILE_ROWS + largest dy spread of any epoch (slab height)
    int pad;
};

// Per (chunk, epoch) footprint, packed for one 8-byte scalar load:
//   x = (dy_min << 16) | (dx_min & 0xffff)   origin of the staged region relative to the tile
//   y = (rows   << 16) | cols                64 + dx spread, TILE_ROWS + dy spread
using EpochBox = int2;
constexpr int BOX_NOT_STAGED = (int)0x80008000u;  // word 0 of an epoch that kb_search_lds does not stage
constexpr int LDS_OFF_UNSTAGED = -1;
constexpr int LDS_OFF_PER_LANE = -2;
__host__ __device__ __forceinline__ int box_dx(EpochBox b) { return (int)(short)(b.x & 0xffff); }
__host__ __device__ __forceinline__ int box_dy(EpochBox b) { return b.x >> 16; }
__host__ __device__ __forceinline__ int box_cols(EpochBox b) { return b.y & 0xffff; }
__host__ __device__ __forceinline__ int box_rows(EpochBox b) { return b.y >> 16; }

struct SearchArgs {
    const void* psi_phi;
    const double* times;
    const kb_trajectory* cands;
    kb_trajectory* results;
    const int2* table;         // [n_chunks][T][C] integer shifts (dx, dy)
    const ChunkInfo* chunks;   // [n_chunks]
    const EpochBox* boxes;     // [n_chunks][T]
    const int64_t* origins;    // [n_chunks][T] byte offset of the slab origin inside the padded copy, relative
                               // to the tile's own pixel (kb_slab_origin_kernel)
    const int* lds_off;        // [n_chunks][T][C] byte offset of the shifted tile inside plane A
    const int* global_box;     // {dx_min, dx_max, dy_min, dy_max, rows_max} over every (candidate, epoch)
    const void* padded;        // [T][Hp][Wp] raw pairs, apron = NO_DATA (kb_search_lds only)
    int Wp, Hp, px0, py0;      // padded pitch / height, position of image pixel (0,0) inside the padded frame
    const int* n_invalid;      // device counter: NO_DATA pixels inside the image (written by kb_pad_kernel)
    int all_staged;            // every (chunk, epoch) is staged through LDS
    kb_psi_phi_meta meta;
    kb_search_params params;
    int T, W, H;
    int n_cands, n_chunks;
    int sw, sh;
    int tiles_x, tiles_y, n_tiles;
    int K;
    int force_exact;
    int fast_decode;    // uint8/uint16: the fp32-FMA decode was verified bit-identical for every code
    float* sg_scratch;  // sigma-G per-lane scratch, or null
};

// Encoded sample -> float.  The reference decodes in double with two roundings
// (search_math.h decode_code).  (code - 1) * scale is exact in double, so the
// value is fl32(fl64(S)) with S = (code-1)*scale + min exact; a single fp32 FMA
// gives fl32(S).  The host checks all 2^(8*bs)-1 codes of the array's scale
// parameters once per search and enables the FMA form only if every code agrees
// bit for bit (verify_fast_decode); otherwise the double form is used.
__device__ __forceinline__ float decode_fast_or_exact(unsigned code, float scale, float min_val, int fast) {
    if (fast) return fmaf((float)code - 1.0f, scale, min_val);
    return decode_code((float)code, scale, min_val);
}

// ---------------------------------------------------------------------------
// shift table
// ---------------------------------------------------------------------------
// kind: 0 = the shift is uniform over all start pixels; 1 = v*t + 0.5 sits inside the guard band
// of a rounding boundary, each pixel lands on fl - 1, fl or fl + 1; 2 = out of the proven range.
__device__ __forceinline__ int uniform_shift(float v, double t, int* kind) {
    const double a = __dmul_rn((double)v, t);
    const double g = __dadd_rn(a, 0.5);
    const double fl = floor(g);
    const double frac = g - fl;
    // Guard band 2^-20 around the rounding boundary and |a| < 2^22: with start
    // coordinates |x| < 2^22 the two extra roundings of (x + a) + 0.5 move the
    // value by < 2^-28, so floor() cannot change ("shift table").
    if (!(fabs(a) < 4194304.0)) {
        *kind = 2;
        return 0;
    }
    if (!(frac >= 9.5367431640625e-07 && frac <= 1.0 - 9.5367431640625e-07)) *kind = max(*kind, 1);
    return (int)fl;
}

template <int C>
__global__ __launch_bounds__(256) void kb_shift_table_kernel(const kb_trajectory* __restrict__ cands,
                                                             const double* __restrict__ times, int n_cands,
                                                             int T, int2* __restrict__ table,
                                                             ChunkInfo* __restrict__ chunks,
                                                             EpochBox* __restrict__ boxes,
                                                             int* __restrict__ lds_off,
                                                             int* __restrict__ n_not_lds,
                                                             int* __restrict__ global_box) {
    // One workgroup per chunk, one thread per epoch (strided): the thread owns the
    // C shifts of its epoch, their bounding box and the LDS offsets derived from it.
    const int chunk = blockIdx.x;
    int dx_min = INT32_MAX, dx_max = INT32_MIN, dy_min = INT32_MAX, dy_max = INT32_MIN, any_unsafe = 0, lds_bad = 0;
    int rows_max = TILE_ROWS;
    int sx_min = INT32_MAX, sx_max = INT32_MIN, sy_min = INT32_MAX, sy_max = INT32_MIN;  // staged epochs only
    int n_per_lane = 0;
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const double tm = times[t];
        int2 sh[C];
        int ex0 = INT32_MAX, ex1 = INT32_MIN, ey0 = INT32_MAX, ey1 = INT32_MIN;
        bool epoch_unsafe = false;  // some candidate has no uniform shift here
        bool epoch_wild = false;    // ... and not even a bounded one
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int ci = chunk * C + c;
            sh[c] = make_int2(0, 0);
            if (ci < n_cands) {
                int kx = 0, ky = 0;
                sh[c].x = uniform_shift(cands[ci].vx, tm, &kx);
                sh[c].y = uniform_shift(cands[ci].vy, tm, &ky);
                if (kx == 2 || ky == 2) {
                    epoch_wild = true;
                } else {
                    // footprint: one pixel of slack on an axis whose shift is only known to +-1
                    ex0 = min(ex0, sh[c].x - (kx ? 1 : 0));
                    ex1 = max(ex1, sh[c].x + (kx ? 1 : 0));
                    ey0 = min(ey0, sh[c].y - (ky ? 1 : 0));
                    ey1 = max(ey1, sh[c].y + (ky ? 1 : 0));
                }
                if (kx != 0 || ky != 0) {
                    sh[c].x = SHIFT_UNSAFE;
                    epoch_unsafe = true;
                }
            }
        }
        // The slab starts on a multiple of LDS_ALIGN_PX columns of the padded frame (the host places the
        // image so that x_start_min + px0 is one): 16-byte pieces of a slab row are then 16-byte aligned
        // in HBM, 64-byte aligned for float pairs.  Measured on MI355X this made NO difference (8.46 ms
        // before and after); it is kept because it costs nothing and removes one variable, not because
        // the memory pipe was shown to need it.
        if (ex0 <= ex1) ex0 -= ((ex0 % LDS_ALIGN_PX) + LDS_ALIGN_PX) % LDS_ALIGN_PX;
        // A slab of (TILE_ROWS + dy spread) x LDS_COLS 8-byte pairs must fit one group buffer.
        const bool fits = !epoch_wild && ex0 <= ex1 && (ex1 - ex0) <= (LDS_COLS - WAVE) &&
                          (TILE_ROWS + ey1 - ey0) * LDS_COLS * 8 <= LDS_GROUP_BYTES && ex0 > -30000 && ex1 < 30000 &&
                          ey0 > -30000 && ey1 < 30000;
        EpochBox box = make_int2(0, (TILE_ROWS << 16) | WAVE);
        if (fits) {
            box.x = (ey0 << 16) | (ex0 & 0xffff);
            box.y = ((TILE_ROWS + ey1 - ey0) << 16) | (WAVE + ex1 - ex0);
            rows_max = max(rows_max, TILE_ROWS + ey1 - ey0);
            sx_min = min(sx_min, ex0);
            sx_max = max(sx_max, ex1);
            sy_min = min(sy_min, ey0);
            sy_max = max(sy_max, ey1);
        } else {
            box.x = BOX_NOT_STAGED;  // kb_search_lds evaluates this epoch per lane from the array itself
            lds_bad += 1;
        }
        boxes[(size_t)chunk * T + t] = box;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const size_t e = ((size_t)chunk * T + t) * C + c;
            table[e] = sh[c];
            const bool real = (chunk * C + c) < n_cands;
            // >= 0: slab offset of the uniformly shifted tile; LDS_OFF_PER_LANE: staged, but the lanes
            // find their own pixel inside the slab; LDS_OFF_UNSTAGED: not staged at all
            lds_off[e] = !fits ? LDS_OFF_UNSTAGED
                               : (epoch_unsafe ? LDS_OFF_PER_LANE
                                               : (real ? ((sh[c].y - ey0) * LDS_COLS + (sh[c].x - ex0)) * 8 : 0));
        }
        if (epoch_unsafe) any_unsafe = 1;
        if (epoch_unsafe && fits) n_per_lane += 1;
        if (ex0 <= ex1) {
            dx_min = min(dx_min, ex0);
            dx_max = max(dx_max, ex1);
            dy_min = min(dy_min, ey0);
            dy_max = max(dy_max, ey1);
        }
    }
    __shared__ int red[12][256];
    red[0][threadIdx.x] = dx_min;
    red[1][threadIdx.x] = dx_max;
    red[2][threadIdx.x] = dy_min;
    red[3][threadIdx.x] = dy_max;
    red[4][threadIdx.x] = any_unsafe;
    red[5][threadIdx.x] = lds_bad;
    red[6][threadIdx.x] = rows_max;
    red[7][threadIdx.x] = sx_min;
    red[8][threadIdx.x] = sx_max;
    red[9][threadIdx.x] = sy_min;
    red[10][threadIdx.x] = sy_max;
    red[11][threadIdx.x] = n_per_lane;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red[0][threadIdx.x] = min(red[0][threadIdx.x], red[0][threadIdx.x + s]);
            red[1][threadIdx.x] = max(red[1][threadIdx.x], red[1][threadIdx.x + s]);
            red[2][threadIdx.x] = min(red[2][threadIdx.x], red[2][threadIdx.x + s]);
            red[3][threadIdx.x] = max(red[3][threadIdx.x], red[3][threadIdx.x + s]);
            red[4][threadIdx.x] |= red[4][threadIdx.x + s];
            red[5][threadIdx.x] += red[5][threadIdx.x + s];
            red[6][threadIdx.x] = max(red[6][threadIdx.x], red[6][threadIdx.x + s]);
            red[7][threadIdx.x] = min(red[7][threadIdx.x], red[7][threadIdx.x + s]);
            red[8][threadIdx.x] = max(red[8][threadIdx.x], red[8][threadIdx.x + s]);
            red[9][threadIdx.x] = min(red[9][threadIdx.x], red[9][threadIdx.x + s]);
            red[10][threadIdx.x] = max(red[10][threadIdx.x], red[10][threadIdx.x + s]);
            red[11][threadIdx.x] += red[11][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        ChunkInfo ci;
        ci.dx_min = red[0][0];
        ci.dx_max = red[1][0];
        ci.dy_min = red[2][0];
        ci.dy_max = red[3][0];
        ci.unsafe = red[4][0];
        ci.lds_ok = (red[5][0] == 0) ? 1 : 0;
        ci.rows_max = red[6][0];
        ci.pad = 0;
        chunks[chunk] = ci;
        if (red[5][0] != 0) atomicAdd(n_not_lds, red[5][0]);  // (chunk, epoch) pairs that are not staged
        if (red[11][0] != 0) atomicAdd(&global_box[5], red[11][0]);  // ... staged, but summed per lane
        atomicMax(&global_box[4], ci.rows_max);
        if (red[7][0] <= red[8][0]) {  // shift box of the staged epochs: sizes the apron of the padded copy
            atomicMin(&global_box[0], red[7][0]);
            atomicMax(&global_box[1], red[8][0]);
            atomicMin(&global_box[2], red[9][0]);
            atomicMax(&global_box[3], red[10][0]);
        }
    }
}

// ---------------------------------------------------------------------------
// sample decode
// ---------------------------------------------------------------------------
template <int NB>
struct RawPair;
template <>
struct RawPair<4> {
    using type = float2;
    __device__ static __forceinline__ type invalid() { return make_float2(NAN, NAN); }
    __device__ static __forceinline__ void decode(type r, const SearchArgs&, float* psi, float* phi) {
        *psi = r.x;
        *phi = r.y;
    }
};
template <>
struct RawPair<2> {
    using type = ushort2;
    __device__ static __forceinline__ type invalid() { return make_ushort2(0, 0); }
    __device__ static __forceinline__ void decode(type r, const SearchArgs& a, float* psi, float* phi) {
        *psi = (r.x == 0) ? NAN : decode_code((float)r.x, a.meta.psi_scale, a.meta.psi_min_val);
        *phi = (r.y == 0) ? NAN : decode_code((float)r.y, a.meta.phi_scale, a.meta.phi_min_val);
    }
};
template <>
struct RawPair<1> {
    using type = uchar2;
    __device__ static __forceinline__ type invalid() { return make_uchar2(0, 0); }
    __device__ static __forceinline__ void decode(type r, const SearchArgs& a, float* psi, float* phi) {
        *psi = (r.x == 0) ? NAN : decode_code((float)r.x, a.meta.psi_scale, a.meta.psi_min_val);
        *phi = (r.y == 0) ? NAN : decode_code((float)r.y, a.meta.phi_scale, a.meta.phi_min_val);
    }
};

// Fast formats (NB = 20 / 10): uint16 / uint8 with the verified fp32-FMA decode and
// validity taken from the codes alone (the host also verified that every code
// decodes to a finite value).
template <>
struct RawPair<20> {
    using type = ushort2;
    __device__ static __forceinline__ type invalid() { return make_ushort2(0, 0); }
    __device__ static __forceinline__ void decode(type r, const SearchArgs& a, float* psi, float* phi) {
        const bool ok = (r.x != 0) && (r.y != 0);
        *psi = ok ? fmaf((float)r.x - 1.0f, a.meta.psi_scale, a.meta.psi_min_val) : NAN;
        *phi = fmaf((float)r.y - 1.0f, a.meta.phi_scale, a.meta.phi_min_val);
    }
};
template <>
struct RawPair<10> {
    using type = uchar2;
    __device__ static __forceinline__ type invalid() { return make_uchar2(0, 0); }
    __device__ static __forceinline__ void decode(type r, const SearchArgs& a, float* psi, float* phi) {
        const bool ok = (r.x != 0) && (r.y != 0);
        *psi = ok ? fmaf((float)r.x - 1.0f, a.meta.psi_scale, a.meta.psi_min_val) : NAN;
        *phi = fmaf((float)r.y - 1.0f, a.meta.phi_scale, a.meta.phi_min_val);
    }
};
// Bytes per encoded value of a format tag.
__host__ __device__ constexpr int fmt_bytes(int nb) { return nb >= 10 ? nb / 10 : nb; }

template <int NB>
__device__ __forceinline__ void load_sample(const char* base, uint32_t voff, const SearchArgs& a, float* psi,
                                            float* phi) {
    using R = RawPair<NB>;
    R::decode(*reinterpret_cast<const typename R::type*>(base + voff), a, psi, phi);
}

__device__ __forceinline__ void accumulate(float psi, float phi, bool ok, float& ps, float& ph, int& n) {
    const bool valid = ok && __builtin_isfinite(psi) && __builtin_isfinite(phi);
    // Adding +0.0f is the identity here: the running sums start at +0.0f and can
    // therefore never be -0.0f.
    ps += valid ? psi : 0.0f;
    ph += valid ? phi : 0.0f;
    n += valid ? 1 : 0;
}

// ---------------------------------------------------------------------------
// shared pieces of both search kernels
// ---------------------------------------------------------------------------
struct TileCoords {
    int tx, ty, lane, wv;
    int x_i, y_i, x, y, tile_x0, tile_y0;
    bool row_active;
};

__device__ __forceinline__ TileCoords tile_coords(const SearchArgs& a) {
    // XCD-aware tile order: workgroup b runs on XCD (b % 8); give each XCD a
    // contiguous band of tiles so that its private L2 sees one image region.
    TileCoords c;
    const int b = blockIdx.x;
    const int xcd = b & 7, local = b >> 3;
    const int q = a.n_tiles >> 3, r = a.n_tiles & 7;
    const int tile = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    c.ty = tile / a.tiles_x;
    c.tx = tile - c.ty * a.tiles_x;
    c.lane = threadIdx.x & (WAVE - 1);
    c.wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.y_i = c.ty * TILE_ROWS + c.wv;
    c.x_i = c.tx * WAVE + c.lane;
    c.x = c.x_i + a.params.x_start_min;
    c.y = c.y_i + a.params.y_start_min;
    c.tile_x0 = c.tx * WAVE + a.params.x_start_min;
    c.tile_y0 = c.ty * TILE_ROWS + a.params.y_start_min;
    c.row_active = c.y_i < a.sh;
    return c;
}

template <int KS>
struct TopK {
    float lh[KS];
    int id[KS];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            lh[s] = -FLT_MAX;
            id[s] = -1;
        }
    }
    // kernels.cu:323-330: strict '>' swap-down, reproduced slot by slot.
    __device__ __forceinline__ void insert(float cand_lh, int cand) {
        if (cand_lh > lh[KS - 1]) {
            float cl = cand_lh;
            int cid = cand;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bool g = cl > lh[s];
                const float tl = lh[s];
                const int ti = id[s];
                lh[s] = g ? cl : tl;
                id[s] = g ? cid : ti;
                cl = g ? tl : cl;
                cid = g ? ti : cid;
            }
        }
    }
};

__device__ __forceinline__ SigmaGScratch<WAVE> make_scratch(const SearchArgs& a, const TileCoords& tc) {
    // Lane-interleaved sigma-G scratch: element i of this lane at base[i * 64].
    SigmaGScratch<WAVE> s;
    const size_t wave_id = (size_t)blockIdx.x * TILE_ROWS + tc.wv;
    float* base = a.sg_scratch + wave_id * (size_t)(4 * a.T) * WAVE + tc.lane;
    s.psi.p = base;
    s.phi.p = base + (size_t)a.T * WAVE;
    s.lc.p = base + (size_t)2 * a.T * WAVE;
    s.idx.p = reinterpret_cast<int*>(base + (size_t)3 * a.T * WAVE);
    return s;
}

// Sigma-G clip of ONE trajectory by the whole wavefront (T <= 64): lane t gathers epoch t, the valid
// samples' psi/phi ratios are sorted across the lanes (bitonic network on 64-bit keys
// (ordered ratio, epoch)), the percentile bounds and keep range follow kernels.cu:77-147, and the
// clipped sums are accumulated in sorted order, one fp32 add after the other, exactly like the
// per-lane code of evaluate_trajectory_full.  The reference's exchange sort leaves a particular (not
// stable) permutation among EQUAL ratios, which decides their summation order: when two valid
// ratios are equal the function declines (returns false) and the caller runs the literal per-lane
// code.  All 64 lanes must be active; x, y, vx, vy are wave-uniform.
__device__ __forceinline__ uint32_t ratio_sort_key(float v) {
    const uint32_t b = __float_as_uint((v == 0.0f) ? 0.0f : v);  // -0 and +0 compare equal
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ratio_from_key(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ float lane_value(float v, int lane) {  // lane: wave-uniform
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__device__ __forceinline__ bool clip_lh_wave(const kb_psi_phi_meta& meta, const void* __restrict__ psi_phi,
                                          const double* __restrict__ times, float sgl0, float sgl1, float sigmag_coeff,
                                          int x, int y, float vx, float vy, float* lh_out) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int T = (int)meta.num_times;
    float psi = NAN, phi = NAN;
    if (lane < T) {
        const double t = times[lane];
        int cx, cy;
        const bool okx = predict_index(x, vx, t, &cx);
        const bool oky = predict_index(y, vy, t, &cy);
        if (okx && oky) read_psi_phi(meta, psi_phi, (uint64_t)lane, cy, cx, &psi, &phi);
    }
    const bool valid = __builtin_isfinite(psi) && __builtin_isfinite(phi);
    const int n = __popcll(__ballot(valid));
    if (n == 0) return false;
    const float lc = valid ? ((phi != 0.0f) ? (psi / phi) : 0.0f) : 0.0f;
    uint32_t khi = valid ? ratio_sort_key(lc) : 0xffffffffu;  // invalid samples sort behind every ratio
    uint32_t klo = (uint32_t)lane;
    for (int k = 2; k <= WAVE; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const uint32_t phi_k = __shfl_xor(khi, j), plo_k = __shfl_xor(klo, j);
            const bool mine_less = (khi < phi_k) || (khi == phi_k && klo < plo_k);
            const bool keep_min = ((lane & k) == 0) == ((lane & j) == 0);
            if (keep_min != mine_less) {
                khi = phi_k;
                klo = plo_k;
            }
        }
    }
    // lane i now holds the i-th smallest ratio and the epoch it came from
    const float sv = ratio_from_key(khi);
    const float sv_next = __shfl_down(sv, 1);
    if (__ballot((lane + 1 < n) && (sv == sv_next)) != 0) return false;  // equal ratios: literal code decides
    const float spsi = __shfl(psi, (int)klo), sphi = __shfl(phi, (int)klo);

    if ((double)sgl0 < 0.0001) sgl0 = (float)0.0001;
    if ((double)sgl1 > 0.9999) sgl1 = (float)0.9999;
    int pct_L = (int)((double)ceilf((float)n * sgl0) + 0.001) - 1;
    pct_L = (pct_L < 0) ? 0 : pct_L;
    pct_L = (pct_L >= n) ? (n - 1) : pct_L;
    int pct_H = (int)((double)ceilf((float)n * sgl1) + 0.001) - 1;
    pct_H = (pct_H < 0) ? 0 : pct_H;
    pct_H = (pct_H >= n) ? (n - 1) : pct_H;
    int median_ind = (int)(ceil((double)n * 0.5) + 0.001) - 1;
    median_ind = (median_ind < 0) ? 0 : median_ind;
    median_ind = (median_ind >= n) ? (n - 1) : median_ind;
    pct_L = __builtin_amdgcn_readfirstlane(pct_L);
    pct_H = __builtin_amdgcn_readfirstlane(pct_H);
    median_ind = __builtin_amdgcn_readfirstlane(median_ind);
    const float sigma_g = sigmag_coeff * (lane_value(sv, pct_H) - lane_value(sv, pct_L));
    const float wsg = 2.0f * sigma_g;
    const float vmed = lane_value(sv, median_ind);
    const float min_value = vmed - wsg;
    const float max_value = vmed + wsg;
    // the ratios are ascending, so both tests are true on a prefix of the lanes: the reference's two
    // linear scans (kernels.cu:136-146) become population counts
    const int below = __popcll(__ballot((lane < n) && (sv < min_value)));
    const int upto = __popcll(__ballot((lane < n) && (sv <= max_value)));
    const int min_keep = min(below, median_ind);
    const int max_keep = max(median_ind + 1, upto) - 1;
    float new_psi = 0.0f, new_phi = 0.0f;
    for (int i = min_keep; i <= max_keep; ++i) {  // sorted-value order
        new_psi += lane_value(spsi, i);
        new_phi += lane_value(sphi, i);
    }
    *lh_out = lh_from_sums(new_psi, new_phi);
    return true;
}

// Threshold / sigma-G / insertion of one chunk's C finished candidates.
template <int KS, int C, bool SIGMAG>
__device__ __forceinline__ void finish_chunk(const SearchArgs& a, const TileCoords& tc, int chunk,
                                             const float (&ps)[C], const float (&ph)[C], const int (&cnt)[C],
                                             TopK<KS>& top, const SigmaGScratch<WAVE>& scratch) {
    float lh[C];
    bool take[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        lh[c] = lh_from_sums(ps[c], ph[c]);
        take[c] = !(cnt[c] < a.params.min_observations);
    }
    if constexpr (SIGMAG) {
        // kernels.cu:201-203: only trajectories that pass the unclipped thresholds are clipped (rare:
        // min_lh rejects the noise); the rest either fail kernels.cu:318-320 or are the obs_count == 0
        // corner.  Up to 64 epochs the wavefront clips them one at a time, together (clip_lh_wave).
        uint64_t need[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const bool real = (chunk * C + c) < a.n_cands;  // uniform
            need[c] = __ballot(real && take[c] && (cnt[c] != 0) && !(lh[c] < a.params.min_lh));
        }
        for (int c = 0; c < C; ++c) {  // not unrolled: one copy of the clip code
            uint64_t m = 0;
#pragma unroll
            for (int cc = 0; cc < C; ++cc) m = (cc == c) ? need[cc] : m;
            if (m == 0) continue;  // uniform
            const int cand = chunk * C + c;
            const float vx = a.cands[cand].vx, vy = a.cands[cand].vy;
            float clipped = 0.0f;
#ifndef KB_COOP_MAX
#define KB_COOP_MAX 64  // ablation knob: clip cooperatively only when at most this many lanes need it
#endif
            // Measured on cfg2 + sigma-G (float array): every clip by the whole wave 54 ms, every clip by its
            // own lane 295 ms, mixtures in between.  Quantised (uint8/uint16) arrays produce equal ratios all
            // the time and end up in the per-lane code anyway (300 ms).
            const bool cooperative = a.T <= WAVE && __popcll(m) <= KB_COOP_MAX;
            uint64_t literal = cooperative ? 0 : m;  // lanes that run the literal per-lane code (all at once)
            while (cooperative && m != 0) {
                const int L = __ffsll((unsigned long long)m) - 1;
                m &= m - 1;
                float r = 0.0f;
                if (clip_lh_wave(a.meta, a.psi_phi, a.times, a.params.sgl_L, a.params.sgl_H, a.params.sigmag_coeff,
                                 tc.tile_x0 + L, tc.y, vx, vy, &r)) {
                    if (tc.lane == L) clipped = r;
                } else {
                    literal |= 1ull << L;  // equal ratios: the exchange sort's own order decides
                }
            }
            if ((literal >> tc.lane) & 1) {
                kb_trajectory trj;
                trj.x = tc.x;
                trj.y = tc.y;
                trj.vx = vx;
                trj.vy = vy;
                evaluate_trajectory_full<WAVE>(a.meta, a.psi_phi, a.times, a.params, &trj, &scratch);
                clipped = trj.lh;
            }
            uint64_t mc = 0;
#pragma unroll
            for (int cc = 0; cc < C; ++cc) mc = (cc == c) ? need[cc] : mc;
            const bool mine = (mc >> tc.lane) & 1;
#pragma unroll
            for (int cc = 0; cc < C; ++cc) {
                if (cc == c && mine) lh[cc] = clipped;
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) take[c] = take[c] && !(lh[c] < a.params.min_lh);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int cand = chunk * C + c;
        if (cand >= a.n_cands) break;  // uniform
#ifndef KB_SKIP_INSERT
        if (take[c]) top.insert(lh[c], cand);
#else
        if (take[c] && lh[c] == 12345.0f) top.insert(lh[c], cand);
#endif
    }
}

// Epilogue: the K winners are re-evaluated with exact per-lane positions to
// produce flux / obs_count (and the clipped values when sigma-G is on); the
// likelihood this yields is bit-identical to the one that won the slot.
template <int KS, bool SIGMAG>
__device__ __forceinline__ void write_results(const SearchArgs& a, const TileCoords& tc, const TopK<KS>& top,
                                              const SigmaGScratch<WAVE>& scratch) {
    if (tc.x_i >= a.sw || !tc.row_active) return;
    kb_trajectory* out = a.results + ((size_t)tc.y_i * a.sw + tc.x_i) * a.K;
    for (int s = 0; s < a.K; ++s) {
        int id_s = -1;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            if (k == s) id_s = top.id[k];
        }
        kb_trajectory res;
        res.x = tc.x;
        res.y = tc.y;
        if (id_s < 0) {  // kernels.cu:293-301 placeholder
            res.vx = 0.0f;
            res.vy = 0.0f;
            res.lh = -FLT_MAX;
            res.flux = 0.0f;
            res.obs_count = 0;
        } else {
            res.vx = a.cands[id_s].vx;
            res.vy = a.cands[id_s].vy;
#ifndef KB_SKIP_EPILOGUE
            evaluate_trajectory_full<WAVE>(a.meta, a.psi_phi, a.times, a.params, &res, SIGMAG ? &scratch : nullptr);
#endif
        }
        out[s] = res;
    }
}

// ---------------------------------------------------------------------------
// direct-load kernel (fallback)
// ---------------------------------------------------------------------------
// MODE 0: interior wave, table shifts, no per-lane bounds test.
// MODE 1: table shifts with per-lane bounds test (image edges / off-image starts).
// MODE 2: exact per-lane double positions (chunks with unproven shifts, or forced).
template <int C, int NB, int MODE>
__device__ __forceinline__ void accumulate_chunk_direct(const SearchArgs& a, int chunk, int x, int y, int pix0,
                                                        float (&ps)[C], float (&ph)[C], int (&cnt)[C]) {
    using R = RawPair<NB>;
    constexpr int BYTES = 2 * fmt_bytes(NB);
    const int2* __restrict__ tab = a.table + (size_t)chunk * a.T * C;
    const uint64_t image_bytes = a.meta.pixels_per_image * (uint64_t)BYTES;
    const char* base = reinterpret_cast<const char*>(a.psi_phi);
#pragma unroll 2
    for (int t = 0; t < a.T; ++t) {
        // Phase 1: all C loads of this epoch are issued before anything consumes them.
        typename R::type raw[C];
        bool ok[C];
        if constexpr (MODE == 2) {
            const double tm = a.times[t];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int ci = min(chunk * C + c, a.n_cands - 1);
                int cx, cy;
                bool in = predict_index(x, a.cands[ci].vx, tm, &cx);
                in = predict_index(y, a.cands[ci].vy, tm, &cy) && in;
                ok[c] = in && ((unsigned)cx < (unsigned)a.W) && ((unsigned)cy < (unsigned)a.H);
                const uint32_t voff = ok[c] ? (uint32_t)(cy * a.W + cx) * (uint32_t)BYTES : 0u;
                raw[c] = *reinterpret_cast<const typename R::type*>(base + voff);
            }
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int2 s = tab[t * C + c];  // wave-uniform -> scalar loads
                if constexpr (MODE == 0) {
                    ok[c] = true;
                    const uint32_t voff = (uint32_t)(pix0 + s.y * a.W + s.x) * (uint32_t)BYTES;
                    raw[c] = *reinterpret_cast<const typename R::type*>(base + voff);
                } else {
                    const int cx = x + s.x, cy = y + s.y;
                    ok[c] = ((unsigned)cx < (unsigned)a.W) && ((unsigned)cy < (unsigned)a.H);
                    const uint32_t voff = ok[c] ? (uint32_t)(cy * a.W + cx) * (uint32_t)BYTES : 0u;
                    raw[c] = *reinterpret_cast<const typename R::type*>(base + voff);
                }
            }
        }
        // Phase 2: decode + accumulate in candidate order (each candidate's sums stay in epoch order).
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float psi, phi;
            R::decode(raw[c], a, &psi, &phi);
            accumulate(psi, phi, ok[c], ps[c], ph[c], cnt[c]);
        }
        base += image_bytes;
    }
}

template <int KS, int C, int NB, bool SIGMAG>
__global__ __launch_bounds__(256, (KS <= 8 ? 4 : (KS <= 16 ? 3 : 2))) void kb_search_direct(const SearchArgs a) {
    const TileCoords tc = tile_coords(a);
    if (!tc.row_active) return;  // whole wave (no barriers in this kernel)
    const int pix0 = tc.y * a.W + tc.x;
    TopK<KS> top;
    top.init();
    SigmaGScratch<WAVE> scratch = {};
    if constexpr (SIGMAG) scratch = make_scratch(a, tc);

    for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
        float ps[C], ph[C];
        int cnt[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            ps[c] = 0.0f;
            ph[c] = 0.0f;
            cnt[c] = 0;
        }
        const ChunkInfo ci = a.chunks[chunk];
        const bool exact = a.force_exact || ci.unsafe;
        const bool interior = (tc.tile_x0 + ci.dx_min >= 0) && (tc.tile_x0 + WAVE - 1 + ci.dx_max < a.W) &&
                              (tc.y + ci.dy_min >= 0) && (tc.y + ci.dy_max < a.H);
        if (exact) {
            accumulate_chunk_direct<C, NB, 2>(a, chunk, tc.x, tc.y, pix0, ps, ph, cnt);
        } else if (interior) {
            accumulate_chunk_direct<C, NB, 0>(a, chunk, tc.x, tc.y, pix0, ps, ph, cnt);
        } else {
            accumulate_chunk_direct<C, NB, 1>(a, chunk, tc.x, tc.y, pix0, ps, ph, cnt);
        }
        finish_chunk<KS, C, SIGMAG>(a, tc, chunk, ps, ph, cnt, top, scratch);
    }
    write_results<KS, SIGMAG>(a, tc, top, scratch);
}

// ---------------------------------------------------------------------------
// LDS-staged kernel (LDS-DMA from a padded copy, several epochs per barrier)
// ---------------------------------------------------------------------------
// The padded copy [T][Hp][Wp] holds the image at (px0, py0) and NO_DATA everywhere
// else, so that no staged slab ever needs a bounds test.  Two forms:
//  * CANON (floats; the default, also for encoded arrays when HBM allows): each
//    sample is decoded once, here, with the search's own decode, and NO_DATA becomes
//    the pair (+0, -0).  Adding either zero to a running sum that started at +0
//    leaves it bit-identical (such a sum is never -0), so the search adds every
//    sample unconditionally -- one packed add for (psi, phi) -- and tests only the
//    marker phi == -0 for obs_count.  A valid phi of -0 becomes +0, equally neutral.
//  * encoded (NB bytes per value, apron = code 0): the search decodes per sample.
// n_invalid counts the NO_DATA pixels inside the image.
template <int NB, bool CANON>
__global__ __launch_bounds__(256) void kb_pad_kernel(const SearchArgs a, void* __restrict__ padded,
                                                     int* __restrict__ n_invalid) {
    using R = RawPair<NB>;
    const int t = blockIdx.z;
    const int y = blockIdx.y;
    const int sy = y - a.py0;
    int bad = 0;
    for (int x = blockIdx.x * 256 + threadIdx.x; x < a.Wp; x += gridDim.x * 256) {
        const int sx = x - a.px0;
        const bool in = sx >= 0 && sx < a.W && sy >= 0 && sy < a.H;
        const size_t d = ((size_t)t * a.Hp + y) * a.Wp + x;
        const size_t sidx = ((size_t)t * a.H + (in ? sy : 0)) * a.W + (in ? sx : 0);
        const typename R::type raw = in ? reinterpret_cast<const typename R::type*>(a.psi_phi)[sidx] : R::invalid();
        float psi, phi;
        R::decode(raw, a, &psi, &phi);
        const bool valid = in && __builtin_isfinite(psi) && __builtin_isfinite(phi);
        bad += (in && !valid) ? 1 : 0;
        if (CANON) {
            float2 v = make_float2(0.0f, -0.0f);
            if (valid) {
                v = make_float2(psi, phi);
                if (__float_as_uint(v.y) == 0x80000000u) v.y = 0.0f;
            }
            reinterpret_cast<float2*>(padded)[d] = v;
        } else {
            reinterpret_cast<typename R::type*>(padded)[d] = raw;
        }
    }
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
    if ((threadIdx.x & 63) == 0 && bad != 0) atomicAdd(n_invalid, bad);
}

// Tables are read through the constant address space: the DMA writes and barriers of
// the main loop would otherwise make the compiler fetch them with vector loads, whose
// vmcnt wait also waits for the slab DMA in flight.
typedef const __attribute__((address_space(4))) int* ConstIntPtr;
template <typename P>
__device__ __forceinline__ ConstIntPtr as_const_ints(const P* p) {
    return (ConstIntPtr)(uintptr_t)p;
}

// Slab origins as byte offsets, once the host has fixed the padded frame: the search kernel adds
// the tile's own offset and is spared the 64-bit index arithmetic per (chunk, epoch).
__global__ __launch_bounds__(256) void kb_slab_origin_kernel(const EpochBox* __restrict__ boxes, int64_t n, int T, int Hp,
                                                             int Wp, int px0, int py0, int pair_bytes,
                                                             int64_t* __restrict__ origins) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const EpochBox box = boxes[i];
    const int t = (int)(i % T);
    // an epoch that is not staged copies the slab at the tile's own pixel (inside the frame, never read by
    // the sums): the search loop is spared a test per epoch
    origins[i] = (box.x == BOX_NOT_STAGED)
                         ? (((int64_t)t * Hp + py0) * Wp + px0) * (int64_t)pair_bytes
                         : (((int64_t)t * Hp + box_dy(box) + py0) * Wp + box_dx(box) + px0) * (int64_t)pair_bytes;
}

// Staging map.  A slab (rows x LDS_COLS raw pairs, dense) is copied in workgroup-wide steps of
// 4 KiB: in step j thread tid moves the 16 bytes at slab offset o = 16 * (tid + 256 j), i.e.
// pixel p = o / BYTES = (row, col) = divmod(p, LDS_COLS) of the slab, from the padded array at
// the slab origin plus (row * Wp + col) * BYTES.  The copy goes through registers
// (global_load_dwordx4 -> ds_write_b128): measured on MI355X the LDS-DMA form of the same copy
// (global_load_lds_dwordx4) sustains only ~12 B/clk/CU and stalls the issuing wave.
struct StageLane {
    uint32_t goff[LDS_SLOTS];
};
typedef uint32_t Piece __attribute__((ext_vector_type(4)));
template <int ALIGN>
struct __attribute__((packed, aligned(ALIGN))) PieceMem {
    uint32_t w[4];
};
struct SlabRegs {
    Piece v[LDS_SLOTS];
};

// Offsets of a thread for slabs of slab_bytes: 0 (re-read the slab's first bytes) past the slab's end.
__device__ __forceinline__ StageLane clip_lane(const StageLane& sl, int slab_bytes) {
    StageLane out;
#pragma unroll
    for (int j = 0; j < LDS_SLOTS; ++j) {
        out.goff[j] = (16 * ((int)threadIdx.x + 256 * j) < slab_bytes) ? sl.goff[j] : 0u;
    }
    return out;
}

// Issue the loads of one epoch's slab; `base` = its origin in the padded copy (uniform).  All LDS_SLOTS
// loads are issued whatever the slab size (no branch, no exec mask -- the compiler would serialise
// masked loads with vmcnt(0)): threads past the end of the slab re-read its first bytes and do not
// write them to LDS.
template <int BYTES>
__device__ __forceinline__ void load_slab(const SearchArgs& a, const StageLane& sl, const char* base, int slab_bytes,
                                          SlabRegs& regs, int j0 = 0) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < LDS_SLOTS; ++j) {
        uint32_t goff = sl.goff[j];
        if (j0 != 0) {  // uniform, rare: rounds after the first compute their map on the fly
            const int p = 16 * (tid + 256 * (j0 + j)) / BYTES;
            const int r = p / LDS_COLS, c = p - r * LDS_COLS;
            goff = (uint32_t)(r * a.Wp + c) * (uint32_t)BYTES;
        }
        // (round 0: sl already holds 0 for threads past the end of this chunk's slabs, see clip_lane)
        const uint32_t off = (j0 == 0 || 16 * (tid + 256 * (j0 + j)) < slab_bytes) ? goff : 0u;
        // only BYTES-aligned: the hardware takes unaligned 16-byte global loads
        const PieceMem<(BYTES < 4 ? BYTES : 4)>* src = reinterpret_cast<const PieceMem<(BYTES < 4 ? BYTES : 4)>*>(base + off);
        regs.v[j] = Piece{src->w[0], src->w[1], src->w[2], src->w[3]};
    }
}

__device__ __forceinline__ void write_slab(char* dst, int slab_bytes, const SlabRegs& regs, int j0 = 0) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < LDS_SLOTS; ++j) {
        if (4096 * (j0 + j) < slab_bytes) {  // uniform
            const int o = 16 * (tid + 256 * (j0 + j));
            if (o < slab_bytes) *reinterpret_cast<Piece*>(dst + o) = regs.v[j];
        }
    }
}

// Rounds after the first of a slab larger than LDS_SLOTS x 4 KiB (load, then write, no overlap).
template <int BYTES>
__device__ __forceinline__ void copy_slab_tail(const SearchArgs& a, const StageLane& sl, const char* base, int slab_bytes,
                                               char* dst, SlabRegs& regs) {
    for (int j0 = LDS_SLOTS; 4096 * j0 < slab_bytes; j0 += LDS_SLOTS) {
        load_slab<BYTES>(a, sl, base, slab_bytes, regs, j0);
        write_slab(dst, slab_bytes, regs, j0);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    }
}

typedef float PairF __attribute__((ext_vector_type(2)));

// One epoch that is not staged (a shift inside the guard band of a rounding boundary,
// or a footprint larger than a slab): every lane predicts its own pixel exactly and
// reads the array itself, like kb_search_direct's exact mode.
template <int C, int NB>
__device__ __forceinline__ void unstaged_epoch(const SearchArgs& a, int x, int y, int chunk, int t, PairF (&acc)[C],
                                            int (&cnt)[C]) {
    using R = RawPair<NB>;
    constexpr int BYTES = 2 * fmt_bytes(NB);
    const double tm = a.times[t];
    const char* base = reinterpret_cast<const char*>(a.psi_phi) + (uint64_t)t * a.meta.pixels_per_image * BYTES;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int ci = min(chunk * C + c, a.n_cands - 1);
        int cx, cy;
        bool in = predict_index(x, a.cands[ci].vx, tm, &cx);
        in = predict_index(y, a.cands[ci].vy, tm, &cy) && in;
        const bool ok = in && ((unsigned)cx < (unsigned)a.W) && ((unsigned)cy < (unsigned)a.H);
        const uint32_t voff = ok ? (uint32_t)(cy * a.W + cx) * (uint32_t)BYTES : 0u;
        const typename R::type raw = *reinterpret_cast<const typename R::type*>(base + voff);
        float psi, phi;
        R::decode(raw, a, &psi, &phi);
        float s0 = acc[c].x, s1 = acc[c].y;
        accumulate(psi, phi, ok, s0, s1, cnt[c]);
        acc[c] = PairF{s0, s1};
    }
}

// One staged epoch in which some candidate's shift is known only to +-1 pixel (v*t + 0.5 on a
// rounding boundary): every lane predicts its own pixel with the reference's arithmetic and reads
// it from the slab, which was sized with that slack.  No global memory traffic.
template <int C, int SF, bool CANON>
__device__ __forceinline__ void per_lane_epoch(const SearchArgs& a, const TileCoords& tc, int chunk, int t, int box_word,
                                               int slab_bytes, const char* slab, PairF (&acc)[C], int (&cnt)[C]) {
    using R = RawPair<SF>;
    constexpr int BYTES = 2 * fmt_bytes(SF);
    typedef const __attribute__((address_space(4))) double* ConstDoublePtr;
    const double tm = ((ConstDoublePtr)(uintptr_t)a.times)[t];
    const EpochBox box = make_int2(box_word, 0);
    const int ox = tc.tile_x0 + box_dx(box), oy = tc.tile_y0 + box_dy(box);  // image coordinates of slab pixel (0, 0)
    const int rows = slab_bytes / (LDS_COLS * BYTES);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int ci = min(chunk * C + c, a.n_cands - 1);
        const ConstIntPtr cw = as_const_ints(a.cands + ci);  // {vx, vy, ...}
        int cx, cy;
        bool in = predict_index(tc.x, __int_as_float(cw[0]), tm, &cx);
        in = predict_index(tc.y, __int_as_float(cw[1]), tm, &cy) && in;
        const int rx = cx - ox, ry = cy - oy;
        const bool ok = in && ((unsigned)rx < (unsigned)LDS_COLS) && ((unsigned)ry < (unsigned)rows);
        const int off = ok ? (ry * LDS_COLS + rx) * BYTES : 0;
        const typename R::type raw = *reinterpret_cast<const typename R::type*>(slab + off);
        float psi, phi;
        R::decode(raw, a, &psi, &phi);
        if (CANON) {
            acc[c] += ok ? PairF{psi, phi} : PairF{0.0f, 0.0f};
            cnt[c] += (ok && __float_as_uint(phi) != 0x80000000u) ? 1 : 0;
        } else {
            float s0 = acc[c].x, s1 = acc[c].y;
            accumulate(psi, phi, ok, s0, s1, cnt[c]);
            acc[c] = PairF{s0, s1};
        }
    }
}

// Staging schedule of one chunk.
struct ChunkPlan {
    int slab_bytes;  // rows_max * LDS_COLS * BYTES
    int E;           // epochs per group
    int clean;       // every epoch is staged with uniform shifts: the summing loop needs no per-epoch test
};
template <int BYTES>
__device__ __forceinline__ ChunkPlan chunk_plan(const SearchArgs& a, int chunk) {
    ChunkPlan p;
    const ConstIntPtr ci = as_const_ints(&a.chunks[chunk]);  // {dx_min, dx_max, dy_min, dy_max, unsafe, lds_ok, rows_max}
    p.slab_bytes = ci[6] * LDS_COLS * BYTES;
    p.E = max(1, min(a.T, LDS_GROUP_BYTES / p.slab_bytes));
    p.clean = (ci[4] == 0 && ci[5] != 0) ? 1 : 0;
    return p;
}

// The whole search of one tile.  One flat software pipeline over (chunk, group): while
// group g is summed out of one LDS buffer, group g+1 -- possibly the first group of the next
// chunk -- is copied into the other, one slab per summed epoch: its loads are issued before
// the epoch's sums and written to LDS after them.
// FAST: no sample of this tile can be NO_DATA (the tile stays inside the image under
// every shift, the array has no NO_DATA pixel, every epoch is staged): obs_count is T.
template <int KS, int C, int NB, bool CANON, bool SIGMAG, bool FAST>
__device__ __forceinline__ void lds_search_tile(const SearchArgs& a, const TileCoords& tc, char* smem,
                                                const StageLane& sl, TopK<KS>& top, SigmaGScratch<WAVE>& scratch) {
    constexpr int SF = CANON ? 4 : NB;  // staged format
    using R = RawPair<SF>;
    constexpr int BYTES = 2 * fmt_bytes(SF);
    const int T = a.T;
    const int lane_b = (tc.wv * LDS_COLS + tc.lane) * BYTES;  // this lane's start pixel inside a slab

    PairF acc[C];  // (psi_sum, phi_sum) as pairs: one v_pk_add_f32 per sample
    int cnt[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        acc[c] = PairF{0.0f, 0.0f};
        cnt[c] = 0;
    }

    int chunk = 0, t0 = 0, buf = 0;
    ChunkPlan plan = chunk_plan<BYTES>(a, 0);
    SlabRegs regs;
    typedef const __attribute__((address_space(4))) int64_t* ConstI64Ptr;
    // this tile's own pixel inside the padded copy
    const char* tile_base = reinterpret_cast<const char*>(a.padded) + ((int64_t)tc.tile_y0 * a.Wp + tc.tile_x0) * BYTES;
    StageLane n_sl = clip_lane(sl, plan.slab_bytes);  // staging map of the group being copied
    {
        const ConstI64Ptr org = (ConstI64Ptr)(uintptr_t)a.origins;
        const int n = min(plan.E, T);
        for (int e = 0; e < n; ++e) {
            const int64_t o = org[e];
            load_slab<BYTES>(a, n_sl, tile_base + o, plan.slab_bytes, regs);
            write_slab(smem + e * plan.slab_bytes, plan.slab_bytes, regs);
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            copy_slab_tail<BYTES>(a, sl, tile_base + o, plan.slab_bytes, smem + e * plan.slab_bytes, regs);
        }
    }
    __syncthreads();

    while (chunk < a.n_chunks) {
        // next group in flight during this group's arithmetic
        int n_chunk = chunk, n_t0 = t0 + plan.E;
        ChunkPlan n_plan = plan;
        if (n_t0 >= T) {
            n_chunk = chunk + 1;
            n_t0 = 0;
            if (n_chunk < a.n_chunks) {
                n_plan = chunk_plan<BYTES>(a, n_chunk);
                n_sl = clip_lane(sl, n_plan.slab_bytes);
            }
        }
        const int n_next = (n_chunk < a.n_chunks) ? min(n_plan.E, T - n_t0) : 0;
        const ConstI64Ptr n_org = (ConstI64Ptr)(uintptr_t)(a.origins + (size_t)min(n_chunk, a.n_chunks - 1) * T + n_t0);
        char* nb = smem + (1 - buf) * LDS_GROUP_BYTES;
        // slab e of the next group: loads issued before, LDS writes after the sums of epoch e
        const char* n_base = tile_base;
        auto next_load = [&](int e) -> bool {
#ifdef KB_ABL_NO_DMA
            return false;
#endif
            if (e >= n_next) return false;
            n_base = tile_base + n_org[e];
            load_slab<BYTES>(a, n_sl, n_base, n_plan.slab_bytes, regs);
            return true;
        };
        auto next_write = [&](int e) {
#ifdef KB_ABL_NO_WRITE
            asm volatile("" ::"v"(regs.v[0]), "v"(regs.v[1]), "v"(regs.v[2]));
            return;
#endif
            write_slab(nb + e * n_plan.slab_bytes, n_plan.slab_bytes, regs);
            if (n_plan.slab_bytes > LDS_SLOTS * 4096) {  // uniform, rare
                __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
                copy_slab_tail<BYTES>(a, sl, n_base, n_plan.slab_bytes, nb + e * n_plan.slab_bytes, regs);
            }
        };

        const ConstIntPtr offs = as_const_ints(a.lds_off + ((size_t)chunk * T + t0) * C);  // offsets for 8-byte pairs
        const char* cb = smem + buf * LDS_GROUP_BYTES + lane_b;
        const int n_cur = min(plan.E, T - t0);
        // C samples of one staged epoch with uniform shifts: slab offsets o[] (scalars) -> LDS reads -> sums
        auto sum_epoch = [&](const int (&o)[C], int e) {
            typename R::type raw[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int off = (BYTES == 8) ? o[c] : (o[c] >> 3) * BYTES;
#ifdef KB_ABL_NO_LDSREAD
                if constexpr (CANON) {
                    raw[c] = make_float2(__int_as_float(off + tc.lane), 1.0f);
                    continue;
                }
#endif
                raw[c] = *reinterpret_cast<const typename R::type*>(cb + e * plan.slab_bytes + off);
            }
            // one wait for the C reads instead of the compiler's one per read (instruction issue is the bound)
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if constexpr (CANON) {
#ifdef KB_ABL_SCALAR_ADD
                    float ax = acc[c].x, ay = acc[c].y;
                    ax += raw[c].x;
                    asm volatile("" : "+v"(ax));
                    ay += raw[c].y;
                    acc[c] = PairF{ax, ay};
#else
                    acc[c] += PairF{raw[c].x, raw[c].y};
#endif
                    if (!FAST) cnt[c] += (__float_as_uint(raw[c].y) != 0x80000000u) ? 1 : 0;
                } else {
                    float psi, phi;
                    R::decode(raw[c], a, &psi, &phi);
                    if (FAST) {
                        acc[c] += PairF{psi, phi};
                    } else {
                        float s0 = acc[c].x, s1 = acc[c].y;
                        accumulate(psi, phi, true, s0, s1, cnt[c]);
                        acc[c] = PairF{s0, s1};
                    }
                }
            }
        };
#ifdef KB_ABL_NO_SUM
        if (false) {
#else
        // Keeps the epoch's sums in front of the LDS writes of the staged slab: left alone the compiler
        // sinks the adds behind the writes, whose vmcnt(0) then waits out the loads with nothing to overlap.
        auto pin_sums = [&]() {
            static_assert(C == 8, "operand list below");
            asm volatile(""
                         : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                           "+v"(acc[7]), "+v"(cnt[0]), "+v"(cnt[1]), "+v"(cnt[2]), "+v"(cnt[3]), "+v"(cnt[4]), "+v"(cnt[5]),
                           "+v"(cnt[6]), "+v"(cnt[7])
                         :
                         : "memory");
        };
        if (FAST || plan.clean) {
#endif
            // A block alone on its CU is bound by the chain scalar table fetch -> LDS read -> adds -> slab
            // landed -> LDS write of one epoch (measured 4.9 ms with one block per CU against 7.4 ms with
            // four).  The table words of epoch e + 1 (slab offsets, slab origin) are therefore fetched at
            // the END of epoch e, behind the adds: they travel while the slab loads are waited for and
            // written, and the lgkmcnt wait of epoch e + 1's LDS reads finds them done.  The empty asm
            // pins the fetch behind pin_sums(); both tables have slack behind their last entry.
            int o_cur[C];
#pragma unroll
            for (int c = 0; c < C; ++c) o_cur[c] = offs[c];
            int64_t org_cur = n_org[0];
            for (int e = 0; e < n_cur; ++e) {
                const bool staging = e < n_next;
#ifndef KB_ABL_NO_DMA
                if (staging) {
                    n_base = tile_base + org_cur;
                    load_slab<BYTES>(a, n_sl, n_base, n_plan.slab_bytes, regs);
                }
#endif
                sum_epoch(o_cur, e);
                pin_sums();
                ConstIntPtr po = offs + (e + 1) * C;
                ConstI64Ptr pg = n_org + (e + 1);
                asm volatile("" : "+s"(po), "+s"(pg)::"memory");
#pragma unroll
                for (int c = 0; c < C; ++c) o_cur[c] = po[c];
                org_cur = pg[0];
#ifndef KB_ABL_NO_DMA
                if (staging) next_write(e);
#endif
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            }
        } else {
#ifdef KB_ABL_NO_SUM
            for (int e = 0; e < 0; ++e) {
#else
            for (int e = 0; e < n_cur; ++e) {
#endif
                const bool staging = next_load(e);
                int o[C];
#pragma unroll
                for (int c = 0; c < C; ++c) o[c] = offs[e * C + c];
                if (o[0] >= 0) {
                    sum_epoch(o, e);
                } else if (o[0] == LDS_OFF_UNSTAGED) {
                    unstaged_epoch<C, NB>(a, tc.x, tc.y, chunk, t0 + e, acc, cnt);
                } else {
                    const int bw = as_const_ints(a.boxes + (size_t)chunk * T + t0 + e)[0];
                    per_lane_epoch<C, SF, CANON>(a, tc, chunk, t0 + e, bw, plan.slab_bytes,
                                                 smem + buf * LDS_GROUP_BYTES + e * plan.slab_bytes, acc, cnt);
                }
                pin_sums();
                if (staging) next_write(e);
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            }
        }
        for (int e = n_cur; e < n_next; ++e) {  // the next group holds more epochs than this one
            if (next_load(e)) next_write(e);
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        }

        if (n_chunk != chunk) {  // chunk complete: likelihoods + top-K, while the next chunk's first group lands
            if (tc.row_active) {
                float ps[C], ph[C];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    ps[c] = acc[c].x;
                    ph[c] = acc[c].y;
                    if (FAST) cnt[c] = T;
                }
                finish_chunk<KS, C, SIGMAG>(a, tc, chunk, ps, ph, cnt, top, scratch);
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                acc[c] = PairF{0.0f, 0.0f};
                cnt[c] = 0;
            }
        }
#ifndef KB_ABL_NO_BARRIER
        __syncthreads();
#endif
        buf = 1 - buf;
        chunk = n_chunk;
        t0 = n_t0;
        plan = n_plan;
    }
}

constexpr int LDS_BLOCK = TILE_ROWS * WAVE;

template <int KS, int C, int NB, bool CANON, bool SIGMAG>
// second launch bound = waves per SIMD: 4 / 3 / 2 four-wave workgroups per CU for K <= 8 / 16 / 32
__global__ __launch_bounds__(LDS_BLOCK, (KS <= 8 ? 4 : (KS <= 16 ? 3 : 2))) void kb_search_lds(const SearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // two group buffers
    constexpr int BYTES = 2 * fmt_bytes(CANON ? 4 : NB);
    const TileCoords tc = tile_coords(a);  // rows past the search area stay alive (barriers)
    TopK<KS> top;
    top.init();
    SigmaGScratch<WAVE> scratch = {};
    if constexpr (SIGMAG) scratch = make_scratch(a, tc);

    StageLane sl;
#pragma unroll
    for (int j = 0; j < LDS_SLOTS; ++j) {
        const int p = 16 * ((int)threadIdx.x + 256 * j) / BYTES;  // first pixel of this thread's 16 bytes
        const int r = p / LDS_COLS, c = p - r * LDS_COLS;
        sl.goff[j] = (uint32_t)(r * a.Wp + c) * (uint32_t)BYTES;
    }

    // Workgroup-uniform: can any sample of this tile be NO_DATA?
    const ConstIntPtr gb = as_const_ints(a.global_box);
    const bool fast = a.all_staged && as_const_ints(a.n_invalid)[0] == 0 && (tc.tile_x0 + gb[0] >= 0) &&
                      (tc.tile_x0 + WAVE + gb[1] <= a.W) && (tc.tile_y0 + gb[2] >= 0) &&
                      (tc.tile_y0 + TILE_ROWS + gb[3] <= a.H);
    if (fast) {
        lds_search_tile<KS, C, NB, CANON, SIGMAG, true>(a, tc, smem, sl, top, scratch);
    } else {
        lds_search_tile<KS, C, NB, CANON, SIGMAG, false>(a, tc, smem, sl, top, scratch);
    }
    write_results<KS, SIGMAG>(a, tc, top, scratch);
}

// ---------------------------------------------------------------------------
// large-K kernel (results_per_pixel > 32, e.g. TrajectoryExplorer's K up to 10 000)
// ---------------------------------------------------------------------------
// One lane per start pixel, candidates evaluated one at a time with exact
// per-lane positions, the K-slot list kept in the result array itself and
// updated with the reference's swap-down (kernels.cu:304-331).  This is the
// reference kernel's own structure; it is only used where the register top-K
// cannot hold the list (few start pixels x many results in practice).
template <bool SIGMAG>
__global__ __launch_bounds__(256) void kb_search_large_k(const SearchArgs a) {
    const TileCoords tc = tile_coords(a);
    if (!tc.row_active || tc.x_i >= a.sw) return;
    SigmaGScratch<WAVE> scratch = {};
    if constexpr (SIGMAG) scratch = make_scratch(a, tc);
    kb_trajectory* slots = a.results + ((size_t)tc.y_i * a.sw + tc.x_i) * a.K;
    for (int s = 0; s < a.K; ++s) {  // kernels.cu:293-301
        kb_trajectory p;
        p.x = tc.x;
        p.y = tc.y;
        p.vx = 0.0f;
        p.vy = 0.0f;
        p.lh = -FLT_MAX;
        p.flux = 0.0f;
        p.obs_count = 0;
        slots[s] = p;
    }
    for (int cand = 0; cand < a.n_cands; ++cand) {
        kb_trajectory cur;
        cur.x = tc.x;
        cur.y = tc.y;
        cur.vx = a.cands[cand].vx;
        cur.vy = a.cands[cand].vy;
        evaluate_trajectory_full<WAVE>(a.meta, a.psi_phi, a.times, a.params, &cur, SIGMAG ? &scratch : nullptr);
        if ((cur.obs_count < a.params.min_observations) || (a.params.do_sigmag_filter && cur.lh < a.params.min_lh))
            continue;  // kernels.cu:318-320
        if (!(cur.lh > slots[a.K - 1].lh)) continue;  // cannot displace anything
        for (int s = 0; s < a.K; ++s) {  // kernels.cu:323-330
            const kb_trajectory t = slots[s];
            if (cur.lh > t.lh) {
                slots[s] = cur;
                cur = t;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// multi-GPU: per-pixel merge of the per-rank top-K lists after the RCCL gather
// ---------------------------------------------------------------------------
// lists[r][pixel][K] (each sorted descending by lh, placeholders lh = -FLT_MAX
// last) -> out[pixel][K].  Ties go to the lower rank, then the lower slot, i.e.
// to the lower global candidate index when ranks own contiguous candidate
// slices -- the order a single-GPU run over the concatenated list would keep
// for distinct likelihoods.
constexpr int MERGE_MAX_LISTS = 64;
__global__ __launch_bounds__(256) void kb_merge_topk_kernel(const kb_trajectory* __restrict__ lists, int n_lists,
                                                            uint64_t n_pixels, int K,
                                                            kb_trajectory* __restrict__ out) {
    const uint64_t pix = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= n_pixels) return;
    unsigned char head[MERGE_MAX_LISTS];
    for (int r = 0; r < n_lists; ++r) head[r] = 0;
    const uint64_t list_stride = n_pixels * (uint64_t)K;
    for (int s = 0; s < K; ++s) {
        int best = -1;
        float best_lh = 0.0f;
        for (int r = 0; r < n_lists; ++r) {
            if (head[r] >= K) continue;
            const float lh = lists[(uint64_t)r * list_stride + pix * K + head[r]].lh;
            if (best < 0 || lh > best_lh) {
                best = r;
                best_lh = lh;
            }
        }
        out[pix * K + s] = lists[(uint64_t)best * list_stride + pix * K + head[best]];
        head[best] += 1;
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct Workspace {
    void* ptr = nullptr;
    size_t bytes = 0;
    int device = -1;
};
static std::mutex g_ws_mutex;
static Workspace g_ws[3];  // 0: shift table + chunk info, 1: sigma-G scratch, 2: padded array copy (LDS kernel)

static int ensure_workspace(int which, size_t bytes, void** out) {
    int dev = 0;
    KB_HIP_TRY(hipGetDevice(&dev));
    Workspace& w = g_ws[which];
    if (w.ptr != nullptr && (w.device != dev || w.bytes < bytes)) {
        (void)hipFree(w.ptr);
        w.ptr = nullptr;
        w.bytes = 0;
    }
    if (w.ptr == nullptr) {
        KB_HIP_TRY(hipMalloc(&w.ptr, bytes));
        w.bytes = bytes;
        w.device = dev;
    }
    *out = w.ptr;
    return 0;
}

// True when fmaf(code - 1, scale, min) equals the reference's double-rounded decode for every code.
static bool verify_fast_decode(float scale, float min_val, int num_bytes) {
    const unsigned max_code = (1u << (8 * num_bytes)) - 1u;
    for (unsigned code = 1; code <= max_code; ++code) {
        volatile double prod = ((double)(float)code - 1.0) * (double)scale;
        const float exact = (float)(prod + (double)min_val);
        const float fast = std::fmaf((float)code - 1.0f, scale, min_val);
        if (std::memcmp(&exact, &fast, sizeof(float)) != 0 || !std::isfinite(exact)) return false;
    }
    return true;
}

template <typename KernelT>
static void debug_occupancy(const char* name, KernelT kernel, size_t lds, int block = 256) {
    if (std::getenv("KBMOD_DEBUG") == nullptr) return;
    int blocks = -1;
    hipFuncAttributes attr;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kernel, block, lds);
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(kernel));
    std::fprintf(stderr, "[kbmod_hip] %s: %d blocks/CU, %d VGPRs, %zu B static LDS, %zu B scratch, dyn LDS %zu\n", name,
                 blocks, attr.numRegs, attr.sharedSizeBytes, attr.localSizeBytes, lds);
}

// which: 0 = kb_search_direct, 1 = kb_search_lds on an encoded padded copy, 2 = kb_search_lds on canonical floats
template <int KS, int NB, bool SIGMAG>
static void launch_variant(const SearchArgs& a, int which, hipStream_t stream) {
    debug_occupancy("kb_search_lds", kb_search_lds<KS, CHUNK, NB, true, SIGMAG>, 2 * LDS_GROUP_BYTES, LDS_BLOCK);
    debug_occupancy("kb_search_direct", kb_search_direct<KS, CHUNK, NB, SIGMAG>, 0);
    const dim3 grid(a.n_tiles), block(256), lds_block(LDS_BLOCK);
    if (which == 2) {
        hipLaunchKernelGGL((kb_search_lds<KS, CHUNK, NB, true, SIGMAG>), grid, lds_block, 2 * LDS_GROUP_BYTES, stream,
                           a);
    } else if (which == 1) {
        if constexpr (NB != 4) {
            hipLaunchKernelGGL((kb_search_lds<KS, CHUNK, NB, false, SIGMAG>), grid, lds_block, 2 * LDS_GROUP_BYTES,
                               stream, a);
        }
    } else {
        hipLaunchKernelGGL((kb_search_direct<KS, CHUNK, NB, SIGMAG>), grid, block, 0, stream, a);
    }
}

template <int KS, int NB>
static void launch_sigmag(const SearchArgs& a, bool sigmag, int which, hipStream_t stream) {
    if (sigmag)
        launch_variant<KS, NB, true>(a, which, stream);
    else
        launch_variant<KS, NB, false>(a, which, stream);
}

// Format code of the array for the templates: 4 = float, 2 / 1 = encoded with the reference's
// double-precision decode, 20 / 10 = encoded with the verified single-FMA decode.
static int format_code(const SearchArgs& a) {
    if (a.meta.num_bytes == 1) return a.fast_decode ? 10 : 1;
    if (a.meta.num_bytes == 2) return a.fast_decode ? 20 : 2;
    return 4;
}

template <int KS>
static void launch_search(const SearchArgs& a, bool sigmag, int which, hipStream_t stream) {
    switch (format_code(a)) {
        case 1:
            launch_sigmag<KS, 1>(a, sigmag, which, stream);
            break;
        case 10:
            launch_sigmag<KS, 10>(a, sigmag, which, stream);
            break;
        case 2:
            launch_sigmag<KS, 2>(a, sigmag, which, stream);
            break;
        case 20:
            launch_sigmag<KS, 20>(a, sigmag, which, stream);
            break;
        default:
            launch_sigmag<KS, 4>(a, sigmag, which, stream);
            break;
    }
}

template <int NB>
static void launch_pad_fmt(const SearchArgs& a, bool canon, void* padded, int* n_invalid, hipStream_t stream) {
    const dim3 grid((unsigned)std::min<int64_t>(((int64_t)a.Wp + 255) / 256, 64), (unsigned)a.Hp, (unsigned)a.T);
    if (canon)
        hipLaunchKernelGGL((kb_pad_kernel<NB, true>), grid, dim3(256), 0, stream, a, padded, n_invalid);
    else
        hipLaunchKernelGGL((kb_pad_kernel<NB, false>), grid, dim3(256), 0, stream, a, padded, n_invalid);
}

static void launch_pad(const SearchArgs& a, bool canon, void* padded, int* n_invalid, hipStream_t stream) {
    switch (format_code(a)) {
        case 1:
            launch_pad_fmt<1>(a, canon, padded, n_invalid, stream);
            break;
        case 10:
            launch_pad_fmt<10>(a, canon, padded, n_invalid, stream);
            break;
        case 2:
            launch_pad_fmt<2>(a, canon, padded, n_invalid, stream);
            break;
        case 20:
            launch_pad_fmt<20>(a, canon, padded, n_invalid, stream);
            break;
        default:
            launch_pad_fmt<4>(a, true, padded, n_invalid, stream);
            break;
    }
}

}  // namespace kb

extern "C" {

int kb_device_search_filter(const kb_psi_phi_meta* meta, const void* psi_phi_dev, const double* times_dev,
                            kb_search_params params, const kb_trajectory* cands_dev, uint64_t n_cands,
                            kb_trajectory* results_dev, uint64_t n_results, uint32_t flags, void* stream_v,
                            kb_search_stats* stats_out) {
    using namespace kb;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    if (meta == nullptr) return fail("deviceSearchFilter: null meta data");
    // kernels.cu:337-340
    if (meta->num_times > KB_MAX_NUM_IMAGES) {
        return fail("Number of images exceeds GPU maximum " + std::to_string(KB_MAX_NUM_IMAGES));
    }
    if (meta->num_times == 0) return fail("PsiPhi data has no images.");
    // kernels.cu:346-354
    if (psi_phi_dev == nullptr) return fail("PsiPhi data has not been created.");
    if (times_dev == nullptr) return fail("GPU time data has not been created.");
    if (cands_dev == nullptr) return fail("Invalid test list pointer.");
    if (results_dev == nullptr) return fail("Invalid result list pointer.");
    if (kb_device_count() == 0) return fail("GPU is not available for search.");

    // kernels.cu:371-378
    const int64_t sw = (int64_t)params.x_start_max - params.x_start_min;
    const int64_t sh = (int64_t)params.y_start_max - params.y_start_min;
    if (sw <= 0 || sh <= 0) {
        return fail("Invalid search bounds x=[" + std::to_string(params.x_start_min) + ", " +
                    std::to_string(params.x_start_max) + "] y=[" + std::to_string(params.y_start_min) + ", " +
                    std::to_string(params.y_start_max) + "]");
    }
    if (params.results_per_pixel == 0) return fail("Invalid results per pixel. Got 0");
    // kernels.cu:383-389
    const uint64_t expected = (uint64_t)params.results_per_pixel * (uint64_t)sw * (uint64_t)sh;
    params.total_results = expected;
    if (n_results < expected) {
        return fail("Not enough space allocated for results. Requires: " + std::to_string(expected) +
                    ". Received: " + std::to_string(n_results));
    }
    if (meta->pixels_per_image * 8ull > 0xffffffffull) {
        return fail("Image too large for 32-bit in-image offsets (H*W*8 must fit 4 GiB).");
    }
    if (meta->width > (1u << 22) || meta->height > (1u << 22) || std::abs((long)params.x_start_min) > (1 << 21) ||
        std::abs((long)params.x_start_max) > (1 << 21) || std::abs((long)params.y_start_min) > (1 << 21) ||
        std::abs((long)params.y_start_max) > (1 << 21)) {
        flags |= 1u;  // start coordinates outside the proven range of the shift table
    }

    SearchArgs a;
    a.psi_phi = psi_phi_dev;
    a.times = times_dev;
    a.cands = cands_dev;
    a.results = results_dev;
    a.meta = *meta;
    a.params = params;
    a.T = (int)meta->num_times;
    a.W = (int)meta->width;
    a.H = (int)meta->height;
    a.n_cands = (int)n_cands;
    a.n_chunks = (int)((n_cands + CHUNK - 1) / CHUNK);
    a.sw = (int)sw;
    a.sh = (int)sh;
    a.tiles_x = (a.sw + WAVE - 1) / WAVE;
    a.tiles_y = (a.sh + TILE_ROWS - 1) / TILE_ROWS;
    a.n_tiles = a.tiles_x * a.tiles_y;
    a.K = (int)params.results_per_pixel;
    a.force_exact = (flags & 1u) ? 1 : 0;
    a.fast_decode = 0;
    if (meta->num_bytes != 4 && (flags & 8u) == 0) {  // bit 3: force the double-precision decode
        a.fast_decode = (verify_fast_decode(meta->psi_scale, meta->psi_min_val, meta->num_bytes) &&
                         verify_fast_decode(meta->phi_scale, meta->phi_min_val, meta->num_bytes))
                                ? 1
                                : 0;
    }
    a.sg_scratch = nullptr;
    a.table = nullptr;
    a.chunks = nullptr;
    a.boxes = nullptr;
    a.lds_off = nullptr;
    a.global_box = nullptr;
    a.origins = nullptr;
    a.padded = nullptr;
    a.Wp = a.Hp = a.px0 = a.py0 = 0;
    a.n_invalid = nullptr;
    a.all_staged = 0;

    EventTimer table_timer(stream, stats_out != nullptr);
    EventTimer search_timer(stream, stats_out != nullptr);
    std::lock_guard<std::mutex> lock(g_ws_mutex);

    float table_ms = 0.0f, search_ms = 0.0f;
    // Kernel choice.  kb_search_lds (LDS-DMA staging from a padded copy) is the default for K <= 32;
    // flags bit 1 forces kb_search_direct, bit 2 insists on kb_search_lds even for few candidates,
    // bit 4 keeps an encoded array encoded in the padded copy.  Measured on MI355X (profiles/r01_*):
    // the direct kernel is bound by the vector-memory pipe (every sample is its own 512-byte wave
    // load), the staged kernel reads each slab once per workgroup and sums out of LDS.
    int which = 0;
    const bool want_lds = (flags & 2u) == 0 && (flags & 1u) == 0 && a.K <= 32 &&
                          (a.n_chunks >= 4 || (flags & 4u) != 0);
    if (n_cands > 0) {
        const size_t table_bytes = (size_t)a.n_chunks * a.T * CHUNK * sizeof(int2);
        const size_t off_bytes = ((size_t)a.n_chunks * a.T * CHUNK + 4 * CHUNK) * sizeof(int);  // + prefetch slack
        const size_t box_bytes = (size_t)a.n_chunks * a.T * sizeof(EpochBox);
        const size_t org_bytes = ((size_t)a.n_chunks * a.T + 8) * sizeof(int64_t);  // + prefetch slack
        const size_t chunk_bytes = (size_t)a.n_chunks * sizeof(ChunkInfo);
        // NO_DATA pixel counter [1], unstaged (chunk, epoch) counter [1], staged shift box + tallest slab [5],
        // per-lane (chunk, epoch) counter [1]
        const size_t inv_bytes = 8 * sizeof(int);
        void* ws = nullptr;
        if (ensure_workspace(0, table_bytes + off_bytes + box_bytes + chunk_bytes + inv_bytes + org_bytes, &ws)) return 1;
        char* wsc = reinterpret_cast<char*>(ws);
        a.table = reinterpret_cast<const int2*>(wsc);
        a.lds_off = reinterpret_cast<const int*>(wsc + table_bytes);
        a.boxes = reinterpret_cast<const EpochBox*>(wsc + table_bytes + off_bytes);
        a.chunks = reinterpret_cast<const ChunkInfo*>(wsc + table_bytes + off_bytes + box_bytes);
        int* inv = reinterpret_cast<int*>(wsc + table_bytes + off_bytes + box_bytes + chunk_bytes);
        int64_t* origins = reinterpret_cast<int64_t*>(wsc + table_bytes + off_bytes + box_bytes + chunk_bytes + inv_bytes);
        a.origins = origins;
        int* n_invalid = inv;
        int* n_not_lds = inv + 1;
        int* gbox = inv + 2;
        a.n_invalid = n_invalid;
        a.global_box = gbox;
        table_timer.begin();
        static const int inv_init[8] = {0, 0, INT32_MAX, INT32_MIN, INT32_MAX, INT32_MIN, 0, 0};
        KB_HIP_TRY(hipMemcpyAsync(inv, inv_init, sizeof(inv_init), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL((kb_shift_table_kernel<CHUNK>), dim3(a.n_chunks), dim3(256), 0, stream, cands_dev,
                           times_dev, a.n_cands, a.T, reinterpret_cast<int2*>(wsc),
                           reinterpret_cast<ChunkInfo*>(wsc + table_bytes + off_bytes + box_bytes),
                           reinterpret_cast<EpochBox*>(wsc + table_bytes + off_bytes),
                           reinterpret_cast<int*>(wsc + table_bytes), n_not_lds, gbox);
        KB_HIP_TRY(hipGetLastError());
        if (want_lds) {
            // The choice and the apron of the padded copy need six ints back.
            int back[7] = {0, 0, 0, 0, 0, 0, 0};  // unstaged epochs, dx_min, dx_max, dy_min, dy_max, rows_max, per-lane epochs
            KB_HIP_TRY(hipMemcpyAsync(back, n_not_lds, sizeof(back), hipMemcpyDeviceToHost, stream));
            KB_HIP_TRY(hipStreamSynchronize(stream));
            // An unstaged epoch costs several times a staged one: above 10 % the direct kernel wins.
            const uint64_t n_epochs = (uint64_t)a.n_chunks * (uint64_t)a.T;
            if (std::getenv("KBMOD_DEBUG") != nullptr) {
                std::fprintf(stderr,
                             "[kbmod_hip] (chunk, epoch) pairs: %llu, unstaged %d, per-lane %d; staged shift box x [%d, %d] "
                             "y [%d, %d], tallest slab %d rows\n",
                             (unsigned long long)n_epochs, back[0], back[6], back[1], back[2], back[3], back[4], back[5]);
            }
            if (back[1] <= back[2] && (uint64_t)back[0] * 10ull <= n_epochs) {
                // Every slab [origin, origin + rows_max) x [origin, origin + LDS_COLS) lies inside the padded frame.
                // (the zero shift is included: unstaged epochs copy the slab at the tile's own pixel)
                back[1] = std::min(back[1], 0);
                back[2] = std::max(back[2], 0);
                back[3] = std::min(back[3], 0);
                back[4] = std::max(back[4], 0);
                const int64_t x_lo = (int64_t)params.x_start_min + back[1];
                const int64_t x_hi = (int64_t)params.x_start_min + (int64_t)WAVE * (a.tiles_x - 1) + back[2] + LDS_COLS;
                const int64_t y_lo = (int64_t)params.y_start_min + back[3];
                const int64_t y_hi =
                        (int64_t)params.y_start_min + (int64_t)TILE_ROWS * (a.tiles_y - 1) + back[4] + back[5];
                int64_t px0 = std::max<int64_t>(0, -x_lo), py0 = std::max<int64_t>(0, -y_lo);
                // slab alignment (kb_shift_table_kernel): x_start_min + px0 is a multiple of LDS_ALIGN_PX,
                // the row pitch a multiple of 16 pixels
                px0 += (((-(px0 + (int64_t)params.x_start_min)) % LDS_ALIGN_PX) + LDS_ALIGN_PX) % LDS_ALIGN_PX;
                int64_t Wp = px0 + std::max<int64_t>(a.W, x_hi), Hp = py0 + std::max<int64_t>(a.H, y_hi);
                Wp = (Wp + 15) / 16 * 16;
                const uint64_t frame = (uint64_t)a.T * (uint64_t)Hp * (uint64_t)Wp;
                const uint64_t image = (uint64_t)a.T * (uint64_t)a.H * (uint64_t)a.W;
                // Canonical floats unless the caller keeps the array encoded or HBM is short.
                bool canon = meta->num_bytes == 4 || (flags & 16u) == 0;
                if (canon && meta->num_bytes != 4) {
                    size_t free_b = 0, total_b = 0;
                    KB_HIP_TRY(hipMemGetInfo(&free_b, &total_b));
                    const uint64_t have = g_ws[2].ptr != nullptr ? g_ws[2].bytes : 0;
                    if (frame * 8ull + 64 > have && frame * 8ull + (2ull << 30) > (uint64_t)free_b + have) canon = false;
                }
                const uint64_t pair_bytes = canon ? 8ull : 2ull * (uint64_t)meta->block_size;
                const uint64_t padded_bytes = frame * pair_bytes + 64;
                // Per-lane DMA offsets are 32-bit; an apron that outweighs the image 3:1 is not worth staging.
                if ((uint64_t)back[5] * (uint64_t)Wp * pair_bytes <= 0x7fffffffull && frame <= 4ull * image + (8ull << 20)) {
                    void* padded = nullptr;
                    if (ensure_workspace(2, padded_bytes, &padded)) return 1;
                    a.padded = padded;
                    a.Wp = (int)Wp;
                    a.Hp = (int)Hp;
                    a.px0 = (int)px0;
                    a.py0 = (int)py0;
                    // bit 5 (debug): never take the count-free specialisation
                    a.all_staged = (back[0] == 0 && back[6] == 0 && (flags & 32u) == 0) ? 1 : 0;
                    launch_pad(a, canon, padded, n_invalid, stream);
                    KB_HIP_TRY(hipGetLastError());
                    const int64_t n_org = (int64_t)a.n_chunks * a.T;
                    hipLaunchKernelGGL(kb_slab_origin_kernel, dim3((unsigned)((n_org + 255) / 256)), dim3(256), 0, stream,
                                       a.boxes, n_org, a.T, a.Hp, a.Wp, a.px0, a.py0, (int)pair_bytes, origins);
                    KB_HIP_TRY(hipGetLastError());
                    which = canon ? 2 : 1;
                }
            }
        }
        table_ms = table_timer.end();
    }

    const bool sigmag = params.do_sigmag_filter != 0;
    if (sigmag) {
        const size_t waves = (size_t)a.n_tiles * TILE_ROWS;
        const size_t bytes = waves * (size_t)(4 * a.T) * WAVE * sizeof(float);
        void* sg = nullptr;
        if (ensure_workspace(1, bytes, &sg)) return 1;
        a.sg_scratch = reinterpret_cast<float*>(sg);
    }

    search_timer.begin();
    int variant;
    if (a.K > 32) {
        if (sigmag)
            hipLaunchKernelGGL((kb_search_large_k<true>), dim3(a.n_tiles), dim3(256), 0, stream, a);
        else
            hipLaunchKernelGGL((kb_search_large_k<false>), dim3(a.n_tiles), dim3(256), 0, stream, a);
        variant = 99;
    } else if (a.K <= 8) {
        launch_search<8>(a, sigmag, which, stream);
        variant = 8;
    } else if (a.K <= 16) {
        launch_search<16>(a, sigmag, which, stream);
        variant = 16;
    } else {
        launch_search<32>(a, sigmag, which, stream);
        variant = 32;
    }
    KB_HIP_TRY(hipGetLastError());
    search_ms = search_timer.end();
    if (which != 0 && std::getenv("KBMOD_DEBUG") != nullptr) {
        int bad = -1;
        KB_HIP_TRY(hipMemcpyAsync(&bad, a.n_invalid, sizeof(int), hipMemcpyDeviceToHost, stream));
        KB_HIP_TRY(hipStreamSynchronize(stream));
        std::fprintf(stderr, "[kbmod_hip] padded frame %d x %d (image at %d, %d), NO_DATA pixels %d, all_staged %d\n", a.Wp,
                     a.Hp, a.px0, a.py0, bad, a.all_staged);
    }

    if (stats_out != nullptr) {
        const uint64_t S = (uint64_t)sw * (uint64_t)sh;
        stats_out->search_kernel_ms = search_ms;
        stats_out->table_kernel_ms = table_ms;
        stats_out->num_evals = S * n_cands * meta->num_times;
        stats_out->algorithmic_bytes = stats_out->num_evals * 2ull * (uint64_t)meta->block_size +
                                       S * (uint64_t)a.K * 28ull + n_cands * 28ull + meta->num_times * 8ull;
        stats_out->kernel_variant = which * 10000 + variant * 100 + meta->num_bytes * 10 + (sigmag ? 1 : 0);
        stats_out->num_search_launches = 1;
    } else {
        // kernels.cu:396 -- the reference call is synchronous.
        KB_HIP_TRY(hipStreamSynchronize(stream));
    }
    return 0;
}

int kb_release_workspaces(void) {
    using namespace kb;
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    for (Workspace& w : g_ws) {
        if (w.ptr != nullptr) {
            int prev = 0;
            KB_HIP_TRY(hipGetDevice(&prev));
            KB_HIP_TRY(hipSetDevice(w.device));
            KB_HIP_TRY(hipFree(w.ptr));
            KB_HIP_TRY(hipSetDevice(prev));
        }
        w = Workspace();
    }
    return 0;
}

int kb_merge_topk(const kb_trajectory* lists_dev, int32_t n_lists, uint64_t n_pixels, int32_t K,
                  kb_trajectory* out_dev, void* stream_v) {
    using namespace kb;
    if (lists_dev == nullptr || out_dev == nullptr) return fail("merge_topk: null pointer");
    if (n_lists <= 0 || n_lists > MERGE_MAX_LISTS) return fail("merge_topk: unsupported number of lists");
    if (K <= 0 || K > 255) return fail("merge_topk: unsupported K");
    if (n_pixels == 0) return 0;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    const unsigned blocks = (unsigned)((n_pixels + 255) / 256);
    hipLaunchKernelGGL(kb_merge_topk_kernel, dim3(blocks), dim3(256), 0, stream, lists_dev, n_lists, n_pixels, K,
                       out_dev);
    KB_HIP_TRY(hipGetLastError());
    return 0;
}

// kernels.cu:154-242 called on host pointers (stack_search.cpp:203-204).
int kb_evaluate_trajectory_host(const kb_psi_phi_meta* meta, const void* psi_phi_host, const double* times_host,
                                kb_search_params params, kb_trajectory* candidate) {
    using namespace kb;
    if (meta == nullptr || psi_phi_host == nullptr || times_host == nullptr || candidate == nullptr) {
        return fail("evaluateTrajectory: null argument");
    }
    if (meta->num_times > KB_MAX_NUM_IMAGES) {
        return fail("Too many images to evaluate on GPU. Max = " + std::to_string(KB_MAX_NUM_IMAGES));
    }
    const size_t T = (size_t)meta->num_times;
    std::vector<float> buf(3 * T + 1);
    std::vector<int> idx(T + 1);
    SigmaGScratch<1> sc{{buf.data()}, {buf.data() + T}, {buf.data() + 2 * T}, {idx.data()}};
    evaluate_trajectory_full<1>(*meta, psi_phi_host, times_host, params, candidate, &sc);
    return 0;
}

void kb_sigmag_filtered_indices(const float* values, int num_values, float sgl0, float sgl1, float sigmag_coeff,
                                float width, int* idx_array, int* min_keep_idx, int* max_keep_idx) {
    // kernels.cu:84: ignore the call rather than touch invalid memory.
    if ((idx_array == nullptr) || ((min_keep_idx == nullptr) && (max_keep_idx == nullptr))) return;
    kb::StridedView<const float, 1> v{values};
    kb::StridedView<int, 1> ix{idx_array};
    kb::sigmag_filtered_indices_t(v, num_values, sgl0, sgl1, sigmag_coeff, width, ix, min_keep_idx, max_keep_idx);
}

}  // extern "C"
