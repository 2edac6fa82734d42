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

// LDS stage buffer: at most LDS_ROWS x LDS_COLS pixels per (chunk, epoch).
// Plane A: (psi, phi) float2 with NO_DATA replaced by (+0, +0); plane B: validity
// (1 / 0) for obs_count.
constexpr int LDS_COLS = 84;  // pitch in pixels: 64 start columns + up to 20 of dx spread
constexpr int LDS_ROWS = 13;  // 4 start rows + up to 9 of dy spread
constexpr int LDS_PLANE_A = LDS_ROWS * LDS_COLS * 8;  // bytes
constexpr int LDS_PLANE_B = LDS_ROWS * LDS_COLS * 4;
constexpr int LDS_BUF = LDS_PLANE_A + LDS_PLANE_B;  // 12.8 KiB
constexpr int LDS_NBUF = 3;                         // ring of three: 38.4 KiB per workgroup, 4 workgroups per CU

struct ChunkInfo {
    int dx_min, dx_max, dy_min, dy_max;  // bounding box of the chunk's shifts over all epochs
    int unsafe;                          // any entry flagged SHIFT_UNSAFE
    int lds_ok;                          // every epoch's footprint fits the LDS stage buffer
    int pad[2];
};

// Per (chunk, epoch) footprint, packed for one 8-byte scalar load:
//   x = (dy_min << 16) | (dx_min & 0xffff)   origin of the staged region relative to the tile
//   y = (rows   << 16) | cols                64 + dx spread, TILE_ROWS + dy spread
using EpochBox = int2;
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
    const int* lds_off;        // [n_chunks][T][C] byte offset of the shifted tile inside plane A
    const int* epoch_invalid;  // [T] number of NO_DATA pixels in image t, then [T] = their total
    const int* global_box;     // {dx_min, dx_max, dy_min, dy_max} over every (candidate, epoch)
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
__device__ __forceinline__ int uniform_shift(float v, double t, bool* unsafe) {
    const double a = __dmul_rn((double)v, t);
    const double g = __dadd_rn(a, 0.5);
    const double fl = floor(g);
    const double frac = g - fl;
    // Guard band 2^-20 around the rounding boundary and |a| < 2^22: with start
    // coordinates |x| < 2^22 the two extra roundings of (x + a) + 0.5 move the
    // value by < 2^-28, so floor() cannot change ("shift table").
    if (!(fabs(a) < 4194304.0) || !(frac >= 9.5367431640625e-07 && frac <= 1.0 - 9.5367431640625e-07)) {
        *unsafe = true;
        return 0;
    }
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
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const double tm = times[t];
        int2 sh[C];
        int ex0 = INT32_MAX, ex1 = INT32_MIN, ey0 = INT32_MAX, ey1 = INT32_MIN;
        bool epoch_unsafe = false;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int ci = chunk * C + c;
            sh[c] = make_int2(0, 0);
            if (ci < n_cands) {
                bool unsafe = false;
                sh[c].x = uniform_shift(cands[ci].vx, tm, &unsafe);
                sh[c].y = uniform_shift(cands[ci].vy, tm, &unsafe);
                if (unsafe) {
                    sh[c].x = SHIFT_UNSAFE;
                    epoch_unsafe = true;
                } else {
                    ex0 = min(ex0, sh[c].x);
                    ex1 = max(ex1, sh[c].x);
                    ey0 = min(ey0, sh[c].y);
                    ey1 = max(ey1, sh[c].y);
                }
            }
        }
        const bool fits = !epoch_unsafe && ex0 <= ex1 && (ex1 - ex0) <= (LDS_COLS - WAVE) &&
                          (ey1 - ey0) <= (LDS_ROWS - TILE_ROWS) && ex0 > -30000 && ex1 < 30000 && ey0 > -30000 &&
                          ey1 < 30000;
        EpochBox box = make_int2(0, (TILE_ROWS << 16) | WAVE);
        if (fits) {
            box.x = (ey0 << 16) | (ex0 & 0xffff);
            box.y = ((TILE_ROWS + ey1 - ey0) << 16) | (WAVE + ex1 - ex0);
        } else {
            lds_bad = 1;
        }
        boxes[(size_t)chunk * T + t] = box;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const size_t e = ((size_t)chunk * T + t) * C + c;
            table[e] = sh[c];
            const bool real = (chunk * C + c) < n_cands;
            lds_off[e] = (fits && real) ? ((sh[c].y - ey0) * LDS_COLS + (sh[c].x - ex0)) * 8 : 0;
        }
        if (epoch_unsafe) any_unsafe = 1;
        if (ex0 <= ex1) {
            dx_min = min(dx_min, ex0);
            dx_max = max(dx_max, ex1);
            dy_min = min(dy_min, ey0);
            dy_max = max(dy_max, ey1);
        }
    }
    __shared__ int red[6][256];
    red[0][threadIdx.x] = dx_min;
    red[1][threadIdx.x] = dx_max;
    red[2][threadIdx.x] = dy_min;
    red[3][threadIdx.x] = dy_max;
    red[4][threadIdx.x] = any_unsafe;
    red[5][threadIdx.x] = lds_bad;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red[0][threadIdx.x] = min(red[0][threadIdx.x], red[0][threadIdx.x + s]);
            red[1][threadIdx.x] = max(red[1][threadIdx.x], red[1][threadIdx.x + s]);
            red[2][threadIdx.x] = min(red[2][threadIdx.x], red[2][threadIdx.x + s]);
            red[3][threadIdx.x] = max(red[3][threadIdx.x], red[3][threadIdx.x + s]);
            red[4][threadIdx.x] |= red[4][threadIdx.x + s];
            red[5][threadIdx.x] |= red[5][threadIdx.x + s];
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
        ci.lds_ok = (red[5][0] == 0 && red[4][0] == 0) ? 1 : 0;
        ci.pad[0] = ci.pad[1] = 0;
        chunks[chunk] = ci;
        if (!ci.lds_ok) atomicAdd(n_not_lds, 1);
        if (ci.dx_min <= ci.dx_max) {
            atomicMin(&global_box[0], ci.dx_min);
            atomicMax(&global_box[1], ci.dx_max);
            atomicMin(&global_box[2], ci.dy_min);
            atomicMax(&global_box[3], ci.dy_max);
        }
    }
}

// Number of NO_DATA pixels per image: an epoch without any lets the LDS path skip
// the validity plane whenever the staged footprint lies inside the image.
template <int NB>
__global__ __launch_bounds__(256) void kb_count_invalid_kernel(const void* __restrict__ psi_phi, uint64_t ppi,
                                                               int* __restrict__ counts) {
    const int t = blockIdx.y;
    int bad = 0;
    for (uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x; p < ppi; p += (uint64_t)gridDim.x * 256) {
        const uint64_t e = (uint64_t)t * ppi + p;
        if (NB == 4) {
            const float2 v = reinterpret_cast<const float2*>(psi_phi)[e];
            bad += !(__builtin_isfinite(v.x) && __builtin_isfinite(v.y));
        } else if (NB == 2) {
            const ushort2 v = reinterpret_cast<const ushort2*>(psi_phi)[e];
            bad += (v.x == 0 || v.y == 0);
        } else {
            const uchar2 v = reinterpret_cast<const uchar2*>(psi_phi)[e];
            bad += (v.x == 0 || v.y == 0);
        }
    }
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
    if ((threadIdx.x & 63) == 0 && bad != 0) {
        atomicAdd(&counts[t], bad);
        atomicAdd(&counts[gridDim.y], bad);  // total over all epochs
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

// Threshold / sigma-G / insertion of one chunk's C finished candidates.
template <int KS, int C, bool SIGMAG>
__device__ __forceinline__ void finish_chunk(const SearchArgs& a, const TileCoords& tc, int chunk,
                                             const float (&ps)[C], const float (&ph)[C], const int (&cnt)[C],
                                             TopK<KS>& top, const SigmaGScratch<WAVE>& scratch) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int cand = chunk * C + c;
        if (cand >= a.n_cands) break;  // uniform
        float lh = lh_from_sums(ps[c], ph[c]);
        bool take = !(cnt[c] < a.params.min_observations);
        if constexpr (SIGMAG) {
            // kernels.cu:201-203: only trajectories that pass the unclipped
            // thresholds are clipped (rare: min_lh rejects the noise); the rest
            // either fail kernels.cu:318-320 or are the obs_count == 0 corner.
            const bool clip = take && (cnt[c] != 0) && !(lh < a.params.min_lh);
            if (clip) {
                kb_trajectory trj;
                trj.x = tc.x;
                trj.y = tc.y;
                trj.vx = a.cands[cand].vx;
                trj.vy = a.cands[cand].vy;
                evaluate_trajectory_full<WAVE>(a.meta, a.psi_phi, a.times, a.params, &trj, &scratch);
                lh = trj.lh;
            }
            take = take && !(lh < a.params.min_lh);
        }
#ifndef KB_SKIP_INSERT
        if (take) top.insert(lh, cand);
#else
        if (take && lh == 12345.0f) top.insert(lh, cand);
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
// LDS-staged kernel
// ---------------------------------------------------------------------------
constexpr int STAGE_RR = (LDS_ROWS + TILE_ROWS - 1) / TILE_ROWS;  // staged rows per wave (stride TILE_ROWS)
constexpr int STAGE_G = 2;                                        // column groups of 64 lanes

// What one (tile, chunk, epoch) footprint needs, all wave-uniform.
struct Footprint {
    int x0, y0, rows, cols;  // image coordinates of the staged region's origin, its size
    bool clean;              // every staged pixel is inside the image and the image has no NO_DATA pixel
};

template <bool FAST>
__device__ __forceinline__ Footprint make_footprint(const SearchArgs& a, const TileCoords& tc, const EpochBox box,
                                                    int invalid_in_epoch) {
    Footprint f;
    f.x0 = tc.tile_x0 + box_dx(box);
    f.y0 = tc.tile_y0 + box_dy(box);
    f.rows = box_rows(box);
    f.cols = box_cols(box);
    f.clean = FAST ? true
                   : (invalid_in_epoch == 0 && f.x0 >= 0 && f.y0 >= 0 && (f.x0 + f.cols) <= a.W &&
                      (f.y0 + f.rows) <= a.H);
    return f;
}

// Issue the global loads of one footprint into registers.  g_lane = this lane's
// byte offset (wv * W + lane) * BYTES inside the footprint; row and column-group
// strides are added on the scalar side / as immediates.
template <int NB, bool FAST>
__device__ __forceinline__ void stage_load(const SearchArgs& a, const Footprint& f, int t, const TileCoords& tc,
                                           uint32_t g_lane, typename RawPair<NB>::type (&raw)[STAGE_RR][STAGE_G]) {
    using R = RawPair<NB>;
    using RT = typename R::type;
    constexpr int BYTES = 2 * fmt_bytes(NB);
    const char* image = reinterpret_cast<const char*>(a.psi_phi) + (uint64_t)t * a.meta.pixels_per_image * (uint64_t)BYTES;
    if (FAST || f.clean) {
        // Whole footprint inside the image: scalar base + one per-lane offset, no bounds tests.
        const char* base = image + ((int64_t)f.y0 * a.W + f.x0) * BYTES;
#pragma unroll
        for (int rr = 0; rr < STAGE_RR; ++rr) {
            if (tc.wv + TILE_ROWS * rr < f.rows) {  // uniform
                const char* row = base + (int64_t)rr * (TILE_ROWS * BYTES) * a.W;
                raw[rr][0] = *reinterpret_cast<const RT*>(row + g_lane);
                if (f.cols > WAVE) {  // uniform
                    if (tc.lane < f.cols - WAVE) raw[rr][1] = *reinterpret_cast<const RT*>(row + g_lane + WAVE * BYTES);
                }
            }
        }
    } else {
#pragma unroll
        for (int rr = 0; rr < STAGE_RR; ++rr) {
            const int r = tc.wv + TILE_ROWS * rr;  // wave-uniform
            const int gy = f.y0 + r;
            const bool row_ok = (r < f.rows) && ((unsigned)gy < (unsigned)a.H);
#pragma unroll
            for (int g = 0; g < STAGE_G; ++g) {
                const int col = tc.lane + WAVE * g;
                const int gx = f.x0 + col;
                const bool ok = row_ok && (col < f.cols) && ((unsigned)gx < (unsigned)a.W);
                raw[rr][g] = R::invalid();
                if (ok) raw[rr][g] = *reinterpret_cast<const RT*>(image + (uint32_t)(gy * a.W + gx) * (uint32_t)BYTES);
            }
        }
    }
}

// Publish one staged footprint into LDS ring slot BUF.
// l_lane = this lane's byte offset (wv * LDS_COLS + lane) * 8 in plane A.
template <int NB, bool FAST, int BUF>
__device__ __forceinline__ void stage_write(const SearchArgs& a, const Footprint& f, const TileCoords& tc, char* smem,
                                            int l_lane, const typename RawPair<NB>::type (&raw)[STAGE_RR][STAGE_G]) {
    using R = RawPair<NB>;
    char* pa = smem + BUF * LDS_BUF + l_lane;                       // plane A, this lane
    char* pb = smem + BUF * LDS_BUF + LDS_PLANE_A + (l_lane >> 1);  // plane B, this lane
#pragma unroll
    for (int rr = 0; rr < STAGE_RR; ++rr) {
        if (tc.wv + TILE_ROWS * rr < f.rows) {  // uniform
#pragma unroll
            for (int g = 0; g < STAGE_G; ++g) {
                if (g == 0 || (f.cols > WAVE && tc.lane < f.cols - WAVE)) {
                    constexpr int ROW_A = TILE_ROWS * LDS_COLS * 8, ROW_B = TILE_ROWS * LDS_COLS * 4;
                    float psi, phi;
                    R::decode(raw[rr][g], a, &psi, &phi);
                    if (FAST || f.clean) {
                        *reinterpret_cast<float2*>(pa + rr * ROW_A + g * WAVE * 8) = make_float2(psi, phi);
                    } else {
                        const bool valid = __builtin_isfinite(psi) && __builtin_isfinite(phi);
                        *reinterpret_cast<float2*>(pa + rr * ROW_A + g * WAVE * 8) =
                                valid ? make_float2(psi, phi) : make_float2(0.0f, 0.0f);
                        *reinterpret_cast<int*>(pb + rr * ROW_B + g * WAVE * 4) = valid ? 1 : 0;
                    }
                }
            }
        }
    }
}

// Ring-of-three pipeline state of one chunk.
template <int NB>
struct LdsPipe {
    typename RawPair<NB>::type raw[STAGE_RR][STAGE_G];  // footprint of epoch t+1, loads in flight
    Footprint f_cur;                                     // epoch t   (resident in ring slot t % 3)
    Footprint f_pending;                                 // epoch t+1 (in `raw`)
};

// One epoch t (ring slot BUF):
//   1. publish the footprint of epoch t+1 (its loads were issued one iteration ago) into slot BUF+1,
//   2. issue the loads of epoch t+2,
//   3. read this epoch's C shifted rows from slot BUF and accumulate in epoch order,
//   4. barrier.
template <int C, int NB, bool FAST, int BUF>
__device__ __forceinline__ void lds_epoch(const SearchArgs& a, const TileCoords& tc, char* smem, int t, int T,
                                          const EpochBox* __restrict__ boxes, const int* __restrict__ invalid,
                                          const int* __restrict__ offs, uint32_t g_lane, int l_lane,
                                          LdsPipe<NB>& p, int& clean_epochs, float (&ps)[C], float (&ph)[C],
                                          int (&cnt)[C]) {
    int off[C];
#pragma unroll
    for (int c = 0; c < C; ++c) off[c] = offs[t * C + c];  // scalar loads, consumed in step 3

    if (t + 1 < T) stage_write<NB, FAST, (BUF + 1) % LDS_NBUF>(a, p.f_pending, tc, smem, l_lane, p.raw);
    const Footprint f_now = p.f_cur;
    p.f_cur = p.f_pending;
    if (t + 2 < T) {
        p.f_pending = make_footprint<FAST>(a, tc, boxes[t + 2], FAST ? 0 : invalid[t + 2]);
        stage_load<NB, FAST>(a, p.f_pending, t + 2, tc, g_lane, p.raw);
    }

    const char* cur = smem + BUF * LDS_BUF + l_lane;  // compile-time ring slot
    float2 v[C];
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = *reinterpret_cast<const float2*>(cur + off[c]);
    if (FAST || f_now.clean) {
        if (!FAST) clean_epochs += 1;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            ps[c] += v[c].x;
            ph[c] += v[c].y;
        }
    } else {
        const char* curb = smem + BUF * LDS_BUF + LDS_PLANE_A + (l_lane >> 1);
        int ok[C];
#pragma unroll
        for (int c = 0; c < C; ++c) ok[c] = *reinterpret_cast<const int*>(curb + (off[c] >> 1));
#pragma unroll
        for (int c = 0; c < C; ++c) {
            ps[c] += v[c].x;
            ph[c] += v[c].y;
            cnt[c] += ok[c];
        }
    }
    __syncthreads();
}

// All T epochs of one chunk.  FAST: the workgroup's tile never leaves the image
// under any candidate shift and the stack has no NO_DATA pixel at all, so every
// footprint is clean, nothing is bounds-tested and obs_count is T.
template <int C, int NB, bool FAST>
__device__ __forceinline__ void lds_chunk(const SearchArgs& a, const TileCoords& tc, char* smem, int chunk,
                                          uint32_t g_lane, int l_lane, float (&ps)[C], float (&ph)[C], int (&cnt)[C]) {
    const int T = a.T;
    const EpochBox* __restrict__ boxes = a.boxes + (size_t)chunk * T;
    const int* __restrict__ offs = a.lds_off + (size_t)chunk * T * C;
    const int* __restrict__ invalid = a.epoch_invalid;
    int clean_epochs = 0;

    LdsPipe<NB> p;
    p.f_cur = make_footprint<FAST>(a, tc, boxes[0], FAST ? 0 : invalid[0]);
    stage_load<NB, FAST>(a, p.f_cur, 0, tc, g_lane, p.raw);
    stage_write<NB, FAST, 0>(a, p.f_cur, tc, smem, l_lane, p.raw);
    p.f_pending = p.f_cur;
    if (T > 1) {
        p.f_pending = make_footprint<FAST>(a, tc, boxes[1], FAST ? 0 : invalid[1]);
        stage_load<NB, FAST>(a, p.f_pending, 1, tc, g_lane, p.raw);
    }
    __syncthreads();

    int t = 0;
    for (; t + 2 < T; t += 3) {
        lds_epoch<C, NB, FAST, 0>(a, tc, smem, t, T, boxes, invalid, offs, g_lane, l_lane, p, clean_epochs, ps, ph, cnt);
        lds_epoch<C, NB, FAST, 1>(a, tc, smem, t + 1, T, boxes, invalid, offs, g_lane, l_lane, p, clean_epochs, ps, ph, cnt);
        lds_epoch<C, NB, FAST, 2>(a, tc, smem, t + 2, T, boxes, invalid, offs, g_lane, l_lane, p, clean_epochs, ps, ph, cnt);
    }
    if (t < T) {
        lds_epoch<C, NB, FAST, 0>(a, tc, smem, t, T, boxes, invalid, offs, g_lane, l_lane, p, clean_epochs, ps, ph, cnt);
        if (t + 1 < T)
            lds_epoch<C, NB, FAST, 1>(a, tc, smem, t + 1, T, boxes, invalid, offs, g_lane, l_lane, p, clean_epochs, ps,
                                      ph, cnt);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) cnt[c] += FAST ? T : clean_epochs;
}

template <int KS, int C, int NB, bool SIGMAG>
__global__ __launch_bounds__(256, (KS <= 8 ? 4 : (KS <= 16 ? 3 : 2))) void kb_search_lds(const SearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // ring of LDS_NBUF stage buffers
    const TileCoords tc = tile_coords(a);  // rows past the search area stay alive (barriers)
    TopK<KS> top;
    top.init();
    SigmaGScratch<WAVE> scratch = {};
    if constexpr (SIGMAG) scratch = make_scratch(a, tc);
    const int l_lane = (tc.wv * LDS_COLS + tc.lane) * 8;  // plane A offset of this lane
    const uint32_t g_lane = (uint32_t)(tc.wv * a.W + tc.lane) * (uint32_t)(2 * fmt_bytes(NB));  // footprint offset

    // Workgroup-uniform: can this tile take the validity-free pipeline for the whole search?
    const int* __restrict__ gb = a.global_box;
    const bool fast = a.epoch_invalid[a.T] == 0 && (tc.tile_x0 + gb[0] >= 0) && (tc.tile_x0 + WAVE + gb[1] <= a.W) &&
                      (tc.tile_y0 + gb[2] >= 0) && (tc.tile_y0 + TILE_ROWS + gb[3] <= a.H);

    for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
        float ps[C], ph[C];
        int cnt[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            ps[c] = 0.0f;
            ph[c] = 0.0f;
            cnt[c] = 0;
        }
        if (fast) {
            lds_chunk<C, NB, true>(a, tc, smem, chunk, g_lane, l_lane, ps, ph, cnt);
        } else {
            lds_chunk<C, NB, false>(a, tc, smem, chunk, g_lane, l_lane, ps, ph, cnt);
        }
        if (tc.row_active) finish_chunk<KS, C, SIGMAG>(a, tc, chunk, ps, ph, cnt, top, scratch);
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
static Workspace g_ws[2];  // 0: shift table + chunk info, 1: sigma-G scratch

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
static void debug_occupancy(const char* name, KernelT kernel, size_t lds) {
    if (std::getenv("KBMOD_DEBUG") == nullptr) return;
    int blocks = -1;
    hipFuncAttributes attr;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kernel, 256, lds);
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(kernel));
    std::fprintf(stderr, "[kbmod_hip] %s: %d blocks/CU, %d VGPRs, %zu B static LDS, %zu B scratch, dyn LDS %zu\n", name,
                 blocks, attr.numRegs, attr.sharedSizeBytes, attr.localSizeBytes, lds);
}

template <int KS, int NB, bool SIGMAG>
static void launch_variant(const SearchArgs& a, bool lds, hipStream_t stream) {
    debug_occupancy("kb_search_lds", kb_search_lds<KS, CHUNK, NB, SIGMAG>, LDS_NBUF * LDS_BUF);
    debug_occupancy("kb_search_direct", kb_search_direct<KS, CHUNK, NB, SIGMAG>, 0);
    if (lds) {
        hipLaunchKernelGGL((kb_search_lds<KS, CHUNK, NB, SIGMAG>), dim3(a.n_tiles), dim3(256), LDS_NBUF * LDS_BUF, stream, a);
    } else {
        hipLaunchKernelGGL((kb_search_direct<KS, CHUNK, NB, SIGMAG>), dim3(a.n_tiles), dim3(256), 0, stream, a);
    }
}

template <int KS, int NB>
static void launch_sigmag(const SearchArgs& a, bool sigmag, bool lds, hipStream_t stream) {
    if (sigmag)
        launch_variant<KS, NB, true>(a, lds, stream);
    else
        launch_variant<KS, NB, false>(a, lds, stream);
}

template <int KS>
static void launch_search(const SearchArgs& a, bool sigmag, bool lds, hipStream_t stream) {
    switch (a.meta.num_bytes) {
        case 1:
            if (a.fast_decode)
                launch_sigmag<KS, 10>(a, sigmag, lds, stream);
            else
                launch_sigmag<KS, 1>(a, sigmag, lds, stream);
            break;
        case 2:
            if (a.fast_decode)
                launch_sigmag<KS, 20>(a, sigmag, lds, stream);
            else
                launch_sigmag<KS, 2>(a, sigmag, lds, stream);
            break;
        default:
            launch_sigmag<KS, 4>(a, sigmag, lds, stream);
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
    a.epoch_invalid = nullptr;
    a.global_box = nullptr;

    EventTimer table_timer(stream, stats_out != nullptr);
    EventTimer search_timer(stream, stats_out != nullptr);
    std::lock_guard<std::mutex> lock(g_ws_mutex);

    float table_ms = 0.0f, search_ms = 0.0f;
    // bit 2 selects the LDS-staged kernel.  Measured on MI355X (profiles/r01_*): at 4 waves/SIMD the
    // per-epoch stage -> barrier -> read chain of kb_search_lds is latency-bound (12.9 ms on cfg2) while
    // kb_search_direct is bound by the vector-memory pipe (9.2 ms), so the direct kernel is the default.
    bool use_lds = (flags & 4u) != 0 && (flags & 1u) == 0;
    if (n_cands > 0) {
        const size_t table_bytes = (size_t)a.n_chunks * a.T * CHUNK * sizeof(int2);
        const size_t off_bytes = (size_t)a.n_chunks * a.T * CHUNK * sizeof(int);
        const size_t box_bytes = (size_t)a.n_chunks * a.T * sizeof(EpochBox);
        const size_t chunk_bytes = (size_t)a.n_chunks * sizeof(ChunkInfo);
        // per-epoch NO_DATA counts [T], their total [1], the not-LDS chunk counter [1], the global shift box [4]
        const size_t inv_bytes = ((size_t)a.T + 6) * sizeof(int);
        void* ws = nullptr;
        if (ensure_workspace(0, table_bytes + off_bytes + box_bytes + chunk_bytes + inv_bytes, &ws)) return 1;
        char* wsc = reinterpret_cast<char*>(ws);
        a.table = reinterpret_cast<const int2*>(wsc);
        a.lds_off = reinterpret_cast<const int*>(wsc + table_bytes);
        a.boxes = reinterpret_cast<const EpochBox*>(wsc + table_bytes + off_bytes);
        a.chunks = reinterpret_cast<const ChunkInfo*>(wsc + table_bytes + off_bytes + box_bytes);
        int* inv = reinterpret_cast<int*>(wsc + table_bytes + off_bytes + box_bytes + chunk_bytes);
        a.epoch_invalid = inv;
        int* n_not_lds = inv + a.T + 1;
        int* gbox = inv + a.T + 2;
        a.global_box = gbox;
        table_timer.begin();
        KB_HIP_TRY(hipMemsetAsync(inv, 0, inv_bytes, stream));
        static const int gbox_init[4] = {INT32_MAX, INT32_MIN, INT32_MAX, INT32_MIN};
        KB_HIP_TRY(hipMemcpyAsync(gbox, gbox_init, sizeof(gbox_init), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL((kb_shift_table_kernel<CHUNK>), dim3(a.n_chunks), dim3(256), 0, stream, cands_dev,
                           times_dev, a.n_cands, a.T, reinterpret_cast<int2*>(wsc),
                           reinterpret_cast<ChunkInfo*>(wsc + table_bytes + off_bytes + box_bytes),
                           reinterpret_cast<EpochBox*>(wsc + table_bytes + off_bytes),
                           reinterpret_cast<int*>(wsc + table_bytes), n_not_lds, gbox);
        KB_HIP_TRY(hipGetLastError());
        if (use_lds) {
            const dim3 grid((unsigned)std::min<uint64_t>((meta->pixels_per_image + 255) / 256, 1024), (unsigned)a.T);
            if (meta->num_bytes == 1)
                hipLaunchKernelGGL((kb_count_invalid_kernel<1>), grid, dim3(256), 0, stream, psi_phi_dev,
                                   meta->pixels_per_image, inv);
            else if (meta->num_bytes == 2)
                hipLaunchKernelGGL((kb_count_invalid_kernel<2>), grid, dim3(256), 0, stream, psi_phi_dev,
                                   meta->pixels_per_image, inv);
            else
                hipLaunchKernelGGL((kb_count_invalid_kernel<4>), grid, dim3(256), 0, stream, psi_phi_dev,
                                   meta->pixels_per_image, inv);
            KB_HIP_TRY(hipGetLastError());
            // The kernel choice needs one int back: are all chunks LDS-stageable?
            int not_lds = 0;
            KB_HIP_TRY(hipMemcpyAsync(&not_lds, n_not_lds, sizeof(int), hipMemcpyDeviceToHost, stream));
            KB_HIP_TRY(hipStreamSynchronize(stream));
            if (not_lds != 0) use_lds = false;
        }
        table_ms = table_timer.end();
    } else {
        use_lds = false;
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
        use_lds = false;
        if (sigmag)
            hipLaunchKernelGGL((kb_search_large_k<true>), dim3(a.n_tiles), dim3(256), 0, stream, a);
        else
            hipLaunchKernelGGL((kb_search_large_k<false>), dim3(a.n_tiles), dim3(256), 0, stream, a);
        variant = 99;
    } else if (a.K <= 8) {
        launch_search<8>(a, sigmag, use_lds, stream);
        variant = 8;
    } else if (a.K <= 16) {
        launch_search<16>(a, sigmag, use_lds, stream);
        variant = 16;
    } else {
        launch_search<32>(a, sigmag, use_lds, stream);
        variant = 32;
    }
    KB_HIP_TRY(hipGetLastError());
    search_ms = search_timer.end();

    if (stats_out != nullptr) {
        const uint64_t S = (uint64_t)sw * (uint64_t)sh;
        stats_out->search_kernel_ms = search_ms;
        stats_out->table_kernel_ms = table_ms;
        stats_out->num_evals = S * n_cands * meta->num_times;
        stats_out->algorithmic_bytes = stats_out->num_evals * 2ull * (uint64_t)meta->block_size +
                                       S * (uint64_t)a.K * 28ull + n_cands * 28ull + meta->num_times * 8ull;
        stats_out->kernel_variant = (use_lds ? 10000 : 0) + variant * 100 + meta->num_bytes * 10 + (sigmag ? 1 : 0);
        stats_out->num_search_launches = 1;
    } else {
        // kernels.cu:396 -- the reference call is synchronous.
        KB_HIP_TRY(hipStreamSynchronize(stream));
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
