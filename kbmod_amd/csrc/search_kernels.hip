// Shift-and-stack trajectory search for MI355X (gfx950).
//
// Replaces kernels/kernels.cu:252-397 of the reference (searchFilterImages +
// deviceSearchFilter) with a design built around one observation: for a fixed
// candidate velocity and epoch, floor(x + vx*t + 0.5) = x + d for every start
// pixel x, i.e. the "gather" is an integer translation of the start tile.
//
//   * kb_shift_table_kernel : one thread per (candidate, epoch) computes the
//     integer shift (dx, dy) in the reference's double arithmetic and proves it
//     valid for every start pixel (guard band around the rounding boundary;
//     entries that cannot be proven are flagged and take the exact path).
//   * kb_search_tiles       : one 64-lane wavefront owns 64 consecutive start
//     pixels of one row, so each (candidate, epoch) sample is ONE coalesced
//     512-byte row segment whose offset comes from scalar registers.  C
//     candidates are accumulated together (independent loads in flight, fp32
//     sums in strict epoch order), then pushed through a K-slot register-resident
//     top-K that reproduces the reference's swap-down insertion exactly.
//     The K winners are re-evaluated with exact per-lane positions to produce
//     flux / obs_count (and the sigma-G clipped likelihood).
//
// Numerics: fp32 sums in epoch order, correctly rounded sqrt/divide, double
// position arithmetic without FMA -- results equal the reference's host
// instantiation bit for bit (oracle: oracle/kbmod_oracle.c).
#include <algorithm>
#include <mutex>
#include <vector>

#include "kb_common.h"
#include "search_math.h"

#pragma clang fp contract(off)

namespace kb {

constexpr int SHIFT_UNSAFE = INT32_MIN;  // dx marker: no uniform shift proven for this (candidate, epoch)
constexpr int TILE_ROWS = 4;             // waves (rows) per 256-thread workgroup

struct ChunkInfo {
    int dx_min, dx_max, dy_min, dy_max;  // bounding box of the chunk's shifts over all epochs
    int unsafe;                          // any entry flagged SHIFT_UNSAFE
    int pad[3];
};

struct SearchArgs {
    const void* psi_phi;
    const double* times;
    const kb_trajectory* cands;
    kb_trajectory* results;
    const int2* table;        // [n_chunks][T][C]
    const ChunkInfo* chunks;  // [n_chunks]
    kb_psi_phi_meta meta;
    kb_search_params params;
    int T, W, H;
    int n_cands, n_chunks;
    int sw, sh;
    int tiles_x, tiles_y, n_tiles;
    int K;
    int force_exact;
    float* sg_scratch;  // sigma-G per-lane scratch (see launch code), or null
};

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
                                                             ChunkInfo* __restrict__ chunks) {
    const int chunk = blockIdx.x;
    int dx_min = INT32_MAX, dx_max = INT32_MIN, dy_min = INT32_MAX, dy_max = INT32_MIN, any_unsafe = 0;
    for (int e = threadIdx.x; e < T * C; e += blockDim.x) {
        const int t = e / C, c = e - t * C;
        const int ci = chunk * C + c;
        int2 s = make_int2(0, 0);
        if (ci < n_cands) {
            bool unsafe = false;
            const double tm = times[t];
            s.x = uniform_shift(cands[ci].vx, tm, &unsafe);
            s.y = uniform_shift(cands[ci].vy, tm, &unsafe);
            if (unsafe) {
                s.x = SHIFT_UNSAFE;
                any_unsafe = 1;
            } else {
                dx_min = min(dx_min, s.x);
                dx_max = max(dx_max, s.x);
                dy_min = min(dy_min, s.y);
                dy_max = max(dy_max, s.y);
            }
        }
        table[(size_t)chunk * T * C + e] = s;
    }
    __shared__ int red[5][256];
    red[0][threadIdx.x] = dx_min;
    red[1][threadIdx.x] = dx_max;
    red[2][threadIdx.x] = dy_min;
    red[3][threadIdx.x] = dy_max;
    red[4][threadIdx.x] = any_unsafe;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red[0][threadIdx.x] = min(red[0][threadIdx.x], red[0][threadIdx.x + s]);
            red[1][threadIdx.x] = max(red[1][threadIdx.x], red[1][threadIdx.x + s]);
            red[2][threadIdx.x] = min(red[2][threadIdx.x], red[2][threadIdx.x + s]);
            red[3][threadIdx.x] = max(red[3][threadIdx.x], red[3][threadIdx.x + s]);
            red[4][threadIdx.x] |= red[4][threadIdx.x + s];
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
        ci.pad[0] = ci.pad[1] = ci.pad[2] = 0;
        chunks[chunk] = ci;
    }
}

// ---------------------------------------------------------------------------
// sample decode
// ---------------------------------------------------------------------------
template <int NB>
struct Sample;

template <>
struct Sample<4> {
    static constexpr int BYTES = 8;
    __device__ static __forceinline__ void load(const char* base, uint32_t voff, float* psi, float* phi,
                                                const SearchArgs&) {
        const float2 v = *reinterpret_cast<const float2*>(base + voff);
        *psi = v.x;
        *phi = v.y;
    }
};
template <>
struct Sample<2> {
    static constexpr int BYTES = 4;
    __device__ static __forceinline__ void load(const char* base, uint32_t voff, float* psi, float* phi,
                                                const SearchArgs& a) {
        const ushort2 v = *reinterpret_cast<const ushort2*>(base + voff);
        *psi = (v.x == 0) ? NAN : decode_code((float)v.x, a.meta.psi_scale, a.meta.psi_min_val);
        *phi = (v.y == 0) ? NAN : decode_code((float)v.y, a.meta.phi_scale, a.meta.phi_min_val);
    }
};
template <>
struct Sample<1> {
    static constexpr int BYTES = 2;
    __device__ static __forceinline__ void load(const char* base, uint32_t voff, float* psi, float* phi,
                                                const SearchArgs& a) {
        const uchar2 v = *reinterpret_cast<const uchar2*>(base + voff);
        *psi = (v.x == 0) ? NAN : decode_code((float)v.x, a.meta.psi_scale, a.meta.psi_min_val);
        *phi = (v.y == 0) ? NAN : decode_code((float)v.y, a.meta.phi_scale, a.meta.phi_min_val);
    }
};

__device__ __forceinline__ void accumulate(float psi, float phi, bool ok, float& ps, float& ph, int& n) {
    const bool valid = ok && __builtin_isfinite(psi) && __builtin_isfinite(phi);
    // Adding +0.0f is the identity here: the running sums start at +0.0f and can
    // therefore never be -0.0f.
    ps += valid ? psi : 0.0f;
    ph += valid ? phi : 0.0f;
    n += valid ? 1 : 0;
}

// MODE 0: interior wave, table shifts, no per-lane bounds test.
// MODE 1: table shifts with per-lane bounds test (image edges / off-image starts).
// MODE 2: exact per-lane double positions (chunks with unproven shifts, or forced).
template <int C, int NB, int MODE>
__device__ __forceinline__ void accumulate_chunk(const SearchArgs& a, int chunk, int x, int y, int pix0,
                                                 float (&ps)[C], float (&ph)[C], int (&cnt)[C]) {
    using S = Sample<NB>;
    const int2* __restrict__ tab = a.table + (size_t)chunk * a.T * C;
    const uint64_t image_bytes = a.meta.pixels_per_image * (uint64_t)S::BYTES;
    const char* base = reinterpret_cast<const char*>(a.psi_phi);
#pragma unroll 2
    for (int t = 0; t < a.T; ++t) {
        if constexpr (MODE == 2) {
            const double tm = a.times[t];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int ci = min(chunk * C + c, a.n_cands - 1);
                int cx, cy;
                bool ok = predict_index(x, a.cands[ci].vx, tm, &cx);
                ok = predict_index(y, a.cands[ci].vy, tm, &cy) && ok;
                ok = ok && ((unsigned)cx < (unsigned)a.W) && ((unsigned)cy < (unsigned)a.H);
                const uint32_t voff = ok ? (uint32_t)(cy * a.W + cx) * (uint32_t)S::BYTES : 0u;
                float psi, phi;
                S::load(base, voff, &psi, &phi, a);
                accumulate(psi, phi, ok, ps[c], ph[c], cnt[c]);
            }
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int2 s = tab[t * C + c];  // wave-uniform -> scalar loads
                if constexpr (MODE == 0) {
                    const int off = s.y * a.W + s.x;
                    const uint32_t voff = (uint32_t)(pix0 + off) * (uint32_t)S::BYTES;
                    float psi, phi;
                    S::load(base, voff, &psi, &phi, a);
                    accumulate(psi, phi, true, ps[c], ph[c], cnt[c]);
                } else {
                    const int cx = x + s.x, cy = y + s.y;
                    const bool ok = ((unsigned)cx < (unsigned)a.W) && ((unsigned)cy < (unsigned)a.H);
                    const uint32_t voff = ok ? (uint32_t)(cy * a.W + cx) * (uint32_t)S::BYTES : 0u;
                    float psi, phi;
                    S::load(base, voff, &psi, &phi, a);
                    accumulate(psi, phi, ok, ps[c], ph[c], cnt[c]);
                }
            }
        }
        base += image_bytes;
    }
}

// ---------------------------------------------------------------------------
// the search kernel
// ---------------------------------------------------------------------------
template <int KS, int C, int NB, bool SIGMAG>
__global__ __launch_bounds__(256) void kb_search_tiles(const SearchArgs a) {
    // XCD-aware tile order: workgroup b runs on XCD (b % 8); give each XCD a
    // contiguous band of tiles so that its private L2 sees one image region.
    const int b = blockIdx.x;
    const int xcd = b & 7, local = b >> 3;
    const int q = a.n_tiles >> 3, r = a.n_tiles & 7;
    const int tile = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    const int ty = tile / a.tiles_x;
    const int tx = tile - ty * a.tiles_x;

    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int y_i = ty * TILE_ROWS + wv;
    if (y_i >= a.sh) return;  // whole wave
    const int x_i = tx * WAVE + lane;
    const int x = x_i + a.params.x_start_min;
    const int y = y_i + a.params.y_start_min;
    const int wave_x0 = tx * WAVE + a.params.x_start_min;
    const int pix0 = y * a.W + x;

    float s_lh[KS];
    int s_id[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        s_lh[s] = -FLT_MAX;
        s_id[s] = -1;
    }

    // Lane-interleaved sigma-G scratch: element i of this lane at base[i * 64].
    SigmaGScratch<WAVE> scratch;
    {
        const size_t wave_id = (size_t)blockIdx.x * TILE_ROWS + wv;
        float* base = SIGMAG ? a.sg_scratch + wave_id * (size_t)(4 * a.T) * WAVE + lane : nullptr;
        scratch.psi.p = base;
        scratch.phi.p = base + (size_t)a.T * WAVE;
        scratch.lc.p = base + (size_t)2 * a.T * WAVE;
        scratch.idx.p = reinterpret_cast<int*>(base + (size_t)3 * a.T * WAVE);
    }

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
        const bool interior = (wave_x0 + ci.dx_min >= 0) && (wave_x0 + WAVE - 1 + ci.dx_max < a.W) &&
                              (y + ci.dy_min >= 0) && (y + ci.dy_max < a.H);
        if (exact) {
            accumulate_chunk<C, NB, 2>(a, chunk, x, y, pix0, ps, ph, cnt);
        } else if (interior) {
            accumulate_chunk<C, NB, 0>(a, chunk, x, y, pix0, ps, ph, cnt);
        } else {
            accumulate_chunk<C, NB, 1>(a, chunk, x, y, pix0, ps, ph, cnt);
        }

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
                    trj.x = x;
                    trj.y = y;
                    trj.vx = a.cands[cand].vx;
                    trj.vy = a.cands[cand].vy;
                    evaluate_trajectory_full<WAVE>(a.meta, a.psi_phi, a.times, a.params, &trj, &scratch);
                    lh = trj.lh;
                }
                take = take && !(lh < a.params.min_lh);
            }
            if (take && lh > s_lh[KS - 1]) {
                // kernels.cu:323-330: strict '>' swap-down, reproduced slot by slot.
                float cl = lh;
                int cid = cand;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const bool g = cl > s_lh[s];
                    const float tl = s_lh[s];
                    const int ti = s_id[s];
                    s_lh[s] = g ? cl : tl;
                    s_id[s] = g ? cid : ti;
                    cl = g ? tl : cl;
                    cid = g ? ti : cid;
                }
            }
        }
    }

    if (x_i >= a.sw) return;

    // Epilogue: the K winners are re-evaluated with exact per-lane positions to
    // produce flux / obs_count (and the clipped values when sigma-G is on); the
    // likelihood this yields is bit-identical to the one that won the slot.
    kb_trajectory* out = a.results + ((size_t)y_i * a.sw + x_i) * a.K;
    for (int s = 0; s < a.K; ++s) {
        int id_s = -1;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            if (k == s) id_s = s_id[k];
        }
        kb_trajectory res;
        res.x = x;
        res.y = y;
        if (id_s < 0) {  // kernels.cu:293-301 placeholder
            res.vx = 0.0f;
            res.vy = 0.0f;
            res.lh = -FLT_MAX;
            res.flux = 0.0f;
            res.obs_count = 0;
        } else {
            res.vx = a.cands[id_s].vx;
            res.vy = a.cands[id_s].vy;
            evaluate_trajectory_full<WAVE>(a.meta, a.psi_phi, a.times, a.params, &res,
                                           SIGMAG ? &scratch : nullptr);
        }
        out[s] = res;
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

template <int KS, int C, int NB>
static void launch_search(const SearchArgs& a, bool sigmag, hipStream_t stream) {
    if (sigmag)
        hipLaunchKernelGGL((kb_search_tiles<KS, C, NB, true>), dim3(a.n_tiles), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((kb_search_tiles<KS, C, NB, false>), dim3(a.n_tiles), dim3(256), 0, stream, a);
}

template <int KS, int C>
static void launch_search_nb(const SearchArgs& a, bool sigmag, hipStream_t stream) {
    switch (a.meta.num_bytes) {
        case 1:
            launch_search<KS, C, 1>(a, sigmag, stream);
            break;
        case 2:
            launch_search<KS, C, 2>(a, sigmag, stream);
            break;
        default:
            launch_search<KS, C, 4>(a, sigmag, stream);
            break;
    }
}

constexpr int CHUNK = 8;  // candidates accumulated together per wave

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
    if (params.results_per_pixel > 32) {
        return fail("results_per_pixel > 32 is not supported by the register top-K path yet.");
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
    a.sg_scratch = nullptr;

    EventTimer table_timer(stream, stats_out != nullptr);
    EventTimer search_timer(stream, stats_out != nullptr);
    std::lock_guard<std::mutex> lock(g_ws_mutex);

    float table_ms = 0.0f, search_ms = 0.0f;
    if (n_cands > 0) {
        const size_t table_bytes = (size_t)a.n_chunks * a.T * CHUNK * sizeof(int2);
        const size_t chunk_bytes = (size_t)a.n_chunks * sizeof(ChunkInfo);
        void* ws = nullptr;
        if (ensure_workspace(0, table_bytes + chunk_bytes, &ws)) return 1;
        a.table = reinterpret_cast<const int2*>(ws);
        a.chunks = reinterpret_cast<const ChunkInfo*>(reinterpret_cast<char*>(ws) + table_bytes);
        table_timer.begin();
        hipLaunchKernelGGL((kb_shift_table_kernel<CHUNK>), dim3(a.n_chunks), dim3(256), 0, stream, cands_dev,
                           times_dev, a.n_cands, a.T, reinterpret_cast<int2*>(ws),
                           reinterpret_cast<ChunkInfo*>(reinterpret_cast<char*>(ws) + table_bytes));
        KB_HIP_TRY(hipGetLastError());
        table_ms = table_timer.end();
    } else {
        a.table = nullptr;
        a.chunks = nullptr;
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
    if (a.K <= 8) {
        launch_search_nb<8, CHUNK>(a, sigmag, stream);
        variant = 8;
    } else if (a.K <= 16) {
        launch_search_nb<16, CHUNK>(a, sigmag, stream);
        variant = 16;
    } else {
        launch_search_nb<32, CHUNK>(a, sigmag, stream);
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
        stats_out->kernel_variant = variant * 100 + meta->num_bytes * 10 + (sigmag ? 1 : 0);
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
