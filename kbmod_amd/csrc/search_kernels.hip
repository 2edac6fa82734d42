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

#include "search_common.h"

#pragma clang fp contract(off)

namespace kb {

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
    // |a| < 2^22 and start coordinates |x| < 2^22.  The reference rounds (x + a) and then ((x + a) + 0.5): each by at most
    // 2^-31 (values below 2^23: half an ulp of 2^-30); g itself carries the rounding of a + 0.5, at most 2^-31.  So the
    // value whose floor the reference takes lies within 1.5 * 2^-30 < 2^-29 of x + fl + frac, and floor() is x + fl for
    // every x whenever frac keeps 2^-29 from both ends: the guard band is 2^-27, four times that ("shift
    // table").  And when a has at most 29 fractional bits -- dyadic times, e.g. i / 64 days, times a float velocity --
    // x + a and (x + a) + 0.5 are exact (23 + 1 integer bits, 29 fractional ones: 53), so nothing is rounded at all and
    // the shift is uniform wherever frac lies, an exact half pixel included.
    if (!(fabs(a) < 4194304.0)) {
        *kind = 2;
        return 0;
    }
    const double scaled = a * 536870912.0;  // 2^29: exact
    const bool exact_sums = scaled == floor(scaled);
    if (!exact_sums && !(frac >= 7.450580596923828e-09 && frac <= 1.0 - 7.450580596923828e-09)) *kind = max(*kind, 1);
    return (int)fl;
}

// Tables behind edge_counts (search_lds.h): for chunk `blockIdx.x`, per axis and direction, per distance d = 0 .. D and
// candidate c of the chunk, the number of epochs whose integer shift along that axis and direction is at most d -- i.e. the
// epochs at which a start pixel d pixels from that edge of the image is still on it.  Rows of C uint16 (32 bytes), laid
// out [chunk][x+, x-, y+, y-][d][c].  *ok is cleared when some candidate's shifts do not grow monotonically in magnitude
// with one sign along an axis: only then are the four epoch sets leading runs of the epochs and their intersection the
// shortest of them.
template <int C>
__global__ __launch_bounds__(256) void kb_edge_count_kernel(const int2* __restrict__ table, int n_cands, int T, int D,
                                                            unsigned short* __restrict__ tab, int* __restrict__ ok) {
    // Per (axis, direction, candidate): a histogram of the distance from which an epoch counts, in LDS, filled by all
    // threads over (epoch, candidate) -- a thread per candidate walking its epochs is a chain of T dependent loads, 33 us --,
    // then running sums over d.
    extern __shared__ unsigned int edge_hist[];  // [2 axes][2 directions][C][D + 1]
    const int chunk = (int)blockIdx.x;
    const int2* tc = table + (size_t)chunk * T * C;
    const int D1 = D + 1;
    const int live = min(C, n_cands - chunk * C);
    for (int i = (int)threadIdx.x; i < 4 * C * D1; i += (int)blockDim.x) edge_hist[i] = 0u;
    __syncthreads();
    bool good = true;
    for (int i = (int)threadIdx.x; i < T * C; i += (int)blockDim.x) {
        const int e = i / C, c = i % C;
        if (c >= live) continue;
        const int2 s = tc[i];
        const int2 before = e > 0 ? tc[i - C] : make_int2(0, 0);
#pragma unroll
        for (int axis = 0; axis < 2; ++axis) {
            const int v = axis ? s.y : s.x, u = axis ? before.y : before.x;
            const int mag = v < 0 ? -v : v, mag_before = u < 0 ? -u : u;
            // magnitudes that never shrink and never change sign (a shrinking one would have to pass through zero)
            good = good && mag >= mag_before && mag <= D && (long long)v * (long long)u >= 0;
            // towards +: the epoch counts from distance max(v, 0) on; towards -: from max(-v, 0) on
            atomicAdd(&edge_hist[((size_t)(2 * axis) * C + c) * D1 + min(max(v, 0), D)], 1u);
            atomicAdd(&edge_hist[((size_t)(2 * axis + 1) * C + c) * D1 + min(max(-v, 0), D)], 1u);
        }
    }
    if (!good) atomicExch(ok, 0);
    __syncthreads();
    // running sums over d, one thread per (axis, direction, candidate); rows of C counts out
    if ((int)threadIdx.x < 4 * C) {
        const int c = (int)threadIdx.x % C, k = (int)threadIdx.x / C;  // k = 2 * axis + direction
        const unsigned int* h = edge_hist + ((size_t)k * C + c) * D1;
        unsigned short* rows = tab + ((size_t)chunk * 4 + k) * D1 * C;
        unsigned int run = 0;
        for (int d = 0; d < D1; ++d) {
            run += h[d];
            rows[(size_t)d * C + c] = (unsigned short)run;
        }
    }
}

template <int C>
__global__ __launch_bounds__(256) void kb_shift_table_kernel(const kb_trajectory* __restrict__ cands,
                                                             const double* __restrict__ times, int n_cands,
                                                             int T, int2* __restrict__ table,
                                                             ChunkInfo* __restrict__ chunks,
                                                             EpochBox* __restrict__ boxes,
                                                             int* __restrict__ lds_off,
                                                             int* __restrict__ n_not_lds,
                                                             int* __restrict__ global_box, int tile_rows, int col_quantum,
                                                             int max_cols) {
    // One workgroup per chunk, one thread per epoch (strided): the thread owns the
    // C shifts of its epoch, their bounding box and the LDS offsets derived from it.
    const int chunk = blockIdx.x;
    int dx_min = INT32_MAX, dx_max = INT32_MIN, dy_min = INT32_MAX, dy_max = INT32_MIN, any_unsafe = 0, lds_bad = 0;
    int rows_max = tile_rows;
    int sx_min = INT32_MAX, sx_max = INT32_MIN, sy_min = INT32_MAX, sy_max = INT32_MIN;  // staged epochs only
    int n_per_lane = 0;
    int not_monotone = 0;  // (chunks of XWIDE_CHUNK) a candidate whose shift shrinks or changes sign from one epoch to the next
    int cols_max = WAVE + col_quantum;  // widest staged footprint of the chunk
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
                if constexpr (C == XWIDE_CHUNK) {
                    // What the edge tables need (kb_edge_count_kernel checks it again on the device: the instance for chunks
                    // of 32 has no counting loop to fall back on, so its host must know before the launch): along both axes
                    // the magnitude of a candidate's shift never shrinks and its sign never changes from epoch to epoch.
                    if (t > 0 && kx == 0 && ky == 0) {
                        int jx = 0, jy = 0;
                        const int px = uniform_shift(cands[ci].vx, times[t - 1], &jx), py = uniform_shift(cands[ci].vy, times[t - 1], &jy);
                        const int ax = sh[c].x < 0 ? -sh[c].x : sh[c].x, ay = sh[c].y < 0 ? -sh[c].y : sh[c].y;
                        const int bx = px < 0 ? -px : px, by = py < 0 ? -py : py;
                        if (jx != 0 || jy != 0 || ax < bx || ay < by || (long long)sh[c].x * px < 0 || (long long)sh[c].y * py < 0) {
                            not_monotone += 1;
                        }
                    }
                }
                if (kx != 0 || ky != 0) {
                    sh[c].x = SHIFT_UNSAFE;
                    epoch_unsafe = true;
                }
            }
        }
        // Candidates past the end of the list (last chunk) repeat the chunk's first shift, so that the
        // loads kb_search_direct issues for them stay inside the chunk's bounding box.
#pragma unroll
        for (int c = 1; c < C; ++c) {
            if (chunk * C + c >= n_cands) sh[c] = sh[0];
        }
        // The slab starts on a multiple of LDS_ALIGN_PX columns of the padded frame (the host places the
        // image so that x_start_min + px0 is one): 16-byte pieces of a slab row are then 16-byte aligned
        // in HBM, 64-byte aligned for float pairs.  Measured on MI355X this made NO difference (8.46 ms
        // before and after); it is kept because it costs nothing and removes one variable, not because
        // the memory pipe was shown to need it.
        if (ex0 <= ex1) ex0 -= ((ex0 % LDS_ALIGN_PX) + LDS_ALIGN_PX) % LDS_ALIGN_PX;
        // A slab of (tile rows + dy spread) x max_cols 8-byte pairs (lds_cols of the chunk width) must fit one group buffer.
        const bool fits = !epoch_wild && ex0 <= ex1 && (ex1 - ex0) <= (max_cols - WAVE) &&
                          (tile_rows + ey1 - ey0) * max_cols * 8 <= lds_group_bytes(tile_rows) && ex0 > -30000 && ex1 < 30000 &&
                          ey0 > -30000 && ey1 < 30000;
        EpochBox box = make_int2(0, (tile_rows << 16) | WAVE);
        if (fits) {
            box.x = (ey0 << 16) | (ex0 & 0xffff);
            box.y = ((tile_rows + ey1 - ey0) << 16) | (WAVE + ex1 - ex0);
            rows_max = max(rows_max, tile_rows + ey1 - ey0);
            cols_max = max(cols_max, WAVE + ex1 - ex0);
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
            // (the slab offsets need the chunk's pitch: second pass below; this pass leaves the case)
            lds_off[e] = !fits ? LDS_OFF_UNSTAGED : (epoch_unsafe ? LDS_OFF_PER_LANE : (real ? 1 : 0));
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
    __shared__ int red[14][256];
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
    red[12][threadIdx.x] = cols_max;
    red[13][threadIdx.x] = not_monotone;
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
            red[12][threadIdx.x] = max(red[12][threadIdx.x], red[12][threadIdx.x + s]);
            red[13][threadIdx.x] += red[13][threadIdx.x + s];
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
        ci.cols = min(max_cols, (red[12][0] + col_quantum - 1) / col_quantum * col_quantum);
        {
            const int stride = (ci.rows_max * ci.cols * 8 + 1023) & ~1023;  // (chunk_plan of search_lds.h, float pairs)
            ci.e_even = group_epochs(T, tile_rows, stride, true);
            ci.e_any = group_epochs(T, tile_rows, stride, false);
            ci.t_over_e_even = T / ci.e_even;
            ci.t_over_e_any = T / ci.e_any;
            ci.cols_inv = (int)(((1u << 20) + (unsigned)ci.cols - 1u) / (unsigned)ci.cols);
            ci.pad[0] = ci.pad[1] = ci.pad[2] = 0;
        }
        chunks[chunk] = ci;
        atomicMax(&global_box[6], ci.rows_max * ci.cols);  // largest slab of the search, in pixels
        if (red[13][0] != 0) atomicAdd(&global_box[7], red[13][0]);
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
    // Second pass: slab offsets of the uniformly shifted tile at the chunk's pitch (>= 0; LDS_OFF_PER_LANE: staged,
    // but the lanes find their own pixel inside the slab; LDS_OFF_UNSTAGED: not staged at all).
    const int cols = min(max_cols, (red[12][0] + col_quantum - 1) / col_quantum * col_quantum);
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const EpochBox box = boxes[(size_t)chunk * T + t];  // written by this thread above
        if (box.x == BOX_NOT_STAGED) continue;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const size_t e = ((size_t)chunk * T + t) * C + c;
            if (lds_off[e] == 1) {
                const int2 sh = table[e];
                lds_off[e] = ((sh.y - box_dy(box)) * cols + (sh.x - box_dx(box))) * 8;
            }
        }
    }
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
// n_invalid becomes non-zero when the image holds any NO_DATA pixel (a lower bound of their number).
template <int NB, bool CANON>
__global__ __launch_bounds__(256) void kb_pad_kernel(const SearchArgs a, int Hp, int px0, int py0,
                                                     void* __restrict__ padded, int* __restrict__ n_invalid, int stream_out) {
    using R = RawPair<NB>;
    const int t = blockIdx.z;
    const int y = blockIdx.y;
    const int sy = y - py0;
    int bad = 0;
    // Two pixels of the frame per thread (its pitch is a multiple of 16 pixels): canonical pairs leave as one 16-byte
    // store -- non-temporal when the frame is beyond the Infinity Cache (stream_out: it will not be read before it has
    // been evicted anyway) --, which is what a streaming copy needs to reach the device's rate (tools/ubench/copybench.hip).
    for (int x = 2 * (blockIdx.x * 256 + threadIdx.x); x < a.Wp; x += 2 * gridDim.x * 256) {
        const size_t d = ((size_t)t * Hp + y) * a.Wp + x;
        float2 v[2];
        typename R::type raws[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int sx = x + k - px0;
            const bool in = sx >= 0 && sx < a.W && sy >= 0 && sy < a.H;
            const size_t sidx = ((size_t)t * a.H + (in ? sy : 0)) * a.W + (in ? sx : 0);
            const typename R::type raw = in ? reinterpret_cast<const typename R::type*>(a.psi_phi)[sidx] : R::invalid();
            float psi, phi;
            R::decode(raw, a, &psi, &phi);
            const bool valid = in && __builtin_isfinite(psi) && __builtin_isfinite(phi);
            bad += (in && !valid) ? 1 : 0;
            raws[k] = raw;
            v[k] = make_float2(0.0f, -0.0f);
            if (valid) {
                v[k] = make_float2(psi, phi);
                if (__float_as_uint(v[k].y) == 0x80000000u) v[k].y = 0.0f;
            }
        }
        if (CANON) {
            typedef float Quad __attribute__((ext_vector_type(4)));
            Quad q = {v[0].x, v[0].y, v[1].x, v[1].y};
            Quad* dst = reinterpret_cast<Quad*>(reinterpret_cast<float2*>(padded) + d);
            if (stream_out) {
                __builtin_nontemporal_store(q, dst);
            } else {
                *dst = q;
            }
        } else {
            reinterpret_cast<typename R::type*>(padded)[d] = raws[0];
            reinterpret_cast<typename R::type*>(padded)[d + 1] = raws[1];
        }
    }
    // Only "none at all" matters to the search (its count-free specialisation): once the counter is known to
    // be non-zero further waves skip the atomic, which would otherwise serialise a masked stack's whole copy.
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
    if ((threadIdx.x & 63) == 0 && bad != 0 && __atomic_load_n(n_invalid, __ATOMIC_RELAXED) == 0) atomicAdd(n_invalid, bad);
}

// Slab references, once the host has fixed the padded frame: byte offset of the slab origin relative to
// the tile's own pixel (the search kernel is spared the 64-bit index arithmetic per (chunk, epoch)) and
// the slab's size, one 16-byte scalar load per (chunk, epoch).
__global__ __launch_bounds__(256) void kb_slab_ref_kernel(const EpochBox* __restrict__ boxes,
                                                          const ChunkInfo* __restrict__ chunks, int64_t n, int T, int Hp,
                                                          int Wp, int px0, int py0, int pair_bytes,
                                                          SlabRef* __restrict__ refs, const int* __restrict__ lds_off,
                                                          int tile_rows, int* __restrict__ lds_fold, int chunk_c) {
    const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (slot >= n + SLAB_REF_SLACK) return;
    const int64_t i = slot < n ? slot : n - 1;  // the entries of slack repeat the last one (loaded, never summed)
    const EpochBox box = boxes[i];
    const int t = (int)(i % T);
    SlabRef r;
    // an epoch that is not staged copies the slab at the tile's own pixel (inside the frame, never read by
    // the sums): the search loop is spared a test per epoch
    r.origin = (box.x == BOX_NOT_STAGED)
                       ? (((int64_t)t * Hp + py0) * Wp + px0) * (int64_t)pair_bytes
                       : (((int64_t)t * Hp + box_dy(box) + py0) * Wp + box_dx(box) + px0) * (int64_t)pair_bytes;
    const ChunkInfo ci = chunks[i / T];
    // as tall as THIS epoch's shift box (the hand-scheduled loop skips the pieces behind it); the chunk's pitch
    r.bytes = ((box.x == BOX_NOT_STAGED) ? ci.rows_max : box_rows(box)) * ci.cols * pair_bytes;
    r.pad = 0;
    refs[slot] = r;
    // the offsets of the hand-scheduled loop (float-staged kernels: 8-byte pairs in LDS).  Groups start at multiples of E
    // epochs (chunk_plan of search_lds.h), slab e of a group sits e strides into the group buffer.
    const int stride = (ci.rows_max * ci.cols * 8 + 1023) & ~1023;
    // (the table is read by the hand-scheduled instances only: groups of an even number of epochs for chunks of 8 and 16, of
    // any number for chunks of 32 -- chunk_plan of search_lds.h)
    const int E = group_epochs(T, tile_rows, stride, chunk_c != XWIDE_CHUNK);
    const int place = (t % E) * stride;
#pragma unroll
    for (int c = 0; c < chunk_c; ++c) {
        const int o = lds_off[i * chunk_c + c];
        lds_fold[slot * chunk_c + c] = o >= 0 ? o + place : o;
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

// The same merge on the 16-byte records of kb_device_search_compact (the exchange format between GPUs),
// writing full trajectories.  A workgroup owns 256 consecutive pixels: the n_lists x 256 x K records are
// read as contiguous runs (256 * K * 16 bytes per list), each thread then merges its pixel's lists out
// of registers -- the head record of every list -- advancing one list per output slot.
template <int NL>  // upper bound of n_lists: lists live in registers, every loop is unrolled
__global__ __launch_bounds__(256) void kb_merge_compact_kernel(const kb_compact_result* __restrict__ lists, int n_lists,
                                                               uint64_t n_pixels, int K, int sw, int x_min, int y_min,
                                                               const kb_trajectory* __restrict__ all_cands,
                                                               uint64_t n_all_cands, kb_trajectory* __restrict__ out) {
    const uint64_t pix = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= n_pixels) return;
    const uint64_t list_stride = n_pixels * (uint64_t)K;
    int head[NL];
    float head_lh[NL];
#pragma unroll
    for (int r = 0; r < NL; ++r) {
        head[r] = (r < n_lists) ? 0 : K;  // lists past n_lists are exhausted from the start
        head_lh[r] = (r < n_lists) ? lists[(uint64_t)r * list_stride + pix * K].lh : -FLT_MAX;
    }
    const int y_i = (int)(pix / (uint64_t)sw), x_i = (int)(pix - (uint64_t)y_i * (uint64_t)sw);
    for (int s = 0; s < K; ++s) {
        int best = -1, best_head = 0;
        float best_lh = 0.0f;
#pragma unroll
        for (int r = 0; r < NL; ++r) {  // strict '>': equal likelihoods go to the lower list
            const bool better = head[r] < K && (best < 0 || head_lh[r] > best_lh);
            best = better ? r : best;
            best_head = better ? head[r] : best_head;
            best_lh = better ? head_lh[r] : best_lh;
        }
        const uint64_t at = (uint64_t)best * list_stride + pix * K + best_head;
        const kb_compact_result rec = lists[at];
        const float next_lh = (best_head + 1 < K) ? lists[at + 1].lh : -FLT_MAX;
#pragma unroll
        for (int r = 0; r < NL; ++r) {
            if (r == best) {
                head[r] = best_head + 1;
                head_lh[r] = next_lh;
            }
        }
        kb_trajectory res = placeholder_result(x_i + x_min, y_i + y_min);
        if (rec.cand >= 0 && (uint64_t)rec.cand < n_all_cands) {
            res.vx = all_cands[rec.cand].vx;
            res.vy = all_cands[rec.cand].vy;
            res.lh = rec.lh;
            res.flux = rec.flux;
            res.obs_count = rec.obs_count;
        }
        out[pix * K + s] = res;
    }
}

// Tie-exact merge (kb_merge_compact_exact).  The reference's insertion (kernels.cu:323-330) is not a stable
// shift: a run of equal likelihoods rotates whenever something is inserted in front of it, and loses its first
// member when it sits at the end of a full list, so which members of a tie a pixel keeps depends on the order of
// ALL candidates.  What that order can influence is bounded, though.  With v = the K-th largest likelihood of the
// pixel, the final list is the result of the reference's insertion over just
//     G = { candidates with lh > v }  +  { the first K candidates, by index, with lh == v },
// in candidate order: smaller values never touch the part of the list at or above v; every candidate above v
// enters (there are fewer than K); and a candidate equal to v enters only while fewer than K candidates >= v have
// arrived, so only the first K of them can (search_kernels.hip; brute-force check in
// tests/test_tie_exact_merge.py).  G lies inside the first 2K - 1 entries of the per-pixel list ordered by
// (lh descending, candidate ascending) -- a total order, under which the per-device lists (built by STABLE
// insertion, flag 512, 2K slots each) merge exactly.  So: merge the first K2 entries of that order, then replay G.
// Where the first K + 1 merged values are strictly decreasing nothing can rotate and the merged prefix is the answer.
// (merge_exact_pixel: search_math.h, shared with the host twin)
__global__ __launch_bounds__(64) void kb_merge_compact_exact_kernel(const kb_compact_result* __restrict__ lists, int n_lists,
                                                                    uint64_t n_pixels, int K2, int K, int sw, int x_min,
                                                                    int y_min, const kb_trajectory* __restrict__ all_cands,
                                                                    uint64_t n_all_cands, kb_trajectory* __restrict__ out) {
    const uint64_t pix = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (pix >= n_pixels) return;
    const uint64_t list_stride = n_pixels * (uint64_t)K2;
    const kb_compact_result* mine = lists + pix * (uint64_t)K2;
    auto read = [&](int r, int pos) { return mine[(uint64_t)r * list_stride + pos]; };
    MergedEntry merged[MERGE_EXACT_MAX_K2];
    int heads[MERGE_MAX_LISTS];
    int slots[MERGE_EXACT_MAX_K2];
    const int n_out = merge_exact_pixel(read, n_lists, K2, K, merged, heads, slots);
    const int y_i = (int)(pix / (uint64_t)sw), x_i = (int)(pix - (uint64_t)y_i * (uint64_t)sw);
    for (int s = 0; s < K; ++s) {
        kb_trajectory res = placeholder_result(x_i + x_min, y_i + y_min);
        if (s < n_out && slots[s] >= 0) {
            const uint32_t at = merged[slots[s]].at;
            const kb_compact_result rec = read((int)(at / (uint32_t)K2), (int)(at % (uint32_t)K2));
            if ((uint64_t)rec.cand < n_all_cands) {
                res.vx = all_cands[rec.cand].vx;
                res.vy = all_cands[rec.cand].vy;
                res.lh = rec.lh;
                res.flux = rec.flux;
                res.obs_count = rec.obs_count;
            }
        }
        out[pix * (uint64_t)K + s] = res;
    }
}

// The merge of the exchange with K records per device (kb_merge_compact_repairable; merge_fold_pixel, search_math.h): the
// fold over the devices' lists in candidate order.  Every pixel the fold decides is written; a pixel with a suspect slice
// -- a dropped candidate may tie with the list's last slot -- is appended to `hazards` for kb_repair_pixels.
// KT > 0: K = KT is a compile-time constant (lists of up to 8: what the exchange runs) and everything lives in registers;
// KT = 0: any K up to 32 through merge_fold_pixel itself, its two arrays in scratch memory.
template <int KT>
__global__ __launch_bounds__(256) void kb_merge_compact_repairable_kernel(const kb_compact_result* __restrict__ lists, int n_lists,
                                                                          uint64_t n_pixels, int K_any, int sw, int x_min, int y_min,
                                                                          const kb_trajectory* __restrict__ all_cands,
                                                                          uint64_t n_all_cands, kb_trajectory* __restrict__ out,
                                                                          uint32_t* __restrict__ hazards,
                                                                          unsigned long long* __restrict__ n_hazards) {
    constexpr int KA = KT > 0 ? KT : MERGE_EXACT_MAX_K2;  // array length
    const int K = KT > 0 ? KT : K_any;
    const uint64_t pix_raw = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (pix_raw - (uint64_t)(threadIdx.x & 63u) >= n_pixels) return;  // (whole waves past the end)
    const bool live = pix_raw < n_pixels;  // (the lanes past the end of the last wave stay: they help to stage)
    const uint64_t pix = live ? pix_raw : n_pixels - 1;
    const uint64_t list_stride = n_pixels * (uint64_t)K;
    kb_compact_result state[KA];
    bool hazard = false;
    if constexpr (KT > 0) {
        // merge_fold_pixel with everything in registers: fully unrolled, no array indexed by a run-time value (with K a
        // run-time value the compiler turns even a chain of selects over the slots back into an indexed access, and the lists
        // went to scratch memory: 0.67 ms for 512 x 512 pixels x 8 lists against the search's 2.4).
        // A pixel's K records of one list are ONE run of K x 16 bytes; read record by record, lane by lane, every load touches
        // 64 lines.  Each wave therefore copies its 64 pixels' records of a list -- contiguous in the list -- with coalesced
        // 16-byte loads into its own patch of LDS (pixel pitch K + 1 records: conflict-free for the 16-byte reads) and reads
        // them there.
        __shared__ __attribute__((aligned(16))) kb_compact_result patch[256 / 64][64 * (KT + 1)];
        const int lane = (int)(threadIdx.x & 63u);
        kb_compact_result* mine_patch = patch[threadIdx.x >> 6];
        const uint64_t wave_pix0 = pix_raw - (uint64_t)lane;
        const int wave_recs = (int)(min((uint64_t)64, n_pixels - wave_pix0) * (uint64_t)KT);
        auto stage = [&](int r) {
            const kb_compact_result* src = lists + (uint64_t)r * list_stride + wave_pix0 * (uint64_t)KT;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                const int c = j * 64 + lane;  // record c of the wave's run: pixel c / K, slot c % K
                if (c < wave_recs) mine_patch[(c / KT) * (KT + 1) + (c % KT)] = src[c];
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        const kb_compact_result none = {-FLT_MAX, 0.0f, -1, 0};
        stage(0);
#pragma unroll
        for (int s = 0; s < KT; ++s) {
            const kb_compact_result rec = mine_patch[lane * (KT + 1) + s];
            state[s] = rec.cand >= 0 ? rec : none;  // kernels.cu:293-301
        }
        for (int r = 1; r < n_lists; ++r) {
            stage(r);
            const float tail_before = state[KT - 1].lh;
            kb_compact_result rec[KT];
            bool can[KT];
            bool any = false;
#pragma unroll
            for (int s = 0; s < KT; ++s) {
                rec[s] = mine_patch[lane * (KT + 1) + s];
                can[s] = rec[s].cand >= 0 && rec[s].lh > tail_before;  // (what is not above the last slot moves nothing)
                any = any || can[s];
            }
            if (!any) continue;
            const kb_compact_result last = rec[KT - 1];
            // candidate order: an odd-even transposition network over the records (compile-time pairs: a pick by rank, written
            // as a chain of selects over the array, is turned back into an indexed access and the array into scratch memory);
            // the records that cannot enter sort to the end
            int key[KT];
#pragma unroll
            for (int i = 0; i < KT; ++i) key[i] = can[i] ? rec[i].cand : 0x7fffffff;
#pragma unroll
            for (int pass = 0; pass < KT; ++pass) {
#pragma unroll
                for (int i = pass & 1; i + 1 < KT; i += 2) {
                    const bool swap = key[i + 1] < key[i];
                    const int ka = key[i], kb2 = key[i + 1];
                    key[i] = swap ? kb2 : ka;
                    key[i + 1] = swap ? ka : kb2;
                    const kb_compact_result a = rec[i], b = rec[i + 1];
                    rec[i].lh = swap ? b.lh : a.lh;
                    rec[i].flux = swap ? b.flux : a.flux;
                    rec[i].cand = swap ? b.cand : a.cand;
                    rec[i].obs_count = swap ? b.obs_count : a.obs_count;
                    rec[i + 1].lh = swap ? a.lh : b.lh;
                    rec[i + 1].flux = swap ? a.flux : b.flux;
                    rec[i + 1].cand = swap ? a.cand : b.cand;
                    rec[i + 1].obs_count = swap ? a.obs_count : b.obs_count;
                }
            }
#pragma unroll
            for (int i = 0; i < KT; ++i) {
                if (key[i] == 0x7fffffff) continue;  // (cannot enter; per lane)
                kb_compact_result in = rec[i];
#pragma unroll
                for (int s = 0; s < KT; ++s) {  // kernels.cu:323-330
                    if (in.lh > state[s].lh) {
                        const kb_compact_result t = state[s];
                        state[s] = in;
                        in = t;
                    }
                }
            }
            hazard = hazard || (last.cand >= 0 && last.lh > tail_before && last.lh == state[KT - 1].lh);
        }
    } else {
        const kb_compact_result* mine = lists + pix * (uint64_t)K;
        auto read = [&](int r, int pos) { return mine[(uint64_t)r * list_stride + pos]; };
        kb_compact_result recs[KA];
        uint64_t suspects = 0;
        hazard = merge_fold_pixel(read, n_lists, K, state, recs, &suspects);
    }
    if (!live) return;
    if (hazard) {
        hazards[atomicAdd(n_hazards, 1ull)] = (uint32_t)pix;  // (its slots are re-made from the stack)
        return;
    }
    const int y_i = (int)(pix / (uint64_t)sw), x_i = (int)(pix - (uint64_t)y_i * (uint64_t)sw);
#pragma unroll
    for (int s = 0; s < KA; ++s) {
        if (s >= K) break;
        kb_trajectory res = placeholder_result(x_i + x_min, y_i + y_min);
        const kb_compact_result rec = state[s];
        if (rec.cand >= 0 && (uint64_t)rec.cand < n_all_cands) {
            res.vx = all_cands[rec.cand].vx;
            res.vy = all_cands[rec.cand].vy;
            res.lh = rec.lh;
            res.flux = rec.flux;
            res.obs_count = rec.obs_count;
        }
        out[pix * (uint64_t)K + s] = res;
    }
}

// evaluate_trajectory_full (search_math.h) without the sigma-G branch and with the samples of REPAIR_BATCH epochs requested
// before the first is summed: the same positions (predict_index), the same bounds rule and decode (read_psi_phi), the same
// sums in epoch order -- the same bits --, but the loads of a batch are independent of each other, so a lane waits for one
// memory latency per batch instead of one per epoch (the rolled loop ran 700 pixels in 0.7 ms; this one in under 0.1).
constexpr int REPAIR_BATCH = 32;
template <int NB>
__device__ __forceinline__ void evaluate_plain_batched(const kb_psi_phi_meta& m, const void* __restrict__ arr,
                                                       const double* __restrict__ times, kb_trajectory* c) {
    float psi_sum = 0.0f, phi_sum = 0.0f;
    int num_seen = 0;
    const int T = (int)m.num_times;
    for (int i0 = 0; i0 < T; i0 += REPAIR_BATCH) {
        float psi[REPAIR_BATCH], phi[REPAIR_BATCH];
#pragma unroll
        for (int j = 0; j < REPAIR_BATCH; ++j) {
            const int i = min(i0 + j, T - 1);  // (past the end: a second read of the last epoch, not summed)
            const double t = times[i];
            int cx, cy;
            const bool okx = predict_index(c->x, c->vx, t, &cx);
            const bool oky = predict_index(c->y, c->vy, t, &cy);
            const bool in = okx && oky && cy >= 0 && cx >= 0 && (uint64_t)cy < m.height && (uint64_t)cx < m.width;
            const uint64_t pix = in ? m.pixels_per_image * (uint64_t)i + (uint64_t)cy * m.width + (uint64_t)cx : 0ull;
            if (NB == 4) {
                const float2 v = reinterpret_cast<const float2*>(arr)[pix];
                psi[j] = in ? v.x : NAN;
                phi[j] = in ? v.y : NAN;
            } else {
                float pv, fv;
                if (NB == 1) {
                    const uchar2 v = reinterpret_cast<const uchar2*>(arr)[pix];
                    pv = (float)v.x;
                    fv = (float)v.y;
                } else {
                    const ushort2 v = reinterpret_cast<const ushort2*>(arr)[pix];
                    pv = (float)v.x;
                    fv = (float)v.y;
                }
                psi[j] = (!in || pv == 0.0f) ? NAN : decode_code(pv, m.psi_scale, m.psi_min_val);
                phi[j] = (!in || fv == 0.0f) ? NAN : decode_code(fv, m.phi_scale, m.phi_min_val);
            }
        }
#pragma unroll
        for (int j = 0; j < REPAIR_BATCH; ++j) {
            if (i0 + j < T && __builtin_isfinite(psi[j]) && __builtin_isfinite(phi[j])) {
                psi_sum += psi[j];
                phi_sum += phi[j];
                num_seen += 1;
            }
        }
    }
    c->obs_count = num_seen;
    c->lh = lh_from_sums(psi_sum, phi_sum);
    c->flux = flux_from_sums(psi_sum, phi_sum);
}

// kb_repair_pixels: one workgroup of REPAIR_WAVES wavefronts per listed start pixel.  Lanes = candidates: every wave
// evaluates 64 of a block of REPAIR_WAVES x 64 consecutive candidates -- the reference's evaluateTrajectory at exact
// positions (evaluate_plain_batched: evaluate_trajectory_full's arithmetic, what kb_search_large_k and the epilogues run) --
// into LDS, then wave 0 runs the reference's insertion over the block in candidate order (kernels.cu:304-331): the list
// lives in LDS (K x 16 bytes), the likelihood to beat in a scalar of wave 0.
// With the exchange's lists at hand (`lists` != null, list r = the candidates [begin[r], begin[r + 1]) in job-wide order)
// the kernel walks merge_fold_pixel's fold and evaluates only the SUSPECT slices again; every other list's K records (put
// back into candidate order) stand for its slice: an eighth of the evaluations at eight ranks, and the scattered loads of
// the evaluations are what this kernel costs.
constexpr int REPAIR_WAVES = 8;
constexpr int REPAIR_MAX_LISTS = 64;
struct RepairLists {
    const kb_compact_result* lists;  // [n_lists][n_pixels][K], or null: every candidate is evaluated
    uint64_t n_pixels;
    int n_lists;
    int begin[REPAIR_MAX_LISTS + 1];
};
struct RepairEval {  // one evaluated candidate of the block in flight
    float lh, flux;
    int obs;
};
template <int NB>
__global__ __launch_bounds__(REPAIR_WAVES * WAVE) void kb_repair_pixels_kernel(const kb_psi_phi_meta meta, const void* __restrict__ psi_phi,
                                                                               const double* __restrict__ times, const kb_search_params params,
                                                                               const kb_trajectory* __restrict__ cands, uint64_t n_cands,
                                                                               const uint32_t* __restrict__ pixels, uint64_t n_listed,
                                                                               const RepairLists rl, kb_trajectory* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char repair_smem[];
    constexpr int BLOCK = REPAIR_WAVES * WAVE;
    const int K = (int)params.results_per_pixel;
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & (WAVE - 1));
    // LDS: the block's evaluations, the list, one word of wave 0's decisions
    RepairEval* evals = reinterpret_cast<RepairEval*>(repair_smem);
    kb_compact_result* list = reinterpret_cast<kb_compact_result*>(repair_smem + (size_t)BLOCK * sizeof(RepairEval));
    int* decision = reinterpret_cast<int*>(list + K);
    const int sw = params.x_start_max - params.x_start_min;
    const uint32_t pix = pixels[blockIdx.x];
    const int y_i = (int)(pix / (uint32_t)sw), x_i = (int)(pix - (uint32_t)y_i * (uint32_t)sw);
    const int x = x_i + params.x_start_min, y = y_i + params.y_start_min;
    // (inside wave 0 the list is lane 0's: the other lanes fill and read it with a wavefront fence between)
    auto wave_fence = []() {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    float tail = -FLT_MAX;  // wave 0, uniform: the last slot's likelihood
    // one candidate (uniform arguments; wave 0) through kernels.cu:323-330
    auto insert = [&](float lh, float flux, int cand, int obs) {
        if (!(lh > tail)) return;  // (a candidate that is not above the last slot moves nothing: the swaps need a strict '>')
        float new_tail = tail;
        if (lane == 0) {
            kb_compact_result in = {lh, flux, cand, obs};
            for (int s = 0; s < K; ++s) {
                const kb_compact_result t = list[s];
                if (in.lh > t.lh) {
                    list[s] = in;
                    in = t;
                    if (in.cand < 0) break;  // (a displaced placeholder moves nothing further)
                }
            }
            new_tail = list[K - 1].lh;
        }
        tail = __shfl(new_tail, 0);
    };
    // the candidates [lo, hi) from the stack: every wave evaluates, wave 0 inserts (workgroup-uniform arguments)
    auto evaluate_range = [&](uint64_t lo, uint64_t hi) {
        for (uint64_t base = lo; base < hi; base += BLOCK) {
            const uint64_t ci = base + (uint64_t)threadIdx.x;
            RepairEval e = {-FLT_MAX, 0.0f, -1};  // (obs -1: does not exist / fails the observation count)
            if (ci < hi) {
                kb_trajectory cur = placeholder_result(x, y);
                cur.vx = cands[ci].vx;
                cur.vy = cands[ci].vy;
                evaluate_plain_batched<NB>(meta, psi_phi, times, &cur);
                if (!(cur.obs_count < params.min_observations)) e = RepairEval{cur.lh, cur.flux, cur.obs_count};  // kernels.cu:318-320
            }
            evals[threadIdx.x] = e;
            __syncthreads();
            if (wave == 0) {
                for (int w = 0; w < REPAIR_WAVES; ++w) {
                    const RepairEval mine_e = evals[w * WAVE + lane];
                    uint64_t todo = __ballot(mine_e.obs >= 0 && mine_e.lh > tail);
                    while (todo != 0ull) {  // uniform: lowest candidate first
                        const int src = (int)__builtin_ctzll(todo);
                        todo &= todo - 1ull;
                        insert(__shfl(mine_e.lh, src), __shfl(mine_e.flux, src), (int)(base + (uint64_t)(w * WAVE + src)),
                               __shfl(mine_e.obs, src));
                    }
                }
            }
            __syncthreads();
        }
    };
    if (wave == 0) {
        for (int s = lane; s < K; s += WAVE) list[s] = kb_compact_result{-FLT_MAX, 0.0f, -1, 0};  // kernels.cu:293-301
        wave_fence();
    }
    if (rl.lists == nullptr) {
        evaluate_range(0, n_cands);
    } else {
        // The fold of merge_fold_pixel (wave 0), with the suspect slices evaluated from the stack instead of trusted.
        const int n_lists = rl.n_lists;
        const uint64_t list_stride = rl.n_pixels * (uint64_t)K;
        const kb_compact_result* mine = rl.lists + (uint64_t)pix * (uint64_t)K;
        if (wave == 0) {
            // slice 0: its list IS the reference's state behind it
            for (int s = lane; s < K; s += WAVE) {
                kb_compact_result rec = mine[s];
                if (rec.cand < 0) rec = kb_compact_result{-FLT_MAX, 0.0f, -1, 0};
                list[s] = rec;
            }
            wave_fence();
            float t0 = -FLT_MAX;
            if (lane == 0) t0 = list[K - 1].lh;
            tail = __shfl(t0, 0);
        }
        for (int r = 1; r < n_lists; ++r) {  // workgroup-uniform; the lists cover ascending candidate ranges
            if (wave == 0) {
                const float tail_before = tail;
                wave_fence();
                // the state as it stands (K <= 32 records, one per lane), should the slice turn out suspect
                kb_compact_result saved = {-FLT_MAX, 0.0f, -1, 0};
                if (lane < K) saved = list[lane];
                kb_compact_result rec = {-FLT_MAX, 0.0f, -1, 0};
                if (lane < K) rec = mine[(uint64_t)r * list_stride + (uint64_t)lane];
                const float last_lh = __shfl(rec.lh, K - 1);
                const int last_cand = __shfl(rec.cand, K - 1);
                // the list's records stand for its slice: back into candidate order, then through the insertion
                const bool valid = rec.cand >= 0 && rec.lh > tail_before;
                int rank = 0;
                for (int j = 0; j < K; ++j) {
                    const int cj = __shfl(rec.cand, j);
                    const float lj = __shfl(rec.lh, j);
                    rank += (cj >= 0 && lj > tail_before && cj < rec.cand) ? 1 : 0;
                }
                const int n_valid = (int)__builtin_popcountll(__ballot(valid));
                for (int i = 0; i < n_valid; ++i) {
                    const int src = (int)__builtin_ctzll(__ballot(valid && rank == i));
                    insert(__shfl(rec.lh, src), __shfl(rec.flux, src), __shfl(rec.cand, src), __shfl(rec.obs_count, src));
                }
                const bool suspect = last_cand >= 0 && last_lh > tail_before && last_lh == tail;
                if (suspect) {
                    // (merge_fold_pixel) a dropped candidate may tie with the last slot: the state goes back, the slice comes
                    // from the stack instead
                    wave_fence();
                    if (lane < K) list[lane] = saved;
                    wave_fence();
                    tail = tail_before;
                }
                if (lane == 0) *decision = suspect ? 1 : 0;
            }
            __syncthreads();
            const bool again = *decision != 0;
            __syncthreads();
            if (again) evaluate_range((uint64_t)rl.begin[r], (uint64_t)rl.begin[r + 1]);
        }
    }
    __syncthreads();
    if (wave == 0) {
        wave_fence();
        for (int s = lane; s < K; s += WAVE) {
            kb_trajectory res = placeholder_result(x, y);
            const kb_compact_result rec = list[s];
            if (rec.cand >= 0) {
                res.vx = cands[rec.cand].vx;
                res.vy = cands[rec.cand].vy;
                res.lh = rec.lh;
                res.flux = rec.flux;
                res.obs_count = rec.obs_count;
            }
            out[(uint64_t)pix * (uint64_t)K + s] = res;
        }
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static thread_local char g_kernel_instance[96] = "";
void note_kernel_instance(const char* name) {
    std::snprintf(g_kernel_instance, sizeof(g_kernel_instance), "%s", name);
}

struct Workspace {
    void* ptr = nullptr;
    size_t bytes = 0;
    int device = -1;
};
// One set of workspaces (and one lock) per device: searches on different devices run side by side from
// different host threads (StackSearch's fan-out), searches on one device one after the other.
constexpr int MAX_DEVICES = 64;
static std::mutex g_ws_mutex[MAX_DEVICES];
static Workspace g_ws_all[MAX_DEVICES][8];  // 0: shift table + chunk info, 1: literal sigma-G scratch, 2: padded array copy (LDS kernel),
                            // 3: sigma-G work items + clipped values, 4: second per-pixel list buffer (sigma-G batches),
                            // 5: cold block of the kernel arguments, 6: per-pixel lists of kb_search_lds between chunks,
                            // 7: observation counts per shift for tiles at the image's edge (kb_edge_count_kernel)

// What the padded copy in workspace 2 of a device was made from (flag 256 of the search entry points: the caller
// vouches that the array has not changed since its last search; the copy is then reused when everything else that
// determines it is the same).
struct PaddedKey {
    const void* src = nullptr;
    void* copy = nullptr;
    uint64_t T = 0, H = 0, W = 0;
    int num_bytes = 0, fmt = 0, canon = 0;
    float scale[4] = {0, 0, 0, 0};
    int64_t Hp = 0, Wp = 0, px0 = 0, py0 = 0;
    uint64_t generation = 0;  // of a library-built array when the copy was made (0: an array the caller vouches for, flag 256)
    int n_invalid_host = -1;  // the copy's NO_DATA counter once some search has read it back (-1: not yet); not part of same()
    bool valid = false;
    bool same(const PaddedKey& o) const {
        return valid && o.valid && src == o.src && copy == o.copy && generation == o.generation && T == o.T && H == o.H && W == o.W &&
               num_bytes == o.num_bytes && fmt == o.fmt && canon == o.canon && scale[0] == o.scale[0] &&
               scale[1] == o.scale[1] && scale[2] == o.scale[2] && scale[3] == o.scale[3] && Hp == o.Hp && Wp == o.Wp &&
               px0 == o.px0 && py0 == o.py0;
    }
};
static PaddedKey g_padded_key[MAX_DEVICES];

// ---- arrays built (hence owned) by the library: kb_common.h ----
// Every owned array carries a generation, drawn from one counter and renewed whenever a library call builds or writes the
// array (or the caller says it wrote: kb_note_array_written).  A padded copy remembers the generation it was made from
// (PaddedKey::generation) and stands only while the array still has it -- nothing here reaches into another device's key, so
// the per-device lock of the searches is the only lock a key ever needs.
struct OwnedArray {
    const char* ptr;
    uint64_t bytes;
    uint64_t generation;
};
static std::mutex g_owned_mutex;
static std::vector<OwnedArray> g_owned;
static uint64_t g_next_generation = 1;  // (g_owned_mutex held; 0 = "not an owned array")
void note_array_built(const void* p, uint64_t bytes) {
    std::lock_guard<std::mutex> lock(g_owned_mutex);
    const char* c = static_cast<const char*>(p);
    // whatever was remembered about blocks this one overlaps is about memory that has been handed out again
    for (size_t i = 0; i < g_owned.size();) {
        if (g_owned[i].ptr < c + bytes && c < g_owned[i].ptr + g_owned[i].bytes) {
            g_owned.erase(g_owned.begin() + (long)i);
        } else {
            ++i;
        }
    }
    g_owned.push_back(OwnedArray{c, bytes, g_next_generation++});
}
void note_array_written(const void* p) {
    std::lock_guard<std::mutex> lock(g_owned_mutex);
    const char* c = static_cast<const char*>(p);
    for (OwnedArray& o : g_owned) {
        if (c >= o.ptr && c < o.ptr + o.bytes) o.generation = g_next_generation++;
    }
}
void note_array_gone(const void* p) {
    std::lock_guard<std::mutex> lock(g_owned_mutex);
    const char* c = static_cast<const char*>(p);
    for (size_t i = 0; i < g_owned.size(); ++i) {
        if (g_owned[i].ptr == c) {
            g_owned.erase(g_owned.begin() + (long)i);
            return;
        }
    }
}
uint64_t array_generation(const void* p) {
    std::lock_guard<std::mutex> lock(g_owned_mutex);
    for (const OwnedArray& o : g_owned) {
        if (o.ptr == static_cast<const char*>(p)) return o.generation;
    }
    return 0;
}
bool array_is_library_owned(const void* p) { return array_generation(p) != 0; }

static int current_device_slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    return (dev >= 0 && dev < MAX_DEVICES) ? dev : 0;
}

static int ensure_workspace(int which, size_t bytes, void** out) {
    const int dev = current_device_slot();
    Workspace& w = g_ws_all[dev][which];
    if (w.ptr != nullptr && w.bytes < bytes) {
        (void)hipFree(w.ptr);
        w.ptr = nullptr;
        w.bytes = 0;
        if (which == 2) g_padded_key[dev].valid = false;
    }
    if (w.ptr == nullptr) {
        KB_HIP_TRY(hipMalloc(&w.ptr, bytes));
        w.bytes = bytes;
        w.device = dev;
    }
    *out = w.ptr;
    return 0;
}

// ensure_workspace without the error: false when the allocation fails (the caller has a plan B).
static bool try_workspace(int which, size_t bytes, void** out) {
    if (ensure_workspace(which, bytes, out) == 0) return true;
    (void)hipGetLastError();
    *out = nullptr;
    return false;
}

// True when fmaf(code - 1, scale, min) equals the reference's double-rounded decode for every code.
// (the answer for the scale parameters seen last is kept: a StackSearch searches the same array many times, and 65 535 codes
// twice over were 0.2 ms of host time in front of every search of a uint16 array)
static bool verify_fast_decode(float scale, float min_val, int num_bytes) {
    struct Seen {
        float scale, min_val;
        int num_bytes, answer;
    };
    static std::mutex seen_mutex;
    static Seen seen[4] = {{0.0f, 0.0f, 0, -1}, {0.0f, 0.0f, 0, -1}, {0.0f, 0.0f, 0, -1}, {0.0f, 0.0f, 0, -1}};
    static unsigned next_slot = 0;
    {
        std::lock_guard<std::mutex> lock(seen_mutex);
        for (const Seen& e : seen) {
            if (e.answer >= 0 && e.num_bytes == num_bytes && std::memcmp(&e.scale, &scale, sizeof(float)) == 0 &&
                std::memcmp(&e.min_val, &min_val, sizeof(float)) == 0) {
                return e.answer != 0;
            }
        }
    }
    bool ok = true;
    const unsigned max_code = (1u << (8 * num_bytes)) - 1u;
    for (unsigned code = 1; code <= max_code && ok; ++code) {
        volatile double prod = ((double)(float)code - 1.0) * (double)scale;
        const float exact = (float)(prod + (double)min_val);
        const float fast = std::fmaf((float)code - 1.0f, scale, min_val);
        if (std::memcmp(&exact, &fast, sizeof(float)) != 0 || !std::isfinite(exact)) ok = false;
    }
    std::lock_guard<std::mutex> lock(seen_mutex);
    seen[next_slot++ % 4] = Seen{scale, min_val, num_bytes, ok ? 1 : 0};
    return ok;
}

// Format code of the array for the kernel templates: 4 = float, 2 / 1 = encoded with the reference's
// double-precision decode, 20 / 10 = encoded with the verified single-FMA decode.
static int format_code(int num_bytes, int fast_decode) {
    if (num_bytes == 1) return fast_decode ? 10 : 1;
    if (num_bytes == 2) return fast_decode ? 20 : 2;
    return 4;
}

// The tile grid of a launch: rows per tile differ between the kernels.
static SearchArgs with_tile_rows(SearchArgs a, int rows) {
    a.tiles_y = (a.sh + rows - 1) / rows;
    a.n_tiles = a.tiles_x * a.tiles_y;
    return a;
}

// which: 0 = kb_search_direct, 1 = kb_search_lds on an encoded padded copy, 2 = kb_search_lds on canonical floats
static void launch_search(const SearchArgs& a, int fmt, bool sigmag, int which, int lds_rows, int list_mode,
                          hipStream_t stream) {
    if (which == 2) {
        launch_search_lds_canon(with_tile_rows(a, lds_rows), lds_rows, sigmag, list_mode, stream);
    } else if (which == 1) {
        launch_search_lds_encoded(with_tile_rows(a, lds_rows), lds_rows, fmt, sigmag, stream);
    } else {
        // (list_mode 2 = whole records: in the store of kb_search_lds, in registers here)
        launch_search_direct(with_tile_rows(a, DIRECT_ROWS), fmt, sigmag,
                             list_mode == 2 || list_mode == 3 || (a.K <= 8 && a.n_cands < 65535 && a.T < 65535), stream);
    }
}

template <int NB>
static void launch_pad_fmt(const SearchArgs& a, const SearchCold& cold, bool canon, void* padded, int* n_invalid,
                           hipStream_t stream) {
    const dim3 grid((unsigned)std::min<int64_t>(((int64_t)a.Wp + 511) / 512, 64), (unsigned)cold.Hp, (unsigned)a.T);
    // (a frame beyond the 256 MiB Infinity Cache is written past the caches)
    const int stream_out = (uint64_t)a.T * (uint64_t)cold.Hp * (uint64_t)a.Wp * 8ull > (256ull << 20) ? 1 : 0;
    if (canon)
        hipLaunchKernelGGL((kb_pad_kernel<NB, true>), grid, dim3(256), 0, stream, a, cold.Hp, cold.px0, cold.py0, padded,
                           n_invalid, stream_out);
    else
        hipLaunchKernelGGL((kb_pad_kernel<NB, false>), grid, dim3(256), 0, stream, a, cold.Hp, cold.px0, cold.py0, padded,
                           n_invalid, 0);
}

static void launch_pad(const SearchArgs& a, const SearchCold& cold, int fmt, bool canon, void* padded, int* n_invalid,
                       hipStream_t stream) {
    switch (fmt) {
        case 1:
            launch_pad_fmt<1>(a, cold, canon, padded, n_invalid, stream);
            break;
        case 10:
            launch_pad_fmt<10>(a, cold, canon, padded, n_invalid, stream);
            break;
        case 2:
            launch_pad_fmt<2>(a, cold, canon, padded, n_invalid, stream);
            break;
        case 20:
            launch_pad_fmt<20>(a, cold, canon, padded, n_invalid, stream);
            break;
        default:
            launch_pad_fmt<4>(a, cold, true, padded, n_invalid, stream);
            break;
    }
}

}  // namespace kb

namespace kb {
// kb_device_search_filter / kb_device_search_compact: `sink` says where the per-pixel lists go.
static int search_filter_impl(const kb_psi_phi_meta* meta, const void* psi_phi_dev, const double* times_dev,
                              kb_search_params params, const kb_trajectory* cands_dev, uint64_t n_cands,
                              const ResultSink sink, uint64_t n_results, uint32_t flags, void* stream_v,
                              kb_search_stats* stats_out, int* counts_written = nullptr) {
    if (counts_written != nullptr) *counts_written = 0;
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
    if (sink.full == nullptr && sink.compact == nullptr) return fail("Invalid result list pointer.");
    if (kb_device_count() == 0) return fail("GPU is not available for search.");
    (void)hipGetLastError();  // a stale error of this thread (another library's probing) is not ours

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

    SearchArgs a{};
    SearchCold cold{};
    a.psi_phi = psi_phi_dev;
    cold.times = times_dev;
    cold.cands = cands_dev;
    cold.results = sink;
    cold.meta = *meta;
    cold.params = params;
    a.x_start_min = params.x_start_min;
    a.y_start_min = params.y_start_min;
    a.min_obs = params.min_observations;
    // Sigma-G searches test `lh < min_lh` inside the kernel (kernels.cu:318-320).  Without the filter the reference inserts
    // whatever beats a slot and drops lh < min_lh afterwards (stack_search.cpp:266-270); flag 1024 (the caller will apply
    // that post-filter, or the sparse exchange will) lets the lists ignore such candidates from the start: `min_lh` then is
    // the largest float below params.min_lh, a floor under every list's "likelihood to beat" -- a candidate enters only with
    // lh > floor, i.e. lh >= min_lh.  Slots at or above min_lh are exactly those of the default (the swap-down insertion
    // never lets a smaller value touch the part of a list at or above a larger one); -FLT_MAX = no floor.
    a.min_lh = params.do_sigmag_filter ? params.min_lh
                                       : ((flags & 1024u) != 0 && params.min_lh > -FLT_MAX ? std::nextafterf(params.min_lh, -INFINITY) : -FLT_MAX);
    a.psi_scale = meta->psi_scale;
    a.psi_min_val = meta->psi_min_val;
    a.phi_scale = meta->phi_scale;
    a.phi_min_val = meta->phi_min_val;
    a.T = (int)meta->num_times;
    a.W = (int)meta->width;
    a.H = (int)meta->height;
    a.n_cands = (int)n_cands;
    a.chunk = CHUNK;
    a.n_chunks = (int)((n_cands + CHUNK - 1) / CHUNK);
    a.sw = (int)sw;
    a.sh = (int)sh;
    a.tiles_x = (a.sw + WAVE - 1) / WAVE;
    // Tile height of kb_search_lds: 64 x 16 when the search area
    // gives every CU a tile of that size, 64 x 8 otherwise; flags bits 6 / 7 force one or the other (tests).
    // Encoded staging is built for 64 x 8 only; chunks of XWIDE_CHUNK candidates for 64 x 16 only.
    const bool list_fits = params.do_sigmag_filter != 0 || params.results_per_pixel <= 32;  // (kb_search_large_k beyond)
    const int64_t tall_tiles = (int64_t)a.tiles_x * ((sh + LDS_ROWS_TALL - 1) / LDS_ROWS_TALL);
    auto rows_for = [&](int chunk) {
        const bool keep_encoded = meta->num_bytes != 4 && (flags & 16u) != 0;
        // A short candidate list (up to four chunks of 16) is mostly list-filling: the first chunks of a search insert in
        // nearly every round, the sixteen waves of a 64 x 16 tile then wait at every group change for the wave with the most
        // rounds.  Two 64 x 8 workgroups per CU have separate barriers and overlap one tile's finish with the other's sums, which
        // is worth more there than the taller tile's smaller apron: 128 x 4096 x 4096 with 32 / 64 / 128 / 256 candidates 10.5 /
        // 20.2 / 40.4 / 78.8 ms against 12.7 / 22.1 / 40.4 / 74.8 ms (profiles/r04_tile_height.log).  The sigma-G emit keeps no list.
        // (Chunks of 32: the bytes a slab's apron costs are what that instance exists to save -- tall tiles.)
        if (chunk == XWIDE_CHUNK) return LDS_ROWS_TALL;  // (the one tile height that instance is built for)
        const bool short_list = n_cands <= 64 && params.do_sigmag_filter == 0;
        if (list_fits && !keep_encoded && ((tall_tiles >= 128 && (flags & 128u) == 0 && !short_list) || (flags & 64u) != 0)) {
            return LDS_ROWS_TALL;
        }
        return LDS_ROWS_WIDE_K;
    };
    int lds_rows = rows_for(CHUNK);
    a.tiles_y = (a.sh + lds_rows - 1) / lds_rows;
    a.n_tiles = a.tiles_x * a.tiles_y;
    a.K = (int)params.results_per_pixel;
    a.force_exact = (flags & 1u) ? 1 : 0;
    a.stable_lists = (flags & 512u) ? 1 : 0;  // per-pixel lists as stable top-K (kb_merge_compact_exact)
    cold.fast_decode = 0;
    if (meta->num_bytes != 4 && (flags & 8u) == 0) {  // bit 3: force the double-precision decode
        cold.fast_decode = (verify_fast_decode(meta->psi_scale, meta->psi_min_val, meta->num_bytes) &&
                         verify_fast_decode(meta->phi_scale, meta->phi_min_val, meta->num_bytes))
                                ? 1
                                : 0;
    }
    const int fmt = format_code(meta->num_bytes, cold.fast_decode);

    EventTimer table_timer(stream, stats_out != nullptr);
    EventTimer search_timer(stream, stats_out != nullptr);
    std::lock_guard<std::mutex> lock(g_ws_mutex[current_device_slot()]);
    Workspace* g_ws = g_ws_all[current_device_slot()];

    float table_ms = 0.0f, search_ms = 0.0f;
    // Kernel choice.  kb_search_lds (LDS-DMA staging from a padded copy) is the default for K <= 32;
    // flags bit 1 forces kb_search_direct, bit 2 insists on kb_search_lds even for few candidates,
    // bit 4 keeps an encoded array encoded in the padded copy.  Measured on MI355X (profiles/r01_*):
    // the direct kernel is bound by the vector-memory pipe (every sample is its own 512-byte wave
    // load), the staged kernel reads each slab once per workgroup and sums out of LDS.
    int which = 0;
    int padded_reused = 0;
    uint64_t padded_copy_bytes = 0;
    // (the offset tables of the hand-scheduled instances are indexed with 32-bit byte offsets: a candidate list x epochs beyond
    // that goes to kb_search_direct, which reads none of them, instead of failing)
    const bool fold_fits = ((uint64_t)n_cands + 2 * XWIDE_CHUNK) * (uint64_t)a.T * sizeof(int) <= 0x7fff0000ull;
    const bool want_lds = (flags & 2u) == 0 && (flags & 1u) == 0 && a.K <= 32 && fold_fits &&
                          (n_cands >= 8 || (flags & 4u) != 0);
    // Candidates per chunk.  WIDE_CHUNK for the two instances of kb_search_lds built for it -- float staging; lists of up to
    // 8 as packed records in registers, or none (the in-search sigma-G filter) -- when everything known before the tables
    // says one of them will run; if the tables then say otherwise (too many epochs that cannot be staged, no room for the padded copy),
    // they are rebuilt for CHUNK.  KBMOD_CHUNK = 8 keeps CHUNK (tests, comparisons).
    {
        const bool emitting = params.do_sigmag_filter != 0;  // (the in-search sigma-G filter: the search launch keeps no list)
        bool wide = want_lds && n_cands > (uint64_t)CHUNK && (meta->num_bytes == 4 || (flags & 16u) == 0) &&
                    (emitting || ((a.K <= 8 || (a.K <= 16 && a.stable_lists != 0)) && n_cands < 65535 && a.T < 65535));
        // (list modes of the wide instances: 3 = packed records in registers, K <= 8; 4 = pooled stable lists in the store, K <= 16)
        if (const char* env = std::getenv("KBMOD_LIST_MODE")) wide = wide && (emitting || std::atoi(env) == (a.K <= 8 ? 3 : 4));
        // Chunks of XWIDE_CHUNK: the packed-list instance again (K <= 8, reference or stable insertion), for arrays whose float
        // copy lies beyond the Infinity Cache -- there the kernel is bound by the bytes that cross the fabric, and those fall
        // with the candidates a staged slab serves (configs[3]'s share 88 -> 50 GB, configs[4] 5.2 -> 2.9 TB per launch) --,
        // on 64 x 16 tiles.  The instance is count-free: it takes stacks without NO_DATA pixels whose border tiles get their
        // counts from the edge tables; everything else that the tables must confirm is checked below.
        const uint64_t float_copy = (uint64_t)a.T * (uint64_t)a.H * (uint64_t)a.W * 8ull;
        // Measured (profiles/r05_chunk_width.log; round 6's finish with the generated selection tree moves none of it by more than
        // 1 %): the finish of a chunk of 32 costs 2.5 x that of a chunk of 16 (list registers in scratch memory, stored and
        // reloaded by every round), a fixed cost the sums of a deep stack amortise -- 512 epochs -13 %, 384 -10 %, 256
        // -5 ... -7 %, 128 epochs +7 ... +11 % against chunks of 16; hence from 192 epochs on.
        bool xwide = wide && !emitting && a.K <= 8 && n_cands >= (uint64_t)XWIDE_CHUNK && float_copy > (256ull << 20) && a.T >= 192 &&
                     tall_tiles >= 128 && (flags & (16u | 32u | 128u)) == 0 && params.x_start_min >= 0 && params.y_start_min >= 0 &&
                     params.x_start_max <= a.W && params.y_start_max <= a.H;
        if (const char* env = std::getenv("KBMOD_CHUNK")) {
            const int want = std::atoi(env);
            // (32 asks for the instance wherever it CAN run, cache-resident arrays and small search areas included: tests)
            xwide = want == XWIDE_CHUNK && wide && !emitting && a.K <= 8 && n_cands >= (uint64_t)XWIDE_CHUNK &&
                    (flags & (16u | 32u | 128u)) == 0 && params.x_start_min >= 0 && params.y_start_min >= 0 &&
                    params.x_start_max <= a.W && params.y_start_max <= a.H;
            wide = wide && (want == WIDE_CHUNK || xwide);
        }
        if (const char* env = std::getenv("KBMOD_EDGE_COUNTS")) xwide = xwide && std::atoi(env) != 0;
        {
            // an array a previous search has found NO_DATA pixels in is not tried again (the attempt costs a pad pass)
            const PaddedKey& prev = g_padded_key[current_device_slot()];
            if (prev.valid && prev.src == psi_phi_dev && prev.n_invalid_host > 0 && (flags & 2048u) == 0 &&
                ((flags & 256u) != 0 || (prev.generation != 0 && prev.generation == array_generation(psi_phi_dev)))) {
                xwide = false;
            }
        }
        if (wide) a.chunk = xwide ? XWIDE_CHUNK : WIDE_CHUNK;
    }
    bool wide_has_special = false;
    bool wide_store_failed = false;  // lists of 9 to 16 with wide chunks live in the list store: without it, chunks of CHUNK
    void* wide_lists = nullptr;
    int special_epochs = 0;
    int shift_box[4] = {0, 0, 0, 0};  // staged shift box of the tables that settled: dx_min, dx_max, dy_min, dy_max
    bool xwide_refused = false;  // chunks of XWIDE_CHUNK were asked for and the tables (or the array) said no
    int learned_n_invalid = -1;  // the array's NO_DATA count, once this search has read it back (a property of the array)
    for (bool settled = n_cands == 0; !settled;) {
        a.n_chunks = (int)((n_cands + a.chunk - 1) / a.chunk);
        which = 0;
        lds_rows = rows_for(a.chunk);
        a.tiles_y = (a.sh + lds_rows - 1) / lds_rows;
        a.n_tiles = a.tiles_x * a.tiles_y;
        const size_t table_bytes = (size_t)a.n_chunks * a.T * a.chunk * sizeof(int2);
        const size_t off_bytes = ((size_t)a.n_chunks * a.T * a.chunk + 4 * a.chunk) * sizeof(int);  // + prefetch slack
        const size_t box_bytes = (size_t)a.n_chunks * a.T * sizeof(EpochBox);
        const size_t org_bytes = ((size_t)a.n_chunks * a.T + SLAB_REF_SLACK) * sizeof(SlabRef);
        const size_t fold_bytes = want_lds ? ((size_t)a.n_chunks * a.T + SLAB_REF_SLACK) * a.chunk * sizeof(int) : 0;  // (read by kb_search_lds only)
        const size_t chunk_bytes = (size_t)a.n_chunks * sizeof(ChunkInfo);
        // NO_DATA pixel counter [1], unstaged (chunk, epoch) counter [1], staged shift box + tallest slab [5],
        // per-lane (chunk, epoch) counter [1], largest slab in pixels [1], spare [3]
        const size_t inv_bytes = 12 * sizeof(int);
        void* ws = nullptr;
        if (fold_bytes > 0x7fffff00ull) return fail("deviceSearchFilter: candidate list x epochs too long for the offset tables");
        if (ensure_workspace(0, table_bytes + off_bytes + box_bytes + chunk_bytes + inv_bytes + org_bytes + fold_bytes + 64, &ws)) return 1;
        char* wsc = reinterpret_cast<char*>(ws);
        a.table = reinterpret_cast<const int2*>(wsc);
        a.lds_off = reinterpret_cast<const int*>(wsc + table_bytes);
        cold.boxes = reinterpret_cast<const EpochBox*>(wsc + table_bytes + off_bytes);
        a.chunks = reinterpret_cast<const ChunkInfo*>(wsc + table_bytes + off_bytes + box_bytes);
        int* inv = reinterpret_cast<int*>(wsc + table_bytes + off_bytes + box_bytes + chunk_bytes);
        SlabRef* slab_refs = reinterpret_cast<SlabRef*>(wsc + table_bytes + off_bytes + box_bytes + chunk_bytes + inv_bytes);
        a.slabs = slab_refs;
        int* lds_fold = reinterpret_cast<int*>(wsc + (table_bytes + off_bytes + box_bytes + chunk_bytes + inv_bytes + org_bytes + 63) / 64 * 64);
        a.lds_fold = lds_fold;
        int* n_invalid = inv;
        int* n_not_lds = inv + 1;
        int* gbox = inv + 2;
        a.n_invalid = n_invalid;
        a.global_box = gbox;
        // pitch quantum of the slabs: 16 bytes of raw pairs -- 2 pixels of float pairs (the canonical copy), 8 when
        // the caller keeps an encoded array encoded in the padded copy (flag 16)
        const bool tables_for_encoded = meta->num_bytes != 4 && (flags & 16u) != 0;
        const int col_quantum = tables_for_encoded ? 8 : 2;
        table_timer.begin();
        static const int inv_init[12] = {0, 0, INT32_MAX, INT32_MIN, INT32_MAX, INT32_MIN, 0, 0, 0, 0, 0, 0};
        KB_HIP_TRY(hipMemcpyAsync(inv, inv_init, sizeof(inv_init), hipMemcpyHostToDevice, stream));
        const int max_cols = lds_cols(a.chunk);
        if (a.chunk == XWIDE_CHUNK) {
            hipLaunchKernelGGL((kb_shift_table_kernel<XWIDE_CHUNK>), dim3(a.n_chunks), dim3(256), 0, stream, cands_dev,
                               times_dev, a.n_cands, a.T, reinterpret_cast<int2*>(wsc),
                               reinterpret_cast<ChunkInfo*>(wsc + table_bytes + off_bytes + box_bytes),
                               reinterpret_cast<EpochBox*>(wsc + table_bytes + off_bytes),
                               reinterpret_cast<int*>(wsc + table_bytes), n_not_lds, gbox, lds_rows, col_quantum, max_cols);
        } else if (a.chunk == WIDE_CHUNK) {
            hipLaunchKernelGGL((kb_shift_table_kernel<WIDE_CHUNK>), dim3(a.n_chunks), dim3(256), 0, stream, cands_dev,
                               times_dev, a.n_cands, a.T, reinterpret_cast<int2*>(wsc),
                               reinterpret_cast<ChunkInfo*>(wsc + table_bytes + off_bytes + box_bytes),
                               reinterpret_cast<EpochBox*>(wsc + table_bytes + off_bytes),
                               reinterpret_cast<int*>(wsc + table_bytes), n_not_lds, gbox, lds_rows, col_quantum, max_cols);
        } else {
            hipLaunchKernelGGL((kb_shift_table_kernel<CHUNK>), dim3(a.n_chunks), dim3(256), 0, stream, cands_dev,
                               times_dev, a.n_cands, a.T, reinterpret_cast<int2*>(wsc),
                               reinterpret_cast<ChunkInfo*>(wsc + table_bytes + off_bytes + box_bytes),
                               reinterpret_cast<EpochBox*>(wsc + table_bytes + off_bytes),
                               reinterpret_cast<int*>(wsc + table_bytes), n_not_lds, gbox, lds_rows, col_quantum, max_cols);
        }
        KB_HIP_TRY(hipGetLastError());
        int largest_slab_px = 0;
        if (want_lds) {
            // The choice and the apron of the padded copy need six ints back.
            // unstaged epochs, dx_min, dx_max, dy_min, dy_max, rows_max, per-lane epochs, largest slab, shifts that are not monotone
            int back[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            KB_HIP_TRY(hipMemcpyAsync(back, n_not_lds, sizeof(back), hipMemcpyDeviceToHost, stream));
            KB_HIP_TRY(hipStreamSynchronize(stream));
            largest_slab_px = back[7];
            // (the wide-chunk instance has no path for epochs that are not staged with uniform shifts)
            wide_has_special = back[0] != 0 || back[6] != 0;
            for (int k = 0; k < 4; ++k) shift_box[k] = back[1 + k];
            special_epochs = back[0] + back[6];
            // An unstaged epoch costs several times a staged one: above 10 % the direct kernel wins.
            const uint64_t n_epochs = (uint64_t)a.n_chunks * (uint64_t)a.T;
            if (std::getenv("KBMOD_DEBUG") != nullptr) {
                std::fprintf(stderr,
                             "[kbmod_hip] (chunk, epoch) pairs: %llu, unstaged %d, per-lane %d; staged shift box x [%d, %d] "
                             "y [%d, %d], tallest slab %d rows\n",
                             (unsigned long long)n_epochs, back[0], back[6], back[1], back[2], back[3], back[4], back[5]);
            }
            uint64_t unstaged_limit = 10;  // staged unless more than 1 / limit of the (chunk, epoch) pairs cannot be
            if (const char* env = std::getenv("KBMOD_UNSTAGED_LIMIT")) unstaged_limit = std::max<uint64_t>(1, std::strtoull(env, nullptr, 10));
            if (back[1] <= back[2] && (uint64_t)back[0] * unstaged_limit <= n_epochs) {
                // Every slab [origin, origin + rows_max) x [origin, origin + LDS_COLS) lies inside the padded frame.
                // (the zero shift is included: unstaged epochs copy the slab at the tile's own pixel)
                back[1] = std::min(back[1], 0);
                back[2] = std::max(back[2], 0);
                back[3] = std::min(back[3], 0);
                back[4] = std::max(back[4], 0);
                const int64_t x_lo = (int64_t)params.x_start_min + back[1];
                // (the frame is sized for the widest pitch of any chunk width, so that the copy of one search serves the next
                // whatever instance it launches)
                const int64_t x_hi = (int64_t)params.x_start_min + (int64_t)WAVE * (a.tiles_x - 1) + back[2] + LDS_COLS_XWIDE;
                const int64_t y_lo = (int64_t)params.y_start_min + back[3];
                // rows a slab's staging rounds can touch (whole rounds are loaded, see load_slab), for either
                // staged format
                auto rows_touched = [&](uint64_t pair_b) {
                    // (the widest slab for the rounds, the narrowest for the rows they span: a chunk's pitch
                    // lies between 64 + quantum and LDS_COLS)
                    const uint64_t row_b = (uint64_t)LDS_COLS_XWIDE * pair_b, row_b_min = (uint64_t)(WAVE + col_quantum) * pair_b;
                    const uint64_t round_b = (uint64_t)stage_round(lds_rows);
                    const uint64_t rounds = ((uint64_t)back[5] * row_b + round_b - 1) / round_b;
                    return (int64_t)((rounds * round_b + row_b_min - 1) / row_b_min);
                };
                const int64_t rows_cap = std::max(rows_touched(8), rows_touched(2ull * (uint64_t)meta->block_size));
                const int64_t y_hi =
                        (int64_t)params.y_start_min + (int64_t)lds_rows * (a.tiles_y - 1) + back[4] + rows_cap;
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
                const bool encoded_instance = (params.do_sigmag_filter != 0 || a.K <= 8) && lds_rows == LDS_ROWS_WIDE_K &&
                                              tables_for_encoded;  // search_lds_encoded.hip (slab pitches in its quantum)
                // (the device's free memory is asked for only when the workspace kept from earlier searches is too small:
                // the query costs tens of microseconds per search)
                size_t free_b = 0, total_b = 0;
                bool asked = false;
                const uint64_t have = g_ws[2].ptr != nullptr ? g_ws[2].bytes : 0;
                auto room_for = [&](uint64_t bytes) {
                    if (bytes <= have) return true;
                    if (!asked) {
                        asked = true;
                        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
                    }
                    return bytes + (2ull << 30) <= (uint64_t)free_b + have;
                };
                if (canon && meta->num_bytes != 4 && !room_for(frame * 8ull + 64)) canon = false;
                const uint64_t pair_bytes = canon ? 8ull : 2ull * (uint64_t)meta->block_size;
                const uint64_t padded_bytes = frame * pair_bytes + 64;
                // Per-lane offsets are 32-bit; an apron that outweighs the image 3:1 is not worth staging; and
                // without room for the copy in HBM (or if its allocation fails) the search reads the array itself
                // (kb_search_direct), which needs no workspace beyond the tables.
                void* padded = nullptr;
                if ((canon || encoded_instance) && (uint64_t)rows_cap * (uint64_t)Wp * pair_bytes <= 0x7fffffffull &&
                    frame <= 4ull * image + (8ull << 20) && room_for(padded_bytes) && try_workspace(2, padded_bytes, &padded)) {
                    a.padded = padded;
                    padded_copy_bytes = padded_bytes;
                    a.Wp = (int)Wp;
                    cold.Hp = (int)Hp;
                    cold.px0 = (int)px0;
                    cold.py0 = (int)py0;
                    // bit 5 (debug): never take the count-free specialisation
                    a.all_staged = (back[0] == 0 && (flags & 32u) == 0) ? 1 : 0;
                    // the NO_DATA counter of the copy lives behind it (it has to outlive this search's tables)
                    n_invalid = reinterpret_cast<int*>(static_cast<char*>(padded) + frame * pair_bytes);
                    a.n_invalid = n_invalid;
                    PaddedKey key;
                    key.src = psi_phi_dev;
                    key.copy = padded;
                    key.T = (uint64_t)a.T;
                    key.H = (uint64_t)a.H;
                    key.W = (uint64_t)a.W;
                    key.num_bytes = meta->num_bytes;
                    key.fmt = fmt;
                    key.canon = canon ? 1 : 0;
                    key.scale[0] = meta->psi_scale;
                    key.scale[1] = meta->psi_min_val;
                    key.scale[2] = meta->phi_scale;
                    key.scale[3] = meta->phi_min_val;
                    key.Hp = Hp;
                    key.Wp = Wp;
                    key.px0 = px0;
                    key.py0 = py0;
                    key.valid = true;
                    PaddedKey& have_key = g_padded_key[current_device_slot()];
                    // The copy (and its NO_DATA counter) of the previous search stands when nothing that determines it has
                    // changed and the array is known to be the same bits: the caller vouches for it (flag 256), or the library
                    // built the array itself and no library call has written into it since (kb_common.h).  Flag 2048: never.
                    key.generation = array_generation(psi_phi_dev);  // (read once: a write noted from here on renews it)
                    const bool unchanged = (flags & 256u) != 0 || key.generation != 0;
                    if (unchanged && (flags & 2048u) == 0 && have_key.same(key)) {
                        padded_reused = 1;
                    } else {
                        have_key.valid = false;
                        KB_HIP_TRY(hipMemsetAsync(n_invalid, 0, sizeof(int), stream));
                        launch_pad(a, cold, fmt, canon, padded, n_invalid, stream);
                        KB_HIP_TRY(hipGetLastError());
                        // (the NO_DATA count read back earlier in this search belongs to the frame it was counted in -- the pad
                        // pass counts inside the padded frame only --: it carries over to a re-made copy of the same frame, no further)
                        const bool same_frame = have_key.Hp == key.Hp && have_key.Wp == key.Wp && have_key.px0 == key.px0 &&
                                                have_key.py0 == key.py0 && have_key.src == key.src;
                        have_key = key;
                        have_key.n_invalid_host = same_frame ? learned_n_invalid : -1;
                    }
                    if (a.chunk == XWIDE_CHUNK) {
                        // The instance for chunks of 32 is count-free: what the tables must confirm, and the array.
                        const int D = std::max(std::max(-back[1], back[2]), std::max(std::max(-back[3], back[4]), 0));
                        void* edge_ws = nullptr;
                        if (back[0] != 0 || back[6] != 0 || back[8] != 0 || a.all_staged == 0 || D > 200 ||
                            (uint64_t)largest_slab_px * 8ull > 2ull * (uint64_t)stage_round(LDS_ROWS_TALL) ||
                            !try_workspace(7, (size_t)a.n_chunks * 4 * (size_t)(D + 1) * XWIDE_CHUNK * sizeof(unsigned short) + 64, &edge_ws)) {
                            xwide_refused = true;
                        } else {
                            if (have_key.n_invalid_host < 0) {  // (a sync behind the pad pass; once per copy)
                                int n_inv = 0;
                                KB_HIP_TRY(hipMemcpyAsync(&n_inv, n_invalid, sizeof(int), hipMemcpyDeviceToHost, stream));
                                KB_HIP_TRY(hipStreamSynchronize(stream));
                                have_key.n_invalid_host = n_inv;
                            }
                            learned_n_invalid = have_key.n_invalid_host;
                            if (learned_n_invalid != 0) xwide_refused = true;
                        }
                    }
                    const int64_t n_org = (int64_t)a.n_chunks * a.T;
                    hipLaunchKernelGGL(kb_slab_ref_kernel, dim3((unsigned)((n_org + SLAB_REF_SLACK + 255) / 256)), dim3(256), 0, stream,
                                       cold.boxes, a.chunks, n_org, a.T, cold.Hp, a.Wp, cold.px0, cold.py0, (int)pair_bytes,
                                       slab_refs, a.lds_off, lds_rows, lds_fold, a.chunk);
                    KB_HIP_TRY(hipGetLastError());
                    which = canon ? 2 : 1;
                }
            }
        }
        table_timer.mark_end();  // (read behind the search: the host goes on to launch it)
        if (a.chunk == WIDE_CHUNK && which == 2 && !wide_has_special && params.do_sigmag_filter == 0 && a.K > 8) {
            const SearchArgs at = with_tile_rows(a, lds_rows);
            wide_store_failed = !try_workspace(6, (size_t)at.n_tiles * 16 * block_threads(lds_rows) * 16, &wide_lists);
        }
        if (a.chunk != CHUNK && (which != 2 || wide_has_special || wide_store_failed)) {
            a.chunk = CHUNK;  // the instance the wide chunks are for will not (or cannot) run: tables for the others
        } else if (a.chunk == XWIDE_CHUNK && xwide_refused) {
            a.chunk = WIDE_CHUNK;  // NO_DATA pixels, shifts beyond the edge tables, slabs of more than two rounds: chunks of 16
        } else {
            settled = true;
        }
    }

    const bool sigmag = params.do_sigmag_filter != 0;
    a.chunk_lo = 0;
    a.chunk_hi = a.n_chunks;
    cold.sg = SigmaGWork{};

    // Sigma-G: the search launch emits work items, two more launches resolve them (sigmag_kernels.hip).
    // Work items are bounded by rows x candidates; the candidate list is cut into batches of whole
    // chunks so that one batch's worst case fits the work-item store (KBMOD_SIGMAG_CAP items, default 4 Mi).
    const int n_rows = a.tiles_x * a.sh;  // rows of 64 start pixels
    int batch_chunks = std::max(a.n_chunks, 1), n_batches = 1, resolve_waves = 0;
    void* pingpong = nullptr;  // second list buffer, in the sink's record format
    if (sigmag) {
        int cus = 256;
        int dev = 0;
        KB_HIP_TRY(hipGetDevice(&dev));
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        // literal-clip scratch: one slot per wave of the resolve launch (8 workgroups per CU), or per wave of
        // the bounded kb_search_large_k grid
        resolve_waves = std::max(cus, 1) * 32;
        void* sg = nullptr;
        if (ensure_workspace(1, (size_t)resolve_waves * scratch_words_per_wave(a.T) * sizeof(float), &sg)) return 1;
        cold.sg_scratch = reinterpret_cast<float*>(sg);
    }
    if (sigmag && a.K <= 32) {
        uint64_t cap_limit = 4ull << 20;
        if (const char* env = std::getenv("KBMOD_SIGMAG_CAP")) cap_limit = std::max<uint64_t>(1, std::strtoull(env, nullptr, 10));
        batch_chunks = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::max(a.n_chunks, 1),
                                                                     cap_limit / ((uint64_t)n_rows * a.chunk)));
        n_batches = std::max(1, (a.n_chunks + batch_chunks - 1) / batch_chunks);
        const uint64_t capacity = (uint64_t)n_rows * (uint64_t)batch_chunks * a.chunk;
        const size_t slot_bytes = (capacity * sizeof(uint32_t) + 255) / 256 * 256;
        const size_t entry_bytes = capacity * sizeof(SgEntry);
        const size_t out_bytes = capacity * WAVE * sizeof(float);
        void* w = nullptr;
        if (ensure_workspace(3, slot_bytes + 256 + entry_bytes + 3 * out_bytes, &w)) return 1;
        char* wc = reinterpret_cast<char*>(w);
        cold.sg.slots = reinterpret_cast<uint32_t*>(wc);
        cold.sg.n_entries = reinterpret_cast<int*>(wc + slot_bytes);
        cold.sg.totals = reinterpret_cast<unsigned long long*>(wc + slot_bytes + 64);
        KB_HIP_TRY(hipMemsetAsync(cold.sg.totals, 0, 3 * sizeof(unsigned long long), stream));
        cold.sg.entries = reinterpret_cast<SgEntry*>(wc + slot_bytes + 256);
        cold.sg.lh = reinterpret_cast<float*>(wc + slot_bytes + 256 + entry_bytes);
        cold.sg.flux = reinterpret_cast<float*>(wc + slot_bytes + 256 + entry_bytes + out_bytes);
        cold.sg.obs = reinterpret_cast<int*>(wc + slot_bytes + 256 + entry_bytes + 2 * out_bytes);
        cold.sg.batch_cands = batch_chunks * a.chunk;
        if (n_batches > 1) {
            void* pp = nullptr;
            if (ensure_workspace(4, (size_t)expected * sizeof(kb_trajectory), &pp)) return 1;
            pingpong = pp;
        }
    }

    // Observation counts for tiles at the image's edge out of tables instead of one vector instruction per sample
    // (edge_counts, search_lds.h): for the wide-chunk instances, when every epoch is staged with uniform
    // shifts, the start pixels lie on the image and the shifts are small enough for the tables.  Whether a tile uses
    // them is decided on the device: the stack must hold no NO_DATA pixel (the pad pass counts them) and the shifts
    // must be monotone (the table kernel checks).  KBMOD_EDGE_COUNTS = 0 keeps the counting loops (tests, comparisons).
    cold.edge_tab = nullptr;
    cold.edge_ok = nullptr;
    cold.edge_D = 0;
    int edge_tables = 0;
    if (which == 2 && a.chunk >= WIDE_CHUNK && a.all_staged && shift_box[0] <= shift_box[1] &&
        params.x_start_min >= 0 && params.y_start_min >= 0 && params.x_start_max <= a.W && params.y_start_max <= a.H) {
        const int D = std::max(std::max(-shift_box[0], shift_box[1]), std::max(std::max(-shift_box[2], shift_box[3]), 0));
        const char* env = std::getenv("KBMOD_EDGE_COUNTS");
        const size_t tab_bytes = (size_t)a.n_chunks * 4 * (size_t)(D + 1) * (size_t)a.chunk * sizeof(unsigned short);
        void* et = nullptr;
        if (!(env != nullptr && std::atoi(env) == 0) && D <= 200 && tab_bytes <= (256ull << 20) &&
            try_workspace(7, tab_bytes + 64, &et)) {
            int* ok = reinterpret_cast<int*>(static_cast<char*>(et) + tab_bytes);
            static const int one = 1;
            KB_HIP_TRY(hipMemcpyAsync(ok, &one, sizeof(int), hipMemcpyHostToDevice, stream));
            const size_t hist_bytes = 4 * (size_t)a.chunk * (size_t)(D + 1) * sizeof(unsigned int);
            if (a.chunk == XWIDE_CHUNK) {
                // (up to 103 KB of histograms: beyond the 64 KiB a launch gets without asking)
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kb_edge_count_kernel<XWIDE_CHUNK>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)hist_bytes);
                hipLaunchKernelGGL((kb_edge_count_kernel<XWIDE_CHUNK>), dim3(a.n_chunks), dim3(256), hist_bytes, stream, a.table,
                                   a.n_cands, a.T, D, reinterpret_cast<unsigned short*>(et), ok);
            } else {
                hipLaunchKernelGGL((kb_edge_count_kernel<WIDE_CHUNK>), dim3(a.n_chunks), dim3(256), hist_bytes, stream, a.table,
                                   a.n_cands, a.T, D, reinterpret_cast<unsigned short*>(et), ok);
            }
            KB_HIP_TRY(hipGetLastError());
            cold.edge_tab = reinterpret_cast<const uint4*>(et);
            cold.edge_ok = ok;
            cold.edge_D = D;
            edge_tables = 1;
        }
    }
    if (which == 2 && a.chunk == XWIDE_CHUNK && !edge_tables) {
        return fail("deviceSearchFilter: internal error -- the instance for chunks of 32 candidates was chosen without its edge tables");
    }

    // the cold block of the kernel arguments (search_common.h) lives in device memory
    void* cold_dev = nullptr;
    if (ensure_workspace(5, sizeof(SearchCold), &cold_dev)) return 1;
    a.cold = reinterpret_cast<const SearchCold*>(cold_dev);  // (uploaded below, once the list form -- and with it the sink -- is settled)

    // Where kb_search_lds keeps its per-pixel lists (ListMode, search_lds.h): lists of up to 8 as packed result
    // records in registers (3); longer ones, or candidate indices beyond 16 bits, as whole records in the HBM store
    // when the candidate list is short against the stack depth (2: the store is visited once per chunk of
    // candidates, the alternative costs an exact re-evaluation of every winner, K x T samples per pixel), else as
    // (likelihood, candidate) pairs in that store (1, lists of more than 8) or in registers (0).
    // KBMOD_LIST_MODE = 0 / 1 / 2 / 3 overrides where the pair (K, mode) exists (tests).
    a.lists = nullptr;
    int list_mode = 0;
    if (!sigmag && a.K <= 32) {
        const int ks = a.K <= 8 ? 8 : (a.K <= 16 ? 16 : 32);
        const bool short_list = a.n_chunks <= a.T;
        // (3: packed records in registers, K <= 8 on the float-staged kernel; candidate index and count share a word)
        const bool packable = ks == 8 && which == 2 && a.n_cands < 65535 && a.T < 65535;
        list_mode = ks == 32 ? 1 : (packable ? 3 : (short_list ? 2 : (ks == 16 ? 1 : 0)));
        if (const char* env = std::getenv("KBMOD_LIST_MODE")) {
            const int want = std::atoi(env);
            if ((ks == 8 && (want == 0 || want == 2 || (want == 3 && packable))) || (ks == 16 && (want == 1 || want == 2))) {
                list_mode = want;
            }
        }
        if (which == 2 && ks == 16 && a.chunk == WIDE_CHUNK) {  // the wide-chunk instance for lists of 9 to 16
            list_mode = 4;
            a.lists = reinterpret_cast<uint2*>(wide_lists);
        }
        if (which != 2 && list_mode == 3) list_mode = 0;
        if (which == 1 && list_mode != 0) list_mode = 0;  // encoded staging: registers (K <= 8)
        if (which == 2 && (list_mode == 1 || list_mode == 2)) {
            const SearchArgs at = with_tile_rows(a, lds_rows);
            const size_t slot_bytes = list_mode == 2 ? 16 : 8;  // TileLists::SLOT_BYTES
            void* lists = nullptr;
            if (try_workspace(6, (size_t)at.n_tiles * ks * block_threads(lds_rows) * slot_bytes, &lists)) {
                a.lists = reinterpret_cast<uint2*>(lists);
            } else if (ks == 8) {
                list_mode = packable ? 3 : 0;  // no room for the store: lists of 8 fit the registers
            } else {
                which = 0;  // ... longer ones do in kb_search_direct
            }
        }
    }

    // ResultSink::counts is honoured by the epilogues of kb_search_lds with packed or pooled lists only: any other instance
    // writes every slot and leaves the counts to the caller (kb_sparsify_compact then counts them from the records).
    // With the in-search sigma-G filter the lists are written by kb_sigmag_select_kernel, which honours it for its last batch.
    const bool counts_ok = sink.counts != nullptr && a.K <= 32 &&
                           (sigmag || (which == 2 && (list_mode == 3 || list_mode == 4)));
    if (!counts_ok || sigmag) cold.results.counts = nullptr;
    if (counts_written != nullptr) *counts_written = counts_ok ? 1 : 0;
    KB_HIP_TRY(hipMemcpyAsync(cold_dev, &cold, sizeof(SearchCold), hipMemcpyHostToDevice, stream));

    g_kernel_instance[0] = 0;
    search_timer.begin();
    int variant;
    if (a.K > 32) {
        if (sink.compact != nullptr) return fail("compact results support results_per_pixel <= 32");
        if (sigmag) {
            const SearchArgs ad = with_tile_rows(a, DIRECT_ROWS);
            launch_search_large_k(ad, true, std::max(1, std::min(ad.n_tiles, resolve_waves / DIRECT_ROWS)), stream);
        } else {
            const SearchArgs ad = with_tile_rows(a, DIRECT_ROWS);
            launch_search_large_k(ad, false, ad.n_tiles, stream);
        }
        variant = 99;
    } else if (sigmag) {
        // batch b leaves its lists in buffer (n_batches - 1 - b) % 2: the last one in results_dev
        ResultSink other = sink;
        if (sink.compact != nullptr) {
            other.compact = reinterpret_cast<kb_compact_result*>(pingpong);
        } else {
            other.full = reinterpret_cast<kb_trajectory*>(pingpong);
        }
        const ResultSink bufs[2] = {sink, other};
        const ResultSink* prev = nullptr;
        for (int b = 0; b < n_batches; ++b) {
            a.chunk_lo = b * batch_chunks;
            a.chunk_hi = std::min(a.n_chunks, a.chunk_lo + batch_chunks);
            KB_HIP_TRY(hipMemsetAsync(cold.sg.slots, 0, (size_t)n_rows * cold.sg.batch_cands * sizeof(uint32_t), stream));
            KB_HIP_TRY(hipMemsetAsync(cold.sg.n_entries, 0, sizeof(int), stream));
            if (a.chunk_lo < a.chunk_hi) launch_search(a, fmt, true, which, lds_rows, 0, stream);  // the emitting instances keep no list
            KB_HIP_TRY(hipGetLastError());
            const ResultSink* next = &bufs[(n_batches - 1 - b) % 2];
            // (the lists of a batch that another one follows are read back slot by slot: only the last may leave runs unwritten)
            ResultSink next_sink = *next;
            next_sink.counts = (counts_ok && b == n_batches - 1) ? sink.counts : nullptr;
            if (launch_sigmag_resolve(a, cold, prev, next_sink, resolve_waves, stream)) return 1;
            prev = next;
        }
        variant = a.K <= 8 ? 8 : (a.K <= 16 ? 16 : 32);
    } else {
        launch_search(a, fmt, false, which, lds_rows, list_mode, stream);
        variant = a.K <= 8 ? 8 : (a.K <= 16 ? 16 : 32);
    }
    KB_HIP_TRY(hipGetLastError());
    search_ms = search_timer.end();
    table_ms = n_cands != 0 ? table_timer.elapsed() : 0.0f;
    if (which == 2 && a.chunk == XWIDE_CHUNK && !sigmag && a.K <= 32) {
        // The count-free instance for chunks of 32 leaves a tile it cannot serve alone and says so (search_lds.h); the host's
        // checks above make that unreachable -- if it happens all the same, the caller gets an error, not a partial result.
        // (One 4-byte read-back and a sync behind a launch of tens of milliseconds.)
        int refused = 0;
        KB_HIP_TRY(hipMemcpyAsync(&refused, a.global_box + XWIDE_REFUSAL_WORD, sizeof(int), hipMemcpyDeviceToHost, stream));
        KB_HIP_TRY(hipStreamSynchronize(stream));
        if (refused != 0) {
            return fail("deviceSearchFilter: internal error -- the count-free instance for chunks of 32 candidates met a tile that "
                        "has to count samples (set KBMOD_CHUNK=16 to search with chunks of 16)");
        }
    }
    if (which != 0 && std::getenv("KBMOD_DEBUG") != nullptr) {
        int bad = -1;
        KB_HIP_TRY(hipMemcpyAsync(&bad, a.n_invalid, sizeof(int), hipMemcpyDeviceToHost, stream));
        KB_HIP_TRY(hipStreamSynchronize(stream));
        std::fprintf(stderr, "[kbmod_hip] padded frame %d x %d (image at %d, %d), NO_DATA pixels %d, all_staged %d%s\n", a.Wp,
                     cold.Hp, cold.px0, cold.py0, bad, a.all_staged, padded_reused ? ", copy reused" : "");
    }

    if (stats_out != nullptr) {
        const uint64_t S = (uint64_t)sw * (uint64_t)sh;
        stats_out->search_kernel_ms = search_ms;
        stats_out->table_kernel_ms = table_ms;
        stats_out->num_evals = S * n_cands * meta->num_times;
        stats_out->algorithmic_bytes = stats_out->num_evals * 2ull * (uint64_t)meta->block_size +
                                       S * (uint64_t)a.K * 28ull + n_cands * 28ull + meta->num_times * 8ull;
        stats_out->kernel_variant = which * 10000 + variant * 100 + meta->num_bytes * 10 + (sigmag ? 1 : 0);
        stats_out->num_search_launches = sigmag && a.K <= 32 ? 3 * n_batches : 1;
        stats_out->lds_read_bytes = which == 2 ? stats_out->num_evals * 8ull
                                               : (which == 1 ? stats_out->num_evals * 2ull * (uint64_t)meta->block_size : 0ull);
        stats_out->sigmag_work_items = 0;
        stats_out->sigmag_trajectories = 0;
        stats_out->sigmag_literal = 0;
        std::snprintf(stats_out->kernel_name, sizeof(stats_out->kernel_name), "%s", g_kernel_instance);
        stats_out->padded_copy_reused = padded_reused;
        stats_out->special_epochs = which != 0 ? special_epochs : 0;
        stats_out->edge_count_tables = edge_tables;
        // the environment switches (tests, comparisons) that were set while this search chose its kernels: never silent
        static const char* const kSwitches[] = {"KBMOD_CHUNK", "KBMOD_LIST_MODE", "KBMOD_EDGE_COUNTS", "KBMOD_UNSTAGED_LIMIT",
                                                "KBMOD_SIGMAG_CAP", "KBMOD_DEBUG"};
        stats_out->env_overrides = 0;
        for (int i = 0; i < 6; ++i) stats_out->env_overrides |= std::getenv(kSwitches[i]) != nullptr ? (1 << i) : 0;
        if (cold.sg.totals != nullptr) {
            unsigned long long totals[3] = {0, 0, 0};
            KB_HIP_TRY(hipMemcpyAsync(totals, cold.sg.totals, sizeof(totals), hipMemcpyDeviceToHost, stream));
            KB_HIP_TRY(hipStreamSynchronize(stream));
            stats_out->sigmag_work_items = totals[0];
            stats_out->sigmag_trajectories = totals[1];
            stats_out->sigmag_literal = totals[2];
        }
    } else {
        // kernels.cu:396 -- the reference call is synchronous.
        KB_HIP_TRY(hipStreamSynchronize(stream));
    }
    return 0;
}
void release_result_arenas();  // result_kernels.hip: the scratch arena of kb_filter_sort_results
void release_exchange_arenas();  // exchange_kernels.hip: block totals and offsets of the sparse exchange
}  // namespace kb

extern "C" {

int kb_device_search_filter(const kb_psi_phi_meta* meta, const void* psi_phi_dev, const double* times_dev,
                            kb_search_params params, const kb_trajectory* cands_dev, uint64_t n_cands,
                            kb_trajectory* results_dev, uint64_t n_results, uint32_t flags, void* stream,
                            kb_search_stats* stats_out) {
    const kb::ResultSink sink = {results_dev, nullptr, 0};
    return kb::search_filter_impl(meta, psi_phi_dev, times_dev, params, cands_dev, n_cands, sink, n_results, flags, stream,
                                  stats_out);
}

int kb_device_search_compact(const kb_psi_phi_meta* meta, const void* psi_phi_dev, const double* times_dev,
                             kb_search_params params, const kb_trajectory* cands_dev, uint64_t n_cands,
                             int32_t cand_index_base, kb_compact_result* results_dev, uint64_t n_results, uint32_t flags,
                             void* stream, kb_search_stats* stats_out) {
    const kb::ResultSink sink = {nullptr, results_dev, cand_index_base};
    return kb::search_filter_impl(meta, psi_phi_dev, times_dev, params, cands_dev, n_cands, sink, n_results, flags, stream,
                                  stats_out);
}

int kb_device_search_counted(const kb_psi_phi_meta* meta, const void* psi_phi_dev, const double* times_dev,
                             kb_search_params params, const kb_trajectory* cands_dev, uint64_t n_cands,
                             int32_t cand_index_base, kb_compact_result* results_dev, uint64_t n_results, uint8_t* counts_dev,
                             uint32_t flags, void* stream, kb_search_stats* stats_out, int32_t* counts_written_out) {
    if (counts_dev == nullptr || counts_written_out == nullptr) return kb::fail("device_search_counted: null pointer");
    const kb::ResultSink sink = {nullptr, results_dev, cand_index_base, counts_dev, params.min_lh};
    int written = 0;
    const int rc = kb::search_filter_impl(meta, psi_phi_dev, times_dev, params, cands_dev, n_cands, sink, n_results, flags, stream,
                                          stats_out, &written);
    *counts_written_out = written;
    return rc;
}

int kb_device_search_filter_counted(const kb_psi_phi_meta* meta, const void* psi_phi_dev, const double* times_dev,
                                    kb_search_params params, const kb_trajectory* cands_dev, uint64_t n_cands,
                                    kb_trajectory* results_dev, uint64_t n_results, uint8_t* counts_dev, uint32_t flags, void* stream,
                                    kb_search_stats* stats_out, int32_t* counts_written_out) {
    if (counts_dev == nullptr || counts_written_out == nullptr) return kb::fail("device_search_filter_counted: null pointer");
    const kb::ResultSink sink = {results_dev, nullptr, 0, counts_dev, params.min_lh};
    int written = 0;
    const int rc = kb::search_filter_impl(meta, psi_phi_dev, times_dev, params, cands_dev, n_cands, sink, n_results, flags, stream,
                                          stats_out, &written);
    *counts_written_out = written;
    return rc;
}

int kb_release_workspaces(void) {
    using namespace kb;
    release_result_arenas();
    release_exchange_arenas();
    int prev = 0;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    for (int dev = 0; dev < MAX_DEVICES; ++dev) {
        std::lock_guard<std::mutex> lock(g_ws_mutex[dev]);
        for (Workspace& w : g_ws_all[dev]) {
            if (w.ptr != nullptr) {
                KB_HIP_TRY(hipSetDevice(w.device));
                KB_HIP_TRY(hipFree(w.ptr));
            }
            w = Workspace();
        }
        g_padded_key[dev].valid = false;
    }
    if (have_prev) (void)hipSetDevice(prev);
    return 0;
}

int kb_note_array_written(const void* ptr_dev) {
    if (ptr_dev != nullptr) kb::note_array_written(ptr_dev);
    return 0;
}

int kb_merge_topk(const kb_trajectory* lists_dev, int32_t n_lists, uint64_t n_pixels, int32_t K,
                  kb_trajectory* out_dev, void* stream_v) {
    using namespace kb;
    if (lists_dev == nullptr || out_dev == nullptr) return fail("merge_topk: null pointer");
    if (n_lists <= 0 || n_lists > MERGE_MAX_LISTS) return fail("merge_topk: unsupported number of lists");
    if (K <= 0 || K > 255) return fail("merge_topk: unsupported K");
    if (n_pixels == 0) return 0;
    KB_REQUIRE_DEVICE("the list merge.");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    const unsigned blocks = (unsigned)((n_pixels + 255) / 256);
    hipLaunchKernelGGL(kb_merge_topk_kernel, dim3(blocks), dim3(256), 0, stream, lists_dev, n_lists, n_pixels, K,
                       out_dev);
    KB_HIP_TRY(hipGetLastError());
    return 0;
}

int kb_merge_compact(const kb_compact_result* lists_dev, int32_t n_lists, kb_search_params params,
                     const kb_trajectory* all_cands_dev, uint64_t n_all_cands, kb_trajectory* out_dev, void* stream_v) {
    using namespace kb;
    if (lists_dev == nullptr || out_dev == nullptr || all_cands_dev == nullptr) return fail("merge_compact: null pointer");
    if (n_lists <= 0 || n_lists > MERGE_MAX_LISTS) return fail("merge_compact: unsupported number of lists");
    const int64_t sw = (int64_t)params.x_start_max - params.x_start_min;
    const int64_t sh = (int64_t)params.y_start_max - params.y_start_min;
    const int K = (int)params.results_per_pixel;
    if (sw <= 0 || sh <= 0) return fail("merge_compact: invalid search bounds");
    if (K <= 0 || K > 32) return fail("merge_compact: unsupported K");
    KB_REQUIRE_DEVICE("the list merge.");
    (void)hipGetLastError();
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    const uint64_t n_pixels = (uint64_t)sw * (uint64_t)sh;
    const unsigned blocks = (unsigned)((n_pixels + 255) / 256);
    if (n_lists <= 8) {
        hipLaunchKernelGGL(kb_merge_compact_kernel<8>, dim3(blocks), dim3(256), 0, stream, lists_dev, n_lists, n_pixels, K,
                           (int)sw, params.x_start_min, params.y_start_min, all_cands_dev, n_all_cands, out_dev);
    } else {
        hipLaunchKernelGGL(kb_merge_compact_kernel<MERGE_MAX_LISTS>, dim3(blocks), dim3(256), 0, stream, lists_dev, n_lists,
                           n_pixels, K, (int)sw, params.x_start_min, params.y_start_min, all_cands_dev, n_all_cands, out_dev);
    }
    KB_HIP_TRY(hipGetLastError());
    return 0;
}

int kb_merge_compact_exact(const kb_compact_result* lists_dev, int32_t n_lists, int32_t list_len, kb_search_params params,
                           const kb_trajectory* all_cands_dev, uint64_t n_all_cands, kb_trajectory* out_dev, void* stream_v) {
    using namespace kb;
    if (lists_dev == nullptr || out_dev == nullptr || all_cands_dev == nullptr) return fail("merge_compact_exact: null pointer");
    if (n_lists <= 0 || n_lists > MERGE_MAX_LISTS) return fail("merge_compact_exact: unsupported number of lists");
    const int64_t sw = (int64_t)params.x_start_max - params.x_start_min;
    const int64_t sh = (int64_t)params.y_start_max - params.y_start_min;
    const int K = (int)params.results_per_pixel;
    if (sw <= 0 || sh <= 0) return fail("merge_compact_exact: invalid search bounds");
    if (K <= 0 || list_len < std::max(K, 2 * K - 1) || list_len > MERGE_EXACT_MAX_K2) {
        return fail("merge_compact_exact: lists of " + std::to_string(list_len) + " records per pixel for " + std::to_string(K) +
                    " results (need 2 K - 1 <= list length <= 32: the merge is exact from there on)");
    }
    KB_REQUIRE_DEVICE("the list merge.");
    (void)hipGetLastError();
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    const uint64_t n_pixels = (uint64_t)sw * (uint64_t)sh;
    hipLaunchKernelGGL(kb_merge_compact_exact_kernel, dim3((unsigned)((n_pixels + 63) / 64)), dim3(64), 0, stream, lists_dev,
                       n_lists, n_pixels, (int)list_len, K, (int)sw, params.x_start_min, params.y_start_min, all_cands_dev,
                       n_all_cands, out_dev);
    KB_HIP_TRY(hipGetLastError());
    return 0;
}

int kb_merge_compact_repairable(const kb_compact_result* lists_dev, int32_t n_lists, kb_search_params params,
                                const kb_trajectory* all_cands_dev, uint64_t n_all_cands, kb_trajectory* out_dev,
                                uint32_t* hazard_idx_dev, uint64_t* n_hazard_host, void* stream_v) {
    using namespace kb;
    if (n_hazard_host == nullptr) return fail("merge_compact_repairable: null count pointer");
    *n_hazard_host = 0;
    if (lists_dev == nullptr || out_dev == nullptr || all_cands_dev == nullptr || hazard_idx_dev == nullptr) {
        return fail("merge_compact_repairable: null pointer");
    }
    if (n_lists <= 0 || n_lists > MERGE_MAX_LISTS) return fail("merge_compact_repairable: unsupported number of lists");
    const int64_t sw = (int64_t)params.x_start_max - params.x_start_min;
    const int64_t sh = (int64_t)params.y_start_max - params.y_start_min;
    const int K = (int)params.results_per_pixel;
    if (sw <= 0 || sh <= 0) return fail("merge_compact_repairable: invalid search bounds");
    if (K <= 0 || K > MERGE_EXACT_MAX_K2) return fail("merge_compact_repairable: lists of 1 to 32 records per pixel");
    const uint64_t n_pixels = (uint64_t)sw * (uint64_t)sh;
    if (n_pixels > 0xffffffffull) return fail("merge_compact_repairable: more than 2^32 start pixels");
    KB_REQUIRE_DEVICE("the list merge.");
    (void)hipGetLastError();
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    // (the hazard counter: eight bytes per device, allocated once -- an allocation and a release per call were 0.5 of the
    // 0.7 ms this entry point took)
    // (held until the count is back: two merges on one device would share the counter; the call waits for its stream anyway)
    static std::mutex counter_mutex[MAX_DEVICES];
    static unsigned long long* counters[MAX_DEVICES] = {};
    const int slot = current_device_slot();
    std::lock_guard<std::mutex> lock(counter_mutex[slot]);
    if (counters[slot] == nullptr) KB_HIP_TRY(hipMalloc(&counters[slot], sizeof(unsigned long long)));
    unsigned long long* const counter = counters[slot];
    int rc = 0;
    unsigned long long n_hazard = 0;
    if (hipMemsetAsync(counter, 0, sizeof(unsigned long long), stream) != hipSuccess) rc = fail("merge_compact_repairable: memset failed");
    if (rc == 0) {
        const dim3 grid((unsigned)((n_pixels + 255) / 256));
#define KB_LAUNCH_MERGE(KT)                                                                                                     \
    hipLaunchKernelGGL(kb_merge_compact_repairable_kernel<KT>, grid, dim3(256), 0, stream, lists_dev, n_lists, n_pixels, K,     \
                       (int)sw, params.x_start_min, params.y_start_min, all_cands_dev, n_all_cands, out_dev, hazard_idx_dev, counter)
        switch (K) {
            case 1: KB_LAUNCH_MERGE(1); break;
            case 2: KB_LAUNCH_MERGE(2); break;
            case 3: KB_LAUNCH_MERGE(3); break;
            case 4: KB_LAUNCH_MERGE(4); break;
            case 5: KB_LAUNCH_MERGE(5); break;
            case 6: KB_LAUNCH_MERGE(6); break;
            case 7: KB_LAUNCH_MERGE(7); break;
            case 8: KB_LAUNCH_MERGE(8); break;
            default: KB_LAUNCH_MERGE(0); break;
        }
#undef KB_LAUNCH_MERGE
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(&n_hazard, counter, sizeof(n_hazard), hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) {
            rc = fail("merge_compact_repairable: launch failed");
        }
    }
    *n_hazard_host = n_hazard;
    return rc;
}

int kb_repair_pixels(const kb_psi_phi_meta* meta, const void* psi_phi_dev, const double* times_dev, kb_search_params params,
                     const kb_trajectory* all_cands_dev, uint64_t n_all_cands, const uint32_t* pixel_idx_dev,
                     uint64_t n_listed, const kb_compact_result* lists_dev, int32_t n_lists, const int32_t* list_cand_begin_host,
                     kb_trajectory* out_dev, void* stream_v) {
    using namespace kb;
    if (n_listed == 0) return 0;
    if (meta == nullptr || psi_phi_dev == nullptr || times_dev == nullptr || all_cands_dev == nullptr || pixel_idx_dev == nullptr ||
        out_dev == nullptr) {
        return fail("repair_pixels: null pointer");
    }
    if (meta->num_times == 0 || meta->num_times > KB_MAX_NUM_IMAGES) return fail("repair_pixels: unsupported number of images");
    if (params.do_sigmag_filter) return fail("repair_pixels: the in-search sigma-G filter is not covered (exchange stable lists of 2 K)");
    const int64_t sw = (int64_t)params.x_start_max - params.x_start_min;
    const int64_t sh = (int64_t)params.y_start_max - params.y_start_min;
    const int K = (int)params.results_per_pixel;
    if (sw <= 0 || sh <= 0 || K <= 0 || K > MERGE_EXACT_MAX_K2) return fail("repair_pixels: invalid bounds or list length");
    RepairLists rl{};
    rl.lists = nullptr;
    if (lists_dev != nullptr && list_cand_begin_host != nullptr && n_lists > 0 && n_lists <= REPAIR_MAX_LISTS) {
        bool ascending = list_cand_begin_host[0] == 0 && (uint64_t)list_cand_begin_host[n_lists] == n_all_cands;
        for (int r = 0; r < n_lists; ++r) ascending = ascending && list_cand_begin_host[r] <= list_cand_begin_host[r + 1];
        if (!ascending) return fail("repair_pixels: the lists' candidate ranges must tile [0, n_all_cands) in ascending order");
        rl.lists = lists_dev;
        rl.n_pixels = (uint64_t)sw * (uint64_t)sh;
        rl.n_lists = n_lists;
        for (int r = 0; r <= n_lists; ++r) rl.begin[r] = list_cand_begin_host[r];
    }
    KB_REQUIRE_DEVICE("the pixel repair.");
    (void)hipGetLastError();
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    if (n_listed > 0x7fffffffull) return fail("repair_pixels: too many pixels listed");
    const unsigned blocks = (unsigned)n_listed;  // one workgroup per pixel
    const size_t lds = (size_t)REPAIR_WAVES * WAVE * sizeof(RepairEval) + (size_t)K * sizeof(kb_compact_result) + 16;
    if (meta->num_bytes == 1) {
        hipLaunchKernelGGL(kb_repair_pixels_kernel<1>, dim3(blocks), dim3(REPAIR_WAVES * WAVE), lds, stream, *meta, psi_phi_dev,
                           times_dev, params, all_cands_dev, n_all_cands, pixel_idx_dev, n_listed, rl, out_dev);
    } else if (meta->num_bytes == 2) {
        hipLaunchKernelGGL(kb_repair_pixels_kernel<2>, dim3(blocks), dim3(REPAIR_WAVES * WAVE), lds, stream, *meta, psi_phi_dev,
                           times_dev, params, all_cands_dev, n_all_cands, pixel_idx_dev, n_listed, rl, out_dev);
    } else {
        hipLaunchKernelGGL(kb_repair_pixels_kernel<4>, dim3(blocks), dim3(REPAIR_WAVES * WAVE), lds, stream, *meta, psi_phi_dev,
                           times_dev, params, all_cands_dev, n_all_cands, pixel_idx_dev, n_listed, rl, out_dev);
    }
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

