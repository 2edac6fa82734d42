// In-search sigma-G filter for MI355X (gfx950): the resolve passes behind kb_device_search_filter
// when params.do_sigmag_filter is set.
//
// Reference semantics (kernels/kernels.cu:199-241, 77-147, 304-331): a trajectory that passes the
// unclipped thresholds is re-summed over the samples whose psi/phi ratio lies within
// median +- 2 * coeff * (q_H - q_L), in ascending ratio order, and the per-pixel list is built from
// the clipped likelihoods in candidate order.  Only a small, spatially clustered fraction of the
// (start pixel, candidate) grid passes (the neighbourhood of every bright source), so clipping inside
// the search kernel leaves the few tiles around the sources running long after the rest of the
// device has finished.  Here the search kernels only EMIT the passing trajectories
// (search_kernels.hip, finish_chunk) and two further launches resolve them:
//
//   * kb_sigmag_clip_kernel   : a device-filling grid of waves walks the work items; each trajectory
//     is clipped by a whole wavefront -- lane t gathers epoch t, the ratios are sorted across the
//     lanes with a DPP sorting network (wave_ops.h), the percentile bounds and the keep range become
//     lane reads and population counts, and the clipped sums are chained across the lanes in sorted
//     order, one fp32 add after the other (the arithmetic of the reference's per-thread loop).
//   * kb_sigmag_select_kernel : one wave per row of 64 start pixels replays the candidate-order
//     swap-down insertion (kernels.cu:323-330) over the clipped likelihoods and writes the results.
//
// Exactness.  The reference sorts with an exchange sort whose permutation among EQUAL ratios is
// neither stable nor simple, and that permutation decides the order in which equal-ratio samples
// are added.  If every run of equal ratios consists of bit-identical (psi, phi) pairs -- the only
// kind of tie a quantised (uint8/uint16) array produces in practice, and it produces them all the
// time -- the order inside the run cannot change a single bit of the sums, and the network's order
// is as good as any.  Only a run that mixes different pairs is handed to the literal per-lane
// restatement of the reference code (search_math.h), as are stacks deeper than 64 epochs.
#include <algorithm>
#include <cstdlib>

#include "search_device.h"
#include "wave_ops.h"

#pragma clang fp contract(off)

namespace kb {

struct ResolveArgs {
    kb_psi_phi_meta meta;
    kb_search_params params;
    const void* psi_phi;
    const double* times;
    const kb_trajectory* cands;
    SigmaGWork sg;
    float* sg_scratch;
    ResultSink prev;  // per-pixel lists of the batches before this one (both pointers null: none)
    ResultSink next;  // per-pixel lists after this batch
    int T, K, sw, sh, tiles_x;
    int n_rows;
    int stable_lists;  // SearchArgs::stable_lists
};

// Monotone 32-bit key of a ratio: -0 and +0 share a key, like the comparison they replace.
__device__ __forceinline__ uint32_t ratio_sort_key(float v) {
    const uint32_t b = __float_as_uint((v == 0.0f) ? 0.0f : v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ratio_from_key(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ float lane_value(float v, int lane) {  // lane: wave-uniform
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// One trajectory's samples for the cooperative clip: lane t holds epoch t.
struct Samples {
    float psi, phi;
};

// Clip of ONE trajectory by the whole wavefront (T <= 64, all 64 lanes active).  Returns false when a
// run of equal ratios mixes different (psi, phi) pairs: the exchange sort's own order then matters
// and the caller runs the literal code for that lane.
// *lh_out / *flux_out receive the two clipped SUMS (psi, phi); the caller divides (once per work item).
// keep (may be null): instead of chaining the sums across the lanes, hand the sorted samples (lane i: the i-th smallest ratio's
// pair) and the range [keep->lo, keep->hi] of sorted positions to add back to the caller, which sums a batch of trajectories
// with one lane each (kb_sigmag_clip_kernel).
struct KeepRange {
    float spsi, sphi;  // per lane
    int lo, hi;        // uniform; lo > hi: nothing to add (the sums are 0)
};
__device__ __forceinline__ bool clip_wave(const ResolveArgs& a, int lane, const Samples s, float* lh_out, float* flux_out,
                                          int* obs_out, KeepRange* keep = nullptr) {
    const bool valid = __builtin_isfinite(s.psi) && __builtin_isfinite(s.phi);
    const int n = __popcll(__ballot(valid));
    *obs_out = n;
    if (n == 0) {  // kernels.cu:201: nothing to clip, the unclipped values stand
        *lh_out = 0.0f;  // sums (0, 0): lh_from_sums / flux_from_sums give the -1 of kernels.cu:201
        *flux_out = 0.0f;
        if (keep != nullptr) {
            keep->spsi = 0.0f;
            keep->sphi = 0.0f;
            keep->lo = 1;
            keep->hi = 0;
        }
        return true;
    }
    const float lc = valid ? ((s.phi != 0.0f) ? (s.psi / s.phi) : 0.0f) : 0.0f;
    uint32_t key = valid ? ratio_sort_key(lc) : 0xffffffffu;  // invalid samples sort behind every ratio
    uint32_t src = (uint32_t)lane;
    wave_sort64(key, src, lane);
    // lane i now holds the i-th smallest ratio and the epoch it came from
    const float spsi = __shfl(s.psi, (int)src), sphi = __shfl(s.phi, (int)src);
    const uint32_t key_next = lane_next(key);
    const uint32_t psi_next = lane_next(__float_as_uint(spsi)), phi_next = lane_next(__float_as_uint(sphi));
    const bool mixed_tie = (lane + 1 < n) && (key == key_next) &&
                           (psi_next != __float_as_uint(spsi) || phi_next != __float_as_uint(sphi));
    if (__ballot(mixed_tie) != 0) return false;
    const float sv = ratio_from_key(key);

    float sgl0 = a.params.sgl_L, sgl1 = a.params.sgl_H;
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
    const float sigma_g = a.params.sigmag_coeff * (lane_value(sv, pct_H) - lane_value(sv, pct_L));
    const float wsg = 2.0f * sigma_g;
    const float vmed = lane_value(sv, median_ind);
    const float min_value = vmed - wsg;
    const float max_value = vmed + wsg;
    // the ratios ascend, so both tests hold on a prefix of the lanes: the reference's two linear scans
    // (kernels.cu:136-146) become population counts
    const int below = __popcll(__ballot((lane < n) && (sv < min_value)));
    const int upto = __popcll(__ballot((lane < n) && (sv <= max_value)));
    const int min_keep = min(below, median_ind);
    const int max_keep = max(median_ind + 1, upto) - 1;
    if (keep != nullptr) {
        keep->spsi = spsi;
        keep->sphi = sphi;
        keep->lo = min_keep;
        keep->hi = max_keep;
        *lh_out = 0.0f;
        *flux_out = 0.0f;
        return true;
    }
    // ((0 + v[min_keep]) + v[min_keep + 1]) + ... + v[max_keep], chained across the lanes
    // One v_add_f32_dpp per sum and step (written out: from chain_add the compiler makes two DPP moves and a packed add,
    // 16 cycles of the vector unit per step instead of 8, and a scalar loop of three more instructions around them -- the
    // chain was a third of this kernel's instructions).  A DPP read of a register needs two wait states behind the vector
    // instruction that wrote it: the other sum's add is one, an s_nop the other.
    float acc_psi = 0.0f, acc_phi = 0.0f;
#define KB_CHAIN_STEP                                                                                   \
    "v_add_f32_dpp %0, %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                   \
    "v_add_f32_dpp %1, %1, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 0\n\t"
    int i = min_keep;
    asm volatile("s_nop 1" : "+v"(acc_psi), "+v"(acc_phi));
    for (; i + 3 <= max_keep; i += 4) {
        asm volatile(KB_CHAIN_STEP KB_CHAIN_STEP KB_CHAIN_STEP KB_CHAIN_STEP : "+v"(acc_psi), "+v"(acc_phi) : "v"(spsi), "v"(sphi));
    }
    for (; i <= max_keep; ++i) {
        asm volatile(KB_CHAIN_STEP : "+v"(acc_psi), "+v"(acc_phi) : "v"(spsi), "v"(sphi));
    }
    asm volatile("s_nop 1" : "+v"(acc_psi), "+v"(acc_phi));
#undef KB_CHAIN_STEP
    const float new_psi = lane_value(acc_psi, max_keep), new_phi = lane_value(acc_phi, max_keep);
    *lh_out = new_psi;
    *flux_out = new_phi;
    return true;
}

// The same clip for stacks of 65 .. 64 * E epochs: lane l holds epochs l, l + 64, ... (E slots), the
// network sorts 64 * E keys (wave_ops.h), sorted position p lives in slot p / 64 of lane p % 64, and the
// chained sums carry from one slot's lane 63 into the next slot's lane 0.
template <int E>
__device__ __forceinline__ float slot_value(const float (&v)[E], int pos) {  // pos: wave-uniform
    float r = 0.0f;
#pragma unroll
    for (int s = 0; s < E; ++s) {
        if ((pos >> 6) == s) r = lane_value(v[s], pos & 63);
    }
    return r;
}

template <int E>
__device__ __forceinline__ bool clip_wave_multi(const ResolveArgs& a, int lane, const float (&psi)[E], const float (&phi)[E],
                                                float* lh_out, float* flux_out, int* obs_out) {
    uint32_t key[E], src[E];
    int n = 0;
#pragma unroll
    for (int s = 0; s < E; ++s) {
        const bool valid = __builtin_isfinite(psi[s]) && __builtin_isfinite(phi[s]);
        n += __popcll(__ballot(valid));
        const float lc = valid ? ((phi[s] != 0.0f) ? (psi[s] / phi[s]) : 0.0f) : 0.0f;
        key[s] = valid ? ratio_sort_key(lc) : 0xffffffffu;  // invalid samples sort behind every ratio
        src[s] = (uint32_t)(s * WAVE + lane);
    }
    *obs_out = n;
    if (n == 0) {  // kernels.cu:201: nothing to clip, the unclipped values stand
        *lh_out = 0.0f;  // sums (0, 0): lh_from_sums / flux_from_sums give the -1 of kernels.cu:201
        *flux_out = 0.0f;
        return true;
    }
    wave_sort_multi<E>(key, src, lane);
    // sorted position p = s * 64 + lane now holds the p-th smallest ratio and the epoch it came from
    float spsi[E], sphi[E], sv[E];
#pragma unroll
    for (int s = 0; s < E; ++s) {
        float gp = 0.0f, gf = 0.0f;
#pragma unroll
        for (int u = 0; u < E; ++u) {  // the sample sits in slot u = src / 64 of lane src % 64
            const float cp = __shfl(psi[u], (int)(src[s] & 63u)), cf = __shfl(phi[u], (int)(src[s] & 63u));
            if ((src[s] >> 6) == (uint32_t)u) {
                gp = cp;
                gf = cf;
            }
        }
        spsi[s] = gp;
        sphi[s] = gf;
        sv[s] = ratio_from_key(key[s]);
    }
    // a run of equal ratios made of different (psi, phi) pairs: the exchange sort's own order matters
    bool mixed_tie = false;
#pragma unroll
    for (int s = 0; s < E; ++s) {
        uint32_t k_next = lane_next(key[s]), p_next = lane_next(__float_as_uint(spsi[s])),
                 f_next = lane_next(__float_as_uint(sphi[s]));
        if (s + 1 < E && lane == WAVE - 1) {  // the neighbour of lane 63 is lane 0 of the next slot
            k_next = (uint32_t)__builtin_amdgcn_readlane((int)key[s + 1 < E ? s + 1 : s], 0);
            p_next = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(spsi[s + 1 < E ? s + 1 : s]), 0);
            f_next = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(sphi[s + 1 < E ? s + 1 : s]), 0);
        }
        const int pos = s * WAVE + lane;
        mixed_tie = mixed_tie || ((pos + 1 < n) && key[s] == k_next &&
                                  (p_next != __float_as_uint(spsi[s]) || f_next != __float_as_uint(sphi[s])));
    }
    if (__ballot(mixed_tie) != 0) return false;

    float sgl0 = a.params.sgl_L, sgl1 = a.params.sgl_H;
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
    const float sigma_g = a.params.sigmag_coeff * (slot_value<E>(sv, pct_H) - slot_value<E>(sv, pct_L));
    const float wsg = 2.0f * sigma_g;
    const float vmed = slot_value<E>(sv, median_ind);
    const float min_value = vmed - wsg;
    const float max_value = vmed + wsg;
    int below = 0, upto = 0;
#pragma unroll
    for (int s = 0; s < E; ++s) {
        const int pos = s * WAVE + lane;
        below += __popcll(__ballot((pos < n) && (sv[s] < min_value)));
        upto += __popcll(__ballot((pos < n) && (sv[s] <= max_value)));
    }
    const int min_keep = min(below, median_ind);
    const int max_keep = max(median_ind + 1, upto) - 1;
    // ((0 + v[min_keep]) + v[min_keep + 1]) + ... + v[max_keep], chained across the lanes, slot after slot
    float sum_psi = 0.0f, sum_phi = 0.0f;
#pragma unroll
    for (int s = 0; s < E; ++s) {
        const int lo = max(min_keep, s * WAVE) - s * WAVE, hi = min(max_keep, s * WAVE + WAVE - 1) - s * WAVE;
        if (lo <= hi) {  // uniform
            float acc_psi = 0.0f, acc_phi = 0.0f;
            for (int i = lo; i <= hi; ++i) {
                acc_psi = chain_add_carry(acc_psi, spsi[s], sum_psi);
                acc_phi = chain_add_carry(acc_phi, sphi[s], sum_phi);
            }
            sum_psi = lane_value(acc_psi, hi);
            sum_phi = lane_value(acc_phi, hi);
        }
    }
    *lh_out = sum_psi;
    *flux_out = sum_phi;
    return true;
}

constexpr int CLIP_BLOCK = 256;
constexpr int CLIP_BATCH = 16;  // trajectories whose kept samples are added up side by side (8.3 KB of LDS per wave)

// E: epochs per lane of the cooperative clip (1: up to 64 epochs, 2: up to 128, 4: up to 256); beyond 64 * E epochs
// every trajectory takes the literal per-lane code.
template <int E>
__global__ __launch_bounds__(CLIP_BLOCK) void kb_sigmag_clip_kernel(const ResolveArgs a) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (CLIP_BLOCK / WAVE) + (threadIdx.x >> 6)));
    const int n_waves = gridDim.x * (CLIP_BLOCK / WAVE);
    const int n_entries = *a.sg.n_entries;
    const bool cooperative = a.T <= E * WAVE;
    double tm[E];
#pragma unroll
    for (int s = 0; s < E; ++s) tm[s] = (s * WAVE + lane < a.T) ? a.times[s * WAVE + lane] : 0.0;
    const SigmaGScratch<WAVE> scratch = make_scratch(a.sg_scratch, a.T, (size_t)wave, lane);

    // T <= 64: the sums of the kept samples -- ((0 + v[lo]) + v[lo + 1]) + ... in sorted order, one add after the other like the
    // reference's loop -- are not chained across the lanes trajectory by trajectory (every step of that chain is a wave-wide
    // instruction of which one lane's result counts: a third of this kernel's instructions) but in batches: a trajectory's sorted
    // pairs go to a row of LDS, and when CLIP_BATCH rows are filled, lane r adds up row r, sixteen chains side by side.
    // Adding +0 for a position outside [lo, hi] changes nothing: a sum that starts at +0 is never -0.
    __shared__ float2 batch_rows[CLIP_BLOCK / WAVE][E == 1 ? CLIP_BATCH : 1][WAVE + 1];  // (+1: sixteen lanes, sixteen pairs of banks)
    float2(*rows)[WAVE + 1] = batch_rows[threadIdx.x >> 6];
    int b_lo = 1, b_hi = 0, b_obs = 0;  // of the trajectory in row `lane`
    size_t b_dest = 0;                  // where its results go
    int n_batched = 0;                  // (uniform) rows filled
    auto flush_batch = [&]() {
        if (n_batched == 0) return;
        __builtin_amdgcn_wave_barrier();  // (the rows were written by this wave's own lanes)
        const bool mine = lane < n_batched;
        const int lo = mine ? b_lo : 1, hi = mine ? b_hi : 0;
        const int from = wave_min_i32(mine ? lo : WAVE), to = wave_max_i32(mine ? hi : -1);  // (uniform) the positions anybody adds
        float acc_psi = 0.0f, acc_phi = 0.0f;
        const float2* row = rows[mine ? lane : 0];
        for (int i = from; i <= to; i += 4) {
            float2 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = row[min(i + j, WAVE - 1)];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool in = (unsigned)(i + j - lo) <= (unsigned)(hi - lo) && hi >= lo;
                acc_psi += in ? v[j].x : 0.0f;
                acc_phi += in ? v[j].y : 0.0f;
            }
        }
        if (mine) {
            a.sg.lh[b_dest] = lh_from_sums(acc_psi, acc_phi);
            a.sg.flux[b_dest] = flux_from_sums(acc_psi, acc_phi);
            a.sg.obs[b_dest] = b_obs;
        }
        n_batched = 0;
    };

    unsigned long long n_clipped = 0, n_literal = 0;
    for (int e = wave; e < n_entries; e += n_waves) {
        const SgEntry ent = a.sg.entries[e];
        const uint64_t mask = ent.mask;
        n_clipped += (unsigned long long)__popcll(mask);
        const int ty = (int)(ent.row / (uint32_t)a.tiles_x), tx = (int)(ent.row - (uint32_t)ty * (uint32_t)a.tiles_x);
        const int x0 = tx * WAVE + a.params.x_start_min, y = ty + a.params.y_start_min;
        const float vx = a.cands[ent.cand].vx, vy = a.cands[ent.cand].vy;
        float lh = 0.0f, flux = 0.0f;
        int obs = 0;
        uint64_t literal = mask;
        uint64_t batched = 0;  // trajectories whose results the batch writes
        if (cooperative) {
            literal = 0;
            int cy[E];
            bool y_ok[E];
#pragma unroll
            for (int s = 0; s < E; ++s) {
                cy[s] = 0;
                y_ok[s] = (s * WAVE + lane < a.T) && predict_index(y, vy, tm[s], &cy[s]);
            }
            auto gather = [&](int x, float (&psi)[E], float (&phi)[E]) {
#pragma unroll
                for (int s = 0; s < E; ++s) {
                    psi[s] = NAN;
                    phi[s] = NAN;
                    const int t = s * WAVE + lane;
                    if (t < a.T) {
                        int cx;
                        const bool x_ok = predict_index(x, vx, tm[s], &cx);
                        if (x_ok && y_ok[s]) read_psi_phi(a.meta, a.psi_phi, (uint64_t)t, cy[s], cx, &psi[s], &phi[s]);
                    }
                }
            };
            // the samples of the next trajectory are fetched while this one is sorted
            uint64_t m = mask;
            int L = __ffsll((unsigned long long)m) - 1;
            float cur_psi[E], cur_phi[E];
            gather(x0 + L, cur_psi, cur_phi);
            while (m != 0) {
                m &= m - 1;
                const int L_next = (m != 0) ? __ffsll((unsigned long long)m) - 1 : L;
                float nxt_psi[E], nxt_phi[E];
#pragma unroll
                for (int s = 0; s < E; ++s) {
                    nxt_psi[s] = cur_psi[s];
                    nxt_phi[s] = cur_phi[s];
                }
                if (m != 0) gather(x0 + L_next, nxt_psi, nxt_phi);
                float r_lh, r_flux;
                int r_obs;
                bool done;
                if constexpr (E == 1) {
                    Samples one;
                    one.psi = cur_psi[0];
                    one.phi = cur_phi[0];
                    KeepRange keep;
                    done = clip_wave(a, lane, one, &r_lh, &r_flux, &r_obs, &keep);
                    if (done) {  // (uniform) into the batch: the row, and its range + destination to the lane that will add it up
                        rows[n_batched][lane] = make_float2(keep.spsi, keep.sphi);
                        if (lane == n_batched) {
                            b_lo = keep.lo;
                            b_hi = keep.hi;
                            b_obs = r_obs;
                            b_dest = (size_t)e * WAVE + (size_t)L;
                        }
                        batched |= 1ull << L;
                        if (++n_batched == CLIP_BATCH) flush_batch();
                    }
                } else {
                    done = clip_wave_multi<E>(a, lane, cur_psi, cur_phi, &r_lh, &r_flux, &r_obs);
                }
                if (done) {
                    if (lane == L) {
                        lh = r_lh;
                        flux = r_flux;
                        obs = r_obs;
                    }
                } else {
                    literal |= 1ull << L;
                }
#pragma unroll
                for (int s = 0; s < E; ++s) {
                    cur_psi[s] = nxt_psi[s];
                    cur_phi[s] = nxt_phi[s];
                }
                L = L_next;
            }
        }
        if (cooperative) {
            // lh / flux of every clipped trajectory of the entry at once: until here lane L carried the two clipped
            // sums of its trajectory (one correctly rounded sqrt and two divides per entry instead of per trajectory)
            const float sum_psi = lh, sum_phi = flux;
            lh = lh_from_sums(sum_psi, sum_phi);
            flux = flux_from_sums(sum_psi, sum_phi);
        }
        n_literal += (unsigned long long)__popcll(literal);
        if ((literal >> lane) & 1) {  // kernels.cu:154-242 as written, one trajectory per lane
            kb_trajectory trj;
            trj.x = x0 + lane;
            trj.y = y;
            trj.vx = vx;
            trj.vy = vy;
            evaluate_trajectory_full<WAVE>(a.meta, a.psi_phi, a.times, a.params, &trj, &scratch);
            lh = trj.lh;
            flux = trj.flux;
            obs = trj.obs_count;
        }
        if (((mask & ~batched) >> lane) & 1) {
            const size_t o = (size_t)e * WAVE + lane;
            a.sg.lh[o] = lh;
            a.sg.flux[o] = flux;
            a.sg.obs[o] = obs;
        }
    }
    flush_batch();
    if (lane == 0) {
        if (wave == 0) atomicAdd(&a.sg.totals[0], (unsigned long long)n_entries);
        if (n_clipped != 0) atomicAdd(&a.sg.totals[1], n_clipped);
        if (n_literal != 0) atomicAdd(&a.sg.totals[2], n_literal);
    }
}

// Per-pixel selection over the clipped likelihoods: kernels.cu:318-331 in candidate order.  One wave
// per row of 64 start pixels; the list holds (lh, reference): reference >= 0 is an entry of this
// batch, <= -2 slot -(reference + 2) of the lists left by the batches before, -1 an empty slot.
template <int KS>
__global__ __launch_bounds__(256) void kb_sigmag_select_kernel(const ResolveArgs a) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int row = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (row >= a.n_rows) return;
    const int ty = row / a.tiles_x, tx = row - ty * a.tiles_x;
    const int x_i = tx * WAVE + lane;
    const bool live = x_i < a.sw;
    const size_t pixel = (size_t)ty * a.sw + x_i;

    TopK<KS> top;
    top.init();
    const bool have_prev = a.prev.full != nullptr || a.prev.compact != nullptr;
    if (have_prev && live) {
        for (int s = 0; s < a.K; ++s) {
            const float old_lh = (a.prev.compact != nullptr) ? a.prev.compact[pixel * a.K + s].lh
                                                              : a.prev.full[pixel * a.K + s].lh;
            const bool filled = !(old_lh == -FLT_MAX);  // placeholders carry -FLT_MAX, which nothing inserted can
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                if (k == s && filled) {
                    top.lh[k] = old_lh;
                    top.id[k] = -(s + 2);
                }
            }
        }
    }

    const uint32_t* slot_row = a.sg.slots + (size_t)row * a.sg.batch_cands;
    // A row that crosses a bright mover holds hundreds of entries, and the launch ends with those rows: what its duration
    // follows is the chain of dependent round trips of ONE wave.  A row's slot words are fetched 1024 at a time (sixteen loads in
    // flight); the entries they name are compacted, in candidate order, into the wave's list in LDS; then batches of AHEAD
    // entries run through a two-stage pipeline -- the masks and likelihoods of the next batch are requested before this batch
    // is inserted (round 6; until then request -> wait -> insert, batch after batch: 0.28 ms at cfg3).
    constexpr int AHEAD = KS <= 8 ? 16 : 8;
    constexpr int BLOCKS = 16;  // blocks of 64 slot words fetched together
    __shared__ uint32_t row_items[4][BLOCKS * WAVE];
    typedef __attribute__((address_space(3))) uint32_t* LdsWords;
    const LdsWords items = (LdsWords)(uint32_t)(uintptr_t)row_items[threadIdx.x >> 6];
    const uint32_t* mask_words = reinterpret_cast<const uint32_t*>(a.sg.entries);  // entry e: words 4 e + 2 (lanes 0-31), 4 e + 3
    struct Batch {
        int e[AHEAD];        // (uniform) entry numbers, -1 past the end
        uint32_t mw[AHEAD];  // this lane's half of the entry's mask
        float lh[AHEAD];
    };
    for (int c_base = 0; c_base < a.sg.batch_cands; c_base += BLOCKS * WAVE) {
        uint32_t slots[BLOCKS];
#pragma unroll
        for (int j = 0; j < BLOCKS; ++j) {
            const int idx = c_base + j * WAVE + lane;
            slots[j] = (idx < a.sg.batch_cands) ? slot_row[idx] : 0u;
        }
        int n = 0;  // (uniform) entries of these 1024 candidates
#pragma unroll
        for (int j = 0; j < BLOCKS; ++j) {
            const uint64_t m = __ballot(slots[j] != 0u);
            const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (slots[j] != 0u) items[n + below] = slots[j] - 1u;
            n += __popcll(m);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the list is written
        __builtin_amdgcn_wave_barrier();
        auto request = [&](int b, Batch& x) {
            const int idx = b * AHEAD + (lane & (AHEAD - 1));
            const int ev = (idx < n) ? (int)items[idx] : -1;
#pragma unroll
            for (int k = 0; k < AHEAD; ++k) {
                x.e[k] = __builtin_amdgcn_readlane(ev, k);
                const size_t es = (size_t)max(x.e[k], 0);  // (past the end: entry 0's words, never looked at)
                x.mw[k] = mask_words[4 * es + 2 + (size_t)(lane >> 5)];
                x.lh[k] = a.sg.lh[es * WAVE + lane];
            }
        };
        auto insert_batch = [&](const Batch& x) {
#pragma unroll
            for (int k = 0; k < AHEAD; ++k) {
                // kernels.cu:318-320 on the clipped value (obs_count was tested before the clip and is unchanged)
                if (x.e[k] >= 0 && ((x.mw[k] >> (lane & 31)) & 1u) && !(x.lh[k] < a.params.min_lh)) {
                    top.insert(x.lh[k], x.e[k], a.stable_lists != 0);
                }
            }
        };
        const int n_batches = (n + AHEAD - 1) / AHEAD;
        Batch even, odd;
        if (n_batches > 0) request(0, even);
        for (int b = 0; b < n_batches; b += 2) {
            if (b + 1 < n_batches) request(b + 1, odd);
            insert_batch(even);
            if (b + 2 < n_batches) request(b + 2, even);
            if (b + 1 < n_batches) insert_batch(odd);
        }
        __builtin_amdgcn_wave_barrier();  // (the list is rewritten by the next 1024 candidates)
    }

    // every entry of a list has passed min_lh: the list's length is the count ResultSink::counts asks for
    int filled = 0;
#pragma unroll
    for (int k = 0; k < KS; ++k) filled += (k < a.K && top.id[k] != -1) ? 1 : 0;
    const int x = x_i + a.params.x_start_min, y = ty + a.params.y_start_min;
    // slot s of this lane's list as a full record and the candidate behind it (-1: the placeholder of an empty slot)
    auto record = [&](int s, kb_trajectory* res, int* cand) {
        int ref = -1;
        float lh_s = -FLT_MAX;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            if (k == s) {
                ref = top.id[k];
                lh_s = top.lh[k];
            }
        }
        *res = placeholder_result(x, y);  // kernels.cu:293-301
        *cand = -1;
        if (ref <= -2) {  // carried over from an earlier batch
            const size_t from = pixel * a.K + (size_t)(-(ref + 2));
            if (a.prev.compact != nullptr) {
                const kb_compact_result old = a.prev.compact[from];
                res->lh = old.lh;
                res->flux = old.flux;
                res->obs_count = old.obs_count;
                *cand = old.cand;  // (already offset by cand_base)
            } else {
                *res = a.prev.full[from];
                *cand = 0;
            }
        } else if (ref >= 0) {
            *cand = (int)a.sg.entries[ref].cand;
            res->vx = a.cands[*cand].vx;
            res->vy = a.cands[*cand].vy;
            res->lh = lh_s;
            res->flux = a.sg.flux[(size_t)ref * WAVE + lane];
            res->obs_count = a.sg.obs[(size_t)ref * WAVE + lane];
            *cand += a.next.cand_base;
        }
    };
    if constexpr (KS <= 16) {
        // the wave's 64 x K records are one contiguous run of the result array: coalesced stores through a patch of LDS
        // (store_wave_records, search_device.h; it also writes the count bytes and leaves out a row of 64 empty lists)
        __shared__ uint32_t patches[4][32 * (7 * KS + 1)];
        char* patch = reinterpret_cast<char*>(patches[threadIdx.x >> 6]);
        const size_t run0 = ((size_t)ty * a.sw + (size_t)(tx * WAVE)) * a.K;
        uint8_t* counts_run = a.next.counts != nullptr ? a.next.counts + ((size_t)ty * a.sw + (size_t)(tx * WAVE)) : nullptr;
        if (a.next.compact != nullptr) {
            store_wave_records<4>(reinterpret_cast<uint32_t*>(a.next.compact + run0), a.K, live, patch, [&](int s, uint32_t (&w)[4]) {
                kb_trajectory res;
                int cand;
                record(s, &res, &cand);
                w[0] = __float_as_uint(res.lh);
                w[1] = __float_as_uint(res.flux);
                w[2] = (uint32_t)cand;
                w[3] = (uint32_t)res.obs_count;
            }, counts_run, filled);
        } else {
            store_wave_records<7>(reinterpret_cast<uint32_t*>(a.next.full + run0), a.K, live, patch, [&](int s, uint32_t (&w)[7]) {
                kb_trajectory res;
                int cand;
                record(s, &res, &cand);
                w[0] = __float_as_uint(res.vx);
                w[1] = __float_as_uint(res.vy);
                w[2] = __float_as_uint(res.lh);
                w[3] = __float_as_uint(res.flux);
                w[4] = (uint32_t)res.x;
                w[5] = (uint32_t)res.y;
                w[6] = (uint32_t)res.obs_count;
            }, counts_run, filled);
        }
        return;
    }
    if (a.next.counts != nullptr) {  // (uniform)
        if (live) a.next.counts[pixel] = (uint8_t)filled;
        if (__ballot(live && filled != 0) == 0ull) return;  // ... and a row of 64 empty lists writes no slot
    }
    if (!live) return;
    for (int s = 0; s < a.K; ++s) {
        kb_trajectory res;
        int cand;
        record(s, &res, &cand);
        const size_t slot = pixel * a.K + s;
        if (a.next.compact != nullptr) {
            kb_compact_result out;
            out.lh = res.lh;
            out.flux = res.flux;
            out.cand = cand;
            out.obs_count = res.obs_count;
            a.next.compact[slot] = out;
        } else {
            a.next.full[slot] = res;
        }
    }
}

int launch_sigmag_resolve(const SearchArgs& s, const SearchCold& cold, const ResultSink* prev, const ResultSink& next,
                          int scratch_waves, hipStream_t stream) {
    ResolveArgs a;
    a.meta = cold.meta;
    a.params = cold.params;
    a.psi_phi = s.psi_phi;
    a.times = cold.times;
    a.cands = cold.cands;
    a.sg = cold.sg;
    a.sg_scratch = cold.sg_scratch;
    a.prev = (prev != nullptr) ? *prev : ResultSink{nullptr, nullptr, 0};
    a.next = next;
    a.T = s.T;
    a.K = s.K;
    a.sw = s.sw;
    a.sh = s.sh;
    a.tiles_x = s.tiles_x;
    a.n_rows = s.tiles_x * s.sh;
    a.stable_lists = s.stable_lists;
    const int clip_blocks = std::max(1, scratch_waves / (CLIP_BLOCK / WAVE));
    if (a.T <= WAVE || a.T > 4 * WAVE) {  // beyond 256 epochs: the literal code only
        hipLaunchKernelGGL(kb_sigmag_clip_kernel<1>, dim3(clip_blocks), dim3(CLIP_BLOCK), 0, stream, a);
    } else if (a.T <= 2 * WAVE) {
        hipLaunchKernelGGL(kb_sigmag_clip_kernel<2>, dim3(clip_blocks), dim3(CLIP_BLOCK), 0, stream, a);
    } else {
        hipLaunchKernelGGL(kb_sigmag_clip_kernel<4>, dim3(clip_blocks), dim3(CLIP_BLOCK), 0, stream, a);
    }
    KB_HIP_TRY(hipGetLastError());
    const dim3 grid((unsigned)((a.n_rows + 3) / 4)), block(256);
    if (a.K <= 8) {
        hipLaunchKernelGGL((kb_sigmag_select_kernel<8>), grid, block, 0, stream, a);
    } else if (a.K <= 16) {
        hipLaunchKernelGGL((kb_sigmag_select_kernel<16>), grid, block, 0, stream, a);
    } else {
        hipLaunchKernelGGL((kb_sigmag_select_kernel<32>), grid, block, 0, stream, a);
    }
    KB_HIP_TRY(hipGetLastError());
    return 0;
}

// Self-test kernels of the cross-lane primitives (kb_debug_wave_ops).
__global__ __launch_bounds__(WAVE) void kb_debug_wave_sort_kernel(const uint32_t* __restrict__ keys,
                                                                  uint32_t* __restrict__ keys_out,
                                                                  uint32_t* __restrict__ src_out) {
    const int lane = threadIdx.x;
    uint32_t key = keys[(size_t)blockIdx.x * WAVE + lane], src = (uint32_t)lane;
    wave_sort64(key, src, lane);
    keys_out[(size_t)blockIdx.x * WAVE + lane] = key;
    src_out[(size_t)blockIdx.x * WAVE + lane] = src;
}

__global__ __launch_bounds__(WAVE) void kb_debug_chain_sum_kernel(const float* __restrict__ values,
                                                                  const int* __restrict__ bounds,
                                                                  float* __restrict__ sums) {
    const int lane = threadIdx.x;
    const float v = values[(size_t)blockIdx.x * WAVE + lane];
    const int lo = bounds[2 * blockIdx.x], hi = bounds[2 * blockIdx.x + 1];
    float acc = 0.0f;
    for (int i = lo; i <= hi; ++i) acc = chain_add(acc, v);
    const float r = lane_value(acc, hi);
    if (lane == 0) sums[blockIdx.x] = r;
}

}  // namespace kb

extern "C" int kb_debug_wave_ops(const uint32_t* keys_dev, uint32_t* keys_out_dev, uint32_t* src_out_dev,
                                 const float* values_dev, const int32_t* bounds_dev, float* sums_dev, uint64_t n_waves,
                                 void* stream_v) {
    using namespace kb;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    if (n_waves == 0) return 0;
    KB_REQUIRE_DEVICE("the wave-ops self-test.");
    (void)hipGetLastError();  // a stale error of this thread (another library's probing) is not ours
    if (keys_dev != nullptr) {
        hipLaunchKernelGGL(kb_debug_wave_sort_kernel, dim3((unsigned)n_waves), dim3(WAVE), 0, stream, keys_dev, keys_out_dev,
                           src_out_dev);
    }
    if (values_dev != nullptr) {
        hipLaunchKernelGGL(kb_debug_chain_sum_kernel, dim3((unsigned)n_waves), dim3(WAVE), 0, stream, values_dev, bounds_dev,
                           sums_dev);
    }
    KB_HIP_TRY(hipGetLastError());
    KB_HIP_TRY(hipStreamSynchronize(stream));
    return 0;
}
