// Arithmetic shared by the device search kernel and its host instantiation.
// Everything here must reproduce the reference's IEEE operation sequence bit
// for bit; the library is built with
// -ffp-contract=off and the position formula additionally uses the explicit
// round-to-nearest intrinsics on the device so that no FMA can appear.
//
// References (relative to /root/reference/src/kbmod/search/):
//   predict_index             kernels/kernels.cu:33-35, cpu_search_algorithms.cpp:35-36
//   read_encoded_psi_phi      kernels/kernels.cu:37-71, psi_phi_array.cpp:172-205
//   SigmaGFilteredIndicesCU   kernels/kernels.cu:77-147
//   evaluateTrajectory        kernels/kernels.cu:154-242
#ifndef KB_SEARCH_MATH_H_
#define KB_SEARCH_MATH_H_

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif

#include <cfloat>
#include <cmath>
#include <cstdint>

#include "kbmod_hip.h"

#pragma clang fp contract(off)

namespace kb {

// The header also serves the host module (g++): there the functions are plain inline host functions.
#if defined(__HIPCC__)
#define KB_HD __host__ __device__ __forceinline__
#else
#define KB_HD inline
#endif

// (int)floor((double)pos0 + (double)vel0 * time + 0.5) with every operation
// rounded separately.  Returns false when the value does not fit an int (the
// reference then indexes out of bounds and reads NO_DATA either way).
KB_HD bool predict_index(int pos0, float vel0, double time, int* out) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double prod = __dmul_rn((double)vel0, time);
    const double s = __dadd_rn((double)pos0, prod);
    const double v = floor(__dadd_rn(s, 0.5));
#else
    volatile double prod = (double)vel0 * time;  // volatile: keep the separate roundings on the host too
    volatile double s = (double)pos0 + prod;
    const double v = std::floor(s + 0.5);
#endif
    if (!(v >= -2147483648.0 && v <= 2147483647.0)) {
        *out = -1;
        return false;
    }
    *out = (int)v;
    return true;
}

// decode_uint_scalar (psi_phi_array_ds.h:45-47): double arithmetic, one
// rounding to float at the end.  code != 0.
KB_HD float decode_code(float code, float scale, float min_val) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (float)__dadd_rn(__dmul_rn(__dadd_rn((double)code, -1.0), (double)scale), (double)min_val);
#else
    volatile double p = ((double)code - 1.0) * (double)scale;
    return (float)(p + (double)min_val);
#endif
}

// One (psi, phi) sample with the reference's bounds rule; NaN = NO_DATA.
KB_HD void read_psi_phi(const kb_psi_phi_meta& m, const void* arr, uint64_t t, int row, int col, float* psi,
                        float* phi) {
    *psi = NAN;
    *phi = NAN;
    if (row < 0 || col < 0 || (uint64_t)row >= m.height || (uint64_t)col >= m.width || t >= m.num_times ||
        arr == nullptr)
        return;
    const uint64_t start = 2 * (m.pixels_per_image * t + (uint64_t)row * m.width + (uint64_t)col);
    if (m.num_bytes == 4) {
        *psi = reinterpret_cast<const float*>(arr)[start];
        *phi = reinterpret_cast<const float*>(arr)[start + 1];
        return;
    }
    float pv, fv;
    if (m.num_bytes == 1) {
        pv = (float)reinterpret_cast<const uint8_t*>(arr)[start];
        fv = (float)reinterpret_cast<const uint8_t*>(arr)[start + 1];
    } else {
        pv = (float)reinterpret_cast<const uint16_t*>(arr)[start];
        fv = (float)reinterpret_cast<const uint16_t*>(arr)[start + 1];
    }
    *psi = (pv == 0.0f) ? NAN : decode_code(pv, m.psi_scale, m.psi_min_val);
    *phi = (fv == 0.0f) ? NAN : decode_code(fv, m.phi_scale, m.phi_min_val);
}

KB_HD float lh_from_sums(float psi_sum, float phi_sum) {
    return (phi_sum > 0.0f) ? (psi_sum / sqrtf(phi_sum)) : -1.0f;
}
KB_HD float flux_from_sums(float psi_sum, float phi_sum) {
    return (phi_sum > 0.0f) ? (psi_sum / phi_sum) : -1.0f;
}

// Strided views: the device keeps per-lane scratch interleaved across the 64
// lanes of a wave (element i of a lane at p[i * 64]); the host uses stride 1.
template <typename T, int STRIDE>
struct StridedView {
    T* p;
    KB_HD T& operator[](int i) const { return p[(size_t)i * STRIDE]; }
};

// kernels.cu:77-147.  The exchange sort is reproduced literally (idx[j] is held
// in a register across the inner loop, which does not change the sequence of
// comparisons or swaps): the permutation it leaves among equal values decides
// the summation order of the clipped sums.
template <typename VAL, typename IDX>
KB_HD void sigmag_filtered_indices_t(const VAL& values, int num_values, float sgl0, float sgl1, float sigmag_coeff,
                                     float width, const IDX& idx_array, int* min_keep_idx, int* max_keep_idx) {
    if (num_values == 0) {
        *min_keep_idx = 0;
        *max_keep_idx = -1;
        return;
    }
    if ((double)sgl0 < 0.0001) sgl0 = (float)0.0001;
    if ((double)sgl1 > 0.9999) sgl1 = (float)0.9999;

    for (int j = 0; j < num_values; j++) idx_array[j] = j;
    for (int j = 0; j < num_values; j++) {
        int ij = idx_array[j];
        float vj = values[ij];
        for (int k = j + 1; k < num_values; k++) {
            const int ik = idx_array[k];
            const float vk = values[ik];
            if (vj > vk) {
                idx_array[k] = ij;
                ij = ik;
                vj = vk;
            }
        }
        idx_array[j] = ij;
    }
    int pct_L = (int)((double)ceilf((float)num_values * sgl0) + 0.001) - 1;
    pct_L = (pct_L < 0) ? 0 : pct_L;
    pct_L = (pct_L >= num_values) ? (num_values - 1) : pct_L;
    int pct_H = (int)((double)ceilf((float)num_values * sgl1) + 0.001) - 1;
    pct_H = (pct_H < 0) ? 0 : pct_H;
    pct_H = (pct_H >= num_values) ? (num_values - 1) : pct_H;
    int median_ind = (int)(ceil((double)num_values * 0.5) + 0.001) - 1;
    median_ind = (median_ind < 0) ? 0 : median_ind;
    median_ind = (median_ind >= num_values) ? (num_values - 1) : median_ind;

    const float sigma_g = sigmag_coeff * (values[idx_array[pct_H]] - values[idx_array[pct_L]]);
    const float wsg = width * sigma_g;
    const float vmed = values[idx_array[median_ind]];
    const float min_value = vmed - wsg;
    const float max_value = vmed + wsg;

    int start = 0;
    while ((start < median_ind) && (values[idx_array[start]] < min_value)) ++start;
    *min_keep_idx = start;
    int end = median_ind + 1;
    while ((end < num_values) && (values[idx_array[end]] <= max_value)) ++end;
    *max_keep_idx = end - 1;
}

// The same exchange sort and bounds for the in-kernel clip, on (value, index) PAIRS sorted in place:
// sv[k] carries values[idx[k]] along with idx[k], so that the inner loop reads two independent,
// consecutive streams instead of an index and then the value it points to, and its loads can be
// issued eight iterations ahead (iteration k only writes position k and the carried (ij, vj)).
// The sequence of comparisons and swaps -- hence the permutation left among equal values -- is
// that of sigmag_filtered_indices_t.  On exit sv is sorted and idx is the permutation.
template <typename VAL, typename IDX>
KB_HD void sigmag_sorted_bounds_t(const VAL& sv, int num_values, float sgl0, float sgl1, float sigmag_coeff, float width,
                                  const IDX& idx_array, int* min_keep_idx, int* max_keep_idx) {
    if (num_values == 0) {
        *min_keep_idx = 0;
        *max_keep_idx = -1;
        return;
    }
    if ((double)sgl0 < 0.0001) sgl0 = (float)0.0001;
    if ((double)sgl1 > 0.9999) sgl1 = (float)0.9999;

    for (int j = 0; j < num_values; j++) idx_array[j] = j;
    constexpr int B = 8;
    for (int j = 0; j < num_values; j++) {
        int ij = idx_array[j];
        float vj = sv[j];
        int k = j + 1;
        for (; k + B <= num_values; k += B) {
            int ik[B];
            float vk[B];
#pragma unroll
            for (int u = 0; u < B; ++u) {
                ik[u] = idx_array[k + u];
                vk[u] = sv[k + u];
            }
#pragma unroll
            for (int u = 0; u < B; ++u) {
                if (vj > vk[u]) {
                    idx_array[k + u] = ij;
                    sv[k + u] = vj;
                    ij = ik[u];
                    vj = vk[u];
                }
            }
        }
        for (; k < num_values; k++) {
            const int ik = idx_array[k];
            const float vk = sv[k];
            if (vj > vk) {
                idx_array[k] = ij;
                sv[k] = vj;
                ij = ik;
                vj = vk;
            }
        }
        idx_array[j] = ij;
        sv[j] = vj;
    }
    int pct_L = (int)((double)ceilf((float)num_values * sgl0) + 0.001) - 1;
    pct_L = (pct_L < 0) ? 0 : pct_L;
    pct_L = (pct_L >= num_values) ? (num_values - 1) : pct_L;
    int pct_H = (int)((double)ceilf((float)num_values * sgl1) + 0.001) - 1;
    pct_H = (pct_H < 0) ? 0 : pct_H;
    pct_H = (pct_H >= num_values) ? (num_values - 1) : pct_H;
    int median_ind = (int)(ceil((double)num_values * 0.5) + 0.001) - 1;
    median_ind = (median_ind < 0) ? 0 : median_ind;
    median_ind = (median_ind >= num_values) ? (num_values - 1) : median_ind;

    const float sigma_g = sigmag_coeff * (sv[pct_H] - sv[pct_L]);
    const float wsg = width * sigma_g;
    const float vmed = sv[median_ind];
    const float min_value = vmed - wsg;
    const float max_value = vmed + wsg;

    int start = 0;
    while ((start < median_ind) && (sv[start] < min_value)) ++start;
    *min_keep_idx = start;
    int end = median_ind + 1;
    while ((end < num_values) && (sv[end] <= max_value)) ++end;
    *max_keep_idx = end - 1;
}

// Scratch for one clipped evaluation: 3 floats + 1 index per epoch.
template <int STRIDE>
struct SigmaGScratch {
    StridedView<float, STRIDE> psi, phi, lc;
    StridedView<int, STRIDE> idx;
};

// Full evaluateTrajectory (kernels.cu:154-242) for one trajectory with exact
// per-sample positions.  scratch may be null when !do_sigmag_filter.
template <int STRIDE>
KB_HD void evaluate_trajectory_full(const kb_psi_phi_meta& m, const void* arr, const double* times,
                                    const kb_search_params& p, kb_trajectory* c,
                                    const SigmaGScratch<STRIDE>* scratch) {
    float psi_sum = 0.0f, phi_sum = 0.0f;
    c->obs_count = 0;
    c->lh = -1.0f;
    c->flux = -1.0f;
    const bool keep = p.do_sigmag_filter && scratch != nullptr;
    int num_seen = 0;
    const int T = (int)m.num_times;
    for (int i = 0; i < T; ++i) {
        const double t = times[i];
        int cx, cy;
        const bool okx = predict_index(c->x, c->vx, t, &cx);
        const bool oky = predict_index(c->y, c->vy, t, &cy);
        float psi = NAN, phi = NAN;
        if (okx && oky) read_psi_phi(m, arr, (uint64_t)i, cy, cx, &psi, &phi);
        if (__builtin_isfinite(psi) && __builtin_isfinite(phi)) {
            psi_sum += psi;
            phi_sum += phi;
            if (keep) {
                scratch->psi[num_seen] = psi;
                scratch->phi[num_seen] = phi;
            }
            num_seen += 1;
        }
    }
    c->obs_count = num_seen;
    c->lh = lh_from_sums(psi_sum, phi_sum);
    c->flux = flux_from_sums(psi_sum, phi_sum);

    if ((c->obs_count < p.min_observations) || (c->obs_count == 0) ||
        (p.do_sigmag_filter && c->lh < p.min_lh))
        return;
    if (!keep) return;

    for (int i = 0; i < num_seen; ++i) {
        const float f = scratch->phi[i];
        scratch->lc[i] = (f != 0.0f) ? (scratch->psi[i] / f) : 0.0f;
    }
    int min_keep = 0, max_keep = num_seen - 1;
    sigmag_sorted_bounds_t(scratch->lc, num_seen, p.sgl_L, p.sgl_H, p.sigmag_coeff, 2.0f, scratch->idx, &min_keep,
                           &max_keep);
    if (min_keep < 0) min_keep = 0;
    if (max_keep >= num_seen) max_keep = num_seen - 1;
    float new_psi = 0.0f, new_phi = 0.0f;
    for (int i = min_keep; i <= max_keep; i++) {  // sorted-value order
        const int id = scratch->idx[i];
        new_psi += scratch->psi[id];
        new_phi += scratch->phi[id];
    }
    c->lh = lh_from_sums(new_psi, new_phi);
    c->flux = flux_from_sums(new_psi, new_phi);
}

// ---------------------------------------------------------------------------------------------------------------
// Tie-exact merge of per-device lists (kb_merge_compact_exact and its host twin; see search_kernels.hip).
// read(list, position) -> kb_compact_result of this pixel; merged / heads / slots: caller's scratch of
// MERGE_EXACT_MAX_K2 / n_lists / MERGE_EXACT_MAX_K2 entries.  On return slots[0 .. n) index `merged`: the pixel's
// final list.
// ---------------------------------------------------------------------------------------------------------------
constexpr int MERGE_EXACT_MAX_K2 = 32;
// Exact from lists of 2 K - 1 records on (G lies inside the first 2 K - 1 entries of the union under that order); the entry
// points refuse shorter lists.  (A K-record form with a "hidden tie" mark on a list's last record was built, measured and
// taken out again in round 4 -- LABNOTES.md --; nothing of it is left here.)
struct MergedEntry {
    float lh;
    int cand;
    uint32_t at;  // list * K2 + position: where the record sits
};

template <typename ReadRecord>
KB_HD int merge_exact_pixel(const ReadRecord& read, int n_lists, int K2, int K, MergedEntry* merged,
                                                 int* heads, int* slots) {
    // (1) the first entries of the union by (lh descending, candidate ascending) -- K2 of them, 2 K - 1 when the lists are
    // shorter than that --; every list is in that order
    for (int r = 0; r < n_lists; ++r) heads[r] = 0;
    int n = 0;
    int want = 2 * K - 1 > K2 ? 2 * K - 1 : K2;
    if (want > MERGE_EXACT_MAX_K2) want = MERGE_EXACT_MAX_K2;
    while (n < want) {
        int best = -1;
        float best_lh = 0.0f;
        int best_cand = 0;
        for (int r = 0; r < n_lists; ++r) {
            if (heads[r] >= K2) continue;
            const kb_compact_result rec = read(r, heads[r]);
            if (rec.cand < 0) {  // placeholders close a list
                heads[r] = K2;
                continue;
            }
            if (best < 0 || rec.lh > best_lh || (rec.lh == best_lh && rec.cand < best_cand)) {
                best = r;
                best_lh = rec.lh;
                best_cand = rec.cand;
            }
        }
        if (best < 0) break;
        merged[n].lh = best_lh;
        merged[n].cand = best_cand;
        merged[n].at = (uint32_t)(best * K2 + heads[best]);
        heads[best] += 1;
        n += 1;
    }
    // (2) nothing equal among the first K + 1: the prefix is the list
    bool strict = true;
    for (int i = 0; i + 1 < n && i < K; ++i) strict = strict && (merged[i].lh > merged[i + 1].lh);
    const int n_out = n < K ? n : K;
    if (strict) {
        for (int s = 0; s < n_out; ++s) slots[s] = s;
        return n_out;
    }
    // (3) replay G in candidate order with the reference's insertion
    const bool full = n >= K;
    const float v = full ? merged[K - 1].lh : 0.0f;
    int n_equal = 0;
    uint32_t in_g = 0;  // bit i: merged[i] belongs to G (K2 <= 32)
    for (int i = 0; i < n; ++i) {
        bool take = true;
        if (full) {
            if (merged[i].lh == v) {
                take = n_equal < K;
                n_equal += 1;
            } else {
                take = merged[i].lh > v;
            }
        }
        if (take) in_g |= 1u << i;
    }
    int filled = 0;
    for (int s = 0; s < K; ++s) slots[s] = -1;
    while (in_g != 0u) {
        int pick = -1;
        for (int i = 0; i < n; ++i) {
            if (((in_g >> i) & 1u) && (pick < 0 || merged[i].cand < merged[pick].cand)) pick = i;
        }
        in_g &= ~(1u << pick);
        int cur = pick;
        for (int s = 0; s < K; ++s) {  // kernels.cu:323-330: strict '>' against a slot, an empty slot loses
            if (slots[s] < 0 || merged[cur].lh > merged[slots[s]].lh) {
                const int t = slots[s];
                slots[s] = cur;
                cur = t;
                if (cur < 0) break;
            }
        }
        if (filled < K) filled += 1;
    }
    return filled < n_out ? filled : n_out;
}

// ---------------------------------------------------------------------------------------------------------------
// The exchange with K records per device (kb_merge_compact_repairable and its host twin; round 6): a FOLD over the
// devices in candidate order.  Device r searched the candidates [begin_r, begin_{r+1}) of the job-wide list -- the slices
// tile the list in ascending order -- and its list L_r is what the reference's insertion (kernels.cu:304-331) leaves over
// that slice alone: K slots, likelihoods non-increasing, runs of equal values in whatever order the rotations left them.
//   * The reference walks the candidates in order, so after slice 0 its list IS L_0: the fold starts there, exactly.
//   * Slice r >= 1 is then inserted into that state.  Of its candidates only L_r is known; the ones that fell off L_r (or
//     never entered it) have lh <= m_r = L_r's last likelihood when L_r is full, and do not exist when it is not.  A
//     candidate at or below the state's last likelihood moves nothing (the swaps need a strict '>'), and one that does
//     enter sits below every entry above it and is itself pushed out again once K entries at or above m_r are in -- which
//     L_r's own K records guarantee by the end of the slice --, disturbing on its way only entries that leave with it.  So
//     inserting L_r's records, put back into candidate order, gives the reference's state after slice r UNLESS a dropped
//     candidate EQUAL to m_r can both enter (m_r above the state's last likelihood before the slice) and stay (m_r is the
//     state's last likelihood after it): then which members of that tie the reference keeps is not in the records.
//     That slice is SUSPECT and the pixel a hazard: the caller re-makes it (kb_repair_pixels evaluates the suspect slices).
// Nothing else can go wrong: a pixel whose candidates all tie (a start pixel at the image's edge: every trajectory leaves
// over the same few samples) folds exactly -- L_0, and nothing after it is strictly above its last slot.
// state / recs: caller's scratch of K entries each; returns true for a hazard (bit r of *suspects: slice r; r < 64).
// ---------------------------------------------------------------------------------------------------------------
template <typename ReadRecord>
KB_HD bool merge_fold_pixel(const ReadRecord& read, int n_lists, int K, kb_compact_result* state, kb_compact_result* recs,
                            uint64_t* suspects) {
    *suspects = 0;
    for (int s = 0; s < K; ++s) {
        state[s] = read(0, s);
        if (state[s].cand < 0) state[s] = kb_compact_result{-FLT_MAX, 0.0f, -1, 0};  // kernels.cu:293-301
    }
    for (int r = 1; r < n_lists; ++r) {
        const float tail_before = state[K - 1].lh;
        const kb_compact_result last = read(r, K - 1);
        const bool full = last.cand >= 0;
        // the slice's records in candidate order (an insertion sort of K entries; placeholders last)
        int n = 0;
        for (int s = 0; s < K; ++s) {
            const kb_compact_result rec = read(r, s);
            if (rec.cand < 0) break;  // placeholders close a list
            if (!(rec.lh > tail_before)) continue;  // (cannot enter now, and the state's last likelihood never falls)
            int at = n;
            while (at > 0 && recs[at - 1].cand > rec.cand) {
                recs[at] = recs[at - 1];
                at -= 1;
            }
            recs[at] = rec;
            n += 1;
        }
        for (int i = 0; i < n; ++i) {
            kb_compact_result in = recs[i];
            for (int s = 0; s < K; ++s) {  // kernels.cu:323-330
                if (in.lh > state[s].lh) {
                    const kb_compact_result t = state[s];
                    state[s] = in;
                    in = t;
                    if (in.cand < 0) break;  // (a displaced placeholder moves nothing further)
                }
            }
        }
        if (full && last.lh > tail_before && last.lh == state[K - 1].lh && r < 64) *suspects |= 1ull << r;
    }
    return *suspects != 0ull;
}


}  // namespace kb
#endif
