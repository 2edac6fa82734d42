// Device-side pieces shared by the search kernels' translation units (search_direct.hip,
// search_lds.hip, search_lds_encoded.hip): finishing a chunk of candidates (threshold, top-K insertion
// or sigma-G work items), the epilogue, and the direct-load accumulation that the fallback kernel and
// the ring kernel's rare paths both use.
#ifndef KB_SEARCH_DEVICE_H_
#define KB_SEARCH_DEVICE_H_

#include <type_traits>

#include "search_common.h"

#pragma clang fp contract(off)

namespace kb {

// Tables are read through the constant address space: the DMA writes and barriers of
// the main loop would otherwise make the compiler fetch them with vector loads, whose
// vmcnt wait also waits for the slab DMA in flight.
typedef const __attribute__((address_space(4))) int* ConstIntPtr;
template <typename P>
__device__ __forceinline__ ConstIntPtr as_const_ints(const P* p) {
    return (ConstIntPtr)(uintptr_t)p;
}


// Screen of a finished candidate against the likelihood its list's last slot holds: true when the candidate CANNOT enter,
// decided without the thirty instructions of the correctly rounded sqrt and divide -- and, since round 5, without the
// quarter-rate v_rsq_f32 of the first form (psi * rsq(phi) < floor: twelve instruction slots a candidate, 40 % of a chunk's
// finish).  x -> x |x| is strictly increasing, so for phi > 0
//     psi / sqrt(phi) > F   <=>   psi |psi| > F |F| phi :
// two multiplies and ONE compare decide every combination of signs.  F is the threshold lowered by a relative 2^-18 and one
// smallest normal (screen_floor); the three roundings of the products move the comparison by less than 2^-22 relative and the
// exact likelihood is good to 2^-22, so a candidate that fails the screen also fails `lh > threshold` with the exact value.
// Everything the products cannot be trusted with is left to the exact test: phi <= 0 or NaN, and a right-hand side that is
// not a normal number (F |F| or F |F| phi underflowed, overflowed or is NaN -- thresholds below 1e-19 in magnitude, i.e. never
// in practice, merely lose their screen).  screen_key(threshold) = F |F|, once per lane and chunk.  The property is held over
// the whole float range by tests/test_screen_property.py (a numpy restatement, operation for operation).
__device__ __forceinline__ float screen_floor(float threshold) {
    return threshold - fabsf(threshold) * 3.814697265625e-06f - 1.17549435e-38f;  // -FLT_MAX -> -inf, NaN stays NaN
}
__device__ __forceinline__ float screen_key(float threshold) {
    float f = screen_floor(threshold);
    // A threshold too close to zero for its square -- min_lh = 0 under a list that is not full yet is the common one -- is
    // lowered to -2^-40: lower is always safe, the key stays a normal number, and the screen keeps rejecting what it used to
    // reject there: every candidate with a (really) negative likelihood.
    if (fabsf(f) < 0x1p-40f) f = -0x1p-40f;
    const float k = f * fabsf(f);
    // (a key that is not a normal number -- F |F| underflowed into the denormals, where it keeps a bit or two, or is infinite:
    // an empty slot -- becomes NaN: every product with it is then "not trusted")
    return __builtin_amdgcn_classf(k, 0x108) ? k : __builtin_nanf("");
}
__device__ __forceinline__ bool screened_out(float psi_sum, float phi_sum, float key) {
    // straight-line on purpose (bitwise, not short-circuit: as branches the screen of a chunk was ~360 instructions)
    const float s = psi_sum * fabsf(psi_sum);
    const float t = key * phi_sum;
    const bool trusted = (phi_sum > 0.0f) & __builtin_amdgcn_classf(t, 0x108);  // +-normal
    return trusted & (s <= t);                                                     // NaN compares false: left to the exact test
}

// The other side of the screen, for the emitting instances (a threshold that is the same for every candidate): a candidate
// the products put ABOVE the threshold raised by a relative 2^-18 and one smallest normal also passes the exact test
// `!(lh < threshold)`, so the correctly rounded sqrt and divide run only when some lane lies in the band between the two keys
// (or has sums the products cannot be trusted with).  Same argument, same guards, mirrored; tests/test_screen_property.py.
__device__ __forceinline__ float sure_key(float threshold) {
    float f = threshold + fabsf(threshold) * 3.814697265625e-06f + 1.17549435e-38f;  // FLT_MAX -> inf, NaN stays NaN
    if (fabsf(f) < 0x1p-40f) f = 0x1p-40f;  // (higher is always safe)
    const float k = f * fabsf(f);
    return __builtin_amdgcn_classf(k, 0x108) ? k : __builtin_nanf("");
}
__device__ __forceinline__ bool surely_in(float psi_sum, float phi_sum, float key_hi) {
    const float s = psi_sum * fabsf(psi_sum);
    const float t = key_hi * phi_sum;
    const bool trusted = (phi_sum > 0.0f) & __builtin_amdgcn_classf(t, 0x108);  // +-normal
    return trusted & (s > t);                                                     // NaN compares false: left to the exact test
}

// Threshold / insertion of one chunk's C finished candidates.  With the sigma-G filter on nothing is
// inserted here: the ballot of the lanes that pass the unclipped thresholds (kernels.cu:201-203 and
// :318-320; this includes the obs_count == 0 corner, which the clip leaves alone) becomes one work item
// per (row, candidate) for the resolve passes of sigmag_kernels.hip.
template <int KS, int C, bool SIGMAG>
__device__ __forceinline__ void finish_chunk(const SearchArgs& a, const TileCoords& tc, int chunk,
                                             const float (&ps)[C], const float (&ph)[C], const int (&cnt)[C],
                                             TopK<KS>& top) {
    // The candidates are finished strictly one after the other: each likelihood (a correctly rounded
    // sqrt and divide, a dozen temporaries) is tied by an empty asm to the state the previous candidate
    // left, so that the compiler cannot run the C of them side by side and charge the kernel's main
    // loop with their registers.
    if constexpr (SIGMAG) {
        const bool live = tc.x_i < a.sw;  // lanes past the right edge of the search area own no pixel
        // `lh < min_lh` is decided on the approximate likelihood wherever that is safe: a candidate screened_out rejects also
        // fails the exact test, one surely_in accepts also passes it, and the correctly rounded sqrt and divide run only when
        // some lane lies in the band between the two (or has sums the products cannot be trusted with).  Around a bright mover
        // most candidates of a wave pass in SOME lane: with the lower screen alone the exact likelihoods of all 1024 of them
        // made such a wave -- and with it its tile, and with one tile per CU the launch -- a fifth slower than the rest.
        // One pass: the ballot of candidate c is parked in lane c of a register pair (C 64-bit masks kept in scalar registers
        // would cost the surrounding loop its own), and the chunk's work items leave in ONE store by the lanes
        // that hold one (sixteen entries written one after the other by lane 0 were a tenth of such a wave's instructions).
        static_assert(C <= WAVE, "one lane per candidate of the chunk");
        const float floor_lh = screen_key(a.min_lh), ceil_lh = sure_key(a.min_lh);
        int need_lo = 0, need_hi = 0;  // lane c: the ballot of candidate c
        int n_items = 0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float p = ps[c], f = ph[c];
            asm volatile("" : "+v"(p), "+v"(f), "+v"(need_lo), "+v"(need_hi));
            const bool real = (chunk * C + c) < a.n_cands;  // uniform
            bool pass = real & live & !(cnt[c] < a.min_obs) & !screened_out(p, f, floor_lh);
            const bool sure = surely_in(p, f, ceil_lh);  // (straight-line: under a branch on the ballot of `pass` the launch was 2 % slower)
            if (__ballot(pass & !sure) != 0) {  // uniform
                const float lh = lh_from_sums(p, f);
                pass = pass & (sure | !(lh < a.min_lh));
            }
            const uint64_t b = __ballot(pass);
            need_lo = (tc.lane == c) ? (int)(uint32_t)b : need_lo;
            need_hi = (tc.lane == c) ? (int)(uint32_t)(b >> 32) : need_hi;
            n_items += (b != 0) ? 1 : 0;
        }
        if (n_items == 0) return;  // uniform
        const SigmaGWork& sg = a.cold->sg;
        uint32_t base = 0;
        if (tc.lane == 0) base = (uint32_t)atomicAdd(sg.n_entries, n_items);
        base = __builtin_amdgcn_readfirstlane(base);
        const uint32_t row = (uint32_t)(tc.y_i * a.tiles_x + tc.tx);
        uint32_t* slot_row = sg.slots + (size_t)row * sg.batch_cands + (chunk - a.chunk_lo) * C;
        const bool has = (need_lo | need_hi) != 0;  // (lanes from C on hold nothing)
        const uint64_t holders = __ballot(has);
        // entries in candidate order, like the items' numbers before: lane c's is behind those of the lanes below it
        const uint32_t at = base + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(holders >> 32),
                                                                       __builtin_amdgcn_mbcnt_lo((uint32_t)holders, 0u));
        if (has) {
            SgEntry e;
            e.row = row;
            e.cand = (uint32_t)(chunk * C) + (uint32_t)tc.lane;
            e.mask = ((uint64_t)(uint32_t)need_hi << 32) | (uint64_t)(uint32_t)need_lo;
            sg.entries[at] = e;
            slot_row[tc.lane] = at + 1;
        }
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int cand = chunk * C + c;
            if (cand >= a.n_cands) break;  // uniform
            float p = ps[c], f = ph[c];
            asm volatile("" : "+v"(p), "+v"(f), "+v"(top.lh[KS - 1]));
            const float lh = lh_from_sums(p, f);
            if (!(cnt[c] < a.min_obs) && lh > a.min_lh) top.insert(lh, cand, a.stable_lists != 0);  // (min_lh: the lists' floor, flag 1024)
        }
    }
}

// finish_chunk for kb_search_lds without the sigma-G filter.  The per-pixel lists are touched once per chunk of C
// candidates, i.e. once per C x T samples: kept in registers they would cost the summing loop 2 KS registers for
// nothing.  They live in a lane-interleaved store in HBM (L2-resident in practice) instead; the loop carries the
// likelihood to beat.  A lane none of whose candidates is above its threshold does not touch the store.
template <int KS, int C, bool RECORDS>
__device__ __forceinline__ void finish_chunk_stored(const SearchArgs& a, int chunk, float (&ps)[C], float (&ph)[C],
                                                    const int (&cnt)[C], ListState& ls, char* tile_list, uint32_t lane_off,
                                                    int stride_bytes) {
    // likelihoods one after the other (see finish_chunk); a candidate that may not enter (past the end of the
    // list, too few observations) becomes -FLT_MAX, which the insertion's strict '>' never admits
    bool beats = false;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float p = ps[c], f = ph[c];
        asm volatile("" : "+v"(p), "+v"(f), "+v"(ls.threshold));
        const bool real = (chunk * C + c) < a.n_cands;  // uniform
        float lh = (real && !(cnt[c] < a.min_obs)) ? lh_from_sums(p, f) : -FLT_MAX;
        lh = lh > a.min_lh ? lh : -FLT_MAX;  // (min_lh: the lists' floor, flag 1024; else -FLT_MAX)
        if constexpr (RECORDS) ph[c] = flux_from_sums(p, f);
        ps[c] = lh;
        beats = beats || (lh > ls.threshold);
    }
    // per lane: a lane none of whose candidates beats its threshold neither reads nor writes its list (the
    // memory operations below run under the exec mask of the lanes that do)
    if (!beats) return;
    if constexpr (RECORDS) {
        TopKRecords<KS> top;
        if (ls.stored) {
            top.load(tile_list, lane_off, stride_bytes);
        } else {
            top.init();
        }
#pragma unroll
        for (int c = 0; c < C; ++c) top.insert(ps[c], chunk * C + c, ph[c], cnt[c], a.stable_lists != 0);
        top.store(tile_list, lane_off, stride_bytes);
        ls.threshold = fmaxf(top.lh[KS - 1], a.min_lh);  // (never below the lists' floor, flag 1024)
    } else {
        TopK<KS> top;
        if (ls.stored) {
            top.load(tile_list, lane_off, stride_bytes);
        } else {
            top.init();
        }
#pragma unroll
        for (int c = 0; c < C; ++c) top.insert(ps[c], chunk * C + c, a.stable_lists != 0);
        top.store(tile_list, lane_off, stride_bytes);
        ls.threshold = fmaxf(top.lh[KS - 1], a.min_lh);
    }
    ls.stored = 1;
}

// The sums and count of candidate `sel` (per lane) out of the C register sets of a chunk, halving by the bits of `sel`: four
// bit tests and C - 1 selects per field instead of C - 1 compares and as many selects (sel >= C selects set sel % C).
template <int C>
__device__ __forceinline__ void select_by_bits(const float (&ps)[C], const float (&ph)[C], const int (&cnt)[C], uint32_t sel,
                                               float* p_out, float* f_out, int* n_out) {
    float p[C], f[C];
    int n[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        p[c] = ps[c];
        f[c] = ph[c];
        n[c] = cnt[c];
    }
#pragma unroll
    for (int w = C / 2, bit = 1; w >= 1; w >>= 1, bit <<= 1) {
        const bool odd = (sel & (uint32_t)bit) != 0u;
#pragma unroll
        for (int i = 0; i < w; ++i) {
            p[i] = odd ? p[2 * i + 1] : p[2 * i];
            f[i] = odd ? f[2 * i + 1] : f[2 * i];
            n[i] = odd ? n[2 * i + 1] : n[2 * i];
        }
    }
    *p_out = p[0];
    *f_out = f[0];
    *n_out = n[0];
}

// finish_chunk / epilogue for packed records in registers (kb_search_lds, lists of up to 8, long candidate lists).
// Two steps.  (1) Screen all C candidates (above): per lane a bit mask of those that may enter its list.  (2) While
// any lane has a bit left, EVERY lane takes its own lowest candidate -- selected out of the C register sets --,
// computes the exact likelihood and runs the reference's swap-down insertion (kernels.cu:318-330).  Each lane
// still sees its candidates in list order, but the wave walks max-over-lanes(#passing) rounds per chunk (one or
// two once the lists have filled) instead of one insertion round per candidate that passes in ANY lane (most of
// the C): the insertion and the exact likelihoods were 0.68 of cfg2's 4.62 ms.
template <int KS, int C>
__device__ __forceinline__ void finish_chunk_packed(const SearchArgs& a, int cand_base, const float (&ps)[C], const float (&ph)[C],
                                                    const int (&cnt)[C], TopKPacked<KS>& top) {
    static_assert(C <= 16, "chunks of 32 have their own finish (finish_chunk32_packed, search_lds.h): the compiler's selection "
                           "out of 32 register sets next to 64 live sums put the sums into scratch memory");
    const float floor_lh = screen_key(fmaxf(top.lh[KS - 1], a.min_lh));  // (min_lh: the lists' floor, flag 1024; else -FLT_MAX)
    uint32_t pending = 0;
#pragma unroll
    for (int c = C - 1; c >= 0; --c) {  // (downwards: the mask is built by shifting)
        const bool out = (cnt[c] < a.min_obs) | screened_out(ps[c], ph[c], floor_lh);
        pending = (pending << 1) | (out ? 0u : 1u);
    }
    {
        const int left = a.n_cands - cand_base;  // uniform: candidates of this (half) chunk that exist
        if (left < C) pending &= left > 0 ? (1u << left) - 1u : 0u;
    }
    while (__ballot(pending != 0u) != 0ull) {  // uniform
        const int c_sel = (int)__builtin_ctz(pending | (1u << C));  // (C: a lane with nothing left selects nothing)
        float p, f;
        int n;
        select_by_bits<C>(ps, ph, cnt, (uint32_t)c_sel, &p, &f, &n);
        const float lh = lh_from_sums(p, f);
        if (pending != 0u && lh > top.lh[KS - 1]) {
            top.insert(lh, flux_from_sums(p, f), (uint32_t)(cand_base + c_sel) | ((uint32_t)n << 16), a.stable_lists != 0);
        }
        pending &= pending - 1u;
    }
}
// The K result records of every lane of a wave -- 64 consecutive start pixels of one row, hence one contiguous run of
// 64 * K records in the result array -- leave as coalesced stores: each half of the wave lays its records down in the wave's
// own patch of LDS (the group buffers are dead by the epilogue), then all 64 lanes write the patch out linearly, 256
// contiguous bytes per store instruction.  The per-lane form stores 7 (or 4) dwords per slot at a lane stride of K records:
// 3.76 GB of results of a 4096 x 4096 search took 1.5 ms of a 20 ms kernel that way (KB_EXP_NO_RESULTS), against 0.85 ms
// for a plain fill of as many bytes.  R = dwords per record (7: kb_trajectory, 4: kb_compact_result); rec(s, w) fills
// record s of the calling lane; `patch`: at least 32 * (R * K + 1) dwords of LDS owned by this wave.
template <int R, typename MakeRecord>
__device__ __forceinline__ void store_wave_records(uint32_t* run /* record 0 of the wave's lane 0 */, int K, bool live, char* patch,
                                                   const MakeRecord& rec, uint8_t* counts_run = nullptr, int kept = 0) {
    typedef __attribute__((address_space(3))) uint32_t* LdsWords;
    const LdsWords lds = (LdsWords)(uint32_t)(uintptr_t)patch;
    const int lane = threadIdx.x & (WAVE - 1);
    if (counts_run != nullptr) {  // (uniform) ResultSink::counts: the count byte of every pixel; nothing else for a wave that keeps nothing
        if (live) counts_run[lane] = (uint8_t)kept;
        if (__ballot(live && kept != 0) == 0ull) return;
    }
    const uint32_t rk = (uint32_t)(R * K), stride = rk | 1u;           // (odd pitch: the 32 lanes of a half hit 32 different banks)
    const uint32_t inv = ((1u << 20) + rk - 1u) / rk;                  // i / rk == (i * inv) >> 20 for i < 32 * rk, rk <= 188
    const int n_live = __popcll(__ballot(live));                       // (live lanes are a prefix of the wave)
    for (int h = 0; h < 2; ++h) {
        if (live && (lane >> 5) == h) {
            for (int s = 0; s < K; ++s) {
                uint32_t w[R];
                rec(s, w);
#pragma unroll
                for (int j = 0; j < R; ++j) lds[(uint32_t)(lane & 31) * stride + (uint32_t)(s * R + j)] = w[j];
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        const uint32_t total = (uint32_t)max(0, min(32, n_live - 32 * h)) * rk;
        uint32_t* dst = run + (size_t)(32 * h) * rk;
        for (uint32_t i = (uint32_t)lane; i < total; i += WAVE) {
            const uint32_t src = (i * inv) >> 20;
            dst[i] = lds[src * stride + (i - src * rk)];
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
}

// Stable lists of 16 in the list store of kb_search_lds, pooled (LIST_STORE_POOLED): what the tie-exact exchange between
// devices runs with chunks of WIDE_CHUNK candidates (2 K = 16 records per pixel, flag 512).  A thread's list is
//   * 16 likelihoods in list order (four 16-byte rows of the lane-interleaved store),
//   * a 64-bit word of 16 four-bit cell numbers in list order (slot s in bits [4 s, 4 s + 4)), and
//   * a pool of 16 cells of (flux, candidate | count << 16) that never move: a new entry takes the cell of the entry that
//     falls off the end (cells are numbered by the slots that first used them).
// In registers the list is 18 words instead of the 48 of packed records -- next to 32 sums, 8 counts and the state of the
// summing loop, 48 put 150 registers into scratch memory whose round trips made the finish 130 us per chunk --, and an
// insertion moves one word per slot instead of three.  Stable insertion only (TopK::insert with `stable`): the new entry
// goes behind every entry that is not smaller and everything below shifts by one, i.e. the list is the top 16 by (likelihood
// descending, candidate ascending); the reference's swap-down, which rotates runs of equal likelihoods, stays with the
// register and record lists.  The finish is the one of the packed register lists: screen, then rounds in which every lane
// inserts its own lowest passing candidate; only lanes with a candidate past the screen touch the store.
struct PooledLayout {
    // byte offsets inside a tile's block of the store (threads = ROWS * WAVE); 16 * 16 bytes per thread are allocated
    uint32_t threads;
    __device__ __forceinline__ uint32_t lh_row(int r, uint32_t tid) const { return (uint32_t)r * threads * 16u + tid * 16u; }   // r < 4
    __device__ __forceinline__ uint32_t cells(uint32_t tid) const { return 4u * threads * 16u + tid * 16u; }
    __device__ __forceinline__ uint32_t pool(uint32_t cell, uint32_t tid) const { return 5u * threads * 16u + cell * threads * 8u + tid * 8u; }
};
struct PooledList {
    float lh[16];
    uint64_t cells;
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int s = 0; s < 16; ++s) lh[s] = -FLT_MAX;  // (an entry's likelihood is above this: -FLT_MAX marks an empty slot)
        cells = 0xfedcba9876543210ull;
    }
    __device__ __forceinline__ void load(const char* tile_list, const PooledLayout& lay, uint32_t tid) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint4 v = *reinterpret_cast<const uint4*>(tile_list + lay.lh_row(r, tid));
            lh[4 * r] = __uint_as_float(v.x);
            lh[4 * r + 1] = __uint_as_float(v.y);
            lh[4 * r + 2] = __uint_as_float(v.z);
            lh[4 * r + 3] = __uint_as_float(v.w);
        }
        const uint2 c = *reinterpret_cast<const uint2*>(tile_list + lay.cells(tid));
        cells = ((uint64_t)c.y << 32) | (uint64_t)c.x;
    }
    __device__ __forceinline__ void store(char* tile_list, const PooledLayout& lay, uint32_t tid) const {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            *reinterpret_cast<uint4*>(tile_list + lay.lh_row(r, tid)) =
                    make_uint4(__float_as_uint(lh[4 * r]), __float_as_uint(lh[4 * r + 1]), __float_as_uint(lh[4 * r + 2]),
                               __float_as_uint(lh[4 * r + 3]));
        }
        *reinterpret_cast<uint2*>(tile_list + lay.cells(tid)) = make_uint2((uint32_t)cells, (uint32_t)(cells >> 32));
    }
    // Stable insertion of a candidate known to beat the last slot; returns the pool cell its record goes to.
    __device__ __forceinline__ uint32_t insert(float cand_lh) {
        const uint32_t cell = (uint32_t)(cells >> 60);  // the cell of the entry that falls off
        // g[s] = the candidate is above slot s: false ... false, true ... true down the (descending) list
        uint32_t pos = 0;  // slots that stay in front of the candidate
        float prev = lh[0];
        bool g_prev = cand_lh > prev;
        lh[0] = g_prev ? cand_lh : prev;
        pos += g_prev ? 0u : 1u;
#pragma unroll
        for (int s = 1; s < 16; ++s) {
            const float cur = lh[s];
            const bool g = cand_lh > cur;
            lh[s] = g ? (g_prev ? prev : cand_lh) : cur;
            pos += g ? 0u : 1u;
            prev = cur;
            g_prev = g;
        }
        const uint32_t sh = 4u * pos;                      // pos <= 15: the caller tested the last slot
        const uint64_t below = cells & ((1ull << sh) - 1ull);
        const uint64_t above = (cells >> sh) << sh << 4;  // the top nibble leaves the word
        cells = below | ((uint64_t)cell << sh) | above;
        return cell;
    }
};
template <int C, bool FAST>
__device__ __forceinline__ void finish_chunk_pooled(const SearchArgs& a, int chunk, const float (&ps)[C], const float (&ph)[C],
                                                    const uint32_t (&cntp)[C / 2], ListState& ls, char* tile_list,
                                                    const PooledLayout lay, uint32_t tid) {
    const float floor_lh = screen_key(ls.threshold);
    uint32_t pending = 0;
#pragma unroll
    for (int c = C - 1; c >= 0; --c) {
        const int n = FAST ? a.T : (int)((cntp[c >> 1] >> (16 * (c & 1))) & 0xffffu);
        const bool out = (n < a.min_obs) | screened_out(ps[c], ph[c], floor_lh);
        pending = (pending << 1) | (out ? 0u : 1u);
    }
    {
        const int left = a.n_cands - chunk * C;  // uniform
        if (left < C) pending &= (1u << left) - 1u;
    }
    if (__ballot(pending != 0u) == 0ull) return;  // uniform: nothing of this chunk can enter any list of the wave
    PooledList top;
    top.init();
    const bool mine = pending != 0u;  // per lane: the memory operations below run under the mask of these lanes
    if (mine && ls.stored) top.load(tile_list, lay, tid);
    while (__ballot(pending != 0u) != 0ull) {  // uniform
        const int c_sel = (int)__builtin_ctz(pending | (1u << C));
        float p = ps[0], f = ph[0];
#pragma unroll
        for (int c = 1; c < C; ++c) {
            const bool pick = c_sel == c;
            p = pick ? ps[c] : p;
            f = pick ? ph[c] : f;
        }
        uint32_t n = (uint32_t)a.T;
        if constexpr (!FAST) {
            uint32_t w = cntp[0];
#pragma unroll
            for (int j = 1; j < C / 2; ++j) w = ((c_sel >> 1) == j) ? cntp[j] : w;
            n = (w >> (16 * (c_sel & 1))) & 0xffffu;
        }
        const float lh = lh_from_sums(p, f);
        if (pending != 0u && lh > top.lh[15]) {
            const uint32_t cell = top.insert(lh);
            *reinterpret_cast<uint2*>(tile_list + lay.pool(cell, tid)) =
                    make_uint2(__float_as_uint(flux_from_sums(p, f)), (uint32_t)(chunk * C + c_sel) | (n << 16));
        }
        pending &= pending - 1u;
    }
    if (mine) {
        top.store(tile_list, lay, tid);
        ls.threshold = fmaxf(top.lh[15], a.min_lh);  // (never below the lists' floor, flag 1024)
        ls.stored = 1;
    }
}
__device__ __forceinline__ void write_results_pooled(const SearchArgs& a, const TileCoords& tc, const ListState& ls,
                                                     const char* tile_list, const PooledLayout lay, uint32_t tid,
                                                     char* wave_patch = nullptr) {
    if (wave_patch != nullptr && a.K * 7 <= 128) {  // (uniform) the coalesced form, see store_wave_records
        if (!tc.row_active) return;
        const bool live = tc.x_i < a.sw;
        const ResultSink sink = a.cold->results;
        const kb_trajectory* cands = a.cold->cands;
        const size_t run0 = ((size_t)tc.y_i * a.sw + (size_t)(tc.tx * WAVE)) * a.K;
        uint64_t cells = 0;
        if (live && ls.stored) {
            const uint2 c = *reinterpret_cast<const uint2*>(tile_list + lay.cells(tid));
            cells = ((uint64_t)c.y << 32) | (uint64_t)c.x;
        }
        // (lh, flux, candidate, count) of slot s of this lane's list; candidate < 0: an empty slot
        uint8_t* counts_run = sink.counts != nullptr ? sink.counts + ((size_t)tc.y_i * a.sw + (size_t)(tc.tx * WAVE)) : nullptr;
        int kept = 0;
        if (counts_run != nullptr && live && ls.stored) {
            for (int r = 0; r < 4; ++r) {  // (the list's likelihoods, in list order: -FLT_MAX marks an empty slot)
                const uint4 v = *reinterpret_cast<const uint4*>(tile_list + lay.lh_row(r, tid));
                const float l4[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
                for (int j = 0; j < 4; ++j) kept += (4 * r + j < a.K && l4[j] != -FLT_MAX && !(l4[j] < sink.keep_min_lh)) ? 1 : 0;
            }
        }
        auto slot = [&](int s, float* lh, float* flux, int* id, uint32_t* obs) {
            *lh = -FLT_MAX;
            *flux = 0.0f;
            *id = -1;
            *obs = 0u;
            if (ls.stored) {
                const float v = *reinterpret_cast<const float*>(tile_list + lay.lh_row(s >> 2, tid) + 4 * (s & 3));
                if (v != -FLT_MAX) {
                    const uint32_t cell = (uint32_t)(cells >> (4 * s)) & 15u;
                    const uint2 r = *reinterpret_cast<const uint2*>(tile_list + lay.pool(cell, tid));
                    *lh = v;
                    *flux = __uint_as_float(r.x);
                    *id = (int)(r.y & 0xffffu);
                    *obs = r.y >> 16;
                }
            }
        };
        if (sink.compact != nullptr) {
            store_wave_records<4>(reinterpret_cast<uint32_t*>(sink.compact + run0), a.K, live, wave_patch, [&](int s, uint32_t (&w)[4]) {
                float lh, flux;
                int id;
                uint32_t obs;
                slot(s, &lh, &flux, &id, &obs);
                w[0] = __float_as_uint(lh);
                w[1] = __float_as_uint(flux);
                w[2] = id < 0 ? 0xffffffffu : (uint32_t)(sink.cand_base + id);
                w[3] = obs;
            }, counts_run, kept);
        } else {
            store_wave_records<7>(reinterpret_cast<uint32_t*>(sink.full + run0), a.K, live, wave_patch, [&](int s, uint32_t (&w)[7]) {
                float lh, flux;
                int id;
                uint32_t obs;
                slot(s, &lh, &flux, &id, &obs);
                w[0] = id < 0 ? 0u : __float_as_uint(cands[id].vx);
                w[1] = id < 0 ? 0u : __float_as_uint(cands[id].vy);
                w[2] = __float_as_uint(lh);
                w[3] = __float_as_uint(flux);
                w[4] = (uint32_t)tc.x;
                w[5] = (uint32_t)tc.y;
                w[6] = obs;
            }, counts_run, kept);
        }
        return;
    }
    if (tc.x_i >= a.sw || !tc.row_active) return;
    const size_t slot0 = ((size_t)tc.y_i * a.sw + tc.x_i) * a.K;
    uint64_t cells = 0;
    if (ls.stored) {
        const uint2 c = *reinterpret_cast<const uint2*>(tile_list + lay.cells(tid));
        cells = ((uint64_t)c.y << 32) | (uint64_t)c.x;
    }
    for (int s = 0; s < a.K; ++s) {  // (K <= 16)
        kb_trajectory res = placeholder_result(tc.x, tc.y);  // kernels.cu:293-301
        int id_s = -1;
        if (ls.stored) {
            const float lh = *reinterpret_cast<const float*>(tile_list + lay.lh_row(s >> 2, tid) + 4 * (s & 3));
            if (lh != -FLT_MAX) {
                const uint32_t cell = (uint32_t)(cells >> (4 * s)) & 15u;
                const uint2 r = *reinterpret_cast<const uint2*>(tile_list + lay.pool(cell, tid));
                id_s = (int)(r.y & 0xffffu);
                res.vx = a.cold->cands[id_s].vx;
                res.vy = a.cold->cands[id_s].vy;
                res.lh = lh;
                res.flux = __uint_as_float(r.x);
                res.obs_count = (int)(r.y >> 16);
            }
        }
        store_result(a.cold->results, slot0 + s, res, id_s);
    }
}

template <int KS>
__device__ __forceinline__ void write_packed(const SearchArgs& a, const TileCoords& tc, const TopKPacked<KS>& top,
                                             char* wave_patch = nullptr) {
    if (wave_patch != nullptr) {  // (uniform) the coalesced form
        if (!tc.row_active) return;
        const bool live = tc.x_i < a.sw;
        const ResultSink sink = a.cold->results;
        const kb_trajectory* cands = a.cold->cands;
        const size_t run0 = ((size_t)tc.y_i * a.sw + (size_t)(tc.tx * WAVE)) * a.K;   // the slot of lane 0's first record
        uint8_t* counts_run = sink.counts != nullptr ? sink.counts + ((size_t)tc.y_i * a.sw + (size_t)(tc.tx * WAVE)) : nullptr;
        int kept = 0;
        if (counts_run != nullptr) {
#pragma unroll
            for (int k = 0; k < KS; ++k) kept += (k < a.K && top.io[k] != TopKPacked<KS>::EMPTY && !(top.lh[k] < sink.keep_min_lh)) ? 1 : 0;
        }
        if (sink.compact != nullptr) {
            store_wave_records<4>(reinterpret_cast<uint32_t*>(sink.compact + run0), a.K, live, wave_patch, [&](int s, uint32_t (&w)[4]) {
                uint32_t io = TopKPacked<KS>::EMPTY;
                float lh = -FLT_MAX, flux = 0.0f;
#pragma unroll
                for (int k = 0; k < KS; ++k) {
                    if (k == s) {
                        io = top.io[k];
                        lh = top.lh[k];
                        flux = top.flux[k];
                    }
                }
                const bool empty = io == TopKPacked<KS>::EMPTY;
                w[0] = __float_as_uint(empty ? -FLT_MAX : lh);
                w[1] = __float_as_uint(empty ? 0.0f : flux);
                w[2] = empty ? 0xffffffffu : (uint32_t)(sink.cand_base + (int)(io & 0xffffu));
                w[3] = empty ? 0u : (io >> 16);
            }, counts_run, kept);
        } else {
            store_wave_records<7>(reinterpret_cast<uint32_t*>(sink.full + run0), a.K, live, wave_patch, [&](int s, uint32_t (&w)[7]) {
                uint32_t io = TopKPacked<KS>::EMPTY;
                float lh = -FLT_MAX, flux = 0.0f;
#pragma unroll
                for (int k = 0; k < KS; ++k) {
                    if (k == s) {
                        io = top.io[k];
                        lh = top.lh[k];
                        flux = top.flux[k];
                    }
                }
                const bool empty = io == TopKPacked<KS>::EMPTY;
                const int id = empty ? 0 : (int)(io & 0xffffu);
                w[0] = empty ? 0u : __float_as_uint(cands[id].vx);   // kb_trajectory { vx, vy, lh, flux, x, y, obs_count }; kernels.cu:293-301
                w[1] = empty ? 0u : __float_as_uint(cands[id].vy);
                w[2] = __float_as_uint(empty ? -FLT_MAX : lh);
                w[3] = __float_as_uint(empty ? 0.0f : flux);
                w[4] = (uint32_t)tc.x;
                w[5] = (uint32_t)tc.y;
                w[6] = empty ? 0u : (io >> 16);
            }, counts_run, kept);
        }
        return;
    }
    if (tc.x_i >= a.sw || !tc.row_active) return;
    const size_t slot0 = ((size_t)tc.y_i * a.sw + tc.x_i) * a.K;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (s < a.K) {  // uniform
            kb_trajectory res = placeholder_result(tc.x, tc.y);  // kernels.cu:293-301
            const uint32_t io = top.io[s];
            const int id_s = (io == TopKPacked<KS>::EMPTY) ? -1 : (int)(io & 0xffffu);
            if (id_s >= 0) {
                res.vx = a.cold->cands[id_s].vx;
                res.vy = a.cold->cands[id_s].vy;
                res.lh = top.lh[s];
                res.flux = top.flux[s];
                res.obs_count = (int)(io >> 16);
            }
            store_result(a.cold->results, slot0 + s, res, id_s);
        }
    }
}

// finish_chunk with lists of whole result records in registers (kb_search_direct, lists of up to 16): flux and
// observation count travel with the likelihood, the epilogue copies them out (write_records).
template <int KS, int C>
__device__ __forceinline__ void finish_chunk_records(const SearchArgs& a, int chunk, const float (&ps)[C], const float (&ph)[C],
                                                     const int (&cnt)[C], TopKRecords<KS>& top) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int cand = chunk * C + c;
        if (cand >= a.n_cands) break;  // uniform
        float p = ps[c], f = ph[c];
        asm volatile("" : "+v"(p), "+v"(f), "+v"(top.lh[KS - 1]));  // one candidate after the other (see finish_chunk)
        const float lh = lh_from_sums(p, f);
        if (!(cnt[c] < a.min_obs) && lh > top.lh[KS - 1] && lh > a.min_lh) top.insert(lh, cand, flux_from_sums(p, f), cnt[c], a.stable_lists != 0);
    }
}
template <int KS>
__device__ __forceinline__ void write_records(const SearchArgs& a, const TileCoords& tc, const TopKRecords<KS>& top) {
    if (tc.x_i >= a.sw || !tc.row_active) return;
    const size_t slot0 = ((size_t)tc.y_i * a.sw + tc.x_i) * a.K;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (s < a.K) {  // uniform
            kb_trajectory res = placeholder_result(tc.x, tc.y);  // kernels.cu:293-301
            const int id_s = top.id[s];
            if (id_s >= 0) {
                res.vx = a.cold->cands[id_s].vx;
                res.vy = a.cold->cands[id_s].vy;
                res.lh = top.lh[s];
                res.flux = top.flux[s];
                res.obs_count = top.obs[s];
            }
            store_result(a.cold->results, slot0 + s, res, id_s);
        }
    }
}

// Epilogue (no sigma-G; with it kb_sigmag_select_kernel writes the results): the K winners are
// re-evaluated with exact per-lane positions to produce flux / obs_count; the likelihood this yields
// is bit-identical to the one that won the slot.
template <int KS>
__device__ __forceinline__ void write_results(const SearchArgs& a, const TileCoords& tc, const TopK<KS>& top) {
    if (tc.x_i >= a.sw || !tc.row_active) return;
    const size_t slot0 = ((size_t)tc.y_i * a.sw + tc.x_i) * a.K;
    for (int s = 0; s < a.K; ++s) {
        int id_s = -1;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            if (k == s) id_s = top.id[k];
        }
        kb_trajectory res = placeholder_result(tc.x, tc.y);  // kernels.cu:293-301
        if (id_s >= 0) {
            res.vx = a.cold->cands[id_s].vx;
            res.vy = a.cold->cands[id_s].vy;
            evaluate_trajectory_full<WAVE>(a.cold->meta, a.psi_phi, a.cold->times, a.cold->params, &res,
                                           static_cast<const SigmaGScratch<WAVE>*>(nullptr));
        }
        store_result(a.cold->results, slot0 + s, res, id_s);
    }
}


// The epilogue of kb_search_lds: with lists of records the results are copied out of the store; with (likelihood,
// candidate) lists the winners' candidate indices come out of it slot by slot and each winner is re-evaluated
// (nothing of the list is held across that, the register-hungriest code of the kernel).
template <bool RECORDS>
__device__ __forceinline__ void write_results_stored(const SearchArgs& a, const TileCoords& tc, const ListState& ls,
                                                     const char* tile_list, uint32_t lane_off, int stride_bytes) {
    if (tc.x_i >= a.sw || !tc.row_active) return;
    const size_t slot0 = ((size_t)tc.y_i * a.sw + tc.x_i) * a.K;
    for (int s = 0; s < a.K; ++s) {
        const int id_s = ls.stored ? (int)reinterpret_cast<const uint2*>(tile_list + (size_t)s * stride_bytes + lane_off)->y : -1;
        kb_trajectory res = placeholder_result(tc.x, tc.y);  // kernels.cu:293-301
        if constexpr (RECORDS) {
            if (id_s >= 0) {
                const uint4 r = *reinterpret_cast<const uint4*>(tile_list + (size_t)s * stride_bytes + lane_off);
                res.vx = a.cold->cands[id_s].vx;
                res.vy = a.cold->cands[id_s].vy;
                res.lh = __uint_as_float(r.x);
                res.flux = __uint_as_float(r.z);
                res.obs_count = (int)r.w;
            }
        } else if (id_s >= 0) {
            res.vx = a.cold->cands[id_s].vx;
            res.vy = a.cold->cands[id_s].vy;
            evaluate_trajectory_full<WAVE>(a.cold->meta, a.psi_phi, a.cold->times, a.cold->params, &res,
                                           static_cast<const SigmaGScratch<WAVE>*>(nullptr));
        }
        store_result(a.cold->results, slot0 + s, res, id_s);
    }
}

// MODE 0: interior wave, table shifts, no per-lane bounds test.
// MODE 1: table shifts with per-lane bounds test (image edges / off-image starts).
// MODE 2: exact per-lane double positions (chunks with unproven shifts, or forced).
template <int C, int C0, int HC, int NB, int MODE>
__device__ __forceinline__ void accumulate_chunk_direct(const SearchArgs& a, int chunk, int x, int y, int pix0,
                                                        float (&ps)[C], float (&ph)[C], int (&cnt)[C]) {
    using R = RawPair<NB>;
    constexpr int BYTES = 2 * fmt_bytes(NB);
    const int2* __restrict__ tab = a.table + (size_t)chunk * a.T * C + C0;
    const uint64_t image_bytes = ((uint64_t)a.W * (uint64_t)a.H) * (uint64_t)BYTES;
    const char* base = reinterpret_cast<const char*>(a.psi_phi);
#pragma unroll 2
    for (int t = 0; t < a.T; ++t) {
        // Phase 1: all HC loads of this epoch are issued before anything consumes them.
        typename R::type raw[HC];
        bool ok[HC];
        if constexpr (MODE == 2) {
            const double tm = a.cold->times[t];
#pragma unroll
            for (int c = 0; c < HC; ++c) {
                const int ci = min(chunk * C + C0 + c, a.n_cands - 1);
                int cx, cy;
                bool in = predict_index(x, a.cold->cands[ci].vx, tm, &cx);
                in = predict_index(y, a.cold->cands[ci].vy, tm, &cy) && in;
                ok[c] = in && ((unsigned)cx < (unsigned)a.W) && ((unsigned)cy < (unsigned)a.H);
                const uint32_t voff = ok[c] ? (uint32_t)(cy * a.W + cx) * (uint32_t)BYTES : 0u;
                raw[c] = *reinterpret_cast<const typename R::type*>(base + voff);
            }
        } else {
#pragma unroll
            for (int c = 0; c < HC; ++c) {
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
        for (int c = 0; c < HC; ++c) {
            float psi, phi;
            R::decode(raw[c], a, &psi, &phi);
            accumulate(psi, phi, ok[c], ps[C0 + c], ph[C0 + c], cnt[C0 + c]);
        }
        base += image_bytes;
    }
}

// All C candidates of a chunk, eight at a time.
template <int C, int NB, int MODE>
__device__ __forceinline__ void accumulate_chunk_direct_all(const SearchArgs& a, int chunk, int x, int y, int pix0,
                                                            float (&ps)[C], float (&ph)[C], int (&cnt)[C]) {
    static_assert(C == 8 || C == 16, "chunk size");
    accumulate_chunk_direct<C, 0, 8, NB, MODE>(a, chunk, x, y, pix0, ps, ph, cnt);
    if constexpr (C == 16) accumulate_chunk_direct<C, 8, 8, NB, MODE>(a, chunk, x, y, pix0, ps, ph, cnt);
}

}  // namespace kb
#endif
