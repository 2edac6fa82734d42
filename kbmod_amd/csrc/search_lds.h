// kb_search_lds: the LDS-staged search kernel of libkbmod_hip.so.  Instantiated by search_lds.hip
// (canonical float staging) and search_lds_encoded.hip (encoded staging); see search_kernels.hip for
// the overall design.
#ifndef KB_SEARCH_LDS_H_
#define KB_SEARCH_LDS_H_

#include <cstdio>
#include <type_traits>

#include "search_device.h"
#ifdef KB_ASM_HEADER  // (timing experiments: a variant of the generated statements, tools/gen_lds_loop.py)
#include KB_ASM_HEADER
#else
#include "search_lds_asm.h"
#endif

#pragma clang fp contract(off)

namespace kb {

#ifdef KB_EXP_PROFILE
// timing experiment: shader-clock ticks per phase, summed over waves (read by kb_exp_read_profile)
__device__ unsigned long long kb_exp_prof[8];
#define KB_PROF_MARK(slot)                                   \
    {                                                        \
        const uint64_t now_ = __builtin_amdgcn_s_memtime();  \
        prof_acc[slot] += now_ - prof_t;                     \
        prof_t = now_;                                       \
    }
#else
#define KB_PROF_MARK(slot)
#endif

// Staging map.  A slab (rows x cols raw pairs, dense; cols = the chunk's pitch) is copied in workgroup-wide
// rounds of 16 * ROWS * 64 bytes: in round j thread tid moves the 16 bytes at slab offset
// o = 16 * (tid + ROWS * 64 * j), i.e. pixel p = o / BYTES = (row, col) = divmod(p, cols) of the slab, from the
// padded array at the slab origin plus (row * Wp + col) * BYTES (stage_lanes).  The hand-scheduled STREAM statements
// (search_lds_asm.h, KB_LDS_DMA) request a wave's 1 KiB piece of that map by LDS-DMA -- global_load_lds_dwordx4 with M0 = the
// piece's place in LDS: no staging registers, no ds_write_b128 --; the LOOP statements and every compiler-scheduled path
// below (first group of a launch, encoded staging, slabs of more than LDS_SLOTS rounds) copy through registers
// (global_load_dwordx4 -> ds_write_b128: load_slab / write_slab).
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

// Staging map of a thread for the slabs of one chunk (pitch `cols` pixels, slab_bytes): where its 16 bytes of
// round j sit in the padded copy relative to the slab origin; 0 (re-read the slab's first bytes) past the slab's end.
template <int BYTES, int ROWS>
__device__ __forceinline__ StageLane stage_lanes(const SearchArgs& a, int cols, int slab_bytes, uint32_t inv) {
    StageLane out;
    // p / cols by one multiplication: inv = ceil(2^20 / cols) (ChunkInfo::cols_inv) is exact for p < 2^20 / cols (p < 16 K
    // pixels of two staging rounds, cols <= 112) -- the division itself cost every thread ~25 instructions per slot and chunk
    static_assert(16 * (ROWS * WAVE) * LDS_SLOTS / BYTES <= (1 << 20) / 128, "the reciprocal's range");
#pragma unroll
    for (int j = 0; j < LDS_SLOTS; ++j) {
        const int o = 16 * ((int)threadIdx.x + (ROWS * WAVE) * j);
        const int p = o / BYTES;  // first pixel of this thread's 16 bytes
        const int r = (int)(((uint32_t)p * inv) >> 20), c = p - r * cols;
        out.goff[j] = (o < slab_bytes) ? (uint32_t)(r * a.Wp + c) * (uint32_t)BYTES : 0u;
    }
    return out;
}

// Issue the loads of one epoch's slab; `base` = its origin in the padded copy (uniform).  All LDS_SLOTS
// loads are issued whatever the slab size (no branch, no exec mask -- the compiler would serialise
// masked loads with vmcnt(0)): threads past the end of the slab re-read its first bytes and do not
// write them to LDS.
template <int BYTES, int ROWS>
__device__ __forceinline__ void load_slab(const SearchArgs& a, const StageLane& sl, const char* base, int slab_bytes,
                                          SlabRegs& regs, int j0 = 0, int cols = LDS_COLS) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < LDS_SLOTS; ++j) {
        uint32_t goff = sl.goff[j];
        if (j0 != 0) {  // uniform, rare: rounds after the first compute their map on the fly
            const int p = 16 * (tid + (ROWS * WAVE) * (j0 + j)) / BYTES;
            const int r = p / cols, c = p - r * cols;
            goff = (uint32_t)(r * a.Wp + c) * (uint32_t)BYTES;
        }
        // (round 0: sl already holds 0 for threads past the end of this chunk's slabs, see stage_lanes)
        uint32_t off = (j0 == 0 || 16 * (tid + (ROWS * WAVE) * (j0 + j)) < slab_bytes) ? goff : 0u;
        // the offset stays a 32-bit register across the loop (uniform base + 32-bit lane offset is an addressing mode;
        // hoisted as a 64-bit value it costs an add per load and a register more)
        asm volatile("" : "+v"(off));
        // only BYTES-aligned: the hardware takes unaligned 16-byte global loads
        const PieceMem<(BYTES < 4 ? BYTES : 4)>* src = reinterpret_cast<const PieceMem<(BYTES < 4 ? BYTES : 4)>*>(base + off);
        regs.v[j] = Piece{src->w[0], src->w[1], src->w[2], src->w[3]};
    }
}

template <int ROWS>
__device__ __forceinline__ void write_slab(char* dst, int slab_bytes, const SlabRegs& regs, int j0 = 0) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < LDS_SLOTS; ++j) {
        if (stage_round(ROWS) * (j0 + j) < slab_bytes) {  // uniform
            const int o = 16 * (tid + (ROWS * WAVE) * (j0 + j));
            if (o < slab_bytes) *reinterpret_cast<Piece*>(dst + o) = regs.v[j];
        }
    }
}

// Rounds after the first LDS_SLOTS of a larger slab (load, then write, no overlap).
template <int BYTES, int ROWS>
__device__ __forceinline__ void copy_slab_tail(const SearchArgs& a, const StageLane& sl, const char* base, int slab_bytes,
                                               int cols, char* dst, SlabRegs& regs) {
    for (int j0 = LDS_SLOTS; stage_round(ROWS) * j0 < slab_bytes; j0 += LDS_SLOTS) {
        load_slab<BYTES, ROWS>(a, sl, base, slab_bytes, regs, j0, cols);
        write_slab<ROWS>(dst, slab_bytes, regs, j0);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    }
}

typedef float PairF __attribute__((ext_vector_type(2)));

// finish_chunk_packed (search_device.h) for a whole chunk of 32 candidates: ONE screen, then rounds whose selection out of the
// 32 register pairs is the generated tree KB_SELECT_TREE_32 (search_lds_asm.h: depth first, one temporary per level), and
// an insertion that finishes a slot before it looks at the next (TopKPacked::insert_in_sequence).  Round 5 ran it as two
// halves of 16 with the compiler's own trees, the second half's sums parked in scratch memory by hand; this form measures
// within 1 % of that on deep stacks and 6 % better where border tiles decide (cfg2 forced: 2.83 vs 3.06 ms) -- in either the
// allocator keeps a handful of list registers in scratch memory, stored and reloaded by every round (an interior tile's round:
// 3 loads + 2 stores; phase ticks of the finish 0.79 M per wave against 0.97 M as halves and 0.32 M for chunks of 16 on a
// 128 x 2048 x 2048 stack, LABNOTES r6.6).  Screening all 32 against the list as it stands at the start of the chunk lets a few
// more candidates into the rounds than screening the second half against what the first half left; the exact test in the round
// decides either way, the lists are the same.
// cw: observation counts packed two to a word (candidate c in half c & 1 of cw[c >> 1]); COUNTS = false: every candidate
// has all a.T observations (an interior tile).
template <int KS, bool COUNTS>
__device__ __forceinline__ void finish_chunk32_packed(const SearchArgs& a, int cand_base, const PairF (&acc)[32],
                                                      const uint32_t (&cw)[16], TopKPacked<KS>& top) {
    constexpr int C = 32;
    const float floor_lh = screen_key(fmaxf(top.lh[KS - 1], a.min_lh));
    uint32_t pending = 0;
#pragma unroll
    for (int c = C - 1; c >= 0; --c) {  // (downwards: the mask is built by shifting)
        const int n = COUNTS ? (int)((cw[c >> 1] >> (16 * (c & 1))) & 0xffffu) : a.T;
        const bool out = (n < a.min_obs) | screened_out(acc[c].x, acc[c].y, floor_lh);
        pending = (pending << 1) | (out ? 0u : 1u);
    }
    {
        const int left = a.n_cands - cand_base;  // uniform: candidates of this chunk that exist
        if (left < C) pending &= left > 0 ? (1u << left) - 1u : 0u;
    }
    while (__ballot(pending != 0u) != 0ull) {  // uniform
        const uint32_t sel = (uint32_t)__builtin_ctz(pending | 0x80000000u);  // (a lane with nothing left selects candidate 31 and drops it)
        float p, f;
        KB_SELECT_TREE_32(p, f, sel, acc)
        uint32_t n = (uint32_t)a.T;
        if constexpr (COUNTS) {
            uint32_t w[C / 2];
#pragma unroll
            for (int j = 0; j < C / 2; ++j) w[j] = cw[j];
#pragma unroll
            for (int width = C / 4, bit = 2; width >= 1; width >>= 1, bit <<= 1) {  // (the words are indexed by sel >> 1)
                const bool odd = (sel & (uint32_t)bit) != 0u;
#pragma unroll
                for (int i = 0; i < width; ++i) w[i] = odd ? w[2 * i + 1] : w[2 * i];
            }
            n = (w[0] >> (16u * (sel & 1u))) & 0xffffu;
        }
        const float lh = lh_from_sums(p, f);
        if (pending != 0u && lh > top.lh[KS - 1]) {
            top.insert_in_sequence(lh, flux_from_sums(p, f), (uint32_t)(cand_base + (int)sel) | (n << 16), a.stable_lists != 0);
        }
        pending &= pending - 1u;
    }
}

// One epoch whose samples cannot be read at lane base + scalar offset: either staged with slack because
// some candidate's shift sits on a rounding boundary (every lane predicts its own pixel with the
// reference's arithmetic and reads it from the slab), or not staged at all (a footprint larger than a
// slab, a velocity beyond the table's proven range: the lane reads the array itself, like
// kb_search_direct's exact mode).  Rare, and deliberately compact: the candidates are walked by a real
// loop and routed to their accumulators by a select chain, so that the
// kernel's register budget is set by its main loop and not by this path.
template <int C, int SF, bool CANON>
__device__ __forceinline__ void special_epoch(const SearchArgs& a, const TileCoords& tc, int chunk, int t, bool in_slab,
                                              const char* slab, PairF (&acc)[C], uint32_t (&cntp)[C / 2]) {
    using R = RawPair<SF>;
    constexpr int BYTES = 2 * fmt_bytes(SF);
    typedef const __attribute__((address_space(4))) double* ConstDoublePtr;
    const SearchCold* cold = a.cold;
    const double tm = ((ConstDoublePtr)(uintptr_t)cold->times)[t];
    const int box_word = as_const_ints(cold->boxes + (size_t)chunk * a.T + t)[0];
    const EpochBox box = make_int2(box_word, 0);
    const int ox = tc.tile_x0 + box_dx(box), oy = tc.tile_y0 + box_dy(box);  // image coordinates of slab pixel (0, 0)
    const int rows = as_const_ints(&a.chunks[chunk])[6];                     // rows_max: the slab's height
    const int cols = as_const_ints(&a.chunks[chunk])[7];                     // ... and its pitch
    const kb_trajectory* cands = cold->cands;
#pragma nounroll
    for (int c = 0; c < C; ++c) {
        const int ci = min(chunk * C + c, a.n_cands - 1);
        const ConstIntPtr cw = as_const_ints(cands + ci);  // {vx, vy, ...}
        int cx, cy;
        bool in = predict_index(tc.x, __int_as_float(cw[0]), tm, &cx);
        in = predict_index(tc.y, __int_as_float(cw[1]), tm, &cy) && in;
        float psi = NAN, phi = NAN;
        if (in_slab) {  // uniform
            const int rx = cx - ox, ry = cy - oy;
            const bool ok = in && ((unsigned)rx < (unsigned)cols) && ((unsigned)ry < (unsigned)rows);
            const int off = ok ? (ry * cols + rx) * BYTES : 0;
            const typename R::type raw = *reinterpret_cast<const typename R::type*>(slab + off);
            float p0, p1;
            R::decode(raw, a, &p0, &p1);
            if (CANON && __float_as_uint(p1) == 0x80000000u) p1 = NAN;  // the NO_DATA marker of the canonical copy
            psi = ok ? p0 : NAN;
            phi = ok ? p1 : NAN;
        } else if (in) {
            read_psi_phi(cold->meta, a.psi_phi, (uint64_t)t, cy, cx, &psi, &phi);  // NaN outside the image
        }
        const bool valid = __builtin_isfinite(psi) && __builtin_isfinite(phi);
        const PairF add = valid ? PairF{psi, phi} : PairF{0.0f, 0.0f};
#pragma unroll
        for (int cc = 0; cc < C; ++cc) {
            if (cc == c) {  // uniform
                acc[cc] += add;
                cntp[cc >> 1] += (valid ? 1u : 0u) << (16 * (cc & 1));
            }
        }
    }
}

// The per-pixel lists of a tile, three ways (LM):
//  LIST_REGISTERS  (likelihood, candidate) in registers, 16 per thread for K <= 8; the epilogue re-evaluates every
//                  winner with exact positions for its flux and observation count.
//  LIST_STORE_IDS  the same pairs in a lane-interleaved store in HBM between chunks (finish_chunk_stored): the
//                  summing loop carries only the likelihood to beat, so lists of 16 / 32 keep 4 waves per SIMD.
//  LIST_STORE_RECORDS  whole results (likelihood, candidate, flux, count) in that store: no re-evaluation at all.
// The store is read and written once per chunk of candidates, the re-evaluation costs K x T exact samples per
// pixel: the host takes records when the candidate list is short against the stack depth (cfg4's 64 candidates on
// 128 epochs: 124 -> 53 ms) and registers / ids when it is long (cfg2's 1024 on 64: 5.16 vs 5.61 ms).
//  LIST_REGISTER_RECORDS  whole results in registers at three words per slot (TopKPacked; K <= 8, fewer than 65535
//                  candidates): no store, no re-evaluation; the default for long candidate lists with K <= 8.
//  LIST_STORE_POOLED  stable lists of 16 in that store as likelihoods + cell numbers + a pool of records that never move
//                  (PooledList, search_device.h), fetched only by the lanes with a candidate past the screen: what the
//                  tie-exact exchange between devices runs (2 K = 16 records per pixel) with chunks of WIDE_CHUNK candidates.
enum ListMode { LIST_REGISTERS = 0, LIST_STORE_IDS = 1, LIST_STORE_RECORDS = 2, LIST_REGISTER_RECORDS = 3, LIST_STORE_POOLED = 4 };
template <int KS, int LM>
struct TileLists {
    static constexpr bool STORED = LM == LIST_STORE_IDS || LM == LIST_STORE_RECORDS || LM == LIST_STORE_POOLED;
    static constexpr bool PACKED = LM == LIST_REGISTER_RECORDS;
    static constexpr bool RECORDS = LM == LIST_STORE_RECORDS;
    static constexpr bool STORE_POOLED = LM == LIST_STORE_POOLED;
    static constexpr uint32_t SLOT_BYTES = (RECORDS || STORE_POOLED) ? 16u : 8u;
    TopK<KS> top;     // LIST_REGISTERS
    TopKPacked<KS> packed;  // LIST_REGISTER_RECORDS
    ListState state;  // STORED
    char* store;      // STORED: this tile's block of the store (uniform)
};

// Staging schedule of one chunk.
struct ChunkPlan {
    int cols;        // pitch of the chunk's slabs in pixels: 64 + its widest dx spread (ChunkInfo::cols)
    int slab_bytes;  // rows_max * cols * BYTES
    int stride;      // distance of the group's slabs in LDS: slab_bytes rounded up to a wave's 1 KiB of pieces, so that
                     // a wave can write all 64 of its pieces without a lane mask
    int E;           // epochs per group
    int clean;       // every epoch is staged with uniform shifts: the summing loop needs no per-epoch test
    int whole;       // T / E
    uint32_t inv;    // ceil(2^20 / cols)
};
template <int BYTES, int ROWS, bool EVEN>
__device__ __forceinline__ ChunkPlan chunk_plan(const SearchArgs& a, int chunk) {
    ChunkPlan p;
    // {dx_min, dx_max, dy_min, dy_max, unsafe, lds_ok, rows_max, cols, e_even, e_any, t_over_e_even, t_over_e_any, cols_inv}
    const ConstIntPtr ci = as_const_ints(&a.chunks[chunk]);
    p.cols = ci[7];
    p.slab_bytes = ci[6] * p.cols * BYTES;
    p.stride = (p.slab_bytes + 1023) & ~1023;
    if constexpr (BYTES == 8) {  // (float pairs: what the table kernel worked out for this tile height -- no division here)
        p.E = EVEN ? ci[8] : ci[9];
        p.whole = EVEN ? ci[10] : ci[11];
    } else {
        p.E = group_epochs(a.T, ROWS, p.stride, EVEN);
        p.whole = a.T / p.E;
    }
    p.inv = (uint32_t)ci[12];
    p.clean = (ci[4] == 0 && ci[5] != 0) ? 1 : 0;
    return p;
}

// Observation counts of a chunk's candidates for a lane of a tile at the image's edge, without counting samples.  In a
// stack without a NO_DATA pixel a sample is NO_DATA exactly when it lies off the image, so for start pixel (x, y) and a
// candidate with integer shifts (dx_e, dy_e) the count is the number of epochs with 0 <= x + dx_e < W and 0 <= y + dy_e < H.
// kb_edge_count_kernel has tabulated, per candidate, axis and direction, how many epochs shift at most d pixels; when the
// shifts of every candidate grow monotonically along both axes (its flag; any list of epochs in time order does) each of
// those epoch sets is a leading run of the epochs, and the count is the shortest of the four runs: four 32-byte rows
// (16 candidates each) and 24 packed minima per chunk instead of one vector instruction per sample.
template <int C>
__device__ __forceinline__ void edge_counts(const SearchArgs& a, const TileCoords& tq, int chunk, const uint4* edge_tab,
                                            int edge_D, uint32_t (&cnte)[C / 2]) {
    static_assert(C == 16 || C == 32, "rows of the edge tables hold C counts of 16 bits: whole 16-byte words");
    typedef unsigned short Us2 __attribute__((ext_vector_type(2)));
    constexpr int Q = C / 8;  // 16-byte words per row
    const int D1 = edge_D + 1;
    const uint4* base = edge_tab + (size_t)chunk * 4 * D1 * Q;
    auto clampd = [&](int d) { return min(max(d, 0), edge_D); };
    const int d[4] = {clampd(a.W - 1 - tq.x), clampd(tq.x), clampd(a.H - 1 - tq.y), clampd(tq.y)};
    uint32_t m[C / 2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const uint4 v = base[(k * D1 + d[k]) * Q + q];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (k == 0) {
                    m[4 * q + j] = w[j];
                } else {
                    const Us2 r = __builtin_elementwise_min(__builtin_bit_cast(Us2, m[4 * q + j]), __builtin_bit_cast(Us2, w[j]));
                    m[4 * q + j] = __builtin_bit_cast(uint32_t, r);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < C / 2; ++j) cnte[j] = m[j];
}

// The whole search of one tile.  One flat software pipeline over (chunk, group): while
// group g is summed out of one LDS buffer, group g+1 -- possibly the first group of the next
// chunk -- is copied into the other, one slab per summed epoch: its loads are issued before
// the epoch's sums and written to LDS after them.
// FAST: no sample of this tile can be NO_DATA (the tile stays inside the image under
// every shift, the array has no NO_DATA pixel, every epoch is staged): obs_count is T.
template <int KS, int C, int ROWS, int NB, bool CANON, bool SIGMAG, int LM, bool FAST>
__device__ __forceinline__ void lds_search_tile(const SearchArgs& a, const TileCoords& tc, char* smem,
                                                TileLists<KS, LM>& lists, const uint4* edge_tab = nullptr, int edge_D = 0) {
    constexpr int SF = CANON ? 4 : NB;  // staged format
    using R = RawPair<SF>;
    constexpr int BYTES = 2 * fmt_bytes(SF);
    // the float-staged, depth-1 instances run their epochs through search_lds_asm.h (chunks of XWIDE_CHUNK: their one statement
    // is count-free -- a tile that has to count samples, which the host keeps from this instance where it can, is summed by
    // the compiler-scheduled loops below)
    constexpr bool HAND_SCHEDULED = CANON && (C == 8 || C == 16 || (C == XWIDE_CHUNK && FAST));
    // groups of an even number of epochs: the statements for chunks of 8 and 16 work in pairs of epochs
    constexpr bool EVEN_GROUPS = CANON && (C == 8 || C == 16);
    // (the counting statements that read sixteen samples per wait take eight more registers: not for the pooled lists)
    [[maybe_unused]] constexpr bool KB_LDS_WIDE_COUNT = LM != LIST_STORE_POOLED;
    const int T = a.T;

    PairF acc[C];  // (psi_sum, phi_sum) as pairs: one v_pk_add_f32 per sample
    // observation counts, two 16-bit counts per register (a stack has at most 999 epochs): candidate c in half
    // c & 1 of cntp[c >> 1]
    uint32_t cntp[C / 2];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = PairF{0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < C / 2; ++c) cntp[c] = 0u;

#ifdef KB_EXP_PROFILE
    uint64_t prof_acc[6] = {0, 0, 0, 0, 0, 0};
    uint64_t prof_t = __builtin_amdgcn_s_memtime();
#endif
    int chunk = a.chunk_lo, t0 = 0, buf = 0;
    ChunkPlan plan = chunk_plan<BYTES, ROWS, EVEN_GROUPS>(a, chunk);
    SlabRegs regs;
    typedef int Int4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) Int4* ConstSlabPtr;  // a SlabRef as four dwords
    auto origin_of = [](Int4 r) { return (int64_t)(((uint64_t)(uint32_t)r.y << 32) | (uint64_t)(uint32_t)r.x); };
    // this tile's own pixel inside the padded copy
    const char* tile_base = reinterpret_cast<const char*>(a.padded) + ((int64_t)tc.tile_y0 * a.Wp + tc.tile_x0) * BYTES;
    StageLane n_sl = stage_lanes<BYTES, ROWS>(a, plan.cols, plan.slab_bytes, plan.inv);  // staging map of the group being copied
    {
        const ConstSlabPtr org = (ConstSlabPtr)(uintptr_t)(a.slabs + (size_t)chunk * T);
        const int n = min(plan.E, T);
        for (int e = 0; e < n; ++e) {
            const int64_t o = origin_of(org[e]);
            load_slab<BYTES, ROWS>(a, n_sl, tile_base + o, plan.slab_bytes, regs);
            write_slab<ROWS>(smem + e * plan.stride, plan.slab_bytes, regs);
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            copy_slab_tail<BYTES, ROWS>(a, n_sl, tile_base + o, plan.slab_bytes, plan.cols, smem + e * plan.stride, regs);
        }
    }
    __syncthreads();
    KB_PROF_MARK(0)

    // Operands of the hand-scheduled statements that must sit in scalar registers: read through the first lane, because an
    // "s" constraint on a value the compiler holds (or believes) per-lane is not enforced -- it prints the vector register.
    auto sgpr32 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    auto sgpr64 = [&](uint64_t v) { return ((uint64_t)sgpr32((uint32_t)(v >> 32)) << 32) | (uint64_t)sgpr32((uint32_t)v); };
    while (chunk < a.chunk_hi) {
        // This trip's view of the thread's place.  Everything per-lane below derives from a thread index the compiler
        // cannot see through, so that none of it (lane offsets, the start pixel as doubles for the exact path, ...) is
        // hoisted out of the loop and held in registers across the hand-scheduled statements, which need them.
        int tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid));
        TileCoords tq = tc;
        tq.lane = tid & (WAVE - 1);
        tq.x_i = tc.tx * WAVE + tq.lane;
        tq.x = tq.x_i + a.x_start_min;
        if constexpr (HAND_SCHEDULED) {
            // Whole groups of this chunk whose staged successors are whole groups of this chunk too: ONE asm statement
            // (KB_LDS_STREAM_*), group changes -- LDS writes landed, barrier, buffers swapped -- inside it, so that slab loads
            // and table words stay in flight across them and a group costs a handful of instructions instead of a set-up, a
            // pipeline fill and a drain.  Needs an even number of epochs per group (chunk_plan's EVEN) and slabs of one
            // two staging rounds at most, so that every wave of the block takes this path, with none, one or two pieces per slab.
            // (Every read-write operand of these statements is early-clobber: they are written while inputs are still
            // being read, and without the mark an input of equal value -- pairs per group and pairs to the next barrier --
            // is given the same register.)
#ifdef KB_EXP_NO_STREAM
            const int ng = 0;
#else
            const int ng = (t0 == 0 ? plan.whole : (T - t0) / plan.E) - 1;  // (t0 == 0: the common case, no division)
#endif
            constexpr bool EPOCH_TRIPS = C == XWIDE_CHUNK;  // a trip of the statement is one epoch (chunks of 32), not two
            if (plan.clean && (EPOCH_TRIPS || (plan.E & 1) == 0) && plan.slab_bytes <= 2 * stage_round(ROWS) &&
                ng >= (EPOCH_TRIPS ? 1 : 2)) {
                KB_PROF_MARK(1)
                const ConstSlabPtr first = (ConstSlabPtr)(uintptr_t)(a.slabs + (size_t)chunk * T + t0 + plan.E);
                const int GB = lds_group_bytes(ROWS);
                uint32_t rb = (uint32_t)(uintptr_t)(smem + buf * GB + (tc.wv * plan.cols + tq.lane) * BYTES);
                uint32_t wd = (uint32_t)(uintptr_t)(smem + (1 - buf) * GB + 16 * tid);
                uint32_t dr = sgpr32((uint32_t)((1 - 2 * buf) * GB));
                const uint32_t pg = sgpr32(EPOCH_TRIPS ? (uint32_t)plan.E : (uint32_t)plan.E >> 1);
                uint32_t gc = pg, pairs = sgpr32((uint32_t)ng * pg);
                const uint32_t es = sgpr32((uint32_t)(plan.E * plan.stride)), st = sgpr32((uint32_t)plan.stride);
                const uint32_t go = n_sl.goff[0];
                // (counting statements: groups per 32 epochs -- their NO_DATA shift registers hold 32 samples -- and groups
                // until they are emptied next)
                const uint32_t fg = sgpr32(max(1u, (C == 8 ? 32u : 16u) / (uint32_t)plan.E));
                uint32_t fc = fg;
                const uint64_t ob = sgpr64((uint64_t)(uintptr_t)(a.lds_fold + ((size_t)chunk * T + t0) * C));
#ifdef KB_LDS_DMA
                const uint64_t gb = sgpr64((uint64_t)(uintptr_t)first);  // (the LDS-DMA statements count from one entry earlier)
#else
                const uint64_t gb = sgpr64((uint64_t)(uintptr_t)(first + 1));
#endif
                const uint64_t tb = sgpr64((uint64_t)(uintptr_t)tile_base);
                // (slabs are as tall as their epoch's shift box: a piece behind a slab's end copies the first slab's instead)
                const Int4 ref0 = first[0];
                const uint32_t wp = sgpr32((uint32_t)(1024 * tc.wv));
                const uint32_t dl = sgpr32((uint32_t)ref0.x), dh = sgpr32((uint32_t)ref0.y);
                const uint64_t b0 = sgpr64(tb + (uint64_t)origin_of(ref0));
                const uint32_t tl = (uint32_t)tb, th = (uint32_t)(tb >> 32);
                // (a wave's second piece of a slab of more than one staging round: its lane offset, its first byte)
                const uint32_t gq = n_sl.goff[LDS_SLOTS >= 2 ? 1 : 0];
                const uint32_t wq = sgpr32(wp + (uint32_t)stage_round(ROWS));
                // (pieces of every slab this wave copies: the statement holds a body for each)
                const uint32_t nq = (uint32_t)__builtin_amdgcn_readfirstlane((int)wq < plan.slab_bytes ? 2 : ((int)wp < plan.slab_bytes ? 1 : 0));
                (void)fc; (void)fg;
                KB_LDS_RUN_STREAM
                if constexpr (!FAST) {
                    // the statement subtracted the NO_DATA samples: add its epochs to both counts of every register
                    const uint32_t summed = (uint32_t)(ng * plan.E) * 0x10001u;
#pragma unroll
                    for (int c = 0; c < C / 2; ++c) cntp[c] += summed;
                }
                KB_PROF_MARK(2)
                t0 += ng * plan.E;
                buf ^= ng & 1;
#ifndef KB_EXP_NO_BARRIER
                __syncthreads();
#endif
                KB_PROF_MARK(5)
                continue;
            }
        }
        // next group in flight during this group's arithmetic
        int n_chunk = chunk, n_t0 = t0 + plan.E;
        ChunkPlan n_plan = plan;
        if (n_t0 >= T) {
            n_chunk = chunk + 1;
            n_t0 = 0;
            if (n_chunk < a.chunk_hi) {
                n_plan = chunk_plan<BYTES, ROWS, EVEN_GROUPS>(a, n_chunk);
                n_sl = stage_lanes<BYTES, ROWS>(a, n_plan.cols, n_plan.slab_bytes, n_plan.inv);
            }
        }
        const int n_next = (n_chunk < a.chunk_hi) ? min(n_plan.E, T - n_t0) : 0;
        const ConstSlabPtr n_org = (ConstSlabPtr)(uintptr_t)(a.slabs + (size_t)min(n_chunk, a.chunk_hi - 1) * T + n_t0);
        char* nb = smem + (1 - buf) * lds_group_bytes(ROWS);
        // slab e of the next group: loads issued before, LDS writes after the sums of epoch e
        const char* n_base = tile_base;
        auto next_load = [&](int e) -> bool {
            if (e >= n_next) return false;
            n_base = tile_base + origin_of(n_org[e]);
            load_slab<BYTES, ROWS>(a, n_sl, n_base, n_plan.slab_bytes, regs);
            return true;
        };
        auto next_write = [&](int e) {
            write_slab<ROWS>(nb + e * n_plan.stride, n_plan.slab_bytes, regs);
            if (n_plan.slab_bytes > LDS_SLOTS * stage_round(ROWS)) {  // uniform, rare
                __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
                copy_slab_tail<BYTES, ROWS>(a, n_sl, n_base, n_plan.slab_bytes, n_plan.cols, nb + e * n_plan.stride, regs);
            }
        };

        const ConstIntPtr offs = as_const_ints(a.lds_off + ((size_t)chunk * T + t0) * C);  // offsets for 8-byte pairs
        const char* cb = smem + buf * lds_group_bytes(ROWS) + (tc.wv * plan.cols + tq.lane) * BYTES;  // this lane's start pixel inside a slab
        const int n_cur = min(plan.E, T - t0);
        // C samples of one staged epoch with uniform shifts: slab offsets o[] (scalars) -> LDS reads -> sums
        auto no_hook = []() {};
        auto sum_epoch = [&](const int (&o)[C], const char* rb, auto between) {  // rb: this lane's pixel in the epoch's slab
            // eight reads in flight at a time (a chunk of 16 goes in two halves: the registers of the samples
            // are the ones this kernel is short of)
            constexpr int HALF = 8;
#pragma unroll
            for (int c0 = 0; c0 < C; c0 += HALF) {
                typename R::type raw[HALF];
#pragma unroll
                for (int c = 0; c < HALF; ++c) {
                    const int off = (BYTES == 8) ? o[c0 + c] : (o[c0 + c] >> 3) * BYTES;
                    // (an LDS-typed pointer: through the generic one every read paid a scalar add of the aperture's zero)
                    if constexpr (BYTES == 8) {
                        typedef const __attribute__((address_space(3))) PairF* LdsPair;
                        const PairF v = *(LdsPair)(uint32_t)(uintptr_t)(rb + off);
                        raw[c].x = v.x;
                        raw[c].y = v.y;
                    } else {
                        raw[c] = *reinterpret_cast<const typename R::type*>(rb + off);
                    }
                }
                if (c0 + HALF >= C) {  // (behind the reads of the LAST half: the hook replaces o[])
                    __builtin_amdgcn_sched_barrier(0);
                    between();
                    __builtin_amdgcn_sched_barrier(0);
                }
                // one wait for the reads instead of the compiler's one per read (instruction issue is the bound)
                __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
#pragma unroll
                for (int c = 0; c < HALF; ++c) {
                    if constexpr (CANON) {
                        acc[c0 + c] += PairF{raw[c].x, raw[c].y};
                        if (!FAST) cntp[(c0 + c) >> 1] += ((__float_as_uint(raw[c].y) != 0x80000000u) ? 1u : 0u) << (16 * ((c0 + c) & 1));
                    } else {
                        float psi, phi;
                        R::decode(raw[c], a, &psi, &phi);
                        if (FAST) {
                            acc[c0 + c] += PairF{psi, phi};
                        } else {
                            float s0 = acc[c0 + c].x, s1 = acc[c0 + c].y;
                            int seen = 0;
                            accumulate(psi, phi, true, s0, s1, seen);
                            cntp[(c0 + c) >> 1] += (uint32_t)seen << (16 * ((c0 + c) & 1));
                            acc[c0 + c] = PairF{s0, s1};
                        }
                    }
                }
            }
        };
        // Keeps the epoch's sums in front of the LDS writes of the staged slab: left alone the compiler
        // sinks the adds behind the writes, whose vmcnt(0) then waits out the loads with nothing to overlap.
        auto pin_sums = [&]() {
#pragma unroll
            for (int c = 0; c < C; ++c) asm volatile("" : "+v"(acc[c])::"memory");
#pragma unroll
            for (int c = 0; c < C / 2; ++c) asm volatile("" : "+v"(cntp[c])::"memory");
        };
        KB_PROF_MARK(1)
        if (__builtin_expect(plan.clean != 0, 1)) {  // (the other branch is cold: register spills belong there)
            // A block alone on its CU is bound by the chain scalar table fetch -> LDS read -> adds -> slab
            // landed -> LDS write of one epoch (measured 4.9 ms with one block per CU against 7.4 ms with
            // four).  The table words of epoch e + 1 (slab offsets, slab origin) are therefore fetched at
            // the END of epoch e, behind the adds: they travel while the slab loads are waited for and
            // written, and the lgkmcnt wait of epoch e + 1's LDS reads finds them done.  The empty asm
            // pins the fetch behind pin_sums(); both tables have slack behind their last entry.
            int o_cur[C];
#pragma unroll
            for (int c = 0; c < C; ++c) o_cur[c] = offs[c];
            int64_t org_cur = origin_of(n_org[0]);
            int e = 0;
            // Epochs that both sum and stage, slabs of at most LDS_SLOTS rounds (the rule): the loop is specialised
            // by the number of rounds in which this wave has pieces inside the slab (wave-uniform, fixed for the
            // group), so that its body holds no test at all -- loads, sums, table fetch, LDS writes of whole
            // 1 KiB wave pieces (the slab stride in LDS leaves room for the last one's overhang).
            const int n_both = (n_plan.slab_bytes <= LDS_SLOTS * stage_round(ROWS)) ? min(n_cur, n_next) : 0;
            auto staged_run = [&](auto np_tag) {
                constexpr int NP = decltype(np_tag)::value;
                char* wdst = nb + 16 * tid;
                const char* rptr = cb + e * plan.stride;  // this lane's pixel in the slab being summed
                int64_t org_nxt = origin_of(n_org[1]);  // origin of the slab whose loads are issued next
                auto load = [&](Piece (&v)[LDS_SLOTS], int64_t org) {
#ifdef KB_EXP_NO_STAGE
                    return;
#endif
                    const char* base = tile_base + org;
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        uint32_t off = n_sl.goff[j];
                        asm volatile("" : "+v"(off));  // (see load_slab)
                        const PieceMem<(BYTES < 4 ? BYTES : 4)>* src =
                                reinterpret_cast<const PieceMem<(BYTES < 4 ? BYTES : 4)>*>(base + off);
                        v[j] = Piece{src->w[0], src->w[1], src->w[2], src->w[3]};
                    }
                };
                // sums of epoch e, then the LDS write of staged slab e out of v
                auto step = [&](Piece (&v)[LDS_SLOTS]) {
                    // The table words of the next epoch are fetched right behind the issue of this epoch's LDS reads
                    // and waited for together with them, i.e. BEFORE the slab is written.  The wait for a scalar
                    // load is a wait for every LDS operation of the wave; placed behind the write, as it was, it
                    // holds the next epoch's reads until that write has completed.
                    sum_epoch(o_cur, rptr, [&]() {
#if defined(KB_EXP_TABLE_HIT)  // timing experiment: every epoch re-reads one table entry (no scalar-cache misses, one slab)
                        ConstIntPtr po = offs;
                        ConstSlabPtr pg = n_org;
#else
                        ConstIntPtr po = offs + (e + 1) * C;
                        ConstSlabPtr pg = n_org + (e + 1);
#endif
#pragma unroll
                        for (int c = 0; c < C; ++c) o_cur[c] = po[c];
                        org_nxt = origin_of(pg[0]);
                    });
                    __builtin_amdgcn_sched_barrier(0);  // the sums stay in front of the write and its vmcnt wait
                    if (!FAST) {
                        // (keeps the counts of two unrolled epochs from being merged into three-operand adds that
                        // hold eight more registers across the pair)
#pragma unroll
                        for (int c = 0; c < C / 2; ++c) asm volatile("" : "+v"(cntp[c]));
                    }
#ifndef KB_EXP_NO_STAGE
#pragma unroll
                    for (int j = 0; j < NP; ++j) *reinterpret_cast<Piece*>(wdst + stage_round(ROWS) * j) = v[j];
#endif
                    wdst += n_plan.stride;
                    rptr += plan.stride;
                    ++e;
                };
                {
                    Piece va[LDS_SLOTS];
                    org_nxt = org_cur;
                    while (e < n_both) {
                        load(va, org_nxt);
                        step(va);
                    }
                }
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
                // origin of slab e (for the loop below, and for the next group when this one ends here): the last step's
                // prefetch already holds it
                org_cur = org_nxt;
            };
            // The same run as ONE asm statement (search_lds_asm.h, generated by tools/gen_lds_loop.py) for the float-staged
            // kernels with one slab per wave and epoch at most: two epochs per trip; the table words of epoch e + 2 are
            // fetched BEHIND the wait for epoch e's LDS reads, into the registers those reads just consumed, so the scalar
            // loads -- which share the LDS counter and return out of order: any wait for them waits for everything -- have a
            // whole epoch to land in instead of sitting in front of the sums; two slabs in flight; offsets with the slab's
            // place in its group folded in (SearchArgs::lds_fold), so the lane's read pointer never moves.  Nothing the
            // compiler schedules runs while anything is in flight: the registers involved are fixed and named as clobbers.
            auto asm_run = [&](uint32_t nq_of_wave) {  // pieces of every staged slab this wave copies
                const uint32_t nq = (uint32_t)__builtin_amdgcn_readfirstlane((int)nq_of_wave);
                const int wave_piece_of_run = 1024 * tc.wv;
                static_assert((C == 8 || C == 16 || C == XWIDE_CHUNK) && sizeof(SlabRef) == 16, "search_lds_asm.h");
                constexpr bool EPOCH_TRIPS = C == XWIDE_CHUNK;
                uint32_t pairs = sgpr32(EPOCH_TRIPS ? (uint32_t)(n_both - e) : (uint32_t)(n_both - e) >> 1);
                const uint32_t odd = sgpr32((uint32_t)(n_both - e) & 1u);
                const int done = n_both - e;
                uint32_t wd = (uint32_t)(uintptr_t)(nb + 16 * tid + e * n_plan.stride);
                const uint32_t rb = (uint32_t)(uintptr_t)cb;  // this lane's pixel at the start of the group buffer
                const uint32_t go = n_sl.goff[0];
                const uint64_t ob = sgpr64((uint64_t)(uintptr_t)(a.lds_fold + ((size_t)chunk * T + t0 + e) * C));
                // (the statement for chunks of 32 fetches its slab references from one entry earlier, like every LDS-DMA statement)
                const uint64_t gb = sgpr64((uint64_t)(uintptr_t)(n_org + (EPOCH_TRIPS ? e : e + 1)));
                const uint64_t tb = sgpr64((uint64_t)(uintptr_t)tile_base);
                const uint64_t b0 = sgpr64(tb + (uint64_t)org_cur);
                const uint32_t tl = (uint32_t)tb, th = (uint32_t)(tb >> 32);
                const uint32_t wp = sgpr32((uint32_t)wave_piece_of_run);
                const uint32_t dl = sgpr32((uint32_t)(uint64_t)org_cur), dh = sgpr32((uint32_t)((uint64_t)org_cur >> 32));
                const uint32_t st = sgpr32((uint32_t)n_plan.stride);
                const uint32_t gq = n_sl.goff[LDS_SLOTS >= 2 ? 1 : 0];
                const uint32_t wq = sgpr32(wp + (uint32_t)stage_round(ROWS));
                if constexpr (EPOCH_TRIPS) {
                    // Chunks of 32 have ONE statement, the run of whole groups; a single group that stages into the other
                    // buffer is that run without a group change: the count to the next barrier never reaches zero.
                    uint32_t rbw = rb;
                    uint32_t gc = sgpr32(pairs + 1u), dr = sgpr32(0u);
                    const uint32_t pg = gc, es = sgpr32(0u);
                    uint32_t& rb = rbw;  // (read-write operand of the statement; unchanged without a group change)
                    [[maybe_unused]] uint32_t fc = 0;  // (named by the counting statements of the other chunk widths)
                    [[maybe_unused]] const uint32_t fg = 0;
                    (void)odd;
                    KB_LDS_RUN_STREAM
                } else {
                    KB_LDS_RUN_LOOP
                }
                e += done;
                if constexpr (!FAST) {
                    // (the counting statements subtract the NO_DATA samples from the counts, see search_lds_asm.h)
                    const uint32_t summed = (uint32_t)done * 0x10001u;
#pragma unroll
                    for (int c = 0; c < C / 2; ++c) cntp[c] += summed;
                }
                // hand over to the loop below: plain offsets and origin of epoch / slab e, from the tables it walks.
                // Unconditional on purpose: with e == n_cur the values are never used and the reads land in the tables'
                // slack (off_bytes' 4 * CHUNK ints, SLAB_REF_SLACK references), but the straight-line form measured
                // 4.44 ms against 4.57 ms for the guarded one on the same chip (tools/ab.sh) -- a guard here also splits
                // the live ranges of the loop below.
                {
                    const ConstIntPtr cur = offs + e * C;
#pragma unroll
                    for (int c = 0; c < C; ++c) o_cur[c] = cur[c];
                    org_cur = origin_of(n_org[e]);
                }
            };
            if (n_both > 0) {
                const int wave_piece = 1024 * tc.wv;
                if constexpr (HAND_SCHEDULED) {
                    asm_run((LDS_SLOTS >= 2 && wave_piece + stage_round(ROWS) < n_plan.slab_bytes) ? 2u
                                                                                                    : (wave_piece < n_plan.slab_bytes ? 1u : 0u));
                } else if (LDS_SLOTS >= 2 && wave_piece + stage_round(ROWS) < n_plan.slab_bytes) {
                    staged_run(std::integral_constant<int, (LDS_SLOTS >= 2 ? 2 : 1)>{});
                } else if (wave_piece < n_plan.slab_bytes) {
                    staged_run(std::integral_constant<int, 1>{});
                } else {
                    staged_run(std::integral_constant<int, 0>{});
                }
            }
            KB_PROF_MARK(2)
            // what is left: sums of a group longer than the next one, slabs of more rounds
            for (; e < n_cur; ++e) {
                const bool staging = e < n_next;
                if (staging) {
                    n_base = tile_base + org_cur;
                    load_slab<BYTES, ROWS>(a, n_sl, n_base, n_plan.slab_bytes, regs);
                }
                sum_epoch(o_cur, cb + e * plan.stride, no_hook);
                pin_sums();
                ConstIntPtr po = offs + (e + 1) * C;
                ConstSlabPtr pg = n_org + (e + 1);
                asm volatile("" : "+s"(po), "+s"(pg)::"memory");
#pragma unroll
                for (int c = 0; c < C; ++c) o_cur[c] = po[c];
                org_cur = origin_of(pg[0]);
                if (staging) next_write(e);
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            }
        } else {
            // A chunk with epochs that are not staged, or staged without uniform shifts (rare: a shift within 1e-6 of a
            // half pixel, a footprint larger than a slab).  The instance for chunks of WIDE_CHUNK candidates has no such
            // path: the host gives it candidate lists without such epochs only (search_kernels.hip) -- inlined next to
            // 2 * WIDE_CHUNK accumulators, the per-lane predictions in double precision set the register budget of the
            // whole kernel, and the allocator then keeps accumulators and lists in scratch memory everywhere.
            if constexpr (C == CHUNK) {
                for (int e = 0; e < n_cur; ++e) {
                    const bool staging = next_load(e);
                    int o[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) o[c] = offs[e * C + c];
                    if (o[0] >= 0) {
                        sum_epoch(o, cb + e * plan.stride, no_hook);
                    } else {
                        special_epoch<C, SF, CANON>(a, tq, chunk, t0 + e, o[0] == LDS_OFF_PER_LANE,
                                                    smem + buf * lds_group_bytes(ROWS) + e * plan.stride, acc, cntp);
                    }
                    pin_sums();
                    if (staging) next_write(e);
                    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
                }
            } else {
                __builtin_trap();
            }
        }
        for (int e = n_cur; e < n_next; ++e) {  // the next group holds more epochs than this one
            if (next_load(e)) next_write(e);
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        }
        KB_PROF_MARK(3)
        if (n_chunk != chunk) {  // chunk complete: likelihoods + top-K, while the next chunk's first group lands
            if constexpr (TileLists<KS, LM>::STORE_POOLED) {
                if (tc.row_active) {
                    float ps[C], ph[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        ps[c] = acc[c].x;
                        ph[c] = acc[c].y;
                    }
                    static_assert(KS == 16, "pooled lists hold 16 slots");
                    bool by_table = false;
                    if constexpr (FAST && C >= WIDE_CHUNK) {
                        if (edge_tab != nullptr) {  // (uniform) a tile at the image's edge: counts out of the tables
                            uint32_t cnte[C / 2];
                            edge_counts<C>(a, tq, chunk, edge_tab, edge_D, cnte);
                            finish_chunk_pooled<C, false>(a, chunk, ps, ph, cnte, lists.state, lists.store,
                                                          PooledLayout{(uint32_t)(ROWS * WAVE)}, threadIdx.x);
                            by_table = true;
                        }
                    }
                    if (!by_table) {
                        finish_chunk_pooled<C, FAST>(a, chunk, ps, ph, cntp, lists.state, lists.store,
                                                     PooledLayout{(uint32_t)(ROWS * WAVE)}, threadIdx.x);
                    }
                }
            } else if (tc.row_active) {
                float ps[C], ph[C];
                int cnt[C];
                // (chunks of 32 keep their counts packed two to a word all the way into the rounds: finish_chunk32_packed)
                constexpr bool PACKED_COUNTS = TileLists<KS, LM>::PACKED && C == XWIDE_CHUNK;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    ps[c] = acc[c].x;
                    ph[c] = acc[c].y;
                    if constexpr (!PACKED_COUNTS) cnt[c] = FAST ? T : (int)((cntp[c >> 1] >> (16 * (c & 1))) & 0xffffu);
                }
                if constexpr (FAST && C >= WIDE_CHUNK && !PACKED_COUNTS) {
                    if (edge_tab != nullptr) {  // (uniform) a tile at the image's edge: counts out of the tables
                        uint32_t cnte[C / 2];
                        edge_counts<C>(a, tq, chunk, edge_tab, edge_D, cnte);
#pragma unroll
                        for (int c = 0; c < C; ++c) cnt[c] = (int)((cnte[c >> 1] >> (16 * (c & 1))) & 0xffffu);
                    }
                }
#ifdef KB_EXP_NO_FINISH
                if constexpr (TileLists<KS, LM>::PACKED) {
                    float sink = 0.0f;
#pragma unroll
                    for (int c = 0; c < C; ++c) sink += ps[c] + ph[c] + (PACKED_COUNTS ? 0.0f : (float)cnt[c]);
                    lists.packed.lh[0] = fmaxf(lists.packed.lh[0], sink);
                } else
#endif
                if constexpr (SIGMAG) {
                    TopK<KS> none;  // (the emitting instances keep no list)
                    finish_chunk<KS, C, true>(a, tc, chunk, ps, ph, cnt, none);
                } else if constexpr (TileLists<KS, LM>::STORED) {
                    finish_chunk_stored<KS, C, TileLists<KS, LM>::RECORDS>(a, chunk, ps, ph, cnt, lists.state, lists.store,
                                                                      TileLists<KS, LM>::SLOT_BYTES * threadIdx.x,
                                                                      ROWS * WAVE * TileLists<KS, LM>::SLOT_BYTES);
                } else if constexpr (TileLists<KS, LM>::PACKED && C == XWIDE_CHUNK) {
                    if (edge_tab != nullptr) {  // (uniform) a tile at the image's edge: counts out of the tables
                        uint32_t cnte[C / 2];
                        edge_counts<C>(a, tq, chunk, edge_tab, edge_D, cnte);
                        finish_chunk32_packed<KS, true>(a, chunk * C, acc, cnte, lists.packed);
                    } else {
                        const uint32_t none[C / 2] = {};
                        finish_chunk32_packed<KS, false>(a, chunk * C, acc, none, lists.packed);
                    }
                } else if constexpr (TileLists<KS, LM>::PACKED) {
                    finish_chunk_packed<KS, C>(a, chunk * C, ps, ph, cnt, lists.packed);
                } else {
                    finish_chunk<KS, C, false>(a, tc, chunk, ps, ph, cnt, lists.top);
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                acc[c] = PairF{0.0f, 0.0f};
                cntp[c >> 1] = 0u;
            }
        }
        KB_PROF_MARK(4)
#ifndef KB_EXP_NO_BARRIER
        __syncthreads();
#endif
        KB_PROF_MARK(5)
        buf = 1 - buf;
        chunk = n_chunk;
        t0 = n_t0;
        plan = n_plan;
    }
#ifdef KB_EXP_PROFILE
    if (tc.lane == 0) {
        for (int i = 0; i < 6; ++i) atomicAdd(&kb_exp_prof[i], (unsigned long long)prof_acc[i]);
        atomicAdd(&kb_exp_prof[6], 1ull);
    }
#endif
}


template <int KS, int C, int ROWS, int NB, bool CANON, bool SIGMAG, int LM>
// second launch bound = waves per SIMD: 16 waves per CU (one 64 x 16 or two 64 x 8 workgroups)
__global__ __launch_bounds__(ROWS * WAVE, 4) void kb_search_lds(const SearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // two group buffers
    constexpr int BYTES = 2 * fmt_bytes(CANON ? 4 : NB);
    const TileCoords tc = tile_coords<ROWS>(a);  // rows past the search area stay alive (barriers)
    TileLists<KS, LM> lists;
    lists.top.init();
    lists.packed.init();
    lists.state = {SIGMAG ? -FLT_MAX : a.min_lh, 0};  // the likelihood to beat starts at the lists' floor (flag 1024; else -FLT_MAX)
    lists.store = (SIGMAG || !TileLists<KS, LM>::STORED)
                          ? nullptr
                          : reinterpret_cast<char*>(a.lists) +
                                    ((size_t)(tc.ty * a.tiles_x + tc.tx) * KS) * (ROWS * WAVE * TileLists<KS, LM>::SLOT_BYTES);

    // Workgroup-uniform: can any sample of this tile be NO_DATA?
    const ConstIntPtr gb = as_const_ints(a.global_box);
    const bool fast = a.all_staged && as_const_ints(a.n_invalid)[0] == 0 && (tc.tile_x0 + gb[0] >= 0) &&
                      (tc.tile_x0 + WAVE + gb[1] <= a.W) && (tc.tile_y0 + gb[2] >= 0) &&
                      (tc.tile_y0 + ROWS + gb[3] <= a.H);
    // ... and if it can, only by leaving the image (no NO_DATA pixel in it, every epoch staged with uniform shifts, tables of
    // the candidates' epochs per shift built and valid): the count-free loops + counts out of the tables (edge_counts)
    const uint4* edge_tab = nullptr;
    int edge_D = 0;
    if constexpr (C >= WIDE_CHUNK && (SIGMAG || LM == LIST_REGISTER_RECORDS || LM == LIST_STORE_POOLED)) {
        if (!fast && a.all_staged && as_const_ints(a.n_invalid)[0] == 0) {
            const SearchCold* cold = a.cold;
            const uint4* tab = cold->edge_tab;
            if (tab != nullptr && as_const_ints(cold->edge_ok)[0] != 0) {
                edge_tab = tab;
                edge_D = cold->edge_D;
            }
        }
    }
#ifdef KB_EXP_ALL_FAST
    if (true) {
#else
    if (fast || edge_tab != nullptr) {
#endif
        lds_search_tile<KS, C, ROWS, NB, CANON, SIGMAG, LM, true>(a, tc, smem, lists, edge_tab, edge_D);
    } else if constexpr (C == XWIDE_CHUNK) {
        // The instance for chunks of 32 holds no counting loop (64 accumulators leave no registers for one: compiled in, it put
        // 500 bytes per lane into scratch memory, and a launch with that much scratch pays tens of milliseconds for it).  The
        // host launches it only after it has seen that no tile needs one -- no NO_DATA pixel (read back), every epoch staged
        // with uniform shifts, start pixels on the image, edge tables built, shifts monotone (kb_shift_table_kernel).  Should
        // a tile need one all the same (a drift between the host's checks and this predicate), the launch fails SOFT: the
        // workgroup raises the refusal word of the tables' counter block and leaves its slots alone; the host reads the word
        // back behind the launch and reports an error instead of results (KB_XWIDE_TRAP: abort the context instead, debugging).
#ifdef KB_XWIDE_TRAP
        __builtin_trap();
#else
        if (threadIdx.x == 0) atomicExch(const_cast<int*>(a.global_box) + XWIDE_REFUSAL_WORD, 1);
        return;
#endif
    } else {
        lds_search_tile<KS, C, ROWS, NB, CANON, SIGMAG, LM, false>(a, tc, smem, lists);
    }
#ifdef KB_EXP_NO_RESULTS  // (timing experiment: what the result stores cost; wrong results)
    if (a.K == 12345)
#endif
    if constexpr (!SIGMAG) {
        if constexpr (TileLists<KS, LM>::STORE_POOLED) {
            // (16 slots x 7 dwords + 1 per lane, 32 lanes: 14.5 KB per wave -- eleven waves' worth fits 160 KB, so the sixteen
            // waves of a 64 x 16 tile go in two turns)
            constexpr int PATCH = 32 * (7 * 16 + 1) * 4;
            constexpr int TURNS = (ROWS * PATCH + 2 * lds_group_bytes(ROWS) - 1) / (2 * lds_group_bytes(ROWS));
#pragma unroll
            for (int turn = 0; turn < TURNS; ++turn) {
                const int per_turn = (ROWS + TURNS - 1) / TURNS;
                if (tc.wv / per_turn == turn) {
                    write_results_pooled(a, tc, lists.state, lists.store, PooledLayout{(uint32_t)(ROWS * WAVE)}, threadIdx.x,
                                         smem + (size_t)(tc.wv % per_turn) * PATCH);
                }
                if (TURNS > 1) __syncthreads();
            }
        } else if constexpr (TileLists<KS, LM>::STORED) {
            write_results_stored<TileLists<KS, LM>::RECORDS>(a, tc, lists.state, lists.store, TileLists<KS, LM>::SLOT_BYTES * threadIdx.x,
                                                         ROWS * WAVE * TileLists<KS, LM>::SLOT_BYTES);
        } else if constexpr (TileLists<KS, LM>::PACKED) {
            // (every wave has left the summing loop behind a barrier: the group buffers are free; 32 lanes x (7 K + 1) dwords
            // per wave -- 7.3 KB for K = 8 -- fit sixteen times into them)
            write_packed<KS>(a, tc, lists.packed, smem + (size_t)tc.wv * (32 * (7 * KS + 1) * 4));
        } else {
            write_results<KS>(a, tc, lists.top);
        }
    }
}

// Launch of one instance (two group buffers beyond the default 64 KiB of dynamic LDS need the attribute raised).
template <int KS, int ROWS, int NB, bool CANON, bool SIGMAG, int LM, int C = CHUNK>
static void launch_lds(const SearchArgs& a, hipStream_t stream) {
    constexpr size_t lds_bytes = 2 * lds_group_bytes(ROWS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kb_search_lds<KS, C, ROWS, NB, CANON, SIGMAG, LM>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL((kb_search_lds<KS, C, ROWS, NB, CANON, SIGMAG, LM>), dim3(a.n_tiles), dim3(ROWS * WAVE), lds_bytes,
                       stream, a);
    char name[96];
    std::snprintf(name, sizeof(name), "kb::kb_search_lds<%d, %d, %d, %d, %s, %s, %d>", KS, C, ROWS, NB, CANON ? "true" : "false",
                  SIGMAG ? "true" : "false", LM);
    note_kernel_instance(name);
}

}  // namespace kb
#endif
