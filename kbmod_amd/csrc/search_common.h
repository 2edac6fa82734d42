// Types and device helpers shared by the translation units of the trajectory search
// (search_kernels.hip: shift table, kb_search_lds / kb_search_direct / kb_search_large_k and the
// launcher; sigmag_kernels.hip: the sigma-G resolve passes).
#ifndef KB_SEARCH_COMMON_H_
#define KB_SEARCH_COMMON_H_

#include "kb_common.h"
#include "search_math.h"

#pragma clang fp contract(off)

namespace kb {

constexpr int SHIFT_UNSAFE = INT32_MIN;  // dx marker: no uniform shift proven for this (candidate, epoch)
// Tile geometry of the search kernels.  A workgroup owns 64 x ROWS start pixels, one wavefront per row
// (ROWS is a template parameter of the kernels, a runtime value in the tables built for them), and
// accumulates CHUNK candidates at a time.  kb_search_lds runs 64 x 16 tiles (one 16-wave workgroup per
// CU: the taller the tile, the less of each staged slab is apron) for lists of up to 8 results per pixel
// and 64 x 8 tiles (register budget of 2 waves per SIMD) for longer ones; the direct kernels, which
// share nothing between their waves, keep 64 x 4.
#ifndef KB_CHUNK
#define KB_CHUNK 8
#endif
constexpr int CHUNK = KB_CHUNK;    // candidates accumulated together per wave
// ... and by the packed-list float-staged instance of kb_search_lds (K <= 8, one slab in flight): twice the candidates per
// staged slab halve the passes over the stack, and with them the bytes that cross the fabric (what bounds that kernel)
constexpr int WIDE_CHUNK = 2 * CHUNK;
// ... and by the instance for arrays beyond the 256 MiB Infinity Cache (packed lists, K <= 8, a stack without NO_DATA pixels):
// there the kernel is bound by the bytes that cross the fabric -- N_candidates / chunk passes over the padded copy --, and four
// times the candidates per staged slab are worth more than sixteen samples in flight (64 accumulator registers leave eight)
constexpr int XWIDE_CHUNK = 4 * CHUNK;
// (SearchArgs::global_box[XWIDE_REFUSAL_WORD], the last int of the tables' counter block: raised by a tile of the count-free
// chunk-of-32 instance that would have had to count samples -- search_lds.h; read back by the host behind that launch)
constexpr int XWIDE_REFUSAL_WORD = 9;
#ifndef KB_DIRECT_ROWS
#define KB_DIRECT_ROWS 4
#endif
constexpr int DIRECT_ROWS = KB_DIRECT_ROWS;  // kb_search_direct / kb_search_large_k
constexpr int LDS_ROWS_TALL = 16;  // kb_search_lds, K <= 8 (and the sigma-G emit)
constexpr int LDS_ROWS_WIDE_K = 8; // kb_search_lds, 8 < K <= 32; also small search areas
#ifndef KB_TILE_GROUP_ROWS
#define KB_TILE_GROUP_ROWS 4
#endif
constexpr int TILE_GROUP_ROWS = KB_TILE_GROUP_ROWS;  // tile rows walked together by an XCD (tile_coords)
__host__ __device__ constexpr int block_threads(int rows) { return rows * WAVE; }
__host__ __device__ constexpr int stage_round(int rows) { return rows * WAVE * 16; }  // bytes one staging round moves
#ifndef KB_LDS_GROUP_PER_ROW
#define KB_LDS_GROUP_PER_ROW 5120
#endif
__host__ __device__ constexpr int lds_group_bytes(int rows) { return KB_LDS_GROUP_PER_ROW * rows; }   // one of the two group buffers
// Epochs per group: as many slabs of `stride` bytes as a group buffer holds.  The hand-scheduled instances (search_lds_asm.h)
// take an even number when there are two or more: their run of whole groups works in pairs of epochs.
__host__ __device__ inline int group_epochs(int T, int rows, int stride, bool even) {
    int E = lds_group_bytes(rows) / stride;
    if (E > T) E = T;
    if (E < 1) E = 1;
    if (even && E >= 2) E &= ~1;
    if (even && E > 16) E = 16;  // (their counting form keeps 32 -- chunks of 16: 16 -- samples' NO_DATA bits per register between two tallies)
    return E;
}

// LDS staging (kb_search_lds): per (chunk, epoch) the workgroup stages a slab of
// rows_max(chunk) x cols(chunk) raw pairs -- the union footprint of its 64 x ROWS tile
// under the chunk's shifts -- from a padded HBM copy of the array into a ring of slab slots in LDS.
#ifndef KB_LDS_COLS
#define KB_LDS_COLS 88
#endif
constexpr int LDS_COLS = KB_LDS_COLS;  // widest slab in pixels: 64 start columns + up to 24 of dx spread (each chunk's slabs
                                      // are as wide as its own spread needs: ChunkInfo::cols)
                                      // (88 * {8,4,2} bytes are multiples of the 16-byte DMA granule)
#ifndef KB_LDS_ALIGN_PX
#define KB_LDS_ALIGN_PX 2
#endif
// ... 64 + up to 48 for chunks of XWIDE_CHUNK candidates (a whole row of speeds of the reference's grids: 35 pixels of spread a
// day at 5 .. 40 pixels a day); the padded frame is sized for this width whatever the chunk, so that one copy serves every search
constexpr int LDS_COLS_XWIDE = 112;
__host__ __device__ constexpr int lds_cols(int chunk) { return chunk >= XWIDE_CHUNK ? LDS_COLS_XWIDE : LDS_COLS; }
constexpr int LDS_ALIGN_PX = KB_LDS_ALIGN_PX;  // slab origins are multiples of this many columns of the padded frame
#ifndef KB_LDS_SLOTS
#define KB_LDS_SLOTS 2
#endif
constexpr int LDS_SLOTS = KB_LDS_SLOTS;  // 16-byte pieces a thread holds in registers at once; slabs beyond that many
                                         // rounds are copied in further, non-overlapped rounds
constexpr int SLAB_REF_SLACK = 8;       // valid slab references behind the table's last entry (prefetched, never used)

struct ChunkInfo {
    int dx_min, dx_max, dy_min, dy_max;  // bounding box of the chunk's shifts over all epochs
    int unsafe;                          // any entry flagged SHIFT_UNSAFE
    int lds_ok;                          // every epoch's footprint fits one LDS slab
    int rows_max;                        // tile rows + largest dy spread of any epoch (slab height)
    int cols;                            // slab pitch: 64 + largest dx spread of any staged epoch, rounded up to the
                                         // staging quantum (16 bytes of raw pairs)
    // What kb_search_lds would otherwise divide out per chunk and loop trip on every wave (float-pair staging, the tile height
    // the tables were built for): epochs per group in the even and in the any-number form (group_epochs), the whole groups
    // T holds of each, and ceil(2^20 / cols), the reciprocal stage_lanes multiplies by.
    int e_even, e_any, t_over_e_even, t_over_e_any, cols_inv;
    int pad[3];
};
static_assert(sizeof(ChunkInfo) == 64, "kb_search_lds reads ChunkInfo as sixteen ints");

// Per (chunk, epoch) footprint, packed for one 8-byte scalar load:
//   x = (dy_min << 16) | (dx_min & 0xffff)   origin of the staged region relative to the tile
//   y = (rows   << 16) | cols                64 + dx spread, tile rows + dy spread
using EpochBox = int2;
constexpr int BOX_NOT_STAGED = (int)0x80008000u;  // word 0 of an epoch that kb_search_lds does not stage
constexpr int LDS_OFF_UNSTAGED = -1;
constexpr int LDS_OFF_PER_LANE = -2;
__host__ __device__ __forceinline__ int box_dx(EpochBox b) { return (int)(short)(b.x & 0xffff); }
__host__ __device__ __forceinline__ int box_dy(EpochBox b) { return b.x >> 16; }
__host__ __device__ __forceinline__ int box_cols(EpochBox b) { return b.y & 0xffff; }
__host__ __device__ __forceinline__ int box_rows(EpochBox b) { return b.y >> 16; }

// Per (chunk, epoch): where the slab starts in the padded copy (byte offset relative to the tile's own
// pixel) and how large it is; 16 bytes, one scalar load.
struct SlabRef {
    int64_t origin;
    int32_t bytes;
    int32_t pad;
};

// Sigma-G resolve (sigmag_kernels.hip).  With the in-search sigma-G filter the search kernels do not
// keep a top-K: per (row of 64 start pixels, candidate) they emit the ballot of the lanes that pass the
// unclipped thresholds (kernels.cu:201-203) as one work item; kb_sigmag_clip_kernel clips those
// trajectories spread evenly over the whole device, kb_sigmag_select_kernel runs the per-pixel
// insertion over the clipped likelihoods in candidate order.
struct SgEntry {
    uint32_t row;   // y_i * tiles_x + tx: 64 consecutive start pixels of one row
    uint32_t cand;  // candidate index
    uint64_t mask;  // lanes (start pixels) whose trajectory is clipped
};
struct SigmaGWork {
    uint32_t* slots;   // [rows][batch_cands]: entry index + 1 of (row, candidate), 0 = nothing passed
    SgEntry* entries;  // [capacity]
    int* n_entries;    // device counter of the batch in flight
    unsigned long long* totals;  // {work items, trajectories, trajectories clipped by the literal code} of the whole search
    float* lh;         // [capacity][64] clipped likelihood of the entry's lanes
    float* flux;       // [capacity][64]
    int* obs;          // [capacity][64]
    int batch_cands;   // candidates per batch (row pitch of slots)
};

// Where a per-pixel list goes: 28-byte trajectories (kb_device_search_filter) or the 16-byte records of
// kb_device_search_compact.  Exactly one pointer is set.
struct ResultSink {
    kb_trajectory* full;
    kb_compact_result* compact;
    int cand_base;  // added to the candidate index of a compact record
    // Optional (kb_device_search_counted; the epilogues of kb_search_lds with packed or pooled lists honour it): one byte per
    // start pixel = the number of its records that survive the likelihood post-filter (cand >= 0 and not lh < keep_min_lh, a
    // prefix of the sorted list) -- the header of the sparse exchange form, written by the search itself --, and a wave whose 64
    // pixels keep nothing does NOT write its run of records at all.
    uint8_t* counts;
    float keep_min_lh;
};
// cand < 0: the placeholder of an empty slot (kernels.cu:293-301).
__device__ __forceinline__ void store_result(const ResultSink& sink, size_t slot, const kb_trajectory& res, int cand) {
    if (sink.compact != nullptr) {
        kb_compact_result r;
        r.lh = res.lh;
        r.flux = res.flux;
        r.cand = (cand < 0) ? -1 : sink.cand_base + cand;
        r.obs_count = res.obs_count;
        sink.compact[slot] = r;
    } else {
        sink.full[slot] = res;
    }
}
__device__ __forceinline__ kb_trajectory placeholder_result(int x, int y) {
    kb_trajectory p;
    p.x = x;
    p.y = y;
    p.vx = 0.0f;
    p.vy = 0.0f;
    p.lh = -FLT_MAX;
    p.flux = 0.0f;
    p.obs_count = 0;
    return p;
}

// Arguments of the search kernels, in two parts.  SearchArgs travels by value (kernel-argument
// segment -> SGPRs) and holds only what the accumulation loops touch; everything else -- the
// epilogue, the rare per-lane paths, the sigma-G emit -- sits in a SearchCold block in device memory
// and is fetched through `cold` at the point of use.  (One 440-byte by-value struct kept 50-60
// scalar registers spilled through VGPR lanes inside the loop.)
struct SearchCold {
    kb_psi_phi_meta meta;
    kb_search_params params;
    const double* times;
    const kb_trajectory* cands;
    ResultSink results;
    const EpochBox* boxes;     // [n_chunks][T]
    int Hp, px0, py0;          // padded height, position of image pixel (0,0) inside the padded frame
    int fast_decode;           // uint8/uint16: the fp32-FMA decode was verified bit-identical for every code
    float* sg_scratch;         // sigma-G per-lane scratch of the literal clip, or null
    SigmaGWork sg;
    // Observation counts of tiles at the image's edge without counting samples (kb_edge_count_kernel; null when not built):
    // [chunk][4: x towards +, x towards -, y towards +, y towards -][d = 0 .. edge_D][WIDE_CHUNK counts as uint16, candidate c
    // in half c & 1 of word c >> 1] = epochs of candidate c whose shift along that axis and direction is at most d pixels.
    const uint4* edge_tab;
    const int* edge_ok;        // device flag: every candidate's shifts grow monotonically along both axes (see there)
    int edge_D;
};

struct SearchArgs {
    const SearchCold* cold;
    const void* psi_phi;
    const void* padded;        // [T][Hp][Wp] raw pairs, apron = NO_DATA (kb_search_lds only)
    const int2* table;         // [n_chunks][T][C] integer shifts (dx, dy)
    const ChunkInfo* chunks;   // [n_chunks]
    const SlabRef* slabs;      // [n_chunks][T] slab origin inside the padded copy and slab size (kb_slab_ref_kernel)
    const int* lds_off;        // [n_chunks][T][C] byte offset of the shifted tile inside the slab
    const int* lds_fold;       // [n_chunks][T][C] (+ slack) lds_off with the place of the epoch's slab in its group buffer folded in
                               // (float-staged kernels: the lane's read pointer never moves inside the hand-scheduled loop)
    const int* global_box;     // {dx_min, dx_max, dy_min, dy_max, rows_max} over every (candidate, epoch)
    const int* n_invalid;      // device counter: non-zero when the image holds NO_DATA pixels (kb_pad_kernel)
    uint2* lists;              // kb_search_lds: per-pixel lists between chunks, [tile][slot][thread of the tile] of (lh bits, candidate)
    int T, W, H, Wp;
    int n_cands, n_chunks;
    int chunk;                 // candidates per chunk the tables were built for: CHUNK, or WIDE_CHUNK for the instances that take it
    int chunk_lo, chunk_hi;    // candidate chunks [chunk_lo, chunk_hi) of this launch
    int sw, sh;
    int tiles_x, tiles_y, n_tiles;
    int x_start_min, y_start_min;
    int K;
    int min_obs;
    float min_lh;
    int all_staged;            // every (chunk, epoch) is staged through LDS
    int force_exact;
    int stable_lists;          // per-pixel lists by stable insertion (flag 512: the tie-exact exchange between devices)
    float psi_scale, psi_min_val, phi_scale, phi_min_val;  // decode of encoded samples
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
        *psi = (r.x == 0) ? NAN : decode_code((float)r.x, a.psi_scale, a.psi_min_val);
        *phi = (r.y == 0) ? NAN : decode_code((float)r.y, a.phi_scale, a.phi_min_val);
    }
};
template <>
struct RawPair<1> {
    using type = uchar2;
    __device__ static __forceinline__ type invalid() { return make_uchar2(0, 0); }
    __device__ static __forceinline__ void decode(type r, const SearchArgs& a, float* psi, float* phi) {
        *psi = (r.x == 0) ? NAN : decode_code((float)r.x, a.psi_scale, a.psi_min_val);
        *phi = (r.y == 0) ? NAN : decode_code((float)r.y, a.phi_scale, a.phi_min_val);
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
        *psi = ok ? fmaf((float)r.x - 1.0f, a.psi_scale, a.psi_min_val) : NAN;
        *phi = fmaf((float)r.y - 1.0f, a.phi_scale, a.phi_min_val);
    }
};
template <>
struct RawPair<10> {
    using type = uchar2;
    __device__ static __forceinline__ type invalid() { return make_uchar2(0, 0); }
    __device__ static __forceinline__ void decode(type r, const SearchArgs& a, float* psi, float* phi) {
        const bool ok = (r.x != 0) && (r.y != 0);
        *psi = ok ? fmaf((float)r.x - 1.0f, a.psi_scale, a.psi_min_val) : NAN;
        *phi = fmaf((float)r.y - 1.0f, a.phi_scale, a.phi_min_val);
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

template <int ROWS>
__device__ __forceinline__ TileCoords tile_coords(const SearchArgs& a, int b) {
    // XCD-aware tile order: workgroup b runs on XCD (b % 8); give each XCD a
    // contiguous band of tiles so that its private L2 sees one image region.
    TileCoords c;
    const int xcd = b & 7, local = b >> 3;
    const int q = a.n_tiles >> 3, r = a.n_tiles & 7;
    const int tile = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    // Within an XCD's range the tiles run down groups of TILE_GROUP_ROWS tile rows, column after column: the ~32
    // tiles an XCD holds at any time form a block (8 columns x 4 rows) whose vertical aprons overlap inside that
    // XCD's L2, instead of one long row of tiles whose vertical neighbours come a whole pass later.
    // (a group spans TILE_GROUP_ROWS x 16 image rows whatever the tile height: 8 tile rows of 64 x 8 tiles -- the cfg4 share
    // 19.9 -> 19.7 ms against groups of 4)
    constexpr int GROUP = (TILE_GROUP_ROWS * 16 / ROWS) > 0 ? (TILE_GROUP_ROWS * 16 / ROWS) : 1;
    const int per_group = GROUP * a.tiles_x;
    const int g = tile / per_group, in_group = tile - g * per_group;
    const int rows_here = min(GROUP, a.tiles_y - g * GROUP);
    c.tx = in_group / rows_here;
    c.ty = g * GROUP + (in_group - c.tx * rows_here);
    c.lane = threadIdx.x & (WAVE - 1);
    c.wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.y_i = c.ty * ROWS + c.wv;
    c.x_i = c.tx * WAVE + c.lane;
    c.x = c.x_i + a.x_start_min;
    c.y = c.y_i + a.y_start_min;
    c.tile_x0 = c.tx * WAVE + a.x_start_min;
    c.tile_y0 = c.ty * ROWS + a.y_start_min;
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
    // The list of a thread in the lane-interleaved store of kb_search_lds: slot s of the tile at
    // tile_list + s * stride_bytes (uniform: scalar arithmetic) plus the thread's own 32-bit offset -- one
    // address register for the whole list instead of a 64-bit pointer per slot.
    __device__ __forceinline__ void load(const char* tile_list, uint32_t lane_off, int stride_bytes) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            uint32_t off = lane_off;
            asm volatile("" : "+v"(off));
            const uint2 v = *reinterpret_cast<const uint2*>(tile_list + (size_t)s * stride_bytes + off);
            lh[s] = __uint_as_float(v.x);
            id[s] = (int)v.y;
        }
    }
    __device__ __forceinline__ void store(char* tile_list, uint32_t lane_off, int stride_bytes) const {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            uint32_t off = lane_off;
            asm volatile("" : "+v"(off));
            *reinterpret_cast<uint2*>(tile_list + (size_t)s * stride_bytes + off) =
                    make_uint2(__float_as_uint(lh[s]), (uint32_t)id[s]);
        }
    }
    // kernels.cu:323-330: strict '>' swap-down, reproduced slot by slot.  `stable` (uniform; the tie-exact
    // exchange between devices, kb_merge_compact_exact): once the candidate has found its slot everything below
    // shifts down by one whatever its value, i.e. the list is the top K by (likelihood descending, candidate
    // ascending) -- a total order, which lists of several devices can be merged under.
    __device__ __forceinline__ void insert(float cand_lh, int cand, bool stable = false) {
        if (cand_lh > lh[KS - 1]) {
            float cl = cand_lh;
            int cid = cand;
            bool placed = false;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bool g = (cl > lh[s]) || (stable && placed);
                placed = placed || g;
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

// A list of whole results -- likelihood, candidate, flux, observation count -- as 16-byte records of the list store
// of kb_search_lds: what the search has in hand when a candidate is inserted is exactly what the result needs
// (lh and flux come from the same two sums, kernels.cu:186-190), so lists of records spare the epilogue its
// re-evaluation of every winner.
template <int KS>
struct TopKRecords {
    float lh[KS];
    int id[KS];
    float flux[KS];
    int obs[KS];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            lh[s] = -FLT_MAX;
            id[s] = -1;
            flux[s] = 0.0f;
            obs[s] = 0;
        }
    }
    __device__ __forceinline__ void load(const char* tile_list, uint32_t lane_off, int stride_bytes) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            uint32_t off = lane_off;
            asm volatile("" : "+v"(off));
            const uint4 v = *reinterpret_cast<const uint4*>(tile_list + (size_t)s * stride_bytes + off);
            lh[s] = __uint_as_float(v.x);
            id[s] = (int)v.y;
            flux[s] = __uint_as_float(v.z);
            obs[s] = (int)v.w;
        }
    }
    __device__ __forceinline__ void store(char* tile_list, uint32_t lane_off, int stride_bytes) const {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            uint32_t off = lane_off;
            asm volatile("" : "+v"(off));
            *reinterpret_cast<uint4*>(tile_list + (size_t)s * stride_bytes + off) =
                    make_uint4(__float_as_uint(lh[s]), (uint32_t)id[s], __float_as_uint(flux[s]), (uint32_t)obs[s]);
        }
    }
    // kernels.cu:323-330: strict '>' swap-down, the whole record travels (`stable`: see TopK::insert)
    __device__ __forceinline__ void insert(float cand_lh, int cand, float cand_flux, int cand_obs, bool stable = false) {
        if (cand_lh > lh[KS - 1]) {
            float cl = cand_lh, cf = cand_flux;
            int cid = cand, co = cand_obs;
            bool placed = false;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bool g = (cl > lh[s]) || (stable && placed);
                placed = placed || g;
                const float tl = lh[s], tf = flux[s];
                const int ti = id[s], to = obs[s];
                lh[s] = g ? cl : tl;
                id[s] = g ? cid : ti;
                flux[s] = g ? cf : tf;
                obs[s] = g ? co : to;
                cl = g ? tl : cl;
                cid = g ? ti : cid;
                cf = g ? tf : cf;
                co = g ? to : co;
            }
        }
    }
};

// Whole results in registers at three words per slot: likelihood, flux, and candidate index | observation count << 16
// (both below 65535; the host checks).  What lists of up to 8 cost beyond (likelihood, candidate) pairs is 8
// registers, and the epilogue needs no re-evaluation.
template <int KS>
struct TopKPacked {
    static constexpr uint32_t EMPTY = 0xffffffffu;
    float lh[KS];
    float flux[KS];
    uint32_t io[KS];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            lh[s] = -FLT_MAX;
            flux[s] = 0.0f;
            io[s] = EMPTY;
        }
    }
    // kernels.cu:323-330: strict '>' swap-down (`stable`: see TopK::insert)
    __device__ __forceinline__ void insert(float cand_lh, float cand_flux, uint32_t cand_io, bool stable = false) {
        float cl = cand_lh, cf = cand_flux;
        uint32_t ci = cand_io;
        bool placed = false;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bool g = (cl > lh[s]) || (stable && placed);
            placed = placed || g;
            const float tl = lh[s], tf = flux[s];
            const uint32_t ti = io[s];
            lh[s] = g ? cl : tl;
            flux[s] = g ? cf : tf;
            io[s] = g ? ci : ti;
            cl = g ? tl : cl;
            cf = g ? tf : cf;
            ci = g ? ti : ci;
        }
    }
    // The same insertion with every slot finished before the next is looked at (the empty statement pins the order): left to
    // itself the compiler computes the eight carried triples first and applies them afterwards, twenty more live registers
    // than the three that travel -- next to the 64 sums of a chunk of 32 that put list registers into scratch memory, stored
    // and reloaded by every round.
    __device__ __forceinline__ void insert_in_sequence(float cand_lh, float cand_flux, uint32_t cand_io, bool stable = false) {
        float cl = cand_lh, cf = cand_flux;
        uint32_t ci = cand_io;
        bool placed = false;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bool g = (cl > lh[s]) || (stable && placed);
            placed = placed || g;
            const float tl = lh[s], tf = flux[s];
            const uint32_t ti = io[s];
            lh[s] = g ? cl : tl;
            flux[s] = g ? cf : tf;
            io[s] = g ? ci : ti;
            cl = g ? tl : cl;
            cf = g ? tf : cf;
            ci = g ? ti : ci;
            asm volatile("" : "+v"(cl), "+v"(cf), "+v"(ci), "+v"(lh[s]), "+v"(flux[s]), "+v"(io[s]));
        }
    }
};

// What kb_search_lds keeps in registers of a thread's list while it sums the next chunk: the likelihood a candidate
// has to beat, and whether the list exists in the store yet.
struct ListState {
    float threshold;
    int stored;
};

// Lane-interleaved scratch of the literal sigma-G clip for the wave in slot wave_slot of its launch
// (launches that use it are sized by resident waves, not by the search area): element i of a lane at
// base[i * 64].
__device__ __forceinline__ SigmaGScratch<WAVE> make_scratch(float* sg_scratch, int T, size_t wave_slot, int lane) {
    SigmaGScratch<WAVE> s;
    float* base = sg_scratch + wave_slot * (size_t)(4 * T) * WAVE + lane;
    s.psi.p = base;
    s.phi.p = base + (size_t)T * WAVE;
    s.lc.p = base + (size_t)2 * T * WAVE;
    s.idx.p = reinterpret_cast<int*>(base + (size_t)3 * T * WAVE);
    return s;
}

template <int ROWS>
__device__ __forceinline__ TileCoords tile_coords(const SearchArgs& a) {
    return tile_coords<ROWS>(a, (int)blockIdx.x);
}

// Words of a scratch slot.
__host__ __device__ constexpr size_t scratch_words_per_wave(int T) { return (size_t)(4 * T) * WAVE; }

// Launch geometry of the sigma-G resolve, implemented in sigmag_kernels.hip.
// Clips every entry emitted by the search launch and merges the batch into the per-pixel lists:
// prev (may be null for the first batch) -> next.
// Launchers of the search kernels (search_direct.hip, search_lds.hip, search_lds_encoded.hip).  fmt: 4 = float,
// 2 / 1 = encoded with the reference's double-precision decode, 20 / 10 = encoded with the verified single-FMA decode.
void launch_search_direct(const SearchArgs& a, int fmt, bool sigmag, bool records, hipStream_t stream);
void launch_search_large_k(const SearchArgs& a, bool sigmag, int blocks, hipStream_t stream);
void launch_search_lds_canon(const SearchArgs& a, int rows, bool sigmag, int list_mode, hipStream_t stream);
void launch_search_lds_encoded(const SearchArgs& a, int rows, int fmt, bool sigmag, hipStream_t stream);

// The launchers record which template instance they started, spelled as rocprofv3 prints it (kb_search_stats::kernel_name).
void note_kernel_instance(const char* name);

int launch_sigmag_resolve(const SearchArgs& a, const SearchCold& cold, const ResultSink* prev, const ResultSink& next,
                          int scratch_waves, hipStream_t stream);

}  // namespace kb
#endif
