// Sparse form of the multi-GPU exchange ("shrink traffic by compacting lh >= min_lh first").
//
// The reference removes results with lh < min_lh only AFTER its kernel (stack_search.cpp:266-270); inside the kernel a
// candidate below min_lh still enters the per-pixel list (kernels.cu:318-321 tests the likelihood only under sigma-G).
// But the swap-down insertion (kernels.cu:323-330) never lets a smaller value touch the part of a list at or above a
// larger one: a candidate c goes in front of the first entry below c and everything from there moves down, so the
// entries >= min_lh of the final list are exactly what the same insertion produces over the candidates >= min_lh alone,
// in the same order.  Dropping the records below min_lh BEFORE the exchange therefore changes nothing that survives the
// reference's own post-filter; what is dropped are the slots the post-filter would have removed (and the empty ones).
//
// A device's dense lists [n_pixels][list_len] (kb_device_search_compact) become
//   header: uint8 counts[n_pixels] (records kept per pixel, in list order), padded to a multiple of 16 bytes, followed by
//           the uint64 total (kb_sparse_header_bytes);
//   packed: the kept records, pixel after pixel.
// On a real survey stack nearly every pixel keeps nothing (cfg4: 128 x 4096 x 4096, min_lh = 10 -> 16.8 MB of counts and
// a few MB of records per device instead of 4.3 GB).  The root turns counts into offsets per block of 256 pixels and
// runs the tie-exact merge (merge_exact_pixel, search_math.h) through that indirection: same routine, same result as
// kb_merge_compact_exact on the dense lists wherever a record survives the post-filter.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cstdint>
#include <mutex>
#include <string>

#include "kb_common.h"
#include "search_math.h"

namespace kb {

constexpr int SPARSE_BLOCK = KB_SPARSE_BLOCK;  // pixels per block: the granularity of the record offsets
constexpr int SPARSE_MAX_LISTS = 64;

__device__ __forceinline__ bool sparse_keep(const kb_compact_result& r, float min_lh) {
    return r.cand >= 0 && !(r.lh < min_lh);  // stack_search.cpp:268 removes lh < min_lh; empty slots carry cand = -1
}

__device__ __forceinline__ kb_compact_result load_record(const kb_compact_result* p) {
    const uint4 q = *reinterpret_cast<const uint4*>(p);  // one 16-byte load (the entry points check the alignment)
    kb_compact_result r;
    r.lh = __uint_as_float(q.x);
    r.flux = __uint_as_float(q.y);
    r.cand = (int32_t)q.z;
    r.obs_count = (int32_t)q.w;
    return r;
}

// Exclusive scan of one value per thread over a 256-thread workgroup; *total = the sum (all threads).
__device__ __forceinline__ uint32_t block_scan_256(uint32_t v, uint32_t* wave_sums /* LDS, 4 words */, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t s = wave_sums[w];
        before += (w < wave) ? s : 0u;
        all += s;
    }
    __syncthreads();
    *total = all;
    return before + incl - v;
}

// PHASE 0: counts per pixel + the block's total.  PHASE 1: the kept records to their place (block_base known).
// One workgroup per block of 256 pixels; the 256 * list_len records of a block are read as one contiguous run.
template <int PHASE>
__global__ __launch_bounds__(256) void kb_sparsify_kernel(const kb_compact_result* __restrict__ lists, uint64_t n_pixels,
                                                          int L, float min_lh, uint8_t* __restrict__ counts,
                                                          uint32_t* __restrict__ block_totals,
                                                          const uint64_t* __restrict__ block_base,
                                                          kb_compact_result* __restrict__ packed) {
    __shared__ uint32_t mask[SPARSE_BLOCK];  // bit p: the pixel's record p is kept (list_len <= 32)
    __shared__ uint32_t wave_sums[4];
    const uint64_t pix0 = (uint64_t)blockIdx.x * SPARSE_BLOCK;
    const uint32_t n_here = (uint32_t)std::min<uint64_t>(SPARSE_BLOCK, n_pixels - pix0);
    if (PHASE == 1 && block_totals[blockIdx.x] == 0) return;
    mask[threadIdx.x] = 0;
    __syncthreads();
    const kb_compact_result* src = lists + pix0 * (uint64_t)L;
    const uint32_t n_rec = n_here * (uint32_t)L;
    for (uint32_t i = threadIdx.x; i < n_rec; i += 256) {
        if (sparse_keep(load_record(src + i), min_lh)) atomicOr(&mask[i / (uint32_t)L], 1u << (i % (uint32_t)L));
    }
    __syncthreads();
    const uint32_t c = threadIdx.x < n_here ? (uint32_t)__popc(mask[threadIdx.x]) : 0u;
    uint32_t total = 0;
    const uint32_t before = block_scan_256(c, wave_sums, &total);
    if (PHASE == 0) {
        if (threadIdx.x < n_here) counts[pix0 + threadIdx.x] = (uint8_t)c;
        if (threadIdx.x == 0) block_totals[blockIdx.x] = total;
        return;
    }
    __shared__ uint32_t first[SPARSE_BLOCK];  // where a pixel's records start, relative to the block's base
    first[threadIdx.x] = before;
    __syncthreads();
    const uint64_t base = block_base[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < n_rec; i += 256) {
        const uint32_t p = i / (uint32_t)L, pos = i % (uint32_t)L;
        const uint32_t m = mask[p];
        if ((m >> pos) & 1u) {
            const uint4 q = *reinterpret_cast<const uint4*>(src + i);
            *reinterpret_cast<uint4*>(packed + base + first[p] + __popc(m & ((1u << pos) - 1u))) = q;
        }
    }
}

// The scatter pass when the counts are GIVEN (the search wrote them: ResultSink::counts): a pixel keeps the first counts[p]
// records of its list -- the survivors of a sorted list are a prefix --, and only those are read: the runs of waves that
// kept nothing were never written.
__global__ __launch_bounds__(256) void kb_sparsify_counted_kernel(const kb_compact_result* __restrict__ lists, uint64_t n_pixels,
                                                                  int L, const uint8_t* __restrict__ counts,
                                                                  const uint32_t* __restrict__ block_totals,
                                                                  const uint64_t* __restrict__ block_base,
                                                                  kb_compact_result* __restrict__ packed) {
    __shared__ uint32_t wave_sums[4];
    if (block_totals[blockIdx.x] == 0) return;
    const uint64_t pix = (uint64_t)blockIdx.x * SPARSE_BLOCK + threadIdx.x;
    const uint32_t c = pix < n_pixels ? min((uint32_t)counts[pix], (uint32_t)L) : 0u;
    uint32_t total = 0;
    const uint32_t before = block_scan_256(c, wave_sums, &total);
    const kb_compact_result* src = lists + pix * (uint64_t)L;
    kb_compact_result* dst = packed + block_base[blockIdx.x] + before;
    for (uint32_t k = 0; k < c; ++k) *reinterpret_cast<uint4*>(dst + k) = *reinterpret_cast<const uint4*>(src + k);
}

// The same for whole trajectories (28-byte records: kb_filter_sort_results_counted).  A wave walks its pixels that keep
// something one after the other and copies each one's c x 7 dwords with all lanes: contiguous reads, contiguous writes.
__global__ __launch_bounds__(256) void kb_compact_counted_full_kernel(const kb_trajectory* __restrict__ lists, uint64_t n_pixels,
                                                                      int L, const uint8_t* __restrict__ counts,
                                                                      const uint32_t* __restrict__ block_totals,
                                                                      const uint64_t* __restrict__ block_base,
                                                                      kb_trajectory* __restrict__ packed) {
    __shared__ uint32_t wave_sums[4];
    if (block_totals[blockIdx.x] == 0) return;
    const uint64_t pix = (uint64_t)blockIdx.x * SPARSE_BLOCK + threadIdx.x;
    const uint32_t c = pix < n_pixels ? min((uint32_t)counts[pix], (uint32_t)L) : 0u;
    uint32_t total = 0;
    const uint32_t before = block_scan_256(c, wave_sums, &total);
    const int lane = threadIdx.x & 63;
    const uint64_t wave_pix0 = (uint64_t)blockIdx.x * SPARSE_BLOCK + (uint64_t)(threadIdx.x & ~63);
    const uint32_t* src0 = reinterpret_cast<const uint32_t*>(lists + wave_pix0 * (uint64_t)L);
    uint32_t* dst0 = reinterpret_cast<uint32_t*>(packed + block_base[blockIdx.x]);
    for (unsigned long long todo = __ballot(c != 0); todo != 0ull; todo &= todo - 1ull) {
        const int p = __builtin_ctzll(todo);
        const uint32_t n_dw = 7u * (uint32_t)__shfl((int)c, p), to = 7u * (uint32_t)__shfl((int)before, p);
        const uint32_t from = 7u * (uint32_t)L * (uint32_t)p;
        for (uint32_t i = (uint32_t)lane; i < n_dw; i += 64u) dst0[to + i] = src0[from + i];
    }
}

// Block totals -> exclusive uint64 bases, one workgroup per list (blockIdx.x); grand[list] = the list's total.
__global__ __launch_bounds__(1024) void kb_sparse_scan_kernel(const uint32_t* __restrict__ totals, uint64_t n_blocks,
                                                              uint64_t* __restrict__ bases, uint64_t* __restrict__ grand) {
    __shared__ unsigned long long part[1024];
    const uint32_t* t = totals + (uint64_t)blockIdx.x * n_blocks;
    uint64_t* b = bases + (uint64_t)blockIdx.x * n_blocks;
    const uint64_t per = (n_blocks + 1023) / 1024;
    const uint64_t lo = std::min<uint64_t>(n_blocks, per * threadIdx.x), hi = std::min<uint64_t>(n_blocks, lo + per);
    unsigned long long s = 0;
    for (uint64_t i = lo; i < hi; ++i) s += t[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele over the 1024 partial sums
        const unsigned long long add = threadIdx.x >= o ? part[threadIdx.x - o] : 0ull;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    unsigned long long run = part[threadIdx.x] - s;
    for (uint64_t i = lo; i < hi; ++i) {
        b[i] = run;
        run += t[i];
    }
    if (threadIdx.x == 1023 && grand != nullptr) grand[blockIdx.x] = part[1023];
}

// counts[list][pixel] -> totals[list][block]
__global__ __launch_bounds__(256) void kb_sparse_blocksum_kernel(const uint8_t* __restrict__ headers, uint64_t header_stride,
                                                                 uint64_t n_pixels, uint64_t n_blocks,
                                                                 uint32_t* __restrict__ totals, uint32_t cap) {
    __shared__ uint32_t wave_sums[4];
    const uint64_t blk = blockIdx.x, list = blockIdx.y;
    const uint64_t pix = blk * SPARSE_BLOCK + threadIdx.x;
    // (a count byte is the caller's: one above the list length is cut to it here and in every kernel that reads through the
    // counts, so that no record behind a pixel's list -- or behind the buffer -- is ever touched)
    const uint32_t c = pix < n_pixels ? min((uint32_t)headers[list * header_stride + pix], cap) : 0u;
    uint32_t total = 0;
    (void)block_scan_256(c, wave_sums, &total);
    if (threadIdx.x == 0) totals[list * n_blocks + blk] = total;
}

struct SparseLists {
    const kb_compact_result* packed[SPARSE_MAX_LISTS];
};

// The tie-exact merge of kb_merge_compact_exact_kernel (search_kernels.hip) reading every list through its counts.
// NL = upper bound of n_lists (the per-list cursors live in registers for NL = 8).
template <int NL>
__global__ __launch_bounds__(256) void kb_merge_sparse_exact_kernel(const uint8_t* __restrict__ headers, uint64_t header_stride,
                                                                    const SparseLists lists, const uint64_t* __restrict__ bases,
                                                                    uint64_t n_blocks, int n_lists, uint64_t n_pixels, int K2,
                                                                    int K, int sw, int x_min, int y_min,
                                                                    const kb_trajectory* __restrict__ all_cands,
                                                                    uint64_t n_all_cands, kb_trajectory* __restrict__ out,
                                                                    uint8_t* __restrict__ counts_out) {
    __shared__ uint32_t wave_sums[4];
    const uint64_t pix = (uint64_t)blockIdx.x * SPARSE_BLOCK + threadIdx.x;
    const bool live = pix < n_pixels;
    uint32_t cnt[NL];
    uint32_t off[NL];  // first record of this pixel in list r, relative to the block's base (< 256 * 32)
    uint32_t any = 0;
#pragma unroll
    for (int r = 0; r < NL; ++r) {
        cnt[r] = 0;
        off[r] = 0;
        if (r < n_lists) {  // (uniform)
            const uint32_t c = live ? min((uint32_t)headers[(uint64_t)r * header_stride + pix], (uint32_t)K2) : 0u;
            uint32_t total = 0;
            off[r] = block_scan_256(c, wave_sums, &total);
            cnt[r] = c;
            any += c;
        }
    }
    const int y_i = live ? (int)(pix / (uint64_t)sw) : 0, x_i = live ? (int)(pix - (uint64_t)y_i * (uint64_t)sw) : 0;
    // A wave none of whose 64 pixels is reached by any list -- nearly every wave of a thresholded search -- writes its
    // 64 x K placeholders as ONE contiguous run of 16-byte stores (the per-thread form below stores 7 dwords per slot at a
    // lane stride of K x 28 bytes: 1.7 TB/s for the 3.76 GB of a 4096 x 4096 search, where a plain fill reaches 4.5).
    // counts_out (kb_merge_sparse_exact_counted): the number of merged records per pixel instead of the placeholders -- a wave
    // nothing reaches writes its 64 zero bytes and no slot at all.
    if (counts_out != nullptr && __ballot(any != 0) == 0ull) {
        if (live) counts_out[pix] = 0;
        return;
    }
    if (__ballot(any != 0) == 0ull && (reinterpret_cast<uintptr_t>(out) & 15u) == 0) {
        const int lane = threadIdx.x & 63;
        const uint32_t n_live = (uint32_t)__popcll(__ballot(live));          // (live lanes are a prefix of the wave)
        const uint32_t n_dwords = n_live * 7u * (uint32_t)K;                  // < 64 * 7 * 32
        const uint32_t inv_k = (65536u + (uint32_t)K - 1u) / (uint32_t)K;     // slot / K == (slot * inv_k) >> 16 for slot < 2048, K <= 32
        const uint64_t wave_pix0 = (uint64_t)blockIdx.x * SPARSE_BLOCK + (uint64_t)(threadIdx.x & ~63);
        uint32_t* region = reinterpret_cast<uint32_t*>(out + wave_pix0 * (uint64_t)K);
        // (the wave's pixels follow its first one along the rows of the search area: no division, no cross-lane traffic)
        const int x0 = __builtin_amdgcn_readfirstlane(x_i), y0 = __builtin_amdgcn_readfirstlane(y_i);
        for (uint32_t q = (uint32_t)lane; 4u * q < n_dwords; q += 64u) {
            uint32_t w[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t d = 4u * q + (uint32_t)c;
                const uint32_t slot = (d * 37450u) >> 18;                     // d / 7 (exact below 14 344)
                const uint32_t field = d - 7u * slot;
                int ox = x0 + (int)((slot * inv_k) >> 16), oy = y0;           // the pixel this slot belongs to
                while (ox >= sw) {
                    ox -= sw;
                    oy += 1;
                }
                ox += x_min;
                oy += y_min;
                // kb_trajectory { vx, vy, lh, flux, x, y, obs_count } of an empty slot (kernels.cu:293-301)
                w[c] = field == 2u ? 0xff7fffffu /* -FLT_MAX */ : (field == 4u ? (uint32_t)ox : (field == 5u ? (uint32_t)oy : 0u));
            }
            if (4u * q >= n_dwords) {
                // nothing of this lane's quad lies inside the wave's run
            } else if (4u * q + 3u < n_dwords) {
                *reinterpret_cast<uint4*>(region + 4u * q) = make_uint4(w[0], w[1], w[2], w[3]);
            } else {
                for (uint32_t c = 0; 4u * q + c < n_dwords; ++c) region[4u * q + c] = w[c];
            }
        }
        return;
    }
    if (!live) return;
    kb_trajectory empty;
    empty.x = x_i + x_min;  // kernels.cu:293-301
    empty.y = y_i + y_min;
    empty.vx = 0.0f;
    empty.vy = 0.0f;
    empty.lh = -FLT_MAX;
    empty.flux = 0.0f;
    empty.obs_count = 0;
    kb_trajectory* dst = out + pix * (uint64_t)K;
    if (any == 0) {  // an empty pixel in a wave that holds a reached one
        for (int s = 0; s < K; ++s) dst[s] = empty;
        if (counts_out != nullptr) counts_out[pix] = 0;
        return;
    }
    auto read = [&](int r, int pos) {
        kb_compact_result rec;
        rec.lh = -FLT_MAX;
        rec.flux = 0.0f;
        rec.cand = -1;
        rec.obs_count = 0;
        uint32_t c = 0, o = 0;
#pragma unroll
        for (int q = 0; q < NL; ++q) {  // (register arrays: no dynamic index)
            c = q == r ? cnt[q] : c;
            o = q == r ? off[q] : o;
        }
        if ((uint32_t)pos < c) rec = load_record(lists.packed[r] + bases[(uint64_t)r * n_blocks + blockIdx.x] + o + pos);
        return rec;
    };
    MergedEntry merged[MERGE_EXACT_MAX_K2];
    int heads[SPARSE_MAX_LISTS];
    int slots[MERGE_EXACT_MAX_K2];
    const int n_out = merge_exact_pixel(read, n_lists, K2, K, merged, heads, slots);
    int n_valid = 0;  // (the merged list is filled from the top: the valid slots are a prefix)
    for (int s = 0; s < K; ++s) {
        kb_trajectory res = empty;
        if (s < n_out && slots[s] >= 0) {
            const uint32_t at = merged[slots[s]].at;
            const kb_compact_result rec = read((int)(at / (uint32_t)K2), (int)(at % (uint32_t)K2));
            if ((uint64_t)rec.cand < n_all_cands) {
                res.vx = all_cands[rec.cand].vx;
                res.vy = all_cands[rec.cand].vy;
                res.lh = rec.lh;
                res.flux = rec.flux;
                res.obs_count = rec.obs_count;
                n_valid = (n_valid == s) ? s + 1 : n_valid;
            }
        }
        dst[s] = res;
    }
    if (counts_out != nullptr) counts_out[pix] = (uint8_t)n_valid;
}

// ---- scratch (block totals and bases), one arena per device, kept between calls ----
constexpr int EXCHANGE_DEVICES = 64;
struct ExchangeArena {
    void* ptr = nullptr;
    size_t bytes = 0;
};
static std::mutex g_exchange_mutex[EXCHANGE_DEVICES];
static ExchangeArena g_exchange_arena[EXCHANGE_DEVICES];

static int exchange_device_slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    return (dev >= 0 && dev < EXCHANGE_DEVICES) ? dev : 0;
}

static int exchange_scratch(int slot, size_t bytes, void** out) {
    ExchangeArena& a = g_exchange_arena[slot];
    if (a.ptr != nullptr && a.bytes < bytes) {
        (void)hipFree(a.ptr);
        a = ExchangeArena();
    }
    if (a.ptr == nullptr) {
        KB_HIP_TRY(hipMalloc(&a.ptr, bytes));
        a.bytes = bytes;
    }
    *out = a.ptr;
    return 0;
}

// (called by kb_release_workspaces)
void release_exchange_arenas() {
    int prev = 0;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    for (int dev = 0; dev < EXCHANGE_DEVICES; ++dev) {
        std::lock_guard<std::mutex> lock(g_exchange_mutex[dev]);
        if (g_exchange_arena[dev].ptr != nullptr) {
            (void)hipSetDevice(dev);
            (void)hipFree(g_exchange_arena[dev].ptr);
            g_exchange_arena[dev] = ExchangeArena();
        }
    }
    if (have_prev) (void)hipSetDevice(prev);
}

}  // namespace kb

extern "C" {

uint64_t kb_sparse_header_bytes(uint64_t n_pixels) { return (n_pixels + 15) / 16 * 16 + 16; }

int kb_sparsify_compact(const kb_compact_result* lists_dev, uint64_t n_pixels, int32_t list_len, float min_lh,
                        uint8_t* header_dev, kb_compact_result* packed_dev, uint64_t packed_capacity,
                        uint64_t* total_out_host, void* stream_v) {
    using namespace kb;
    if (total_out_host == nullptr) return fail("sparsify_compact: null count pointer");
    *total_out_host = 0;
    if (lists_dev == nullptr || header_dev == nullptr) return fail("sparsify_compact: null pointer");
    if (list_len <= 0 || list_len > MERGE_EXACT_MAX_K2) return fail("sparsify_compact: lists of 1 to 32 records per pixel");
    if (n_pixels == 0) return fail("sparsify_compact: no pixels");
    if (((uintptr_t)lists_dev | (uintptr_t)packed_dev | (uintptr_t)header_dev) & 15u) {
        return fail("sparsify_compact: buffers must be aligned to 16 bytes");
    }
    KB_REQUIRE_DEVICE("the sparse exchange.");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    const uint64_t n_blocks = (n_pixels + SPARSE_BLOCK - 1) / SPARSE_BLOCK;
    if (n_blocks > 0x7fffffffull) return fail("sparsify_compact: too many pixels for one launch");
    const int slot = exchange_device_slot();
    std::lock_guard<std::mutex> lock(g_exchange_mutex[slot]);
    void* scratch = nullptr;
    if (exchange_scratch(slot, n_blocks * 12 + 64, &scratch)) return 1;
    uint64_t* bases = reinterpret_cast<uint64_t*>(scratch);
    uint32_t* totals = reinterpret_cast<uint32_t*>(static_cast<char*>(scratch) + n_blocks * 8);
    uint64_t* total_dev = reinterpret_cast<uint64_t*>(header_dev + (n_pixels + 15) / 16 * 16);
    // (the padding between the counts and the total travels: keep it defined)
    KB_HIP_TRY(hipMemsetAsync(header_dev + n_pixels, 0, kb_sparse_header_bytes(n_pixels) - n_pixels, stream));
    hipLaunchKernelGGL(kb_sparsify_kernel<0>, dim3((unsigned)n_blocks), dim3(256), 0, stream, lists_dev, n_pixels,
                       (int)list_len, min_lh, header_dev, totals, static_cast<const uint64_t*>(nullptr),
                       static_cast<kb_compact_result*>(nullptr));
    KB_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(kb_sparse_scan_kernel, dim3(1), dim3(1024), 0, stream, totals, n_blocks, bases, total_dev);
    KB_HIP_TRY(hipGetLastError());
    uint64_t total = 0;
    KB_HIP_TRY(hipMemcpyAsync(&total, total_dev, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    KB_HIP_TRY(hipStreamSynchronize(stream));
    *total_out_host = total;
    if (total == 0) return 0;
    if (packed_dev == nullptr || total > packed_capacity) {
        return fail("sparsify_compact: " + std::to_string(total) + " records kept, room for " +
                    std::to_string(packed_dev == nullptr ? 0 : packed_capacity));
    }
    hipLaunchKernelGGL(kb_sparsify_kernel<1>, dim3((unsigned)n_blocks), dim3(256), 0, stream, lists_dev, n_pixels,
                       (int)list_len, min_lh, header_dev, totals, bases, packed_dev);
    KB_HIP_TRY(hipGetLastError());
    KB_HIP_TRY(hipStreamSynchronize(stream));  // the scratch is free for the next call when this one returns
    return 0;
}

int kb_sparsify_counted(const kb_compact_result* lists_dev, uint64_t n_pixels, int32_t list_len, uint8_t* header_dev,
                        kb_compact_result* packed_dev, uint64_t packed_capacity, uint64_t* total_out_host, void* stream_v) {
    using namespace kb;
    if (total_out_host == nullptr) return fail("sparsify_counted: null count pointer");
    *total_out_host = 0;
    if (lists_dev == nullptr || header_dev == nullptr) return fail("sparsify_counted: null pointer");
    if (list_len <= 0 || list_len > MERGE_EXACT_MAX_K2) return fail("sparsify_counted: lists of 1 to 32 records per pixel");
    if (n_pixels == 0) return fail("sparsify_counted: no pixels");
    if (((uintptr_t)lists_dev | (uintptr_t)packed_dev | (uintptr_t)header_dev) & 15u) {
        return fail("sparsify_counted: buffers must be aligned to 16 bytes");
    }
    KB_REQUIRE_DEVICE("the sparse exchange.");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    const uint64_t n_blocks = (n_pixels + SPARSE_BLOCK - 1) / SPARSE_BLOCK;
    if (n_blocks > 0x7fffffffull) return fail("sparsify_counted: too many pixels for one launch");
    const int slot = exchange_device_slot();
    std::lock_guard<std::mutex> lock(g_exchange_mutex[slot]);
    void* scratch = nullptr;
    if (exchange_scratch(slot, n_blocks * 12 + 64, &scratch)) return 1;
    uint64_t* bases = reinterpret_cast<uint64_t*>(scratch);
    uint32_t* totals = reinterpret_cast<uint32_t*>(static_cast<char*>(scratch) + n_blocks * 8);
    uint64_t* total_dev = reinterpret_cast<uint64_t*>(header_dev + (n_pixels + 15) / 16 * 16);
    KB_HIP_TRY(hipMemsetAsync(header_dev + n_pixels, 0, kb_sparse_header_bytes(n_pixels) - n_pixels, stream));
    hipLaunchKernelGGL(kb_sparse_blocksum_kernel, dim3((unsigned)n_blocks, 1), dim3(256), 0, stream, header_dev, (uint64_t)0, n_pixels,
                       n_blocks, totals, (uint32_t)list_len);
    KB_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(kb_sparse_scan_kernel, dim3(1), dim3(1024), 0, stream, totals, n_blocks, bases, total_dev);
    KB_HIP_TRY(hipGetLastError());
    uint64_t total = 0;
    KB_HIP_TRY(hipMemcpyAsync(&total, total_dev, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    KB_HIP_TRY(hipStreamSynchronize(stream));
    *total_out_host = total;
    if (total == 0) return 0;
    if (packed_dev == nullptr || total > packed_capacity) {
        return fail("sparsify_counted: " + std::to_string(total) + " records kept, room for " +
                    std::to_string(packed_dev == nullptr ? 0 : packed_capacity));
    }
    hipLaunchKernelGGL(kb_sparsify_counted_kernel, dim3((unsigned)n_blocks), dim3(256), 0, stream, lists_dev, n_pixels, (int)list_len,
                       header_dev, totals, bases, packed_dev);
    KB_HIP_TRY(hipGetLastError());
    KB_HIP_TRY(hipStreamSynchronize(stream));
    return 0;
}

}  // extern "C"

namespace kb {
// counts[n_pixels] (records kept per start pixel, a prefix of its list of L trajectories: kb_device_search_filter_counted) ->
// the kept trajectories pixel after pixel in out_dev.  *total_host is valid whenever the call returns 0 or fails for want of
// room (capacity = 0 asks for the count alone); synchronises the stream.
int compact_counted_full(const kb_trajectory* lists_dev, uint64_t n_pixels, int L, const uint8_t* counts_dev, kb_trajectory* out_dev,
                         uint64_t capacity, uint64_t* total_host, hipStream_t stream) {
    *total_host = 0;
    const uint64_t n_blocks = (n_pixels + SPARSE_BLOCK - 1) / SPARSE_BLOCK;
    if (n_blocks > 0x7fffffffull) return fail("filter_sort_results_counted: too many pixels for one launch");
    const int slot = exchange_device_slot();
    std::lock_guard<std::mutex> lock(g_exchange_mutex[slot]);
    void* scratch = nullptr;
    if (exchange_scratch(slot, n_blocks * 12 + 64, &scratch)) return 1;
    uint64_t* bases = reinterpret_cast<uint64_t*>(scratch);
    uint32_t* totals = reinterpret_cast<uint32_t*>(static_cast<char*>(scratch) + n_blocks * 8);
    uint64_t* total_dev = reinterpret_cast<uint64_t*>(static_cast<char*>(scratch) + (n_blocks * 12 + 15) / 16 * 16);
    hipLaunchKernelGGL(kb_sparse_blocksum_kernel, dim3((unsigned)n_blocks, 1), dim3(256), 0, stream, counts_dev, (uint64_t)0, n_pixels,
                       n_blocks, totals, (uint32_t)L);
    KB_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(kb_sparse_scan_kernel, dim3(1), dim3(1024), 0, stream, totals, n_blocks, bases, total_dev);
    KB_HIP_TRY(hipGetLastError());
    uint64_t total = 0;
    KB_HIP_TRY(hipMemcpyAsync(&total, total_dev, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    KB_HIP_TRY(hipStreamSynchronize(stream));
    *total_host = total;
    if (total == 0 || capacity == 0) return 0;
    if (total > capacity) return fail("filter_sort_results_counted: " + std::to_string(total) + " records counted, room for " + std::to_string(capacity));
    hipLaunchKernelGGL(kb_compact_counted_full_kernel, dim3((unsigned)n_blocks), dim3(256), 0, stream, lists_dev, n_pixels, L, counts_dev,
                       totals, bases, out_dev);
    KB_HIP_TRY(hipGetLastError());
    KB_HIP_TRY(hipStreamSynchronize(stream));
    return 0;
}
}  // namespace kb

extern "C" {

int kb_merge_sparse_exact(const uint8_t* headers_dev, uint64_t header_stride, const kb_compact_result* const* packed_ptrs_host,
                          int32_t n_lists, int32_t list_len, kb_search_params params, const kb_trajectory* all_cands_dev,
                          uint64_t n_all_cands, kb_trajectory* out_dev, void* stream_v) {
    return kb_merge_sparse_exact_counted(headers_dev, header_stride, packed_ptrs_host, n_lists, list_len, params, all_cands_dev,
                                         n_all_cands, out_dev, nullptr, stream_v);
}

int kb_merge_sparse_exact_counted(const uint8_t* headers_dev, uint64_t header_stride, const kb_compact_result* const* packed_ptrs_host,
                                  int32_t n_lists, int32_t list_len, kb_search_params params, const kb_trajectory* all_cands_dev,
                                  uint64_t n_all_cands, kb_trajectory* out_dev, uint8_t* counts_out_dev, void* stream_v) {
    using namespace kb;
    if (headers_dev == nullptr || packed_ptrs_host == nullptr || out_dev == nullptr || all_cands_dev == nullptr) {
        return fail("merge_sparse_exact: null pointer");
    }
    if (n_lists <= 0 || n_lists > SPARSE_MAX_LISTS) return fail("merge_sparse_exact: unsupported number of lists");
    const int64_t sw = (int64_t)params.x_start_max - params.x_start_min;
    const int64_t sh = (int64_t)params.y_start_max - params.y_start_min;
    const int K = (int)params.results_per_pixel;
    if (sw <= 0 || sh <= 0) return fail("merge_sparse_exact: invalid search bounds");
    if (K <= 0 || list_len < std::max(K, 2 * K - 1) || list_len > MERGE_EXACT_MAX_K2) {
        return fail("merge_sparse_exact: lists of " + std::to_string(list_len) + " records per pixel for " + std::to_string(K) +
                    " results (need 2 K - 1 <= list length <= 32: the merge is exact from there on)");
    }
    const uint64_t n_pixels = (uint64_t)sw * (uint64_t)sh;
    if (header_stride < kb_sparse_header_bytes(n_pixels)) return fail("merge_sparse_exact: header stride shorter than a header");
    SparseLists lists{};
    for (int r = 0; r < n_lists; ++r) {
        lists.packed[r] = packed_ptrs_host[r];  // may be null for a list without records
        if ((uintptr_t)lists.packed[r] & 15u) return fail("merge_sparse_exact: record buffers must be aligned to 16 bytes");
    }
    KB_REQUIRE_DEVICE("the list merge.");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    const uint64_t n_blocks = (n_pixels + SPARSE_BLOCK - 1) / SPARSE_BLOCK;
    if (n_blocks > 0x7fffffffull) return fail("merge_sparse_exact: too many pixels for one launch");
    const int slot = exchange_device_slot();
    std::lock_guard<std::mutex> lock(g_exchange_mutex[slot]);
    void* scratch = nullptr;
    if (exchange_scratch(slot, (uint64_t)n_lists * n_blocks * 12 + 64, &scratch)) return 1;
    uint64_t* bases = reinterpret_cast<uint64_t*>(scratch);
    uint32_t* totals = reinterpret_cast<uint32_t*>(static_cast<char*>(scratch) + (uint64_t)n_lists * n_blocks * 8);
    hipLaunchKernelGGL(kb_sparse_blocksum_kernel, dim3((unsigned)n_blocks, (unsigned)n_lists), dim3(256), 0, stream, headers_dev,
                       header_stride, n_pixels, n_blocks, totals, (uint32_t)list_len);
    KB_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(kb_sparse_scan_kernel, dim3((unsigned)n_lists), dim3(1024), 0, stream, totals, n_blocks, bases,
                       static_cast<uint64_t*>(nullptr));
    KB_HIP_TRY(hipGetLastError());
    if (n_lists <= 8) {
        hipLaunchKernelGGL(kb_merge_sparse_exact_kernel<8>, dim3((unsigned)n_blocks), dim3(256), 0, stream, headers_dev,
                           header_stride, lists, bases, n_blocks, (int)n_lists, n_pixels, (int)list_len, K, (int)sw,
                           params.x_start_min, params.y_start_min, all_cands_dev, n_all_cands, out_dev, counts_out_dev);
    } else {
        hipLaunchKernelGGL(kb_merge_sparse_exact_kernel<SPARSE_MAX_LISTS>, dim3((unsigned)n_blocks), dim3(256), 0, stream,
                           headers_dev, header_stride, lists, bases, n_blocks, (int)n_lists, n_pixels, (int)list_len, K, (int)sw,
                           params.x_start_min, params.y_start_min, all_cands_dev, n_all_cands, out_dev, counts_out_dev);
    }
    KB_HIP_TRY(hipGetLastError());
    KB_HIP_TRY(hipStreamSynchronize(stream));  // the scratch is free for the next call when this one returns
    return 0;
}

}  // extern "C"
