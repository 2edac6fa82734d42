// Device-side post-processing of the per-pixel result slots: the likelihood /
// obs_count filters and the global descending sort that the reference runs on
// the host after every search (stack_search.cpp:266-281 with
// trajectory_list.cpp:96-126: filter_by_likelihood, filter_by_obs_count,
// sort_by_likelihood over up to S*K 28-byte structs).  Doing it in HBM means only
// the survivors cross PCIe.  Compaction and the radix sort come from rocPRIM
// (stable, so equal likelihoods keep their slot order -- one of the orders the
// reference's unstable sort may produce); the kernels around them are ours.
#include <algorithm>
#include <mutex>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <string>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>

#include "kb_common.h"
#include "wave_ops.h"

#pragma clang fp contract(off)

namespace kb {

struct KeepPredicate {
    float min_lh;
    int min_obs;
    // filter_by_likelihood removes lh < min_lh, filter_by_obs_count removes obs < min_obs
    __host__ __device__ bool operator()(const kb_trajectory& t) const {
        return !(t.lh < min_lh) && !(t.obs_count < min_obs);
    }
};

// Stable compaction of the records that pass the two filters, in two passes over blocks of 1024 records (rocprim::select moves
// 28-byte structs at 0.2 TB/s on this device -- 4.6 ms per GB of result slots, twice the search of a 2048 x 2048 stack):
// PHASE 0 counts per block, one scan over the block totals, PHASE 1 writes every survivor to its place.
constexpr int SELECT_BLOCK = 1024;
template <int PHASE>
__global__ __launch_bounds__(256) void kb_select_kernel(const kb_trajectory* __restrict__ in, uint64_t n, KeepPredicate pred,
                                                        uint32_t* __restrict__ block_totals,
                                                        const unsigned long long* __restrict__ block_base,
                                                        kb_trajectory* __restrict__ out) {
    __shared__ uint32_t wave_sums[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t i0 = (uint64_t)blockIdx.x * SELECT_BLOCK + (uint64_t)threadIdx.x * 4;  // four consecutive records per thread
    if (PHASE == 1 && block_totals[blockIdx.x] == 0) return;
    uint32_t keep = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (i0 + j < n) {
            kb_trajectory t;
            t.lh = in[i0 + j].lh;
            t.obs_count = in[i0 + j].obs_count;
            keep |= pred(t) ? (1u << j) : 0u;
        }
    }
    const uint32_t c = (uint32_t)__popc(keep);
    uint32_t incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    uint32_t before = incl - c, all = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        before += (w < wave) ? wave_sums[w] : 0u;
        all += wave_sums[w];
    }
    if (PHASE == 0) {
        if (threadIdx.x == 0) block_totals[blockIdx.x] = all;
        return;
    }
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + block_base[blockIdx.x] + before);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if ((keep >> j) & 1u) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(in + i0 + j);
#pragma unroll
            for (int w = 0; w < 7; ++w) dst[w] = src[w];
            dst += 7;
        }
    }
}

// block totals -> exclusive bases + the grand total (one workgroup)
__global__ __launch_bounds__(1024) void kb_select_scan_kernel(const uint32_t* __restrict__ totals, uint64_t n_blocks,
                                                              unsigned long long* __restrict__ bases,
                                                              unsigned long long* __restrict__ grand) {
    __shared__ unsigned long long part[1024];
    const uint64_t per = (n_blocks + 1023) / 1024;
    const uint64_t lo = n_blocks < per * threadIdx.x ? n_blocks : per * threadIdx.x, hi = n_blocks < lo + per ? n_blocks : lo + per;
    unsigned long long s = 0;
    for (uint64_t i = lo; i < hi; ++i) s += totals[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const unsigned long long add = threadIdx.x >= o ? part[threadIdx.x - o] : 0ull;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    unsigned long long run = part[threadIdx.x] - s;
    for (uint64_t i = lo; i < hi; ++i) {
        bases[i] = run;
        run += totals[i];
    }
    if (threadIdx.x == 1023) *grand = part[1023];
}

__global__ __launch_bounds__(256) void kb_extract_keys_kernel(const kb_trajectory* __restrict__ in, uint64_t n,
                                                              float* __restrict__ keys, uint32_t* __restrict__ idx) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        float lh = in[i].lh;
        // The host comparator (b.lh < a.lh) treats -0.0 and +0.0 as equal; make the radix order agree.
        keys[i] = (lh == 0.0f) ? 0.0f : lh;
        idx[i] = (uint32_t)i;
    }
}

// (first_invalid: the lowest output index whose record fails Trajectory::is_valid -- every float field finite, a non-negative
// count: common.h:82-86, what TrajectoryList::assert_valid walks the list for on the host --, or left alone)
__global__ __launch_bounds__(256) void kb_gather_kernel(const kb_trajectory* __restrict__ in,
                                                        const uint32_t* __restrict__ idx, uint64_t n,
                                                        kb_trajectory* __restrict__ out,
                                                        unsigned long long* __restrict__ first_invalid) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const kb_trajectory t = in[idx[i]];
    out[i] = t;
    if (first_invalid != nullptr) {
        const bool ok = __builtin_isfinite(t.vx) && __builtin_isfinite(t.vy) && __builtin_isfinite(t.lh) &&
                        __builtin_isfinite(t.flux) && t.obs_count >= 0;
        if (!ok) atomicMin(first_invalid, (unsigned long long)i);
    }
}

// psi/phi curves of a list of trajectories (stack_search.cpp:22-39, :302-318): out[i][0..T) = psi,
// out[i][T..2T) = phi, non-finite samples -> 0.  The index is floor() of the FLOAT position
// (Trajectory::get_x_index, common.h:71-79), unlike the search's double formula.
__device__ __forceinline__ int curve_index(int pos0, float vel, double time) {
    const float p = (float)__dadd_rn(__dadd_rn((double)pos0, __dmul_rn(time, (double)vel)), 0.5);
    return (int)floorf(p);
}

__global__ __launch_bounds__(256) void kb_curves_kernel(const kb_psi_phi_meta m, const void* __restrict__ arr,
                                                        const double* __restrict__ times,
                                                        const kb_trajectory* __restrict__ trjs, uint64_t n,
                                                        float* __restrict__ out) {
    const uint64_t T = m.num_times;
    const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * T) return;
    const uint64_t i = e / T, t = e - i * T;
    const kb_trajectory trj = trjs[i];
    const double tm = times[t];
    const int xi = curve_index(trj.x, trj.vx, tm), yi = curve_index(trj.y, trj.vy, tm);
    float psi = 0.0f, phi = 0.0f;
    if (xi >= 0 && yi >= 0 && (uint64_t)xi < m.width && (uint64_t)yi < m.height) {
        const uint64_t s = 2 * (m.pixels_per_image * t + (uint64_t)yi * m.width + (uint64_t)xi);
        float p, f;
        if (m.num_bytes == 4) {
            p = reinterpret_cast<const float*>(arr)[s];
            f = reinterpret_cast<const float*>(arr)[s + 1];
        } else {
            const float pc = (m.num_bytes == 1) ? (float)reinterpret_cast<const uint8_t*>(arr)[s]
                                                : (float)reinterpret_cast<const uint16_t*>(arr)[s];
            const float fc = (m.num_bytes == 1) ? (float)reinterpret_cast<const uint8_t*>(arr)[s + 1]
                                                : (float)reinterpret_cast<const uint16_t*>(arr)[s + 1];
            // psi_phi_array.cpp:195-202: double arithmetic, separately rounded
            p = (pc == 0.0f) ? NAN : (float)__dadd_rn(__dmul_rn(__dadd_rn((double)pc, -1.0), (double)m.psi_scale), (double)m.psi_min_val);
            f = (fc == 0.0f) ? NAN : (float)__dadd_rn(__dmul_rn(__dadd_rn((double)fc, -1.0), (double)m.phi_scale), (double)m.phi_min_val);
        }
        if (__builtin_isfinite(p)) psi = p;
        if (__builtin_isfinite(f)) phi = f;
    }
    out[i * 2 * T + t] = psi;
    out[i * 2 * T + T + t] = phi;
}

struct Scratch {
    void* p = nullptr;
    ~Scratch() {
        if (p) (void)hipFree(p);
    }
};


// ---------------------------------------------------------------------------
// batched sigma-G clipping of likelihood curves
// ---------------------------------------------------------------------------
// SigmaGClipping.compute_clipped_sigma_g_matrix (src/kbmod/filters/sigma_g_filter.py:114-168):
// per row the [low, 50, high] percentiles by torch.nanquantile's linear interpolation in float32
// (sort with NaN last, rank = q * (n_valid - 1), lerp), delta = max(high - low, 1e-5),
// bounds = median -+ n_sigma * coeff * delta, valid = isfinite(lh) && lower < lh < upper.
// One wavefront per row; the row is sorted as order-preserving 32-bit keys in LDS (bitonic).
constexpr int CLIP_ROWS_PER_BLOCK = 4;

__device__ __forceinline__ uint32_t float_sort_key(float v) {
    if (v != v) return 0xffffffffu;  // NaN after everything, +inf included
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(uint32_t k) {
    if (k == 0xffffffffu) return __uint_as_float(0x7fc00000u);
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
// ATen native/Lerp.h, scalar form, float32, separately rounded operations
__device__ __forceinline__ float lerp_aten(float a, float b, float w) {
    const float d = b - a;
    return (w < 0.5f) ? a + w * d : b - d * (1.0f - w);
}

__global__ __launch_bounds__(CLIP_ROWS_PER_BLOCK* WAVE) void kb_sigma_g_clip_kernel(
        const float* __restrict__ lh, uint64_t n_rows, int n_cols, int P, float q_low, float q_high, float scale,
        int clip_negative, uint8_t* __restrict__ valid) {
    extern __shared__ uint32_t keys_all[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = threadIdx.x / WAVE;
    const uint64_t row = (uint64_t)blockIdx.x * CLIP_ROWS_PER_BLOCK + wv;
    uint32_t* keys = keys_all + (size_t)wv * P;
    const bool active = row < n_rows;  // whole wave
    const float* src = lh + (active ? row : 0) * (uint64_t)n_cols;

    int n_valid = 0;
    for (int i = lane; i < P; i += WAVE) {
        float v = (active && i < n_cols) ? src[i] : __uint_as_float(0x7fc00000u);
        if (clip_negative && !(v > 0.0f)) v = __uint_as_float(0x7fc00000u);  // torch.where(lh > 0, lh, nan)
        n_valid += (v == v) ? 1 : 0;
        keys[i] = float_sort_key(v);
    }
    for (int o = WAVE / 2; o > 0; o >>= 1) n_valid += __shfl_xor(n_valid, o);
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < P; i += WAVE) {
                const int p = i ^ j;
                if (p > i) {
                    const uint32_t a = keys[i], b = keys[p];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) {
                        keys[i] = b;
                        keys[p] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    if (!active) return;
    float quant[3];
    const float qs[3] = {q_low, 0.5f, q_high};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float rank = qs[i] * (float)(n_valid - 1);
        if (rank < 0.0f) rank = 0.0f;
        const float below = floorf(rank);
        const int ib = (int)below, ia = (int)ceilf(rank);
        quant[i] = lerp_aten(key_to_float(keys[ib]), key_to_float(keys[ia]), rank - below);
    }
    float delta = quant[2] - quant[0];
    if (delta < 1e-5f) delta = 1e-5f;
    const float n_sigma_g = scale * delta;
    const float lower = quant[1] - n_sigma_g, upper = quant[1] + n_sigma_g;
    uint8_t* dst = valid + row * (uint64_t)n_cols;
    for (int i = lane; i < n_cols; i += WAVE) {
        const float v = src[i];
        dst[i] = (__builtin_isfinite(v) && v < upper && v > lower) ? 1 : 0;
    }
}

// Curves of up to 64 points (the usual stack depth): the row never leaves the registers.  One wavefront
// per row, lane i holds point i as an ordered key, the keys are sorted across the lanes by the DPP /
// permlane-swap network of wave_ops.h (no LDS, no barriers), the quantile neighbours are lane reads.
__global__ __launch_bounds__(CLIP_ROWS_PER_BLOCK* WAVE) void kb_sigma_g_clip64_kernel(
        const float* __restrict__ lh, uint64_t n_rows, int n_cols, float q_low, float q_high, float scale,
        int clip_negative, uint8_t* __restrict__ valid) {
    const int lane = threadIdx.x & (WAVE - 1);
    const uint64_t row = (uint64_t)blockIdx.x * CLIP_ROWS_PER_BLOCK + threadIdx.x / WAVE;
    if (row >= n_rows) return;  // whole wave
    const float* src = lh + row * (uint64_t)n_cols;
    const float own = (lane < n_cols) ? src[lane] : __uint_as_float(0x7fc00000u);
    float v = own;
    if (clip_negative && !(v > 0.0f)) v = __uint_as_float(0x7fc00000u);  // torch.where(lh > 0, lh, nan)
    const int n_valid = __popcll(__ballot(v == v));
    uint32_t key = float_sort_key(v), unused = 0;
    wave_sort64(key, unused, lane);
    float quant[3];
    const float qs[3] = {q_low, 0.5f, q_high};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float rank = qs[i] * (float)(n_valid - 1);
        if (rank < 0.0f) rank = 0.0f;
        const float below = floorf(rank);
        const int ib = __builtin_amdgcn_readfirstlane((int)below), ia = __builtin_amdgcn_readfirstlane((int)ceilf(rank));
        const uint32_t kb_ = (uint32_t)__builtin_amdgcn_readlane((int)key, ib), ka = (uint32_t)__builtin_amdgcn_readlane((int)key, ia);
        quant[i] = lerp_aten(key_to_float(kb_), key_to_float(ka), rank - below);
    }
    float delta = quant[2] - quant[0];
    if (delta < 1e-5f) delta = 1e-5f;
    const float n_sigma_g = scale * delta;
    const float lower = quant[1] - n_sigma_g, upper = quant[1] + n_sigma_g;
    if (lane < n_cols) valid[row * (uint64_t)n_cols + lane] = (__builtin_isfinite(own) && own < upper && own > lower) ? 1 : 0;
}

// ---------------------------------------------------------------------------
// near-duplicate grid filter
// ---------------------------------------------------------------------------
// apply_trajectory_grid_filter / TrajectoryClusterGrid (src/kbmod/filters/clustering_grid.py:58-92,
// 152-175): key = (int(x / w), int(y / w), int((x + dt * vx) / w), int((y + dt * vy) / w)) in double
// arithmetic with truncation; per key the trajectory with the largest lh survives (a later one replaces
// only when strictly larger), keys come out in the order of their first occurrence.  The sequential
// dictionary becomes: stable sort by lh descending, stable sort by key (two 64-bit halves) -> every run
// of equal keys starts with its winner; the runs are then ordered by their smallest original index.
__device__ __forceinline__ bool grid_bin(double v, double w, int32_t* out) {
    const double q = trunc(v / w);
    if (!(q > -2147483648.0 && q < 2147483648.0)) return false;  // also NaN
    *out = (int32_t)q;
    return true;
}

__global__ __launch_bounds__(256) void kb_grid_keys_kernel(const kb_trajectory* __restrict__ trjs, uint64_t n,
                                                           double bin_width, double max_time,
                                                           uint64_t* __restrict__ key_hi, uint64_t* __restrict__ key_lo,
                                                           float* __restrict__ lh, uint32_t* __restrict__ idx,
                                                           int* __restrict__ bad) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const kb_trajectory t = trjs[i];
    int32_t xs = 0, ys = 0, xe = 0, ye = 0;
    bool ok = grid_bin((double)t.x, bin_width, &xs);
    ok = grid_bin((double)t.y, bin_width, &ys) && ok;
    ok = grid_bin(__dadd_rn((double)t.x, __dmul_rn(max_time, (double)t.vx)), bin_width, &xe) && ok;
    ok = grid_bin(__dadd_rn((double)t.y, __dmul_rn(max_time, (double)t.vy)), bin_width, &ye) && ok;
    if (!ok) atomicOr(bad, 1);
    // order-preserving: flip the sign bit of each 32-bit bin
    key_hi[i] = ((uint64_t)((uint32_t)xs ^ 0x80000000u) << 32) | ((uint32_t)ys ^ 0x80000000u);
    key_lo[i] = ((uint64_t)((uint32_t)xe ^ 0x80000000u) << 32) | ((uint32_t)ye ^ 0x80000000u);
    lh[i] = (t.lh == 0.0f) ? 0.0f : t.lh;  // -0 and +0 compare equal in the reference
    idx[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void kb_gather_u64_kernel(const uint64_t* __restrict__ src,
                                                            const uint32_t* __restrict__ idx, uint64_t n,
                                                            uint64_t* __restrict__ dst) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

__global__ __launch_bounds__(256) void kb_run_heads_kernel(const uint64_t* __restrict__ hi,
                                                           const uint64_t* __restrict__ lo, uint64_t n,
                                                           uint32_t* __restrict__ head) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) head[i] = (i == 0 || hi[i] != hi[i - 1] || lo[i] != lo[i - 1]) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void kb_run_reduce_kernel(const uint32_t* __restrict__ head,
                                                            const uint32_t* __restrict__ run_of,  // inclusive scan of head
                                                            const uint32_t* __restrict__ idx, uint64_t n,
                                                            uint32_t* __restrict__ first, uint32_t* __restrict__ best) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = run_of[i] - 1;
    if (head[i]) best[r] = idx[i];
    atomicMin(&first[r], idx[i]);
}

// Temporary storage rocprim wants for a radix sort of n (K, uint32) pairs (host-only query) ...
template <typename K>
static int sort_pairs_bytes(bool descending, size_t n, hipStream_t stream, size_t* bytes) {
    *bytes = 0;
    K* k = nullptr;
    uint32_t* v = nullptr;
    if (descending) {
        KB_HIP_TRY(rocprim::radix_sort_pairs_desc(nullptr, *bytes, k, k, v, v, n, 0, sizeof(K) * 8, stream));
    } else {
        KB_HIP_TRY(rocprim::radix_sort_pairs(nullptr, *bytes, k, k, v, v, n, 0, sizeof(K) * 8, stream));
    }
    return 0;
}
// ... and the sort itself in storage the caller owns (no allocation, no synchronisation: stream order does the rest)
template <typename K>
static int sort_pairs_in(void* tmp, size_t tmp_bytes, bool descending, K* keys_in, K* keys_out, uint32_t* val_in,
                         uint32_t* val_out, size_t n, hipStream_t stream) {
    size_t need = 0;
    if (sort_pairs_bytes<K>(descending, n, stream, &need)) return 1;
    if (need > tmp_bytes) return fail("radix sort of " + std::to_string(n) + " pairs needs " + std::to_string(need) + " bytes of temporary storage, " + std::to_string(tmp_bytes) + " reserved");
    size_t bytes = tmp_bytes;
    if (descending) {
        KB_HIP_TRY(rocprim::radix_sort_pairs_desc(tmp, bytes, keys_in, keys_out, val_in, val_out, n, 0, sizeof(K) * 8, stream));
    } else {
        KB_HIP_TRY(rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, val_in, val_out, n, 0, sizeof(K) * 8, stream));
    }
    return 0;
}

template <typename K>
static int sort_pairs(bool descending, K* keys_in, K* keys_out, uint32_t* val_in, uint32_t* val_out, size_t n,
                      hipStream_t stream) {
    Scratch tmp;
    size_t bytes = 0;
    if (descending) {
        KB_HIP_TRY(rocprim::radix_sort_pairs_desc(nullptr, bytes, keys_in, keys_out, val_in, val_out, n, 0, sizeof(K) * 8, stream));
        KB_HIP_TRY(hipMalloc(&tmp.p, std::max<size_t>(bytes, 16)));
        KB_HIP_TRY(rocprim::radix_sort_pairs_desc(tmp.p, bytes, keys_in, keys_out, val_in, val_out, n, 0, sizeof(K) * 8, stream));
    } else {
        KB_HIP_TRY(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, val_in, val_out, n, 0, sizeof(K) * 8, stream));
        KB_HIP_TRY(hipMalloc(&tmp.p, std::max<size_t>(bytes, 16)));
        KB_HIP_TRY(rocprim::radix_sort_pairs(tmp.p, bytes, keys_in, keys_out, val_in, val_out, n, 0, sizeof(K) * 8, stream));
    }
    KB_HIP_TRY(hipStreamSynchronize(stream));  // tmp is freed on return
    return 0;
}

}  // namespace kb

namespace kb {
// Scratch arena of kb_filter_sort_results, one per device, grow-only; the lock is held for the duration of a call.
struct ResultArena {
    void* ptr = nullptr;
    size_t bytes = 0;
};
constexpr int ARENA_DEVICES = 64;
static std::mutex g_arena_mutex[ARENA_DEVICES];
static ResultArena g_arena[ARENA_DEVICES];
static int arena_slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ARENA_DEVICES) dev = 0;
    return dev;
}
struct ArenaLock {
    int slot;
    std::unique_lock<std::mutex> lock;
    ArenaLock() : slot(arena_slot()), lock(g_arena_mutex[slot]) {}
    int reserve(size_t bytes, char** out) {
        ResultArena& a = g_arena[slot];
        if (a.ptr != nullptr && a.bytes < bytes) {
            (void)hipFree(a.ptr);
            a.ptr = nullptr;
            a.bytes = 0;
        }
        if (a.ptr == nullptr) {
            KB_HIP_TRY(hipMalloc(&a.ptr, bytes));
            a.bytes = bytes;
        }
        *out = static_cast<char*>(a.ptr);
        return 0;
    }
};
// (called by kb_release_workspaces)
void release_result_arenas() {
    int prev = 0;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    for (int dev = 0; dev < ARENA_DEVICES; ++dev) {
        std::lock_guard<std::mutex> lock(g_arena_mutex[dev]);
        if (g_arena[dev].ptr != nullptr) {
            (void)hipSetDevice(dev);
            (void)hipFree(g_arena[dev].ptr);
            g_arena[dev] = ResultArena();
        }
    }
    if (have_prev) (void)hipSetDevice(prev);
}
}  // namespace kb

namespace kb {
static int filter_sort_impl(const kb_trajectory* results_dev, uint64_t n, float min_lh, int32_t min_obs, kb_trajectory* out_dev,
                            uint64_t* n_out_host, int64_t* first_invalid_host, void* stream_v);
static int sort_and_gather(const kb_trajectory* compact, size_t kept, uint64_t n, char* base, size_t rec_bytes, size_t tmp_all,
                           size_t key_bytes, kb_trajectory* out_dev, int64_t* first_invalid_host, hipStream_t stream);
// The records of in[0 .. n) that pass `pred`, in order, to out; their number to *count_dev.  tmp: (n / 1024 + 1) * 12 + 64 bytes.
static int select_records(const kb_trajectory* in, uint64_t n, const KeepPredicate& pred, char* tmp, kb_trajectory* out,
                          unsigned long long* count_dev, hipStream_t stream) {
    const uint64_t n_blocks = (n + SELECT_BLOCK - 1) / SELECT_BLOCK;
    if (n_blocks > 0x7fffffffull) return fail("filter_sort_results: too many records for one launch");
    unsigned long long* bases = reinterpret_cast<unsigned long long*>(tmp);
    uint32_t* totals = reinterpret_cast<uint32_t*>(tmp + n_blocks * 8);
    hipLaunchKernelGGL(kb_select_kernel<0>, dim3((unsigned)n_blocks), dim3(256), 0, stream, in, n, pred, totals, bases, out);
    KB_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(kb_select_scan_kernel, dim3(1), dim3(1024), 0, stream, totals, n_blocks, bases, count_dev);
    KB_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(kb_select_kernel<1>, dim3((unsigned)n_blocks), dim3(256), 0, stream, in, n, pred, totals, bases, out);
    KB_HIP_TRY(hipGetLastError());
    return 0;
}
// exchange_kernels.hip
int compact_counted_full(const kb_trajectory* lists_dev, uint64_t n_pixels, int L, const uint8_t* counts_dev, kb_trajectory* out_dev,
                         uint64_t capacity, uint64_t* total_host, hipStream_t stream);
}
// kb_filter_sort_results_checked for results whose search wrote the per-pixel counts (kb_device_search_filter_counted): only the
// counted records are read -- the runs the search skipped are never touched --, and the working storage is sized by what survives.
extern "C" int kb_filter_sort_results_counted(const kb_trajectory* results_dev, uint64_t n_pixels, int32_t list_len,
                                              const uint8_t* counts_dev, float min_lh, int32_t min_obs, kb_trajectory* out_dev,
                                              uint64_t* n_out_host, int64_t* first_invalid_host, void* stream_v) {
    using namespace kb;
    KB_REQUIRE_DEVICE("the result filter.");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    if (n_out_host == nullptr) return fail("filter_sort_results_counted: null count pointer");
    *n_out_host = 0;
    if (first_invalid_host != nullptr) *first_invalid_host = -1;
    if (n_pixels == 0) return 0;
    if (results_dev == nullptr || out_dev == nullptr || counts_dev == nullptr) return fail("filter_sort_results_counted: null pointer");
    if (list_len <= 0 || list_len > 32) return fail("filter_sort_results_counted: lists of 1 to 32 records per pixel");
    // (the counts leave out the placeholders of empty slots, which filter_by_likelihood keeps when min_lh <= -FLT_MAX)
    if (!(min_lh > -FLT_MAX)) return fail("filter_sort_results_counted: needs a likelihood threshold above -FLT_MAX");
    if (n_pixels * (uint64_t)list_len > 0xffffffffull) return fail("filter_sort_results_counted: more than 2^32 results");
    uint64_t total = 0;
    if (compact_counted_full(results_dev, n_pixels, list_len, counts_dev, nullptr, 0, &total, stream)) return 1;
    if (total == 0) return 0;
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t rec_bytes = up(total * sizeof(kb_trajectory)), key_bytes = up(total * sizeof(float));
    const KeepPredicate pred{min_lh, min_obs};
    size_t tmp_bytes = up(((size_t)total / SELECT_BLOCK + 1) * 12 + 64), tmp2_bytes = 0;
    KB_HIP_TRY(rocprim::radix_sort_pairs_desc(nullptr, tmp2_bytes, static_cast<float*>(nullptr), static_cast<float*>(nullptr),
                                              static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), (size_t)total, 0, 32,
                                              stream));
    const size_t tmp_all = up(std::max<size_t>(std::max(tmp_bytes, tmp2_bytes), 16));
    ArenaLock arena;
    char* base = nullptr;
    // [records kept | count + first-invalid words | temporary storage | 2 x keys | 2 x indices | the counted records]
    if (arena.reserve(rec_bytes + 256 + tmp_all + 4 * key_bytes + rec_bytes, &base)) return 1;
    kb_trajectory* compact = reinterpret_cast<kb_trajectory*>(base);
    kb_trajectory* counted = reinterpret_cast<kb_trajectory*>(base + rec_bytes + 256 + tmp_all + 4 * key_bytes);
    size_t kept = (size_t)total;
    {
        // Both tests run over the counted records, whatever the counts were made with: the counts are the caller's (a search
        // with another min_lh, a merge), and a record they admit that this call's thresholds do not must not come through.
        if (compact_counted_full(results_dev, n_pixels, list_len, counts_dev, counted, total, &total, stream)) return 1;
        unsigned long long* count = reinterpret_cast<unsigned long long*>(base + rec_bytes);
        if (select_records(counted, total, pred, base + rec_bytes + 256, compact, count, stream)) return 1;
        unsigned long long kept_dev = 0;
        KB_HIP_TRY(hipMemcpyAsync(&kept_dev, count, sizeof(kept_dev), hipMemcpyDeviceToHost, stream));
        KB_HIP_TRY(hipStreamSynchronize(stream));
        kept = (size_t)kept_dev;
    }
    *n_out_host = kept;
    if (kept == 0) return 0;
    return sort_and_gather(compact, kept, total, base, rec_bytes, tmp_all, key_bytes, out_dev, first_invalid_host, stream);
}
extern "C" int kb_filter_sort_results(const kb_trajectory* results_dev, uint64_t n, float min_lh, int32_t min_obs,
                                      kb_trajectory* out_dev, uint64_t* n_out_host, void* stream_v) {
    return kb::filter_sort_impl(results_dev, n, min_lh, min_obs, out_dev, n_out_host, nullptr, stream_v);
}
extern "C" int kb_filter_sort_results_checked(const kb_trajectory* results_dev, uint64_t n, float min_lh, int32_t min_obs,
                                              kb_trajectory* out_dev, uint64_t* n_out_host, int64_t* first_invalid_host,
                                              void* stream_v) {
    if (first_invalid_host == nullptr) return kb::fail("filter_sort_results_checked: null index pointer");
    *first_invalid_host = -1;
    return kb::filter_sort_impl(results_dev, n, min_lh, min_obs, out_dev, n_out_host, first_invalid_host, stream_v);
}
static int kb::filter_sort_impl(const kb_trajectory* results_dev, uint64_t n, float min_lh, int32_t min_obs,
                                kb_trajectory* out_dev, uint64_t* n_out_host, int64_t* first_invalid_host, void* stream_v) {
    using namespace kb;
    KB_REQUIRE_DEVICE("the result filter.");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    if (n_out_host == nullptr) return fail("filter_sort_results: null count pointer");
    *n_out_host = 0;
    if (n == 0) return 0;
    if (results_dev == nullptr || out_dev == nullptr) return fail("filter_sort_results: null pointer");
    if (n > 0xffffffffull) return fail("filter_sort_results: more than 2^32 results");

    // One arena per device, kept between calls and grown when a call needs more (kb_release_workspaces returns it): the
    // eight allocations and releases this function used to make per call were most of what search_all spent outside the
    // search kernel when few results survive.  Sized for the worst case (every record survives).
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t rec_bytes = up(n * sizeof(kb_trajectory)), key_bytes = up(n * sizeof(float));
    const KeepPredicate pred{min_lh, min_obs};
    size_t tmp_bytes = 0, tmp2_bytes = 0;
    const uint64_t n_blocks = (n + SELECT_BLOCK - 1) / SELECT_BLOCK;
    tmp_bytes = up(n_blocks * 12 + 64);  // block totals and bases of the compaction
    KB_HIP_TRY(rocprim::radix_sort_pairs_desc(nullptr, tmp2_bytes, static_cast<float*>(nullptr), static_cast<float*>(nullptr),
                                              static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), (size_t)n, 0, 32,
                                              stream));
    const size_t tmp_all = up(std::max<size_t>(std::max(tmp_bytes, tmp2_bytes), 16));
    ArenaLock arena;
    char* base = nullptr;
    if (arena.reserve(rec_bytes + 256 + tmp_all + 4 * key_bytes, &base)) return 1;
    kb_trajectory* compact = reinterpret_cast<kb_trajectory*>(base);
    unsigned long long* count = reinterpret_cast<unsigned long long*>(base + rec_bytes);
    char* tmp = base + rec_bytes + 256;

    // ---- 1. stable compaction into a temporary ----
    if (select_records(results_dev, n, pred, tmp, compact, count, stream)) return 1;
    unsigned long long kept = 0;
    KB_HIP_TRY(hipMemcpyAsync(&kept, count, sizeof(kept), hipMemcpyDeviceToHost, stream));
    KB_HIP_TRY(hipStreamSynchronize(stream));
    *n_out_host = kept;
    if (kept == 0) return 0;

    return sort_and_gather(compact, kept, n, base, rec_bytes, tmp_all, key_bytes, out_dev, first_invalid_host, stream);
}

// Steps 2 and 3 of the filter: `kept` surviving records at `compact` (the head of the arena at `base`, laid out as in
// filter_sort_impl for up to n records) -> sorted by likelihood, descending and stable, in out_dev.
static int kb::sort_and_gather(const kb_trajectory* compact, size_t kept, uint64_t n, char* base, size_t rec_bytes, size_t tmp_all,
                               size_t key_bytes, kb_trajectory* out_dev, int64_t* first_invalid_host, hipStream_t stream) {
    void* tmp = base + rec_bytes + 256;
    float* keys_in = reinterpret_cast<float*>(base + rec_bytes + 256 + tmp_all);
    float* keys_out = reinterpret_cast<float*>(base + rec_bytes + 256 + tmp_all + key_bytes);
    uint32_t* idx_in = reinterpret_cast<uint32_t*>(base + rec_bytes + 256 + tmp_all + 2 * key_bytes);
    uint32_t* idx_out = reinterpret_cast<uint32_t*>(base + rec_bytes + 256 + tmp_all + 3 * key_bytes);
    // ---- 2. stable descending radix sort of (lh, index) ----
    const unsigned blocks = (unsigned)((kept + 255) / 256);
    hipLaunchKernelGGL(kb_extract_keys_kernel, dim3(blocks), dim3(256), 0, stream, compact, (uint64_t)kept, keys_in, idx_in);
    KB_HIP_TRY(hipGetLastError());
    // rocprim picks its sort (block / merge / onesweep) by the element count and each lays out its own temporary storage:
    // ask again for `kept` elements -- a host-only call -- instead of assuming the size for n covers every smaller count
    size_t sort_bytes = 0;
    KB_HIP_TRY(rocprim::radix_sort_pairs_desc(nullptr, sort_bytes, static_cast<float*>(nullptr), static_cast<float*>(nullptr),
                                              static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), kept, 0, 32,
                                              stream));
    if (sort_bytes > tmp_all) return fail("filter_sort_results: sorting " + std::to_string(kept) + " records needs more temporary storage (" +
                                          std::to_string(sort_bytes) + " bytes) than sorting all " + std::to_string(n));
    KB_HIP_TRY(rocprim::radix_sort_pairs_desc(tmp, sort_bytes, keys_in, keys_out, idx_in, idx_out, kept, 0, 32, stream));
    // ---- 3. gather the 28-byte records in sorted order ----
    unsigned long long* bad_dev = nullptr;
    if (first_invalid_host != nullptr) {
        bad_dev = reinterpret_cast<unsigned long long*>(base + rec_bytes + 64);  // (behind the survivor count)
        KB_HIP_TRY(hipMemsetAsync(bad_dev, 0xff, sizeof(unsigned long long), stream));
    }
    hipLaunchKernelGGL(kb_gather_kernel, dim3(blocks), dim3(256), 0, stream, compact, idx_out, (uint64_t)kept, out_dev, bad_dev);
    KB_HIP_TRY(hipGetLastError());
    if (first_invalid_host != nullptr) {
        unsigned long long bad = ~0ull;
        KB_HIP_TRY(hipMemcpyAsync(&bad, bad_dev, sizeof(bad), hipMemcpyDeviceToHost, stream));
        KB_HIP_TRY(hipStreamSynchronize(stream));
        *first_invalid_host = bad == ~0ull ? -1 : (int64_t)bad;
        return 0;
    }
    KB_HIP_TRY(hipStreamSynchronize(stream));
    return 0;
}

extern "C" int kb_psi_phi_curves(const kb_psi_phi_meta* meta, const void* psi_phi_dev, const double* times_dev,
                                 const kb_trajectory* trjs_dev, uint64_t n, float* out_dev, void* stream_v) {
    using namespace kb;
    KB_REQUIRE_DEVICE("the psi/phi curves.");
    if (meta == nullptr || psi_phi_dev == nullptr || times_dev == nullptr) return fail("psi_phi_curves: null input");
    if (n == 0) return 0;
    if (trjs_dev == nullptr || out_dev == nullptr) return fail("psi_phi_curves: null pointer");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    const uint64_t total = n * meta->num_times;
    const uint64_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffull) return fail("psi_phi_curves: too many trajectories for one launch");
    hipLaunchKernelGGL(kb_curves_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, *meta, psi_phi_dev, times_dev,
                       trjs_dev, n, out_dev);
    KB_HIP_TRY(hipGetLastError());
    KB_HIP_TRY(hipStreamSynchronize(stream));
    return 0;
}

extern "C" int kb_sigma_g_clip_matrix(const float* lh_dev, uint64_t n_rows, int32_t n_cols, float low_pct, float high_pct,
                                      float n_sigma, float coeff, int32_t clip_negative, uint8_t* valid_dev,
                                      void* stream_v) {
    using namespace kb;
    if (n_rows == 0 || n_cols == 0) return 0;
    KB_REQUIRE_DEVICE("sigma-G clipping.");
    if (lh_dev == nullptr || valid_dev == nullptr) return fail("sigma_g_clip_matrix: null pointer");
    if (n_cols < 0 || n_cols > 4096) return fail("sigma_g_clip_matrix: curves longer than 4096 points are not supported");
    if (!(low_pct > 0.0f) || !(high_pct < 100.0f) || low_pct > high_pct) {  // sigma_g_filter.py:38-39
        return fail("Invalid bounds [" + std::to_string(low_pct) + ", " + std::to_string(high_pct) + "]");
    }
    if (!(n_sigma > 0.0f)) return fail("Invalid n_sigma " + std::to_string(n_sigma));
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    int P = 1;
    while (P < n_cols) P <<= 1;
    if (P < 2) P = 2;
    const uint64_t blocks = (n_rows + CLIP_ROWS_PER_BLOCK - 1) / CLIP_ROWS_PER_BLOCK;
    if (blocks > 0x7fffffffull) return fail("sigma_g_clip_matrix: too many rows for one launch");
    // the reference forms n_sigma * coeff in double and multiplies the float32 tensor by it
    const float scale = (float)((double)n_sigma * (double)coeff);
    if (n_cols <= WAVE) {
        hipLaunchKernelGGL(kb_sigma_g_clip64_kernel, dim3((unsigned)blocks), dim3(CLIP_ROWS_PER_BLOCK * WAVE), 0, stream,
                           lh_dev, n_rows, (int)n_cols, low_pct / 100.0f, high_pct / 100.0f, scale, (int)clip_negative,
                           valid_dev);
    } else {
        hipLaunchKernelGGL(kb_sigma_g_clip_kernel, dim3((unsigned)blocks), dim3(CLIP_ROWS_PER_BLOCK * WAVE),
                           (size_t)CLIP_ROWS_PER_BLOCK * P * sizeof(uint32_t), stream, lh_dev, n_rows, (int)n_cols, P,
                           low_pct / 100.0f, high_pct / 100.0f, scale, (int)clip_negative, valid_dev);
    }
    KB_HIP_TRY(hipGetLastError());
    KB_HIP_TRY(hipStreamSynchronize(stream));
    return 0;
}

extern "C" int kb_sigma_g_clip_matrix_host(const float* lh_host, uint64_t n_rows, int32_t n_cols, float low_pct,
                                           float high_pct, float n_sigma, float coeff, int32_t clip_negative,
                                           uint8_t* valid_host) {
    using namespace kb;
    if (n_rows == 0 || n_cols == 0) return 0;
    if (lh_host == nullptr || valid_host == nullptr) return fail("sigma_g_clip_matrix: null pointer");
    if (kb_device_count() == 0) return fail("GPU is not available for sigma-G clipping.");
    const uint64_t n = n_rows * (uint64_t)n_cols;
    float* lh_dev = nullptr;
    uint8_t* valid_dev = nullptr;
    KB_HIP_TRY(hipMalloc(&lh_dev, n * sizeof(float)));
    if (hipMalloc(&valid_dev, n) != hipSuccess) {
        (void)hipFree(lh_dev);
        return fail("sigma_g_clip_matrix: out of device memory");
    }
    int rc = 0;
    if (hipMemcpy(lh_dev, lh_host, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        rc = fail("sigma_g_clip_matrix: upload failed");
    }
    if (rc == 0) {
        rc = kb_sigma_g_clip_matrix(lh_dev, n_rows, n_cols, low_pct, high_pct, n_sigma, coeff, clip_negative, valid_dev,
                                    nullptr);
    }
    if (rc == 0 && hipMemcpy(valid_host, valid_dev, n, hipMemcpyDeviceToHost) != hipSuccess) {
        rc = fail("sigma_g_clip_matrix: download failed");
    }
    (void)hipFree(lh_dev);
    (void)hipFree(valid_dev);
    return rc;
}

extern "C" int kb_grid_filter(const kb_trajectory* trjs_dev, uint64_t n, double bin_width, double max_time,
                              uint32_t* kept_idx_dev, uint64_t* n_kept_host, void* stream_v) {
    using namespace kb;
    KB_REQUIRE_DEVICE("the grid filter.");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    if (n_kept_host == nullptr) return fail("grid_filter: null count pointer");
    *n_kept_host = 0;
    // clustering_grid.py:44-50
    if (!(bin_width >= 1.0) || !std::isfinite(bin_width)) return fail("Bin width must be at least 1. Got " + std::to_string(bin_width) + ".");
    if (!(max_time >= 0.0) || !std::isfinite(max_time)) return fail("Max time must be >= 0. Got " + std::to_string(max_time) + ".");
    if (n == 0) return 0;
    if (trjs_dev == nullptr || kept_idx_dev == nullptr) return fail("grid_filter: null pointer");
    if (n >= 0xffffffffull) return fail("grid_filter: more than 2^32 - 1 trajectories");

    // One block of the per-device result arena (kept between calls, shared with kb_filter_sort_results) instead of fourteen
    // allocations and as many releases per call -- they were half of the call's 3.8 ms at 2 M trajectories.
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    size_t t_lh = 0, t_k64 = 0, t_u32 = 0, t_scan = 0;
    if (sort_pairs_bytes<float>(true, n, stream, &t_lh) || sort_pairs_bytes<uint64_t>(false, n, stream, &t_k64) ||
        sort_pairs_bytes<uint32_t>(false, n, stream, &t_u32)) return 1;
    KB_HIP_TRY(rocprim::inclusive_scan(nullptr, t_scan, static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), (size_t)n,
                                       rocprim::plus<uint32_t>(), stream));
    const size_t tmp_bytes = up(std::max(std::max(t_lh, t_k64), std::max(std::max(t_u32, t_scan), (size_t)16)));
    const size_t b8 = up(n * 8), b4 = up(n * 4);
    ArenaLock arena;
    char* base = nullptr;
    if (arena.reserve(4 * b8 + 8 * b4 + 256 + tmp_bytes, &base)) return 1;
    char* at = base;
    auto take = [&](size_t bytes) { char* p = at; at += bytes; return p; };
    uint64_t* key_hi = reinterpret_cast<uint64_t*>(take(b8));
    uint64_t* key_lo = reinterpret_cast<uint64_t*>(take(b8));
    uint64_t* ka = reinterpret_cast<uint64_t*>(take(b8));
    uint64_t* kb_ = reinterpret_cast<uint64_t*>(take(b8));
    float* lh_a = reinterpret_cast<float*>(take(b4));
    float* lh_b = reinterpret_cast<float*>(take(b4));
    uint32_t* ia = reinterpret_cast<uint32_t*>(take(b4));
    uint32_t* ib = reinterpret_cast<uint32_t*>(take(b4));
    uint32_t* head_p = reinterpret_cast<uint32_t*>(take(b4));
    uint32_t* runs_p = reinterpret_cast<uint32_t*>(take(b4));
    uint32_t* first_p = reinterpret_cast<uint32_t*>(take(b4));
    uint32_t* best_p = reinterpret_cast<uint32_t*>(take(b4));
    int* bad_p = reinterpret_cast<int*>(take(256));
    void* tmp = take(tmp_bytes);
    KB_HIP_TRY(hipMemsetAsync(bad_p, 0, 4, stream));
    KB_HIP_TRY(hipMemsetAsync(first_p, 0xff, n * 4, stream));
    const unsigned blocks = (unsigned)((n + 255) / 256);

    hipLaunchKernelGGL(kb_grid_keys_kernel, dim3(blocks), dim3(256), 0, stream, trjs_dev, n, bin_width, max_time, key_hi,
                       key_lo, lh_a, ia, bad_p);
    KB_HIP_TRY(hipGetLastError());
    // 1. lh descending (stable: equal lh keep their original order) -> ib
    if (sort_pairs_in<float>(tmp, tmp_bytes, true, lh_a, lh_b, ia, ib, n, stream)) return 1;
    // 2. stable by the end bins, then by the start bins -> lexicographic (start, end) order
    hipLaunchKernelGGL(kb_gather_u64_kernel, dim3(blocks), dim3(256), 0, stream, key_lo, ib, n, ka);
    if (sort_pairs_in<uint64_t>(tmp, tmp_bytes, false, ka, kb_, ib, ia, n, stream)) return 1;
    hipLaunchKernelGGL(kb_gather_u64_kernel, dim3(blocks), dim3(256), 0, stream, key_hi, ia, n, ka);
    if (sort_pairs_in<uint64_t>(tmp, tmp_bytes, false, ka, kb_, ia, ib, n, stream)) return 1;  // kb_ = sorted start keys, ib = order
    hipLaunchKernelGGL(kb_gather_u64_kernel, dim3(blocks), dim3(256), 0, stream, key_lo, ib, n, ka);  // end keys, same order
    // 3. runs of equal keys: head flags, run numbers, winner (= head) and first occurrence of every run
    hipLaunchKernelGGL(kb_run_heads_kernel, dim3(blocks), dim3(256), 0, stream, kb_, ka, n, head_p);
    size_t scan_bytes = tmp_bytes;
    KB_HIP_TRY(rocprim::inclusive_scan(tmp, scan_bytes, head_p, runs_p, (size_t)n, rocprim::plus<uint32_t>(), stream));
    hipLaunchKernelGGL(kb_run_reduce_kernel, dim3(blocks), dim3(256), 0, stream, head_p, runs_p, ib, n, first_p, best_p);
    KB_HIP_TRY(hipGetLastError());
    uint32_t n_runs = 0;
    int bad_host = 0;
    KB_HIP_TRY(hipMemcpyAsync(&n_runs, runs_p + (n - 1), 4, hipMemcpyDeviceToHost, stream));
    KB_HIP_TRY(hipMemcpyAsync(&bad_host, bad_p, 4, hipMemcpyDeviceToHost, stream));
    KB_HIP_TRY(hipStreamSynchronize(stream));  // (the one synchronisation before the last sort: it needs the number of runs)
    if (bad_host) return fail("grid_filter: a trajectory does not map to a finite 32-bit spatial bin");
    // 4. dictionary order = order of first occurrence
    if (sort_pairs_in<uint32_t>(tmp, tmp_bytes, false, first_p, ia, best_p, kept_idx_dev, (size_t)n_runs, stream)) return 1;
    KB_HIP_TRY(hipStreamSynchronize(stream));
    *n_kept_host = n_runs;
    return 0;
}
