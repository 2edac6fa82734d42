// Device-side post-processing of the per-pixel result slots: the likelihood /
// obs_count filters and the global descending sort that the reference runs on
// the host after every search (stack_search.cpp:266-281 with
// trajectory_list.cpp:96-126: filter_by_likelihood, filter_by_obs_count,
// sort_by_likelihood over up to S*K 28-byte structs).  Doing it in HBM means only
// the survivors cross PCIe.  Compaction and the radix sort come from rocPRIM
// (stable, so equal likelihoods keep their slot order -- one of the orders the
// reference's unstable sort may produce); the kernels around them are ours.
#include <algorithm>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

#include "kb_common.h"

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

__global__ __launch_bounds__(256) void kb_gather_kernel(const kb_trajectory* __restrict__ in,
                                                        const uint32_t* __restrict__ idx, uint64_t n,
                                                        kb_trajectory* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[idx[i]];
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

}  // namespace kb

extern "C" int kb_filter_sort_results(const kb_trajectory* results_dev, uint64_t n, float min_lh, int32_t min_obs,
                                      kb_trajectory* out_dev, uint64_t* n_out_host, void* stream_v) {
    using namespace kb;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    if (n_out_host == nullptr) return fail("filter_sort_results: null count pointer");
    *n_out_host = 0;
    if (n == 0) return 0;
    if (results_dev == nullptr || out_dev == nullptr) return fail("filter_sort_results: null pointer");
    if (n > 0xffffffffull) return fail("filter_sort_results: more than 2^32 results");

    // ---- 1. stable compaction into a temporary ----
    Scratch compact, count, tmp, keys_in, keys_out, idx_in, idx_out;
    KB_HIP_TRY(hipMalloc(&compact.p, n * sizeof(kb_trajectory)));
    KB_HIP_TRY(hipMalloc(&count.p, sizeof(size_t)));
    const KeepPredicate pred{min_lh, min_obs};
    size_t tmp_bytes = 0;
    KB_HIP_TRY(rocprim::select(nullptr, tmp_bytes, results_dev, reinterpret_cast<kb_trajectory*>(compact.p),
                               reinterpret_cast<size_t*>(count.p), (size_t)n, pred, stream));
    KB_HIP_TRY(hipMalloc(&tmp.p, std::max<size_t>(tmp_bytes, 16)));
    KB_HIP_TRY(rocprim::select(tmp.p, tmp_bytes, results_dev, reinterpret_cast<kb_trajectory*>(compact.p),
                               reinterpret_cast<size_t*>(count.p), (size_t)n, pred, stream));
    size_t kept = 0;
    KB_HIP_TRY(hipMemcpyAsync(&kept, count.p, sizeof(size_t), hipMemcpyDeviceToHost, stream));
    KB_HIP_TRY(hipStreamSynchronize(stream));
    *n_out_host = kept;
    if (kept == 0) return 0;

    // ---- 2. stable descending radix sort of (lh, index) ----
    KB_HIP_TRY(hipMalloc(&keys_in.p, kept * sizeof(float)));
    KB_HIP_TRY(hipMalloc(&keys_out.p, kept * sizeof(float)));
    KB_HIP_TRY(hipMalloc(&idx_in.p, kept * sizeof(uint32_t)));
    KB_HIP_TRY(hipMalloc(&idx_out.p, kept * sizeof(uint32_t)));
    const unsigned blocks = (unsigned)((kept + 255) / 256);
    hipLaunchKernelGGL(kb_extract_keys_kernel, dim3(blocks), dim3(256), 0, stream,
                       reinterpret_cast<const kb_trajectory*>(compact.p), (uint64_t)kept,
                       reinterpret_cast<float*>(keys_in.p), reinterpret_cast<uint32_t*>(idx_in.p));
    KB_HIP_TRY(hipGetLastError());
    Scratch tmp2;
    size_t tmp2_bytes = 0;
    KB_HIP_TRY(rocprim::radix_sort_pairs_desc(nullptr, tmp2_bytes, reinterpret_cast<float*>(keys_in.p),
                                              reinterpret_cast<float*>(keys_out.p),
                                              reinterpret_cast<uint32_t*>(idx_in.p),
                                              reinterpret_cast<uint32_t*>(idx_out.p), kept, 0, 32, stream));
    KB_HIP_TRY(hipMalloc(&tmp2.p, std::max<size_t>(tmp2_bytes, 16)));
    KB_HIP_TRY(rocprim::radix_sort_pairs_desc(tmp2.p, tmp2_bytes, reinterpret_cast<float*>(keys_in.p),
                                              reinterpret_cast<float*>(keys_out.p),
                                              reinterpret_cast<uint32_t*>(idx_in.p),
                                              reinterpret_cast<uint32_t*>(idx_out.p), kept, 0, 32, stream));
    // ---- 3. gather the 28-byte records in sorted order ----
    hipLaunchKernelGGL(kb_gather_kernel, dim3(blocks), dim3(256), 0, stream,
                       reinterpret_cast<const kb_trajectory*>(compact.p),
                       reinterpret_cast<const uint32_t*>(idx_out.p), (uint64_t)kept, out_dev);
    KB_HIP_TRY(hipGetLastError());
    KB_HIP_TRY(hipStreamSynchronize(stream));
    return 0;
}

extern "C" int kb_psi_phi_curves(const kb_psi_phi_meta* meta, const void* psi_phi_dev, const double* times_dev,
                                 const kb_trajectory* trjs_dev, uint64_t n, float* out_dev, void* stream_v) {
    using namespace kb;
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
