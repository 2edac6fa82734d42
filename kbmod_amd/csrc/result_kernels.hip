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
