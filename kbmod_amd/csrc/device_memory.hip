// Device discovery and raw HBM block management for libkbmod_hip.so.
// Replaces kernels/kernel_memory.cu:15-136 of the reference (same roles, HIP
// runtime underneath, status codes instead of exceptions across the C ABI).
#include <algorithm>

#include "kb_common.h"

namespace kb {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
int fail(const std::string& msg) {
    set_error(msg);
    return 1;
}
// Streaming copy used to measure what HBM delivers on this device (the "measured peak" next to the nominal 8 TB/s
// in the roofline report).  One 16-byte element per thread, no loop, the store non-temporal: of the forms swept on
// MI355X (tools/ubench/copybench.hip: grid-stride loops of 4 .. 64 Ki workgroups, contiguous spans per workgroup,
// 256 / 1024 threads, non-temporal loads and / or stores, hipMemcpyAsync) this is the one that reaches the rate the
// hardware guide documents for a float4 copy -- 6.3-6.4 TB/s on 4 GiB each way, where the grid-stride loop of 4096
// workgroups this probe used before stops at 4.9.
typedef uint32_t Quad __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void kb_copy_kernel(const Quad* __restrict__ src, Quad* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(src[i], dst + i);
}

// Read-only stream over the same block: what the fabric delivers to the L2s without any write traffic (for a
// block smaller than the Infinity Cache: the rate at which that cache feeds the XCDs).  Four non-temporal 16-byte
// loads in flight per lane, the grid sized so that they cover the block exactly (6.9 TB/s on 4 GiB).
constexpr int READ_PROBE_LOADS = 4;
__global__ __launch_bounds__(256) void kb_read_kernel(const Quad* __restrict__ src, size_t n, uint32_t* __restrict__ sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < READ_PROBE_LOADS; ++k) {
        if (i + k * stride < n) {
            const Quad v = __builtin_nontemporal_load(src + i + k * stride);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;  // never true for the memset pattern; keeps the loads alive
}

// LDS read rate with every CU streaming: each wave of a 1024-thread workgroup issues `iters` rounds of eight
// conflict-free ds_read_b64 (512 bytes per wave-instruction) -- the read mix of kb_search_lds's summing loop without
// its adds.  The figure the roofline block of bench.py holds the search kernel's LDS traffic against.
struct LdsProbeOffsets {
    int o[8];  // byte offsets of the eight reads: runtime values, like the table words of the search kernel (constants
               // would let the compiler fuse pairs of reads into ds_read2_b64, another instruction with another rate)
};
__global__ __launch_bounds__(1024) void kb_lds_read_kernel(int iters, int stride, LdsProbeOffsets offs, uint32_t* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) float2 slab[8192];  // 64 KiB
    for (int i = threadIdx.x; i < 8192; i += 1024) slab[i] = make_float2((float)i, 1.0f);
    __syncthreads();
    typedef float Pair __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(3))) Pair* LdsPair;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t base0 = (uint32_t)(uintptr_t)slab + (wave * 72 + lane) * 8;
    uint32_t base = base0;
    Pair acc = Pair{0.0f, 0.0f};
    for (int it = 0; it < iters; ++it) {
        Pair v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = *(LdsPair)(uintptr_t)(base + offs.o[c]);
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
#pragma unroll
        for (int c = 0; c < 8; ++c) asm volatile("" ::"v"(v[c]));
        base += stride;
        if ((it & 3) == 3) base = base0;
        if (it == iters - 1) acc = v[0];
    }
    if (acc.x == -1.0f) sink[0] = 1u;
}
}  // namespace kb

extern "C" {

const char* kb_last_error(void) { return kb::g_last_error.c_str(); }

// kernel_memory.cu:15-21
int kb_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) {
        (void)hipGetLastError();  // clear the sticky "no device" state
        return 0;
    }
    return count;
}

// kernel_memory.cu:23-48
void kb_print_stats(void) {
    std::printf("\n----- HIP Debugging Log -----\n");
    int count = kb_device_count();
    std::printf("HIP devices = %d\n", count);
    if (count == 0) return;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
        std::printf("Current device = %d (%s, %s, %d CUs)\n", dev, prop.name, prop.gcnArchName,
                    prop.multiProcessorCount);
    }
    size_t free_mem = 0, total_mem = 0;
    if (hipMemGetInfo(&free_mem, &total_mem) == hipSuccess) {
        std::printf("Total Memory = %zu\nFree Memory = %zu\n", total_mem, free_mem);
    }
}

size_t kb_gpu_total_memory(void) {  // kernel_memory.cu:50-58
    if (kb_device_count() == 0) return 0;
    size_t free_mem = 0, total_mem = 0;
    if (hipMemGetInfo(&free_mem, &total_mem) != hipSuccess) return 0;
    return total_mem;
}

size_t kb_gpu_free_memory(void) {  // kernel_memory.cu:60-68
    if (kb_device_count() == 0) return 0;
    size_t free_mem = 0, total_mem = 0;
    if (hipMemGetInfo(&free_mem, &total_mem) != hipSuccess) return 0;
    return free_mem;
}

int kb_check_gpu(size_t req_memory) {  // kernel_memory.cu:70-87
    if (kb_device_count() == 0) return 0;
    return kb_gpu_free_memory() >= req_memory ? 1 : 0;
}

int kb_allocate_gpu_block(uint64_t memory_size, void** out_dev) {  // kernel_memory.cu:93-103
    if (out_dev == nullptr) return kb::fail("allocate_gpu_block: null output pointer");
    *out_dev = nullptr;
    if (memory_size == 0) return kb::fail("Unable to allocate GPU memory: zero bytes requested");
    void* p = nullptr;
    hipError_t err = hipMalloc(&p, memory_size);
    if (err != hipSuccess || p == nullptr) {
        return kb::fail("Unable to allocate GPU memory (" + std::to_string(memory_size) +
                        " bytes): " + hipGetErrorString(err));
    }
    *out_dev = p;
    return 0;
}

int kb_free_gpu_block(void* ptr_dev) {  // kernel_memory.cu:105-110
    if (ptr_dev == nullptr) return kb::fail("Trying to free nullptr.");
    kb::note_array_gone(ptr_dev);  // (a psi/phi array the library built: its padded copy dies with it)
    KB_HIP_TRY(hipFree(ptr_dev));
    return 0;
}

int kb_copy_block_to_gpu(const void* src_host, void* dst_dev, uint64_t memory_size) {  // :112-122
    if (src_host == nullptr) return kb::fail("Invalid CPU pointer");
    if (dst_dev == nullptr) return kb::fail("Invalid GPU pointer");
    kb::note_array_written(dst_dev);
    KB_HIP_TRY(hipMemcpy(dst_dev, src_host, memory_size, hipMemcpyHostToDevice));
    return 0;
}

int kb_copy_block_to_cpu(void* dst_host, const void* src_dev, uint64_t memory_size) {  // :124-134
    if (dst_host == nullptr) return kb::fail("Invalid CPU pointer");
    if (src_dev == nullptr) return kb::fail("Invalid GPU pointer");
    KB_HIP_TRY(hipMemcpy(dst_host, src_dev, memory_size, hipMemcpyDeviceToHost));
    return 0;
}

// kb_copy_block_to_cpu for large blocks into pageable memory: the destination is page-locked for the duration of the copy
// (hipHostRegister), so that the transfer is ONE DMA at the link's rate instead of the runtime's staged pageable path
// (58.7 MB of results: 1.2 ms instead of 3.5); memory that cannot be registered -- or already is -- takes the plain copy.
int kb_copy_block_to_cpu_locked(void* dst_host, const void* src_dev, uint64_t memory_size) {
    if (dst_host == nullptr) return kb::fail("Invalid CPU pointer");
    if (src_dev == nullptr) return kb::fail("Invalid GPU pointer");
    bool locked = false;
    if (memory_size >= (4ull << 20)) {
        locked = hipHostRegister(dst_host, memory_size, hipHostRegisterDefault) == hipSuccess;
        if (!locked) (void)hipGetLastError();
    }
    const hipError_t rc = hipMemcpy(dst_host, src_dev, memory_size, hipMemcpyDeviceToHost);
    if (locked) (void)hipHostUnregister(dst_host);
    if (rc != hipSuccess) return kb::fail(std::string("hipMemcpy (device to host) failed: ") + hipGetErrorString(rc));
    return 0;
}

// Copies `bytes` (rounded down to 16) from one scratch block to another `iters` times after one untimed
// pass and reports read + write bytes over the HIP-event time of the timed passes, in GB/s.
int kb_measure_copy_bandwidth(uint64_t bytes, int32_t iters, void* stream_v, double* gbps_out) {
    using namespace kb;
    if (gbps_out == nullptr || iters <= 0 || bytes < 16) return fail("measure_copy_bandwidth: bad argument");
    KB_REQUIRE_DEVICE("the copy-bandwidth probe.");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    const size_t n = (size_t)(bytes / 16);
    void *src = nullptr, *dst = nullptr;
    KB_HIP_TRY(hipMalloc(&src, n * 16));
    if (hipMalloc(&dst, n * 16) != hipSuccess) {
        (void)hipFree(src);
        return fail("measure_copy_bandwidth: out of device memory");
    }
    (void)hipMemsetAsync(src, 1, n * 16, stream);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    float ms = 0.0f;
    {
        EventTimer timer(stream, true);
        hipLaunchKernelGGL(kb_copy_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const Quad*>(src),
                           reinterpret_cast<Quad*>(dst), n);
        timer.begin();
        for (int i = 0; i < iters; ++i) {
            hipLaunchKernelGGL(kb_copy_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const Quad*>(src),
                               reinterpret_cast<Quad*>(dst), n);
        }
        ms = timer.end();
    }
    (void)hipFree(src);
    (void)hipFree(dst);
    KB_HIP_TRY(hipGetLastError());
    if (!(ms > 0.0f)) return fail("measure_copy_bandwidth: no time measured");
    *gbps_out = 2.0 * (double)(n * 16) * iters / ((double)ms * 1e-3) / 1e9;
    return 0;
}

// Reads `bytes` (rounded down to 16) `iters` times after one untimed pass; *gbps_out = bytes read / time.
int kb_measure_read_bandwidth(uint64_t bytes, int32_t iters, void* stream_v, double* gbps_out) {
    using namespace kb;
    if (gbps_out == nullptr || iters <= 0 || bytes < 16) return fail("measure_read_bandwidth: bad argument");
    KB_REQUIRE_DEVICE("the read-bandwidth probe.");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    const size_t n = (size_t)(bytes / 16);
    void* src = nullptr;
    KB_HIP_TRY(hipMalloc(&src, n * 16 + 16));
    (void)hipMemsetAsync(src, 1, n * 16 + 16, stream);
    uint32_t* sink = reinterpret_cast<uint32_t*>(static_cast<char*>(src) + n * 16);
    const unsigned blocks = (unsigned)((n + 256 * READ_PROBE_LOADS - 1) / (256 * READ_PROBE_LOADS));
    float ms = 0.0f;
    {
        EventTimer timer(stream, true);
        hipLaunchKernelGGL(kb_read_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const Quad*>(src), n, sink);
        timer.begin();
        for (int i = 0; i < iters; ++i) {
            hipLaunchKernelGGL(kb_read_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const Quad*>(src), n,
                               sink);
        }
        ms = timer.end();
    }
    (void)hipFree(src);
    KB_HIP_TRY(hipGetLastError());
    if (!(ms > 0.0f)) return fail("measure_read_bandwidth: no time measured");
    *gbps_out = (double)(n * 16) * iters / ((double)ms * 1e-3) / 1e9;
    return 0;
}

// Aggregate LDS read rate (ds_read_b64, every CU busy): bytes = workgroups x 16 waves x iters x 8 x 512.
int kb_measure_lds_bandwidth(int32_t iters, void* stream_v, double* gbps_out) {
    using namespace kb;
    if (gbps_out == nullptr || iters <= 0) return fail("measure_lds_bandwidth: bad argument");
    KB_REQUIRE_DEVICE("the LDS-bandwidth probe.");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    int dev = 0, cus = 256;
    KB_HIP_TRY(hipGetDevice(&dev));
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    uint32_t* sink = nullptr;
    KB_HIP_TRY(hipMalloc(&sink, 64));
    const unsigned blocks = (unsigned)std::max(cus, 1) * 2;  // two rounds of one workgroup per CU
    float ms = 0.0f;
    {
        EventTimer timer(stream, true);
        LdsProbeOffsets offs;
        for (int c = 0; c < 8; ++c) offs.o[c] = c * 584;
        hipLaunchKernelGGL(kb_lds_read_kernel, dim3(blocks), dim3(1024), 0, stream, 64, 576, offs, sink);
        timer.begin();
        hipLaunchKernelGGL(kb_lds_read_kernel, dim3(blocks), dim3(1024), 0, stream, (int)iters, 576, offs, sink);
        ms = timer.end();
    }
    (void)hipFree(sink);
    KB_HIP_TRY(hipGetLastError());
    if (!(ms > 0.0f)) return fail("measure_lds_bandwidth: no time measured");
    *gbps_out = (double)blocks * 16.0 * (double)iters * 8.0 * 512.0 / ((double)ms * 1e-3) / 1e9;
    return 0;
}

int kb_get_device(void) {
    if (kb_device_count() == 0) return -1;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    return dev;
}

int kb_set_device(int32_t device) {
    if (device < 0 || device >= kb_device_count()) return kb::fail("Invalid device " + std::to_string(device));
    KB_HIP_TRY(hipSetDevice(device));
    return 0;
}

int kb_copy_block_between_gpus(void* dst_dev, int32_t dst_device, const void* src_dev, int32_t src_device,
                               uint64_t memory_size) {
    if (dst_dev == nullptr || src_dev == nullptr) return kb::fail("Invalid GPU pointer");
    if (memory_size == 0) return 0;
    kb::note_array_written(dst_dev);
    if (dst_device == src_device) {
        KB_HIP_TRY(hipMemcpy(dst_dev, src_dev, memory_size, hipMemcpyDeviceToDevice));
    } else {
        KB_HIP_TRY(hipMemcpyPeer(dst_dev, dst_device, src_dev, src_device, memory_size));
    }
    return 0;
}

int kb_device_synchronize(void) {
    KB_HIP_TRY(hipDeviceSynchronize());
    return 0;
}

}  // extern "C"
