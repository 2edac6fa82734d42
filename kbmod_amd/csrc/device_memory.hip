// Device discovery and raw HBM block management for libkbmod_hip.so.
// Replaces kernels/kernel_memory.cu:15-136 of the reference (same roles, HIP
// runtime underneath, status codes instead of exceptions across the C ABI).
#include "kb_common.h"

namespace kb {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
int fail(const std::string& msg) {
    set_error(msg);
    return 1;
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
    KB_HIP_TRY(hipFree(ptr_dev));
    return 0;
}

int kb_copy_block_to_gpu(const void* src_host, void* dst_dev, uint64_t memory_size) {  // :112-122
    if (src_host == nullptr) return kb::fail("Invalid CPU pointer");
    if (dst_dev == nullptr) return kb::fail("Invalid GPU pointer");
    KB_HIP_TRY(hipMemcpy(dst_dev, src_host, memory_size, hipMemcpyHostToDevice));
    return 0;
}

int kb_copy_block_to_cpu(void* dst_host, const void* src_dev, uint64_t memory_size) {  // :124-134
    if (dst_host == nullptr) return kb::fail("Invalid CPU pointer");
    if (src_dev == nullptr) return kb::fail("Invalid GPU pointer");
    KB_HIP_TRY(hipMemcpy(dst_host, src_dev, memory_size, hipMemcpyDeviceToHost));
    return 0;
}

int kb_device_synchronize(void) {
    KB_HIP_TRY(hipDeviceSynchronize());
    return 0;
}

}  // extern "C"
