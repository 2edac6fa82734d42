// Does LDS-DMA (global_load_lds_dwordx4: M0 = LDS base, lane-linear 16-byte pieces) reach every byte of the 160 KB a
// workgroup of kb_search_lds holds?  Each wave copies one 1 KiB piece from global memory to LDS offset `off` for a few offsets
// below and above 64 KiB, the workgroup reads the piece back with ds_read and writes it out; the host compares.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_dma_probe.hip -o /tmp/lds_dma_probe && /tmp/lds_dma_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void probe(const uint4* src, uint4* dst, const unsigned* offsets, int n_off) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int k = 0; k < n_off; ++k) {
        char* base = smem + offsets[k] + 1024 * wv;  // wave-uniform
        __builtin_amdgcn_global_load_lds(src + (size_t)(k * 4 + wv) * 64 + lane, (__attribute__((address_space(3))) void*)base, 16, 0, 0);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        __syncthreads();
        dst[(size_t)(k * 4 + wv) * 64 + lane] = *reinterpret_cast<const uint4*>(base + 16 * lane);
        __syncthreads();
    }
}

int main() {
    const unsigned offs[] = {0u, 4096u, 61440u, 65536u, 81920u, 131072u, 159744u - 3072u};
    const int n_off = sizeof(offs) / sizeof(offs[0]);
    const size_t n = (size_t)n_off * 4 * 64;
    std::vector<uint4> h(n), out(n);
    for (size_t i = 0; i < n; ++i) h[i] = make_uint4((unsigned)i, (unsigned)(i * 7 + 1), ~(unsigned)i, 0xabcd0000u + (unsigned)i);
    uint4 *src, *dst;
    unsigned* doff;
    hipMalloc(&src, n * 16);
    hipMalloc(&dst, n * 16);
    hipMalloc(&doff, sizeof(offs));
    hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
    hipMemset(dst, 0, n * 16);
    hipMemcpy(doff, offs, sizeof(offs), hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 160 * 1024, 0, src, dst, doff, n_off);
    if (hipDeviceSynchronize() != hipSuccess) {
        std::printf("launch failed: %s\n", hipGetErrorString(hipGetLastError()));
        return 1;
    }
    hipMemcpy(out.data(), dst, n * 16, hipMemcpyDeviceToHost);
    for (int k = 0; k < n_off; ++k) {
        size_t bad = 0;
        for (size_t i = (size_t)k * 256; i < (size_t)(k + 1) * 256; ++i) {
            bad += (out[i].x != h[i].x || out[i].y != h[i].y || out[i].z != h[i].z || out[i].w != h[i].w) ? 1 : 0;
        }
        std::printf("LDS offset %6u: %s (%zu of 256 pieces differ)\n", offs[k], bad ? "WRONG" : "ok", bad);
    }
    return 0;
}
