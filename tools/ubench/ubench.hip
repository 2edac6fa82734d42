// Micro-benchmarks of the issue / LDS model of gfx950 that the search kernel's loop design rests on (not shipped;
// tools/ubench/run.sh builds and runs it on the GPU box).  Every kernel runs 256 workgroups x 1024 threads (one
// 16-wave workgroup per CU, 4 waves per SIMD -- the geometry of kb_search_lds) unless stated, and reports shader
// cycles per loop iteration per wave (s_memtime) next to the wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ubench_asm.h"

typedef float PairF __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) int* ConstIntPtr;
typedef const __attribute__((address_space(3))) PairF* LdsPair;
typedef const __attribute__((address_space(3))) float* LdsFloat;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t now() { return __builtin_amdgcn_s_memtime(); }

// MODE 0: 8 x (v_add addr, ds_read_b64) ; wait ; 8 x v_pk_add_f32; offsets live in SGPRs for the whole loop
// MODE 1: the same, offsets fetched per iteration by s_load_dwordx8 from a table (next iteration's fetched behind the reads)
// MODE 2: MODE 0 software-pipelined: reads of iteration i+1 issued before the adds of iteration i (counted lgkmcnt)
// MODE 3: MODE 1 pipelined the same way is impossible (SMEM forces lgkmcnt(0)); instead offsets from LDS (ds_read_b128 x2, broadcast)
// MODE 4: ds_read_addtid_b32 x 2 per sample with M0 = scalar offset (planar psi / phi), no VALU address
// MODE 5: plain v_add_f32 x 16 instead of 8 v_pk_add (reads as MODE 0)
// MODE 6: no LDS at all: 8 v_pk_add per iteration on registers (VALU issue rate)
// MODE 7: MODE 0 without the adds (LDS read rate alone)
template <int MODE>
__global__ __launch_bounds__(1024) void loop_kernel(float* out, const int* table, int iters, int stride, uint64_t* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 40960; i += 1024) reinterpret_cast<float*>(smem)[i] = (float)(i & 7);
    __syncthreads();
    PairF acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = PairF{0.0f, 0.0f};
    const ConstIntPtr tab = (ConstIntPtr)(uintptr_t)table;
    uint32_t base = (uint32_t)(uintptr_t)smem + (wv * 72 + lane) * 8;
    const uint32_t base0 = base;
    int o[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = tab[c];
    const uint64_t t0 = now();
    if constexpr (MODE == 0 || MODE == 5 || MODE == 7) {
        for (int it = 0; it < iters; ++it) {
            PairF raw[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) raw[c] = *(LdsPair)(base + o[c]);
            __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if constexpr (MODE == 0) acc[c] += raw[c];
                if constexpr (MODE == 5) { acc[c].x += raw[c].x; asm volatile("" : "+v"(acc[c].x)); acc[c].y += raw[c].y; asm volatile("" : "+v"(acc[c].y)); }
                if constexpr (MODE == 7) asm volatile("" :: "v"(raw[c]));
            }
            base += stride;
            if ((it & 3) == 3) base = base0;
        }
    } else if constexpr (MODE == 1) {
        int on[8];
        for (int it = 0; it < iters; ++it) {
            PairF raw[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) raw[c] = *(LdsPair)(base + o[c]);
            __builtin_amdgcn_sched_barrier(0);
            ConstIntPtr p = tab + (it + 1) * 8;
#pragma unroll
            for (int c = 0; c < 8; ++c) on[c] = p[c];
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] += raw[c];
#pragma unroll
            for (int c = 0; c < 8; ++c) o[c] = on[c];
            base += stride;
            if ((it & 3) == 3) base = base0;
        }
    } else if constexpr (MODE == 2) {
        PairF ra[8], rb[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) ra[c] = *(LdsPair)(base + o[c]);
        for (int it = 0; it < iters; it += 2) {
            base += stride;
#pragma unroll
            for (int c = 0; c < 8; ++c) rb[c] = *(LdsPair)(base + o[c]);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC87F);  // lgkmcnt(8)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] += ra[c];
            __builtin_amdgcn_sched_barrier(0);
            base += stride;
            if ((it & 2) == 2) base = base0;
#pragma unroll
            for (int c = 0; c < 8; ++c) ra[c] = *(LdsPair)(base + o[c]);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC87F);
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] += rb[c];
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
        for (int c = 0; c < 8; ++c) asm volatile("" :: "v"(ra[c]));
    } else if constexpr (MODE == 3) {
        // offsets of the next iteration from LDS (uniform address: broadcast), all waits counted
        typedef int Int4 __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(3))) Int4* LdsInt4;
        const uint32_t tbase = (uint32_t)(uintptr_t)smem + 163840 - 4096;  // a table of 128 iterations' offsets at the end of LDS
        if (tid < 1024) reinterpret_cast<int*>(smem + 163840 - 4096)[tid] = table[tid & 7];
        __syncthreads();
        Int4 oa = *(LdsInt4)(tbase), ob = *(LdsInt4)(tbase + 16);
        for (int it = 0; it < iters; ++it) {
            PairF raw[8];
            __builtin_amdgcn_s_waitcnt(0xC07F);
            const int ov[8] = {oa.x, oa.y, oa.z, oa.w, ob.x, ob.y, ob.z, ob.w};
#pragma unroll
            for (int c = 0; c < 8; ++c) raw[c] = *(LdsPair)(base + ov[c]);
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t ta = tbase + ((it + 1) & 127) * 32;
            oa = *(LdsInt4)(ta);
            ob = *(LdsInt4)(ta + 16);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC27F);  // lgkmcnt(2): the 8 sample reads are back
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] += raw[c];
            base += stride;
            if ((it & 3) == 3) base = base0;
        }
    } else if constexpr (MODE == 4) {
        // planar: psi plane at smem, phi plane at smem + 81920; M0 carries the wave's row base + the sample's offset
        uint32_t sbase = (uint32_t)(uintptr_t)smem + wv * 72 * 4;
        const uint32_t sbase0 = sbase;
        for (int it = 0; it < iters; ++it) {
            PairF raw[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint32_t m = sbase + (uint32_t)(o[c] >> 1);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tds_read_addtid_b32 %0 offset:0\n\tds_read_addtid_b32 %1 offset:40960"
                             : "=v"(raw[c].x), "=v"(raw[c].y) : "s"(m) : "memory");
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] += raw[c];
            sbase += stride >> 1;
            if ((it & 3) == 3) sbase = sbase0;
        }
    } else if constexpr (MODE == 8 || MODE == 9 || MODE == 10) {
        // MODE 8: sixteen samples per wait (the wide-chunk loop): 16 x (v_add, ds_read_b64); wait; 16 x v_pk_add
        // MODE 9: TWO START-PIXEL ROWS PER WAVE: 8 v_add, 16 ds_read_b64 (the second row's read at an immediate offset from the
        //         same address register); wait; 16 v_pk_add -- run with half the waves
        // MODE 10: two rows per wave, sixteen candidates: 16 v_add, 32 ds_read_b64; wait; 32 v_pk_add (half the waves, 64 accumulators)
        constexpr int NA = MODE == 10 ? 32 : 16;
        PairF acc2[NA];
#pragma unroll
        for (int c = 0; c < NA; ++c) acc2[c] = PairF{0.0f, 0.0f};
        int o2[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) o2[c] = tab[c & 7] + (c >> 3) * 40;
        for (int it = 0; it < iters; ++it) {
            PairF raw[NA];
            if constexpr (MODE == 8) {
#pragma unroll
                for (int c = 0; c < 16; ++c) raw[c] = *(LdsPair)(base + o2[c]);
            } else if constexpr (MODE == 9) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const uint32_t ad = base + o2[c];
                    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:704" : "=&v"(raw[c]), "=&v"(raw[c + 8]) : "v"(ad) : "memory");
                }
            } else {
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const uint32_t ad = base + o2[c];
                    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:704" : "=&v"(raw[c]), "=&v"(raw[c + 16]) : "v"(ad) : "memory");
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
            for (int c = 0; c < NA; ++c) { asm volatile("" : "+v"(raw[c])); acc2[c] += raw[c]; }
            base += stride;
            if ((it & 3) == 3) base = base0;
        }
#pragma unroll
        for (int c = 0; c < NA; ++c) acc[c & 7] += acc2[c];
    } else if constexpr (MODE >= 11 && MODE <= 21) {
        // asm-exact loops (ubench_asm.h): 11 today's mix, 12 two rows x 8 candidates, 13 two rows x 16 candidates, 14 four rows x 8
        int so[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) so[c] = __builtin_amdgcn_readfirstlane(tab[c & 7] + (c >> 3) * 48);
        uint32_t b = base;
        float r = 0.0f;
        const int n = __builtin_amdgcn_readfirstlane(iters);
        if constexpr (MODE == 11) { KB_UB_TODAY(b, n, so, r) }
        if constexpr (MODE == 12) { KB_UB_ROWS2(b, n, so, r) }
        if constexpr (MODE == 13) { KB_UB_ROWS2W(b, n, so, r) }
        if constexpr (MODE == 14) { KB_UB_ROWS4(b, n, so, r) }
        // prefix sharing: 1 + (iteration & 15) of the 16 units per iteration, entered by computed jumps; 16: the jumps with all 16 units
        if constexpr (MODE == 15) { KB_UB_CLASS(b, n, so, r) }
        if constexpr (MODE == 16) { KB_UB_CLASS16(b, n, so, r) }
        // addresses kept in registers, slots as immediates: no / 5 / 16 address updates per 16 samples
        if constexpr (MODE == 17) { KB_UB_FIXED0(b, n, so, r) }
        if constexpr (MODE == 18) { KB_UB_FIXED5(b, n, so, r) }
        if constexpr (MODE == 19) { KB_UB_FIXED16(b, n, so, r) }
        if constexpr (MODE == 20) { KB_UB_BRANCHY5(b, n, so, r) }
        if constexpr (MODE == 21) { KB_UB_BRANCHY0(b, n, so, r) }
        acc[0].x += r;
    } else if constexpr (MODE == 6) {
        PairF one = PairF{1.0f, (float)lane};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int c = 0; c < 8; ++c) { acc[c] += one; asm volatile("" : "+v"(acc[c])); }
        }
    }
    const uint64_t t1 = now();
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += acc[c].x + acc[c].y;
    out[blockIdx.x * 1024 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int threads, int iters, int stride, float* out, int* table, uint64_t* cyc) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(loop_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(loop_kernel<MODE>, dim3(256), dim3(threads), 163840, 0, out, table, iters, stride, cyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
    }
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    uint64_t c = 0;
    CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-44s threads %4d  %8.3f ms  %7.1f memtime-ticks/iter  (%.1f ns/iter)\n", name, threads, ms, (double)c / iters, ms * 1e6 / iters);
}

int main() {
    float* out;
    int* table;
    uint64_t* cyc;
    CHECK(hipMalloc(&out, 256 * 1024 * 4));
    CHECK(hipMalloc(&table, 1 << 20));
    CHECK(hipMalloc(&cyc, 8));
    std::vector<int> h(1 << 18);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (int)((i % 8) * 16 + ((i / 8) % 5) * 8);
    CHECK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int iters = 8192;
    for (int threads : {1024, 512, 256}) {
        run<6>("6 valu only: 8 v_pk_add", threads, iters, 576, out, table, cyc);
        run<7>("7 lds only: 8 (v_add + ds_read_b64)", threads, iters, 576, out, table, cyc);
        run<0>("0 reads + wait + 8 pk_add", threads, iters, 576, out, table, cyc);
        run<5>("5 reads + wait + 16 v_add_f32", threads, iters, 576, out, table, cyc);
        run<1>("1 mode 0 + s_load of next offsets", threads, iters, 576, out, table, cyc);
        run<2>("2 mode 0 software-pipelined", threads, iters, 576, out, table, cyc);
        run<3>("3 offsets from LDS, counted waits", threads, iters, 576, out, table, cyc);
        run<4>("4 ds_read_addtid_b32 x2, M0", threads, iters, 288, out, table, cyc);
    }
    // the same number of evaluations every line: 16 waves x 8192 iterations x 8 samples per CU
    run<0>("0  8 cand x 1 row, 16 waves/CU", 1024, iters, 576, out, table, cyc);
    run<8>("8  16 cand x 1 row, 16 waves/CU (today)", 1024, iters / 2, 576, out, table, cyc);
    run<9>("9  8 cand x 2 rows, 8 waves/CU", 512, iters, 576, out, table, cyc);
    run<10>("10 16 cand x 2 rows, 8 waves/CU", 512, iters / 2, 576, out, table, cyc);
    run<11>("11 asm: 16 cand x 1 row (today), 16 waves", 1024, iters / 2, 576, out, table, cyc);
    run<12>("12 asm: 8 cand x 2 rows, 16 waves", 1024, iters / 2, 576, out, table, cyc);
    run<13>("13 asm: 16 cand x 2 rows (2 batches), 16 waves", 1024, iters / 4, 576, out, table, cyc);
    run<14>("14 asm: 8 cand x 4 rows (2 batches), 16 waves", 1024, iters / 4, 576, out, table, cyc);
    run<15>("15 asm: classes, mean 8.5 of 16 units, 2 jumps", 1024, iters / 2, 576, out, table, cyc);
    run<16>("16 asm: all 16 units + the 2 jumps", 1024, iters / 2, 576, out, table, cyc);
    run<17>("17 asm: addresses kept, 0 updates per 16", 1024, iters / 2, 576, out, table, cyc);
    run<18>("18 asm: addresses kept, 5 updates per 16", 1024, iters / 2, 576, out, table, cyc);
    run<19>("19 asm: addresses kept, 16 updates per 16", 1024, iters / 2, 576, out, table, cyc);
    run<20>("20 asm: kept, 5 of 16 updated by branches", 1024, iters / 2, 576, out, table, cyc);
    run<21>("21 asm: kept, 16 tests, no update taken", 1024, iters / 2, 576, out, table, cyc);
    run<11>("11 asm: 16 cand x 1 row (today), 16 waves", 1024, iters / 2, 576, out, table, cyc);
    run<12>("12 asm: 8 cand x 2 rows, 8 waves", 512, iters, 576, out, table, cyc);
    run<13>("13 asm: 16 cand x 2 rows (2 batches), 8 waves", 512, iters / 2, 576, out, table, cyc);
    run<9>("9  8 cand x 2 rows, 16 waves/CU", 1024, iters / 2, 576, out, table, cyc);
    run<10>("10 16 cand x 2 rows, 16 waves/CU", 1024, iters / 4, 576, out, table, cyc);
    return 0;
}
