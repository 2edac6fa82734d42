// Stamp coadds of result trajectories on the device -- SURVEY.md section 8(f3).
//
// Replaces, for a batch of trajectories, the per-trajectory host loop of append_coadds
// (src/kbmod/filters/stamp_filters.py:72-168): extract_stamp_stack (src/kbmod/core/stamp_utils.py:16-84,
// 352-397: (2r+1)^2 cut-outs, NaN where there is no image) followed by coadd_sum / coadd_mean /
// coadd_median / coadd_weighted (stamp_utils.py:241-344).  The reference forms the stamps in float64
// and reduces over time slice by slice (np.nansum / np.nanmean / np.sum over axis 0 add the slices in
// order, starting from slice 0) and stores float32; the kernel keeps exactly that arithmetic: one
// thread per stamp pixel, a double accumulator, epochs in order, one rounding to float at the end.
// The median is torch.nanmedian's lower median of the non-NaN values.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <mutex>
#include <string>

#include "kb_common.h"
#include "kbmod_hip.h"

namespace kb {

struct CoaddArgs {
    const float* sci;      // [T][H][W]
    const float* var;      // [T][H][W] or null
    const int32_t* x;      // [N][T] stamp centres
    const int32_t* y;      // [N][T]
    const uint8_t* include;  // [N][T] or null (all epochs)
    float* out;            // [N][S][S]
    float* scratch;        // median only, long stacks: [batch][T][S*S]
    int lds_pixels;        // median only: stamp pixels per pass whose T-long columns fit the LDS buffer (0: use scratch)
    uint64_t n0;           // first trajectory of this launch
    int T, H, W, radius, S;
};

__device__ __forceinline__ float stamp_pixel(const float* __restrict__ img, int H, int W, int cx, int cy, int r, int j,
                                             int i) {
    const int xx = cx - r + i, yy = cy - r + j;
    if ((unsigned)xx >= (unsigned)W || (unsigned)yy >= (unsigned)H) return __uint_as_float(0x7fc00000u);
    return img[(size_t)yy * W + xx];
}

// TYPE: KB_COADD_SUM / MEAN / MEDIAN / WEIGHTED.  One workgroup per trajectory.
template <int TYPE>
__global__ __launch_bounds__(256) void kb_coadd_kernel(const CoaddArgs a) {
    const uint64_t n = a.n0 + blockIdx.x;
    const int S2 = a.S * a.S;
    const int32_t* __restrict__ xs = a.x + n * (uint64_t)a.T;
    const int32_t* __restrict__ ys = a.y + n * (uint64_t)a.T;
    const uint8_t* __restrict__ inc = a.include ? a.include + n * (uint64_t)a.T : nullptr;
    const size_t image = (size_t)a.H * a.W;
    // median: the used values of one pixel form a column, kept in LDS ([slot][lds_pixels], this thread's
    // pixel = one bank-conflict-free lane of it) or, for stacks too long for that, in an HBM scratch
    extern __shared__ float lds_col[];
    const bool in_lds = TYPE == KB_COADD_MEDIAN && a.lds_pixels > 0;
    float* __restrict__ col = in_lds ? lds_col + threadIdx.x
                                     : (a.scratch ? a.scratch + (size_t)blockIdx.x * a.T * S2 : nullptr);
    const int col_stride = in_lds ? a.lds_pixels : S2;
    const int per_pass = in_lds ? a.lds_pixels : (int)blockDim.x;
    for (int pix0 = 0; pix0 < S2; pix0 += per_pass) {
        const int pix = pix0 + (int)threadIdx.x;
        if ((int)threadIdx.x >= per_pass || pix >= S2) continue;
        float* __restrict__ mycol = in_lds ? col : col + pix;
        const int j = pix / a.S, i = pix - j * a.S;
        double sum = 0.0, wsum = 0.0;
        int used = 0, n_valid = 0;
        bool first = true;
        for (int t = 0; t < a.T; ++t) {
            if (inc && !inc[t]) continue;
            const float v = stamp_pixel(a.sci + t * image, a.H, a.W, xs[t], ys[t], a.radius, j, i);
            const bool is_nan = v != v;
            n_valid += is_nan ? 0 : 1;
            if constexpr (TYPE == KB_COADD_MEDIAN) {
                mycol[(size_t)used * col_stride] = v;
            } else if constexpr (TYPE == KB_COADD_WEIGHTED) {
                // weights = 1 / var and sci * weights where sci, var are not NaN and var != 0; zeros elsewhere
                const float vv = stamp_pixel(a.var + t * image, a.H, a.W, xs[t], ys[t], a.radius, j, i);
                const bool ok = !is_nan && !(vv != vv) && vv != 0.0f;
                const double w = ok ? 1.0 / (double)vv : 0.0;
                const double ws = ok ? __dmul_rn((double)v, w) : 0.0;
                sum = first ? ws : __dadd_rn(sum, ws);
                wsum = first ? w : __dadd_rn(wsum, w);
            } else {
                const double term = is_nan ? 0.0 : (double)v;  // np.nansum / np.nanmean: NaN -> 0
                sum = first ? term : __dadd_rn(sum, term);
            }
            first = false;
            ++used;
        }
        double res;
        if (used == 0) {
            res = 0.0;  // no epoch selected: zeros (stamp_utils.py:271-272, 294-295)
        } else if constexpr (TYPE == KB_COADD_SUM) {
            res = sum;
        } else if constexpr (TYPE == KB_COADD_MEAN) {
            // a pixel that is NaN at every epoch is set to 0 first (_mask_all_nans): 0 / used
            res = (n_valid == 0) ? 0.0 : sum / (double)n_valid;
        } else if constexpr (TYPE == KB_COADD_WEIGHTED) {
            // all-NaN pixels become zeros with whatever weights their variances give: 0 / sum(w) = 0
            if (n_valid == 0) {
                res = 0.0;
            } else {
                res = sum / ((wsum == 0.0) ? 1e24 : wsum);
            }
        } else {
            // lower median of the non-NaN values: the value with exactly (n_valid - 1) / 2 values before
            // it in the order (value, epoch slot)
            res = 0.0;
            if (n_valid > 0) {
                const int k = (n_valid - 1) / 2;
                for (int c = 0; c < used; ++c) {
                    const float vc = mycol[(size_t)c * col_stride];
                    if (vc != vc) continue;
                    int before = 0;
                    for (int u = 0; u < used; ++u) {
                        const float vu = mycol[(size_t)u * col_stride];
                        before += (vu < vc || (vu == vc && u < c)) ? 1 : 0;
                    }
                    if (before == k) {
                        res = (double)vc;
                        break;
                    }
                }
            }
        }
        a.out[n * (uint64_t)S2 + pix] = (float)res;
    }
}

// Median coadd for stacks of at most 64 epochs: the pixel's values live in 64 registers as
// order-preserving keys (NaN and unused slots = maximum) and go through a fully unrolled bitonic
// network (672 compare-exchanges of v_min_u32 / v_max_u32); the lower median is the key at index
// (n_valid - 1) / 2.  No scratch, no data-dependent loop.
__device__ __forceinline__ uint32_t median_key(float v) {
    if (v != v) return 0xffffffffu;
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ __launch_bounds__(256) void kb_coadd_median64_kernel(const CoaddArgs a) {
    const uint64_t n = a.n0 + blockIdx.x;
    const int S2 = a.S * a.S;
    const int32_t* __restrict__ xs = a.x + n * (uint64_t)a.T;
    const int32_t* __restrict__ ys = a.y + n * (uint64_t)a.T;
    const uint8_t* __restrict__ inc = a.include ? a.include + n * (uint64_t)a.T : nullptr;
    const size_t image = (size_t)a.H * a.W;
    for (int pix = threadIdx.x; pix < S2; pix += blockDim.x) {
        const int j = pix / a.S, i = pix - j * a.S;
        uint32_t key[64];
        int n_valid = 0, used = 0;
#pragma unroll
        for (int t = 0; t < 64; ++t) {
            float v = __uint_as_float(0x7fc00000u);
            if (t < a.T && (!inc || inc[t])) {  // uniform
                v = stamp_pixel(a.sci + t * image, a.H, a.W, xs[t], ys[t], a.radius, j, i);
                ++used;
            }
            key[t] = median_key(v);
            n_valid += (v == v) ? 1 : 0;
        }
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
            for (int jj = k >> 1; jj > 0; jj >>= 1) {
#pragma unroll
                for (int p = 0; p < 64; ++p) {
                    const int q = p ^ jj;
                    if (q > p) {
                        const uint32_t lo = min(key[p], key[q]), hi = max(key[p], key[q]);
                        const bool up = (p & k) == 0;
                        key[p] = up ? lo : hi;
                        key[q] = up ? hi : lo;
                    }
                }
            }
        }
        float res = 0.0f;  // nothing selected or nothing valid: zero (stamp_utils.py:294-302)
        if (n_valid > 0) {
            const int m = (n_valid - 1) / 2;
            uint32_t km = 0;
#pragma unroll
            for (int t = 0; t < 64; ++t) km = (t == m) ? key[t] : km;
            res = __uint_as_float((km & 0x80000000u) ? (km & 0x7fffffffu) : ~km);
        }
        (void)used;
        a.out[n * (uint64_t)S2 + pix] = res;
    }
}

// All stamps of every trajectory (append_all_stamps, stamp_filters.py:171-211): out[n][t] is the
// (2r+1)^2 stamp of epoch t around (x[n][t], y[n][t]), NaN outside the image.  One workgroup per
// trajectory; consecutive threads take consecutive stamp pixels (rows of 2r+1 contiguous image pixels).
__global__ __launch_bounds__(256) void kb_extract_stamps_kernel(const CoaddArgs a) {
    const uint64_t n = a.n0 + blockIdx.x;
    const int S2 = a.S * a.S;
    const int32_t* __restrict__ xs = a.x + n * (uint64_t)a.T;
    const int32_t* __restrict__ ys = a.y + n * (uint64_t)a.T;
    const size_t image = (size_t)a.H * a.W;
    float* __restrict__ out = a.out + n * (uint64_t)a.T * S2;
    for (int e = threadIdx.x; e < a.T * S2; e += blockDim.x) {
        const int t = e / S2, pix = e - t * S2;
        const int j = pix / a.S, i = pix - j * a.S;
        out[e] = stamp_pixel(a.sci + t * image, a.H, a.W, xs[t], ys[t], a.radius, j, i);
    }
}

static std::mutex g_scratch_mutex;
static float* g_scratch = nullptr;
static size_t g_scratch_bytes = 0;
static int g_scratch_device = -1;

}  // namespace kb

extern "C" int kb_coadd_stamps(const float* sci_dev, const float* var_dev, int32_t num_times, int32_t height, int32_t width,
                               const int32_t* x_dev, const int32_t* y_dev, const uint8_t* include_dev, uint64_t n,
                               int32_t radius, int32_t coadd_type, float* out_dev, void* stream_v) {
    using namespace kb;
    if (radius <= 0) return fail("Invalid stamp radius " + std::to_string(radius));  // stamp_filters.py:89-90
    if (n == 0) return 0;
    KB_REQUIRE_DEVICE("stamp coadds.");
    if (sci_dev == nullptr || x_dev == nullptr || y_dev == nullptr || out_dev == nullptr) {
        return fail("coadd_stamps: null pointer");
    }
    if (num_times < 0 || height <= 0 || width <= 0) return fail("coadd_stamps: invalid image stack shape");
    if (coadd_type == KB_COADD_WEIGHTED && var_dev == nullptr) return fail("coadd_stamps: the weighted coadd needs the variance stack");
    if (coadd_type < KB_COADD_SUM || coadd_type > KB_COADD_WEIGHTED) return fail("coadd_stamps: unknown coadd type");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    CoaddArgs a;
    a.sci = sci_dev;
    a.var = var_dev;
    a.x = x_dev;
    a.y = y_dev;
    a.include = include_dev;
    a.out = out_dev;
    a.scratch = nullptr;
    a.T = num_times;
    a.H = height;
    a.W = width;
    a.radius = radius;
    a.S = 2 * radius + 1;
    const uint64_t s2 = (uint64_t)a.S * a.S;
    const unsigned threads = (unsigned)std::min<uint64_t>(256, (s2 + 63) / 64 * 64);
    uint64_t batch = std::min<uint64_t>(n, 1u << 20);
    std::unique_lock<std::mutex> lock(g_scratch_mutex, std::defer_lock);
    constexpr int MEDIAN_LDS_BYTES = 32768;
    a.lds_pixels = 0;
    size_t lds_bytes = 0;
    const bool median64 = coadd_type == KB_COADD_MEDIAN && num_times <= 64;
    if (coadd_type == KB_COADD_MEDIAN && !median64) {
        const int fit = num_times > 0 ? MEDIAN_LDS_BYTES / (int)(sizeof(float) * num_times) : (int)threads;
        if (fit >= 16) {
            a.lds_pixels = std::min<int>((int)threads, fit);
            lds_bytes = (size_t)a.lds_pixels * std::max(1, num_times) * sizeof(float);
        } else {
            // [batch][T][S*S] floats of scratch, at most 256 MiB per launch
            lock.lock();
            const uint64_t per = std::max<uint64_t>(1, (uint64_t)num_times) * s2 * sizeof(float);
            batch = std::max<uint64_t>(1, std::min<uint64_t>(batch, (256ull << 20) / per));
            const size_t need = (size_t)(batch * per);
            int dev = 0;
            KB_HIP_TRY(hipGetDevice(&dev));
            if (g_scratch_bytes < need || g_scratch_device != dev) {
                if (g_scratch) (void)hipFree(g_scratch);
                g_scratch = nullptr;
                g_scratch_bytes = 0;
                KB_HIP_TRY(hipMalloc(&g_scratch, need));
                g_scratch_bytes = need;
                g_scratch_device = dev;
            }
            a.scratch = g_scratch;
        }
    }
    for (uint64_t n0 = 0; n0 < n; n0 += batch) {
        a.n0 = n0;
        const unsigned blocks = (unsigned)std::min<uint64_t>(batch, n - n0);
        switch (coadd_type) {
            case KB_COADD_SUM:
                hipLaunchKernelGGL((kb_coadd_kernel<KB_COADD_SUM>), dim3(blocks), dim3(threads), 0, stream, a);
                break;
            case KB_COADD_MEAN:
                hipLaunchKernelGGL((kb_coadd_kernel<KB_COADD_MEAN>), dim3(blocks), dim3(threads), 0, stream, a);
                break;
            case KB_COADD_MEDIAN:
                if (median64) {
                    hipLaunchKernelGGL(kb_coadd_median64_kernel, dim3(blocks), dim3(threads), 0, stream, a);
                } else {
                    hipLaunchKernelGGL((kb_coadd_kernel<KB_COADD_MEDIAN>), dim3(blocks), dim3(threads), lds_bytes, stream, a);
                }
                break;
            default:
                hipLaunchKernelGGL((kb_coadd_kernel<KB_COADD_WEIGHTED>), dim3(blocks), dim3(threads), 0, stream, a);
                break;
        }
        KB_HIP_TRY(hipGetLastError());
    }
    KB_HIP_TRY(hipStreamSynchronize(stream));
    return 0;
}

extern "C" int kb_extract_stamps(const float* sci_dev, int32_t num_times, int32_t height, int32_t width, const int32_t* x_dev,
                                 const int32_t* y_dev, uint64_t n, int32_t radius, float* out_dev, void* stream_v) {
    using namespace kb;
    if (radius < 1) return fail("Invalid stamp radius: " + std::to_string(radius));  // stamp_filters.py:188-189
    if (n == 0 || num_times == 0) return 0;
    KB_REQUIRE_DEVICE("stamp extraction.");
    if (sci_dev == nullptr || x_dev == nullptr || y_dev == nullptr || out_dev == nullptr) {
        return fail("extract_stamps: null pointer");
    }
    if (num_times < 0 || height <= 0 || width <= 0) return fail("extract_stamps: invalid image stack shape");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    CoaddArgs a{};
    a.sci = sci_dev;
    a.x = x_dev;
    a.y = y_dev;
    a.out = out_dev;
    a.T = num_times;
    a.H = height;
    a.W = width;
    a.radius = radius;
    a.S = 2 * radius + 1;
    const uint64_t batch = 1u << 20;
    for (uint64_t n0 = 0; n0 < n; n0 += batch) {
        a.n0 = n0;
        hipLaunchKernelGGL(kb_extract_stamps_kernel, dim3((unsigned)std::min<uint64_t>(batch, n - n0)), dim3(256), 0, stream, a);
        KB_HIP_TRY(hipGetLastError());
    }
    KB_HIP_TRY(hipStreamSynchronize(stream));
    return 0;
}
