// Host image utilities of kbmod_amd.search: masked PSF correlation and the
// psi/phi pixel preparation.  Mirrors image_utils_cpp.{h,cpp}:20-177 and
// kernel_helpers.cpp:23-117 of the reference (same names and error messages).
// With a GPU present convolve_image / generate_psi / generate_phi run on the
// device (libkbmod_hip.so); the *_cpu variants are the explicit host versions
// the reference also exposes.
#ifndef KBH_IMAGE_UTILS_H_
#define KBH_IMAGE_UTILS_H_

#include <algorithm>
#include <cstdio>

#include "common.h"

namespace search {

// ---- kernel_helpers.cpp:23-79 ------------------------------------------------
inline bool has_gpu() { return kb_device_count() > 0; }
inline void print_cuda_stats() { kb_print_stats(); }
inline size_t get_gpu_total_memory() { return kb_gpu_total_memory(); }
inline size_t get_gpu_free_memory() { return kb_gpu_free_memory(); }
inline std::string stat_gpu_memory_mb() {
    // same text as kernel_helpers.cpp:62-66 ("%f" is what std::to_string prints for a double)
    char line[160];
    std::snprintf(line, sizeof(line), "GPU: %f MB free of %f MB total.", (double)kb_gpu_free_memory() / 1048576.0,
                  (double)kb_gpu_total_memory() / 1048576.0);
    return std::string(line);
}
inline bool validate_gpu(size_t req_memory = 0) { return kb_check_gpu(req_memory) != 0; }

// Test hook of the in-search sigma-G index computation (kernel_helpers.cpp:86-106): the original positions of the
// values that survive the clip, in ascending value order.  Runs the host instantiation of the shared evaluator
// (csrc/search_math.h) through the C ABI; no device needed.
inline std::vector<int> sigmaGFilteredIndices(std::vector<float> values, float sgl0, float sgl1,
                                              float sigma_g_coeff, float width) {
    const int n = (int)values.size();
    std::vector<int> order(n, 0);
    int first = 0, last = n - 1;
    kb_sigmag_filtered_indices(values.data(), n, sgl0, sgl1, sigma_g_coeff, width, order.data(), &first, &last);
    if (last < first) return {};
    return std::vector<int>(order.begin() + first, order.begin() + last + 1);
}

// ---- masked, renormalised correlation on flat row-major buffers ------------------------------------------
// Semantics of image_utils_cpp.cpp:20-68: a non-finite centre passes through; otherwise the taps that fall
// inside the image on finite pixels are summed row by row, left to right (no kernel flip), products and weights
// in separate fp32 accumulators, and the result is (sum * kernel_total) / weight_seen -- NO_DATA when no tap
// counted.  The loops below walk only the part of the kernel that overlaps the image (clipped tap ranges per
// output row / column), which visits the surviving taps in the same order as a per-tap bounds test would, so
// every rounding happens in the same place.  One routine serves the module's convolve_image_cpu and the host
// builders of psi / phi.
inline float kernel_total(const float* k, int64_t kh, int64_t kw) {
    float total = 0.0f;
    for (int64_t p = 0; p < kh * kw; ++p) total += k[p];  // row-major order
    return total;
}

inline void masked_correlate(const float* src, int64_t height, int64_t width, const float* k, int64_t kh, int64_t kw,
                             float* dst) {
    const int64_t reach = (kh - 1) / 2;  // the reference takes the radius from the row count for both axes
    const float total = kernel_total(k, kh, kw);
#pragma omp parallel for schedule(static)
    for (int64_t y = 0; y < height; ++y) {
        const int64_t j_lo = std::max<int64_t>(-reach, -y), j_hi = std::min<int64_t>(reach, height - 1 - y);
        const float* centre_row = src + y * width;
        float* out_row = dst + y * width;
        for (int64_t x = 0; x < width; ++x) {
            const float centre = centre_row[x];
            if (!std::isfinite(centre)) {
                out_row[x] = centre;
                continue;
            }
            const int64_t i_lo = std::max<int64_t>(-reach, -x), i_hi = std::min<int64_t>(reach, width - 1 - x);
            float acc = 0.0f, seen = 0.0f;
            for (int64_t j = j_lo; j <= j_hi; ++j) {
                const float* px = centre_row + j * width + x;
                const float* kr = k + (j + reach) * kw + reach;
                for (int64_t i = i_lo; i <= i_hi; ++i) {
                    const float v = px[i];
                    if (std::isfinite(v)) {
                        const float w = kr[i];
                        seen += w;
                        acc += v * w;
                    }
                }
            }
            out_row[x] = (seen == 0.0f) ? NO_DATA : (acc * total) / seen;
        }
    }
}

inline Image convolve_image_cpu(const Image& img, const Image& psf) {
    Image result(img.rows, img.cols);
    masked_correlate(img.data.data(), img.rows, img.cols, psf.data.data(), psf.rows, psf.cols, result.data.data());
    return result;
}

// image_utils_cpp.cpp:70-101
inline Image convolve_image_gpu(const Image& img, const Image& psf) {
    if (!has_gpu()) throw std::runtime_error("Unable to perform convolve_image_gpu() without GPU.");
    if (psf.rows != psf.cols) throw std::runtime_error("PSF kernel must be square.");
    Image result(img.rows, img.cols);
    const int radius = (int)((psf.rows - 1) / 2);
    // empty_is_nan = 0: the reference's device kernel value (image_kernels.cu:61).
    check_status(kb_device_convolve(img.data.data(), result.data.data(), (int)img.cols, (int)img.rows,
                                    psf.data.data(), radius, 0));
    return result;
}

// image_utils_cpp.cpp:103-108
inline Image convolve_image(const Image& image, const Image& psf) {
    if (has_gpu()) return convolve_image_gpu(image, psf);
    return convolve_image_cpu(image, psf);
}

// image_utils_cpp.cpp:110-120
inline Image square_psf_values(const Image& given_psf) {
    Image psf_sq = given_psf;
    for (size_t i = 0; i < psf_sq.data.size(); ++i) psf_sq.data[i] = given_psf.data[i] * given_psf.data[i];
    return psf_sq;
}

inline void check_same_dims(const Image& sci, const Image& var) {
    if ((sci.rows != var.rows) || (sci.cols != var.cols)) {
        throw std::runtime_error("Science and Variance images must be the same dimensions.  Sci = (" +
                                 std::to_string(sci.rows) + "," + std::to_string(sci.cols) + "), Var = (" +
                                 std::to_string(var.rows) + "," + std::to_string(var.cols) + ").");
    }
}

// Pixel preparation of image_utils_cpp.cpp:142-149 / :165-172: psi0 = sci / var (fp32 divide), phi0 = 1 / var
// (a double divide rounded to float on store); NO_DATA where the variance is non-finite or zero, psi0 also where
// the science pixel is non-finite.  A negative variance is not masked.
inline bool variance_usable(float v) { return std::isfinite(v) && v != 0.0f; }
inline Image prepare_psi(const Image& sci, const Image& var) {
    Image out(sci.rows, sci.cols);
    std::transform(sci.data.begin(), sci.data.end(), var.data.begin(), out.data.begin(), [](float s, float v) {
        return (variance_usable(v) && std::isfinite(s)) ? s / v : NO_DATA;
    });
    return out;
}
inline Image prepare_phi(const Image& var) {
    Image out(var.rows, var.cols);
    std::transform(var.data.begin(), var.data.end(), out.data.begin(),
                   [](float v) { return variance_usable(v) ? (float)(1.0 / (double)v) : NO_DATA; });
    return out;
}

inline Image generate_psi_cpu(const Image& sci, const Image& var, const Image& psf) {
    check_same_dims(sci, var);
    return convolve_image_cpu(prepare_psi(sci, var), psf);
}
inline Image generate_phi_cpu(const Image& var, const Image& psf) {
    return convolve_image_cpu(prepare_phi(var), square_psf_values(psf));
}

// image_utils_cpp.cpp:126-153
inline Image generate_psi(const Image& sci, const Image& var, const Image& psf) {
    check_same_dims(sci, var);
    if (has_gpu()) return convolve_image_gpu(prepare_psi(sci, var), psf);
    return convolve_image_cpu(prepare_psi(sci, var), psf);
}
// image_utils_cpp.cpp:155-177
inline Image generate_phi(const Image& var, const Image& psf) {
    Image psfsq = square_psf_values(psf);
    if (has_gpu()) return convolve_image_gpu(prepare_phi(var), psfsq);
    return convolve_image_cpu(prepare_phi(var), psfsq);
}

}  // namespace search
#endif
