// Host image utilities of kbmod_amd.search: masked PSF correlation and the
// psi/phi pixel preparation.  Mirrors image_utils_cpp.{h,cpp}:20-177 and
// kernel_helpers.cpp:23-117 of the reference (same names and error messages).
// With a GPU present convolve_image / generate_psi / generate_phi run on the
// device (libkbmod_hip.so); the *_cpu variants are the explicit host versions
// the reference also exposes.
#ifndef KBH_IMAGE_UTILS_H_
#define KBH_IMAGE_UTILS_H_

#include "common.h"

namespace search {

// ---- kernel_helpers.cpp:23-79 ------------------------------------------------
inline bool has_gpu() { return kb_device_count() > 0; }
inline void print_cuda_stats() { kb_print_stats(); }
inline size_t get_gpu_total_memory() { return kb_gpu_total_memory(); }
inline size_t get_gpu_free_memory() { return kb_gpu_free_memory(); }
inline std::string stat_gpu_memory_mb() {
    double total_mb = (double)get_gpu_total_memory() / 1048576.0;
    double free_mb = (double)get_gpu_free_memory() / 1048576.0;
    return ("GPU: " + std::to_string(free_mb) + " MB free of " + std::to_string(total_mb) + " MB total.");
}
inline bool validate_gpu(size_t req_memory = 0) { return kb_check_gpu(req_memory) != 0; }

// kernel_helpers.cpp:86-106 (runs the host instantiation; no device needed)
inline std::vector<int> sigmaGFilteredIndices(std::vector<float> values, float sgl0, float sgl1,
                                              float sigma_g_coeff, float width) {
    int num_values = values.size();
    std::vector<int> idx_array(num_values, 0);
    int min_keep_idx = 0;
    int max_keep_idx = num_values - 1;
    kb_sigmag_filtered_indices(values.data(), num_values, sgl0, sgl1, sigma_g_coeff, width, idx_array.data(),
                               &min_keep_idx, &max_keep_idx);
    std::vector<int> result;
    for (int i = min_keep_idx; i <= max_keep_idx; ++i) result.push_back(idx_array[i]);
    return result;
}

// ---- image_utils_cpp.cpp:20-68 -------------------------------------------------
inline Image convolve_image_cpu(const Image& img, const Image& psf) {
    const int64_t img_height = img.rows;
    const int64_t img_width = img.cols;
    Image result(img_height, img_width);
    const int psf_num_rows = (int)psf.rows;
    const int psf_num_cols = (int)psf.cols;
    const int psf_rad = (int)((psf_num_rows - 1) / 2);

    float psf_total = 0.0f;
    for (int r = 0; r < psf_num_rows; ++r)
        for (int c = 0; c < psf_num_cols; ++c) psf_total += psf(r, c);

#pragma omp parallel for schedule(static)
    for (int64_t y = 0; y < img_height; ++y) {
        for (int64_t x = 0; x < img_width; ++x) {
            if (!pixel_value_valid(img(y, x))) {
                result(y, x) = img(y, x);
                continue;
            }
            float sum = 0.0f;
            float psf_portion = 0.0f;
            for (int j = -psf_rad; j <= psf_rad; j++) {
                for (int i = -psf_rad; i <= psf_rad; i++) {
                    if ((x + i >= 0) && (x + i < img_width) && (y + j >= 0) && (y + j < img_height)) {
                        float current_pixel = img(y + j, x + i);
                        if (pixel_value_valid(current_pixel)) {
                            float current_psf = psf(j + psf_rad, i + psf_rad);
                            psf_portion += current_psf;
                            sum += current_pixel * current_psf;
                        }
                    }
                }
            }
            if (psf_portion == 0) {
                result(y, x) = NO_DATA;
            } else {
                result(y, x) = (sum * psf_total) / psf_portion;
            }
        }
    }
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

// image_utils_cpp.cpp:142-149 (pixel preparation only)
inline Image prepare_psi(const Image& sci, const Image& var) {
    Image result(sci.rows, sci.cols);
    const size_t n = result.data.size();
    for (size_t p = 0; p < n; ++p) {
        float var_pix = var.data[p];
        if (std::isfinite(var_pix) && var_pix != 0.0 && std::isfinite(sci.data[p])) {
            result.data[p] = sci.data[p] / var_pix;
        } else {
            result.data[p] = NO_DATA;
        }
    }
    return result;
}
// image_utils_cpp.cpp:165-172
inline Image prepare_phi(const Image& var) {
    Image result(var.rows, var.cols);
    const size_t n = result.data.size();
    for (size_t p = 0; p < n; ++p) {
        float var_pix = var.data[p];
        if (std::isfinite(var_pix) && var_pix != 0.0) {
            result.data[p] = 1.0 / var_pix;  // double divide, stored to float
        } else {
            result.data[p] = NO_DATA;
        }
    }
    return result;
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
