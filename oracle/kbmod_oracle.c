/*
 * kbmod_oracle.c -- TEST INFRASTRUCTURE ONLY (see kbmod_oracle.h).
 *
 * CPU restatement, in plain C, of the reference algorithm for the
 * shift-and-stack search path.  Every function cites the reference file:line
 * (relative to /root/reference/src/kbmod/search/) whose arithmetic it follows:
 * operand types, promotion to double, rounding points and summation order are
 * kept exactly, and this file must be compiled with -ffp-contract=off so that
 * no multiply-add is fused (the reference CPU build targets baseline x86-64,
 * CMakeLists.txt:56-61, which has no FMA).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load the library built from this file.
 */
#include "kbmod_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline int value_valid(float v) { return isfinite(v); } /* common.h:41 */

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ */
/* psi / phi generation                                                 */
/* ------------------------------------------------------------------ */

/* image_utils_cpp.cpp:20-68; gpu flavour: kernels/image_kernels.cu:29-66. */
void orc_convolve(const float* img, int height, int width, const float* psf, int psf_dim, float* out,
                  int gpu_flavour) {
    const int psf_rad = (psf_dim - 1) / 2; /* image_utils_cpp.cpp:27 */

    float psf_total = 0.0f; /* :30-35, row-major order */
    for (int r = 0; r < psf_dim; ++r)
        for (int c = 0; c < psf_dim; ++c) psf_total += psf[r * psf_dim + c];

    for (int y = 0; y < height; ++y) {
        for (int x = 0; x < width; ++x) {
            const float centre = img[(size_t)y * width + x];
            if (!value_valid(centre)) { /* :41-44 invalid centre passes through */
                out[(size_t)y * width + x] = centre;
                continue;
            }
            float sum = 0.0f;
            float psf_portion = 0.0f;
            for (int j = -psf_rad; j <= psf_rad; j++) {     /* :48 rows outer */
                for (int i = -psf_rad; i <= psf_rad; i++) { /* :49 cols inner */
                    if ((x + i >= 0) && (x + i < width) && (y + j >= 0) && (y + j < height)) {
                        const float current_pixel = img[(size_t)(y + j) * width + (x + i)];
                        if (value_valid(current_pixel)) {
                            const float current_psf = psf[(j + psf_rad) * psf_dim + (i + psf_rad)];
                            psf_portion += current_psf;
                            sum += current_pixel * current_psf; /* separate mul, add */
                        }
                    }
                }
            }
            if (psf_portion == 0) { /* :60-61 NaN on CPU; image_kernels.cu:61 0.0 on GPU */
                out[(size_t)y * width + x] = gpu_flavour ? 0.0f : NAN;
            } else {
                out[(size_t)y * width + x] = (sum * psf_total) / psf_portion; /* :63 */
            }
        }
    }
}

/* image_utils_cpp.cpp:110-120 */
void orc_square_psf(const float* psf, int n, float* out) {
    for (int i = 0; i < n; ++i) out[i] = psf[i] * psf[i];
}

/* image_utils_cpp.cpp:126-153 */
void orc_generate_psi(const float* sci, const float* var, int height, int width, const float* psf,
                      int psf_dim, float* out, int gpu_flavour) {
    const size_t n = (size_t)height * width;
    float* tmp = (float*)malloc(n * sizeof(float));
    for (size_t p = 0; p < n; ++p) {
        const float var_pix = var[p];
        if (isfinite(var_pix) && var_pix != 0.0 && isfinite(sci[p])) { /* :144 */
            tmp[p] = sci[p] / var_pix;                                   /* float divide :145 */
        } else {
            tmp[p] = NAN;
        }
    }
    orc_convolve(tmp, height, width, psf, psf_dim, out, gpu_flavour);
    free(tmp);
}

/* image_utils_cpp.cpp:155-177 */
void orc_generate_phi(const float* var, int height, int width, const float* psf, int psf_dim, float* out,
                      int gpu_flavour) {
    const size_t n = (size_t)height * width;
    float* tmp = (float*)malloc(n * sizeof(float));
    for (size_t p = 0; p < n; ++p) {
        const float var_pix = var[p];
        if (isfinite(var_pix) && var_pix != 0.0) {
            tmp[p] = (float)(1.0 / (double)var_pix); /* :168: 1.0 is double -> f64 divide, stored f32 */
        } else {
            tmp[p] = NAN;
        }
    }
    float* psfsq = (float*)malloc((size_t)psf_dim * psf_dim * sizeof(float));
    orc_square_psf(psf, psf_dim * psf_dim, psfsq);
    orc_convolve(tmp, height, width, psfsq, psf_dim, out, gpu_flavour);
    free(psfsq);
    free(tmp);
}

/* ------------------------------------------------------------------ */
/* encoding                                                             */
/* ------------------------------------------------------------------ */

/* psi_phi_array_ds.h:40-43.  std::min(a,b) = (b<a)?b:a ; std::max(a,b) = (a<b)?b:a. */
float orc_encode_uint_scalar(float value, float min_val, float max_val, float scale) {
    if (!value_valid(value)) return 0.0f;
    const float lo = (max_val < value) ? max_val : value;
    const float clamped = (lo < min_val) ? min_val : lo;
    return (float)((double)((clamped - min_val) / scale) + 1.0); /* f32 sub, f32 div, +1.0 in f64 */
}

/* psi_phi_array_ds.h:45-47 */
float orc_decode_uint_scalar(float value, float min_val, float scale) {
    if (value == 0.0) return NAN;
    return (float)(((double)value - 1.0) * (double)scale + (double)min_val);
}

/* psi_phi_array.cpp:219-245 */
void orc_scale_params(const float* imgs, uint64_t n_values, int num_bytes, float out[3]) {
    float min_val = FLT_MAX;
    float max_val = -FLT_MAX;
    for (uint64_t i = 0; i < n_values; ++i) {
        const float elem = imgs[i];
        if (value_valid(elem)) {
            min_val = (elem < min_val) ? elem : min_val; /* std::min(min_val, elem) */
            max_val = (max_val < elem) ? elem : max_val; /* std::max(max_val, elem) */
        }
    }
    float scale = 1.0f;
    if (num_bytes == 1 || num_bytes == 2) {
        float width = (max_val - min_val);
        if (width < 1e-6) width = 1e-6; /* :238 double literal, stored to float */
        const uint64_t num_values = (1 << (8 * num_bytes)) - 1;
        scale = (float)(width / (double)num_values); /* :241 */
    }
    out[0] = min_val;
    out[1] = max_val;
    out[2] = scale;
}

/* psi_phi_array.cpp:247-319 */
void orc_fill_array(const float* psi_imgs, const float* phi_imgs, const orc_meta* meta, void* out) {
    const uint64_t n = meta->num_times * meta->height * meta->width;
    if (meta->num_bytes == 4) {
        float* enc = (float*)out;
        for (uint64_t p = 0; p < n; ++p) {
            enc[2 * p] = psi_imgs[p];
            enc[2 * p + 1] = phi_imgs[p];
        }
        return;
    }
    /* :264-265 -- double expression stored to float */
    const float safe_max_psi = (float)((double)meta->psi_max_val - (double)meta->psi_scale / 100.0);
    const float safe_max_phi = (float)((double)meta->phi_max_val - (double)meta->phi_scale / 100.0);
    for (uint64_t p = 0; p < n; ++p) {
        const float e_psi = orc_encode_uint_scalar(psi_imgs[p], meta->psi_min_val, safe_max_psi, meta->psi_scale);
        const float e_phi = orc_encode_uint_scalar(phi_imgs[p], meta->phi_min_val, safe_max_phi, meta->phi_scale);
        if (meta->num_bytes == 1) { /* :284-285 static_cast<T>(float): truncation */
            ((uint8_t*)out)[2 * p] = (uint8_t)e_psi;
            ((uint8_t*)out)[2 * p + 1] = (uint8_t)e_phi;
        } else {
            ((uint16_t*)out)[2 * p] = (uint16_t)e_psi;
            ((uint16_t*)out)[2 * p + 1] = (uint16_t)e_phi;
        }
    }
}

/* psi_phi_array.cpp:172-205 */
void orc_read_psi_phi(const orc_meta* meta, const void* arr, uint64_t t, int row, int col, float* psi,
                      float* phi) {
    *psi = NAN;
    *phi = NAN;
    if ((arr == NULL) || (row < 0) || (col < 0) || ((uint64_t)row >= meta->height) ||
        ((uint64_t)col >= meta->width)) {
        return;
    }
    const uint64_t ppi = meta->width * meta->height;
    const uint64_t start = 2 * (ppi * t + ((uint64_t)row * meta->width + (uint64_t)col));
    if (meta->num_bytes == 4) {
        *psi = ((const float*)arr)[start];
        *phi = ((const float*)arr)[start + 1];
    } else {
        const float pv = (meta->num_bytes == 1) ? (float)((const uint8_t*)arr)[start]
                                                : (float)((const uint16_t*)arr)[start];
        *psi = (pv == 0.0) ? NAN : (float)(((double)pv - 1.0) * (double)meta->psi_scale + (double)meta->psi_min_val);
        const float fv = (meta->num_bytes == 1) ? (float)((const uint8_t*)arr)[start + 1]
                                                : (float)((const uint16_t*)arr)[start + 1];
        *phi = (fv == 0.0) ? NAN : (float)(((double)fv - 1.0) * (double)meta->phi_scale + (double)meta->phi_min_val);
    }
}

/* ------------------------------------------------------------------ */
/* trajectory evaluation                                                */
/* ------------------------------------------------------------------ */

/* cpu_search_algorithms.cpp:35-36 == kernels.cu:33-35: int + float*double + 0.5f,
 * two separately rounded f64 operations, then +0.5, floor, truncate to int. */
static inline int predict_index(int pos0, float vel0, double time) {
    const double prod = (double)vel0 * time;
    const double s = (double)pos0 + prod;
    return (int)floor(s + (double)0.5f);
}

/* cpu_search_algorithms.cpp:20-50 */
void orc_evaluate_trajectory_cpu(const orc_meta* meta, const void* arr, const double* times,
                                 orc_trajectory* trj) {
    float psi_sum = 0.0f, phi_sum = 0.0f;
    trj->obs_count = 0;
    trj->lh = -1.0f;
    trj->flux = -1.0f;
    int num_seen = 0;
    for (uint64_t i = 0; i < meta->num_times; ++i) {
        const double t = times[i];
        const int cx = predict_index(trj->x, trj->vx, t);
        const int cy = predict_index(trj->y, trj->vy, t);
        float psi, phi;
        orc_read_psi_phi(meta, arr, i, cy, cx, &psi, &phi);
        if (isfinite(psi) && isfinite(phi)) {
            psi_sum += psi;
            phi_sum += phi;
            num_seen += 1;
        }
    }
    trj->obs_count = num_seen;
    trj->lh = (phi_sum > 0) ? (psi_sum / sqrtf(phi_sum)) : -1.0f; /* :48 float sqrt overload */
    trj->flux = (phi_sum > 0) ? (psi_sum / phi_sum) : -1.0f;      /* :49 */
}

/* kernels/kernels.cu:77-147 */
void orc_sigmag_filtered_indices(const float* values, int n, float sgl0, float sgl1, float coeff,
                                 float width, int* idx, int* min_keep, int* max_keep) {
    if (idx == NULL || (min_keep == NULL && max_keep == NULL)) return; /* :84 */
    if (n == 0) {                                                      /* :87-92 */
        *min_keep = 0;
        *max_keep = -1;
        return;
    }
    if (sgl0 < 0.0001) sgl0 = 0.0001; /* :95-96 */
    if (sgl1 > 0.9999) sgl1 = 0.9999;

    for (int j = 0; j < n; j++) idx[j] = j;
    for (int j = 0; j < n; j++) { /* :104-112 exchange sort, ascending */
        for (int k = j + 1; k < n; k++) {
            if (values[idx[j]] > values[idx[k]]) {
                const int tmp = idx[j];
                idx[j] = idx[k];
                idx[k] = tmp;
            }
        }
    }
    /* :117-127 -- int*float -> float, ceil(float) -> float, +0.001 in double, truncate */
    int pct_L = (int)((double)ceilf((float)n * sgl0) + 0.001) - 1;
    pct_L = (pct_L < 0) ? 0 : pct_L;
    pct_L = (pct_L >= n) ? (n - 1) : pct_L;
    int pct_H = (int)((double)ceilf((float)n * sgl1) + 0.001) - 1;
    pct_H = (pct_H < 0) ? 0 : pct_H;
    pct_H = (pct_H >= n) ? (n - 1) : pct_H;
    int median_ind = (int)(ceil((double)n * 0.5) + 0.001) - 1;
    median_ind = (median_ind < 0) ? 0 : median_ind;
    median_ind = (median_ind >= n) ? (n - 1) : median_ind;

    const float sigma_g = coeff * (values[idx[pct_H]] - values[idx[pct_L]]); /* :130 */
    const float min_value = values[idx[median_ind]] - width * sigma_g;
    const float max_value = values[idx[median_ind]] + width * sigma_g;

    int start = 0; /* :135-139 */
    while ((start < median_ind) && (values[idx[start]] < min_value)) ++start;
    *min_keep = start;
    int end = median_ind + 1; /* :142-146 */
    while ((end < n) && (values[idx[end]] <= max_value)) ++end;
    *max_keep = end - 1;
}

/* kernels/kernels.cu:154-242 */
void orc_evaluate_trajectory_kernel(const orc_meta* meta, const void* arr, const double* times,
                                    const orc_params* params, orc_trajectory* trj, int max_images) {
    if (arr == NULL || times == NULL || trj == NULL) return;
    if (meta->num_times >= (uint64_t)max_images) return; /* :160 */

    const int T = (int)meta->num_times;
    float* psi_array = (float*)malloc(sizeof(float) * (size_t)(T + 1));
    float* phi_array = (float*)malloc(sizeof(float) * (size_t)(T + 1));
    float psi_sum = 0.0f, phi_sum = 0.0f;
    trj->obs_count = 0;
    trj->lh = -1.0f;
    trj->flux = -1.0f;

    int num_seen = 0;
    for (int i = 0; i < T; ++i) {
        const double t = times[i];
        /* kernels.cu:33-35 takes float pos0; exact for |x| < 2^24 */
        const int cx = predict_index(trj->x, trj->vx, t);
        const int cy = predict_index(trj->y, trj->vy, t);
        float psi, phi;
        orc_read_psi_phi(meta, arr, (uint64_t)i, cy, cx, &psi, &phi);
        if (isfinite(psi) && isfinite(phi)) {
            psi_sum += psi;
            phi_sum += phi;
            psi_array[num_seen] = psi;
            phi_array[num_seen] = phi;
            num_seen += 1;
        }
    }
    trj->obs_count = num_seen;
    trj->lh = (phi_sum > 0) ? (psi_sum / sqrtf(phi_sum)) : -1.0f;
    trj->flux = (phi_sum > 0) ? (psi_sum / phi_sum) : -1.0f;

    if ((trj->obs_count < params->min_observations) || (trj->obs_count == 0) ||
        (params->do_sigmag_filter && trj->lh < params->min_lh)) { /* :201-203 */
        free(psi_array);
        free(phi_array);
        return;
    }

    if (params->do_sigmag_filter) { /* :213-241 */
        float* lc = (float*)malloc(sizeof(float) * (size_t)num_seen);
        int* idx = (int*)malloc(sizeof(int) * (size_t)num_seen);
        for (int i = 0; i < num_seen; ++i) {
            lc[i] = (phi_array[i] != 0) ? (psi_array[i] / phi_array[i]) : 0;
            idx[i] = i;
        }
        int min_keep = 0, max_keep = num_seen - 1;
        orc_sigmag_filtered_indices(lc, num_seen, params->sgl_L, params->sgl_H, params->sigmag_coeff, 2.0f,
                                    idx, &min_keep, &max_keep);
        if (min_keep < 0) min_keep = 0;
        if (max_keep >= num_seen) max_keep = num_seen - 1;
        float new_psi = 0.0f, new_phi = 0.0f;
        for (int i = min_keep; i <= max_keep; i++) { /* sorted-value order :233-237 */
            new_psi += psi_array[idx[i]];
            new_phi += phi_array[idx[i]];
        }
        trj->lh = (new_phi > 0) ? (new_psi / sqrtf(new_phi)) : -1.0f;
        trj->flux = (new_phi > 0) ? (new_psi / new_phi) : -1.0f;
        free(lc);
        free(idx);
    }
    free(psi_array);
    free(phi_array);
}

/* ------------------------------------------------------------------ */
/* searches                                                             */
/* ------------------------------------------------------------------ */

/* cpu_search_algorithms.cpp:57-124.  evaluate_single_pixel evaluates every
 * candidate, sorts descending by lh and keeps the first R.  The reference sort
 * (trajectory_list.cpp:96-107) is unstable, so its order among equal lh is
 * unspecified; here ties keep candidate order (== a stable sort). */
void orc_search_cpu(const orc_meta* meta, const void* arr, const double* times, const orc_params* params,
                    const orc_trajectory* cands, uint64_t n_cand, orc_trajectory* results) {
    const int64_t sh = (int64_t)params->y_start_max - params->y_start_min;
    const int64_t sw = (int64_t)params->x_start_max - params->x_start_min;
    const uint64_t R = (n_cand < params->results_per_pixel) ? n_cand : params->results_per_pixel; /* :99 */
    if (sh <= 0 || sw <= 0 || R == 0) return;
    memset(results, 0, sizeof(orc_trajectory) * R * (uint64_t)sh * (uint64_t)sw); /* :101-102 */

#pragma omp parallel
    {
        orc_trajectory* all = (orc_trajectory*)malloc(sizeof(orc_trajectory) * n_cand);
        unsigned char* taken = (unsigned char*)malloc(n_cand);
#pragma omp for collapse(2) schedule(dynamic, 16)
        for (int64_t y_i = 0; y_i < sh; ++y_i) {
            for (int64_t x_i = 0; x_i < sw; ++x_i) {
                for (uint64_t c = 0; c < n_cand; ++c) { /* :69-81 */
                    all[c].x = (int32_t)(x_i + params->x_start_min);
                    all[c].y = (int32_t)(y_i + params->y_start_min);
                    all[c].vx = cands[c].vx;
                    all[c].vy = cands[c].vy;
                    all[c].flux = 0.0f;
                    all[c].obs_count = 0;
                    orc_evaluate_trajectory_cpu(meta, arr, times, &all[c]);
                }
                /* :84-85 sort descending + first R, as R stable selections */
                memset(taken, 0, n_cand);
                orc_trajectory* dst = results + ((uint64_t)y_i * (uint64_t)sw + (uint64_t)x_i) * R; /* :117 */
                for (uint64_t r = 0; r < R; ++r) {
                    int64_t best = -1;
                    for (uint64_t c = 0; c < n_cand; ++c) {
                        if (taken[c]) continue;
                        if (best < 0 || all[best].lh < all[c].lh) best = (int64_t)c; /* comparator b.lh < a.lh */
                    }
                    taken[best] = 1;
                    dst[r] = all[best];
                }
            }
        }
        free(all);
        free(taken);
    }
}

/* kernels/kernels.cu:252-332, one start pixel per loop iteration. */
void orc_search_kernel_semantics(const orc_meta* meta, const void* arr, const double* times,
                                 const orc_params* params, const orc_trajectory* cands, uint64_t n_cand,
                                 orc_trajectory* results, int max_images) {
    const int64_t sw = (int64_t)params->x_start_max - params->x_start_min; /* :274-275 */
    const int64_t sh = (int64_t)params->y_start_max - params->y_start_min;
    const uint32_t K = params->results_per_pixel;
    if (sw <= 0 || sh <= 0) return;

#pragma omp parallel for collapse(2) schedule(dynamic, 16)
    for (int64_t y_i = 0; y_i < sh; ++y_i) {
        for (int64_t x_i = 0; x_i < sw; ++x_i) {
            const int x = (int)(x_i + params->x_start_min); /* :281-282 */
            const int y = (int)(y_i + params->y_start_min);
            orc_trajectory* slots = results + ((uint64_t)y_i * (uint64_t)sw + (uint64_t)x_i) * K; /* :286 */
            for (uint32_t r = 0; r < K; ++r) { /* :293-301 */
                slots[r].x = x;
                slots[r].y = y;
                slots[r].vx = 0.0f;
                slots[r].vy = 0.0f;
                slots[r].lh = -FLT_MAX;
                slots[r].flux = 0.0f;
                slots[r].obs_count = 0;
            }
            for (uint64_t t = 0; t < n_cand; ++t) { /* :304-331 */
                orc_trajectory cur;
                cur.x = x;
                cur.y = y;
                cur.vx = cands[t].vx;
                cur.vy = cands[t].vy;
                cur.obs_count = 0;
                cur.lh = 0.0f;
                cur.flux = 0.0f;
                orc_evaluate_trajectory_kernel(meta, arr, times, params, &cur, max_images);
                if ((cur.obs_count < params->min_observations) ||
                    (params->do_sigmag_filter && cur.lh < params->min_lh))
                    continue; /* :318-320 */
                for (uint32_t r = 0; r < K; ++r) { /* :323-330 strict >, swap down */
                    if (cur.lh > slots[r].lh) {
                        const orc_trajectory tmp = slots[r];
                        slots[r] = cur;
                        cur = tmp;
                    }
                }
            }
        }
    }
}

/* stack_search.cpp:266-281 with trajectory_list.cpp:96-126. */
static void merge_sort_desc(orc_trajectory* a, orc_trajectory* tmp, uint64_t n) {
    if (n < 2) return;
    const uint64_t h = n / 2;
    merge_sort_desc(a, tmp, h);
    merge_sort_desc(a + h, tmp, n - h);
    uint64_t i = 0, j = h, k = 0;
    while (i < h && j < n) {
        if (a[i].lh < a[j].lh) tmp[k++] = a[j++]; /* right strictly greater goes first */
        else tmp[k++] = a[i++];
    }
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, sizeof(orc_trajectory) * n);
}

uint64_t orc_filter_sort(orc_trajectory* results, uint64_t n, float min_lh, int min_obs) {
    uint64_t m = 0;
    for (uint64_t i = 0; i < n; ++i) /* filter_by_likelihood: remove a.lh < min_lh */
        if (!(results[i].lh < min_lh)) results[m++] = results[i];
    n = m;
    m = 0;
    for (uint64_t i = 0; i < n; ++i) /* filter_by_obs_count: remove obs < min_obs */
        if (!(results[i].obs_count < min_obs)) results[m++] = results[i];
    n = m;
    if (n > 1) {
        orc_trajectory* tmp = (orc_trajectory*)malloc(sizeof(orc_trajectory) * n);
        merge_sort_desc(results, tmp, n);
        free(tmp);
    }
    return n;
}

/* stack_search.cpp:22-39 with common.h:71-79: the index is floor() of a FLOAT
 * position (get_x_pos returns float), unlike the search's double formula. */
void orc_psi_phi_curve(const orc_meta* meta, const void* arr, const double* times,
                       const orc_trajectory* trj, float* out) {
    const uint64_t T = meta->num_times;
    for (uint64_t i = 0; i < 2 * T; ++i) out[i] = 0.0f;
    for (uint64_t i = 0; i < T; ++i) {
        const double t = times[i];
        const float xpos = (float)((double)trj->x + t * (double)trj->vx + (double)0.5f);
        const float ypos = (float)((double)trj->y + t * (double)trj->vy + (double)0.5f);
        const int xi = (int)floorf(xpos);
        const int yi = (int)floorf(ypos);
        float psi, phi;
        orc_read_psi_phi(meta, arr, i, yi, xi, &psi, &phi);
        if (value_valid(psi)) out[i] = psi;
        if (value_valid(phi)) out[i + T] = phi;
    }
}
