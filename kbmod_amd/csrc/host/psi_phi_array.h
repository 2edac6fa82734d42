// PsiPhiArray: the interleaved, optionally uint8/uint16-encoded psi/phi store.
//
// Mirrors psi_phi_array_ds.h:34-141 and psi_phi_array.cpp:15-410 of the
// reference (names, validation, error messages, encode/decode arithmetic) with
// one MI355X-first change in ownership: the array may be BORN in HBM (built by
// the fused device kernel, image_kernels.hip) and the host copy is materialised
// lazily, only when a host function (read_psi_phi, the CPU search, the
// psi/phi-curve gather) needs it.  The Python-visible `on_gpu` /
// `gpu_array_allocated` flags keep the reference's logical meaning (true only
// between move_to_gpu()/preload and clear_from_gpu()/unload).
#ifndef KBH_PSI_PHI_ARRAY_H_
#define KBH_PSI_PHI_ARRAY_H_

#include <array>
#include <cfloat>
#include <cstdlib>
#include <cstring>

#include "../search_math.h"
#include "common.h"
#include "image_utils.h"

namespace search {

struct PsiPhi {  // psi_phi_array_ds.h:34-37
    float psi = 0.0;
    float phi = 0.0;
};

// psi_phi_array_ds.h:40-47
inline float encode_uint_scalar(float value, float min_val, float max_val, float scale) {
    return !pixel_value_valid(value) ? 0
                                     : (std::max(std::min(value, max_val), min_val) - min_val) / scale + 1.0;
}
inline float decode_uint_scalar(float value, float min_val, float scale) {
    return (value == 0.0) ? NO_DATA : (value - 1.0) * scale + min_val;
}

using PsiPhiArrayMeta = kb_psi_phi_meta;

class PsiPhiArray {
public:
    explicit PsiPhiArray() { reset_meta(true); }
    virtual ~PsiPhiArray() { clear(); }
    PsiPhiArray(const PsiPhiArray&) = delete;
    PsiPhiArray& operator=(const PsiPhiArray&) = delete;

    // psi_phi_array.cpp:23-47
    void clear() {
        if (cpu_array_ptr != nullptr) {
            free(cpu_array_ptr);
            cpu_array_ptr = nullptr;
        }
        cpu_time_array.clear();
        release_device();
        data_on_gpu = false;
        reset_meta(false);
    }

    inline PsiPhiArrayMeta& get_meta_data() { return meta_data; }
    inline bool on_gpu() const { return data_on_gpu; }
    inline int get_num_bytes() const { return meta_data.num_bytes; }
    inline uint64_t get_num_times() const { return meta_data.num_times; }
    inline uint64_t get_width() const { return meta_data.width; }
    inline uint64_t get_height() const { return meta_data.height; }
    inline uint64_t get_pixels_per_image() const { return meta_data.pixels_per_image; }
    inline uint64_t get_num_entries() const { return meta_data.num_entries; }
    inline uint64_t get_total_array_size() const { return meta_data.total_array_size; }
    inline uint64_t get_block_size() const { return meta_data.block_size; }
    inline float get_psi_min_val() const { return meta_data.psi_min_val; }
    inline float get_psi_max_val() const { return meta_data.psi_max_val; }
    inline float get_psi_scale() const { return meta_data.psi_scale; }
    inline float get_phi_min_val() const { return meta_data.phi_min_val; }
    inline float get_phi_max_val() const { return meta_data.phi_max_val; }
    inline float get_phi_scale() const { return meta_data.phi_scale; }
    // True once the data exists on the host OR can be fetched from the device copy.
    inline bool cpu_array_allocated() const { return cpu_array_ptr != nullptr || gpu_array_ptr != nullptr; }
    inline bool gpu_array_allocated() const { return data_on_gpu && gpu_array_ptr != nullptr; }
    inline bool device_resident() const { return gpu_array_ptr != nullptr; }

    // psi_phi_array.cpp:172-205
    PsiPhi read_psi_phi(uint64_t time, int row, int col) {
        PsiPhi sample = {NO_DATA, NO_DATA};
        if (cpu_array_allocated()) {  // bounds, decode and the NO_DATA rule live in search_math.h, shared with the kernels
            kb::read_psi_phi(meta_data, host_ptr(), time, row, col, &sample.psi, &sample.phi);
        }
        return sample;
    }

    // psi_phi_array.cpp:207-212
    double read_time(uint64_t time_index) const {
        if (time_index >= meta_data.num_times) {
            throw std::runtime_error("Out of bounds read for time step. [" + std::to_string(time_index) + "]");
        }
        return cpu_time_array[time_index];
    }

    // psi_phi_array.cpp:113-148
    void set_meta_data(int new_num_bytes, uint64_t new_num_times, uint64_t new_height, uint64_t new_width) {
        const bool known_size = new_num_bytes == -1 || new_num_bytes == 1 || new_num_bytes == 2 || new_num_bytes == 4;
        if (!known_size) {
            throw std::runtime_error("Invalid setting of num_bytes. Must be (-1 [use default], 1, 2, or 4). Got " +
                                     std::to_string(new_num_bytes));
        }
        const std::pair<const char*, uint64_t> extents[] = {
                {"num_times", new_num_times}, {"width", new_width}, {"height", new_height}};
        for (const auto& e : extents) {
            if (e.second == 0) {
                throw std::runtime_error(std::string("Invalid ") + e.first + " passed to set_meta_data: 0");
            }
        }
        if (cpu_array_ptr != nullptr || gpu_array_ptr != nullptr) {
            throw std::runtime_error("Cannot change meta data with allocated arrays. Call clear() first.");
        }
        const int bytes = (new_num_bytes == 1 || new_num_bytes == 2) ? new_num_bytes : 4;  // -1 and 4: float32
        meta_data.num_bytes = bytes;
        meta_data.block_size = (uint64_t)bytes;
        meta_data.num_times = new_num_times;
        meta_data.width = new_width;
        meta_data.height = new_height;
        meta_data.pixels_per_image = new_width * new_height;
        meta_data.num_entries = 2 * meta_data.pixels_per_image * new_num_times;
        meta_data.total_array_size = meta_data.block_size * meta_data.num_entries;
    }

    // psi_phi_array.cpp:150-168
    void set_psi_scaling(float min_val, float max_val, float scale_val) {
        check_scaling(min_val, max_val, scale_val);
        meta_data.psi_min_val = min_val;
        meta_data.psi_max_val = max_val;
        meta_data.psi_scale = scale_val;
    }
    void set_phi_scaling(float min_val, float max_val, float scale_val) {
        check_scaling(min_val, max_val, scale_val);
        meta_data.phi_min_val = min_val;
        meta_data.phi_max_val = max_val;
        meta_data.phi_scale = scale_val;
    }
    void set_time_array(const std::vector<double>& times) {
        cpu_time_array = times;
        if (gpu_time_ptr != nullptr) {  // keep a resident copy coherent
            (void)kb_free_gpu_block(gpu_time_ptr);
            gpu_time_ptr = nullptr;
        }
    }

    // psi_phi_array.cpp:74-111
    void move_to_gpu() {
        if (data_on_gpu) {
            if ((gpu_array_ptr == nullptr) || (gpu_time_ptr == nullptr)) {
                throw std::runtime_error("Inconsistent GPU flags and pointers");
            }
            return;
        }
        assert_sizes_equal(cpu_time_array.size(), meta_data.num_times, "psi-phi number of times");
        if (!has_gpu()) return;  // the reference silently does nothing without a device (:108-110)
        ensure_device();
        data_on_gpu = true;
    }

    // psi_phi_array.cpp:49-72: an explicit request frees the HBM copy (the data
    // survives on the host).
    void clear_from_gpu() {
        if (gpu_array_ptr != nullptr && meta_data.total_array_size > 0) ensure_host();
        release_device();
        data_on_gpu = false;
    }

    // End of a non-preloaded search: drop the logical flag but keep the HBM copy
    // cached for the next search (288 GB of HBM; no PCIe round trip per search).
    void end_device_use() { data_on_gpu = false; }

    // Make sure the encoded array and the times are resident in HBM.
    void ensure_device() {
        if (!has_gpu()) throw std::runtime_error("No GPU onto which to move the PsiPhi array.");
        if (gpu_array_ptr == nullptr) {
            if (cpu_array_ptr == nullptr) throw std::runtime_error("CPU data not allocated.");
            logging::getLogger("kbmod.search.psi_phi_array")
                    ->debug("Allocating PsiPhiArray on GPU: " +
                            std::to_string(get_total_array_size() / (1024 * 1024)) + " MB");
            check_status(kb_allocate_gpu_block(get_total_array_size(), &gpu_array_ptr));
            check_status(kb_copy_block_to_gpu(cpu_array_ptr, gpu_array_ptr, get_total_array_size()));
        }
        if (gpu_time_ptr == nullptr) {
            assert_sizes_equal(cpu_time_array.size(), meta_data.num_times, "psi-phi number of times");
            void* p = nullptr;
            check_status(kb_allocate_gpu_block(cpu_time_array.size() * sizeof(double), &p));
            gpu_time_ptr = reinterpret_cast<double*>(p);
            check_status(kb_copy_block_to_gpu(cpu_time_array.data(), gpu_time_ptr,
                                              cpu_time_array.size() * sizeof(double)));
        }
    }

    // Make sure a host copy of the encoded array exists (D2H on first use).
    void ensure_host() {
        if (cpu_array_ptr != nullptr) return;
        if (gpu_array_ptr == nullptr) throw std::runtime_error("PsiPhi data has not been created.");
        void* p = malloc(get_total_array_size());
        if (p == nullptr) throw std::runtime_error("Unable to allocate space for CPU PsiPhi array.");
        try {
            check_status(kb_copy_block_to_cpu(p, gpu_array_ptr, get_total_array_size()));
        } catch (...) {
            free(p);
            throw;
        }
        cpu_array_ptr = p;
    }
    const void* host_ptr() {
        ensure_host();
        return cpu_array_ptr;
    }

    // Should ONLY be called by the utility functions.
    inline void* get_cpu_array_ptr() { return cpu_array_ptr; }
    inline void* get_gpu_array_ptr() { return gpu_array_ptr; }
    inline void set_cpu_array_ptr(void* new_ptr) { cpu_array_ptr = new_ptr; }
    inline double* get_cpu_time_array_ptr() { return cpu_time_array.data(); }
    inline double* get_gpu_time_array_ptr() { return gpu_time_ptr; }
    // Adopt an array built in HBM by the device builder.
    void adopt_device_array(const kb_psi_phi_meta& meta, void* dev_ptr) {
        if (cpu_array_ptr != nullptr || gpu_array_ptr != nullptr) {
            throw std::runtime_error("PsiPhi array already allocated.");
        }
        meta_data = meta;
        gpu_array_ptr = dev_ptr;
    }

private:
    void reset_meta(bool with_encoding) {
        meta_data.num_times = 0;
        meta_data.width = 0;
        meta_data.height = 0;
        meta_data.pixels_per_image = 0;
        meta_data.num_entries = 0;
        meta_data.total_array_size = 0;
        if (with_encoding) {
            meta_data.block_size = 0;
            meta_data.num_bytes = 4;
        }
        meta_data.psi_min_val = FLT_MAX;
        meta_data.psi_max_val = -FLT_MAX;
        meta_data.psi_scale = 1.0;
        meta_data.phi_min_val = FLT_MAX;
        meta_data.phi_max_val = -FLT_MAX;
        meta_data.phi_scale = 1.0;
    }
    static void check_scaling(float min_val, float max_val, float scale_val) {
        if (min_val > max_val)
            throw std::runtime_error("Min value needs to be < max value. Got " + std::to_string(min_val) + " and " +
                                     std::to_string(max_val));
        if (scale_val <= 0)
            throw std::runtime_error("Scale value must be greater than zero. Got " + std::to_string(scale_val));
    }
    void release_device() {
        if (gpu_array_ptr != nullptr) {
            (void)kb_free_gpu_block(gpu_array_ptr);
            gpu_array_ptr = nullptr;
        }
        if (gpu_time_ptr != nullptr) {
            (void)kb_free_gpu_block(gpu_time_ptr);
            gpu_time_ptr = nullptr;
        }
    }

    PsiPhiArrayMeta meta_data;
    bool data_on_gpu = false;
    void* cpu_array_ptr = nullptr;
    void* gpu_array_ptr = nullptr;
    std::vector<double> cpu_time_array;
    double* gpu_time_ptr = nullptr;
};

// ---- utility functions: psi_phi_array.cpp:219-410 -----------------------------

// psi_phi_array.cpp:219-245: finite range of a set of images and the code width that spans it.
inline std::array<float, 3> compute_scale_params_from_image_vect(const std::vector<Image>& imgs, int num_bytes) {
    float lo = FLT_MAX, hi = -FLT_MAX;
    for (const Image& im : imgs) {
        for (const float v : im.data) {
            if (!pixel_value_valid(v)) continue;
            lo = std::min(lo, v);
            hi = std::max(hi, v);
        }
    }
    float scale = 1.0;
    if (num_bytes == 1 || num_bytes == 2) {
        const float span = std::max(hi - lo, 1e-6f);
        const uint64_t codes = (1ull << (8 * num_bytes)) - 1;
        scale = span / (double)codes;
    }
    return {lo, hi, scale};
}

// The host build of the array (psi_phi_array.cpp:247-319): psi and phi of every pixel side by side, epoch after
// epoch, each value passed through `code` (the identity for float arrays, encode_uint_scalar + truncation for
// encoded ones).
template <typename T, typename PsiCode, typename PhiCode>
void interleave_on_host(PsiPhiArray& data, const std::vector<Image>& psi_imgs, const std::vector<Image>& phi_imgs,
                        PsiCode psi_code, PhiCode phi_code) {
    if (data.get_cpu_array_ptr() != nullptr) throw std::runtime_error("CPU PsiPhi already allocated.");
    T* out = static_cast<T*>(malloc(data.get_total_array_size()));
    if (out == nullptr) throw std::runtime_error("Unable to allocate space for CPU PsiPhi array.");
    T* cursor = out;
    for (uint64_t t = 0; t < data.get_num_times(); ++t) {
        const std::vector<float>& psi = psi_imgs[t].data;
        const std::vector<float>& phi = phi_imgs[t].data;
        for (size_t p = 0; p < psi.size(); ++p) {
            *cursor++ = static_cast<T>(psi_code(psi[p]));
            *cursor++ = static_cast<T>(phi_code(phi[p]));
        }
    }
    data.set_cpu_array_ptr(out);
}

template <typename T>
void set_encode_cpu_psi_phi_array(PsiPhiArray& data, const std::vector<Image>& psi_imgs,
                                  const std::vector<Image>& phi_imgs) {
    // psi_phi_array.cpp:264-265: the largest value stays one hundredth of a code below the top
    const float psi_lo = data.get_psi_min_val(), psi_w = data.get_psi_scale(), psi_top = data.get_psi_max_val() - psi_w / 100.0;
    const float phi_lo = data.get_phi_min_val(), phi_w = data.get_phi_scale(), phi_top = data.get_phi_max_val() - phi_w / 100.0;
    interleave_on_host<T>(
            data, psi_imgs, phi_imgs, [=](float v) { return encode_uint_scalar(v, psi_lo, psi_top, psi_w); },
            [=](float v) { return encode_uint_scalar(v, phi_lo, phi_top, phi_w); });
}

inline void set_float_cpu_psi_phi_array(PsiPhiArray& data, const std::vector<Image>& psi_imgs,
                                        const std::vector<Image>& phi_imgs) {
    interleave_on_host<float>(data, psi_imgs, phi_imgs, [](float v) { return v; }, [](float v) { return v; });
}

// psi_phi_array.cpp:321-372
inline void fill_psi_phi_array(PsiPhiArray& result_data, int num_bytes, const std::vector<Image>& psi_imgs,
                               const std::vector<Image>& phi_imgs, const std::vector<double> zeroed_times) {
    if (result_data.get_cpu_array_ptr() != nullptr || result_data.device_resident()) return;
    const uint64_t num_times = psi_imgs.size();
    if (num_times == 0) throw std::runtime_error("Trying to fill PsiPhi from empty vectors.");
    assert_sizes_equal(phi_imgs.size(), num_times, "psi and phi arrays");
    const int64_t rows = phi_imgs[0].rows, cols = phi_imgs[0].cols;
    for (uint64_t t = 0; t < num_times; ++t) {
        const bool same = psi_imgs[t].rows == rows && psi_imgs[t].cols == cols && phi_imgs[t].rows == rows &&
                          phi_imgs[t].cols == cols;
        if (!same) throw std::runtime_error("All psi and phi images must have the same dimensions.");
    }
    result_data.set_meta_data(num_bytes, num_times, (uint64_t)rows, (uint64_t)cols);

    const int bytes = result_data.get_num_bytes();
    if (bytes == 4) {
        set_float_cpu_psi_phi_array(result_data, psi_imgs, phi_imgs);
    } else {
        const std::array<float, 3> psi_range = compute_scale_params_from_image_vect(psi_imgs, bytes);
        const std::array<float, 3> phi_range = compute_scale_params_from_image_vect(phi_imgs, bytes);
        result_data.set_psi_scaling(psi_range[0], psi_range[1], psi_range[2]);
        result_data.set_phi_scaling(phi_range[0], phi_range[1], phi_range[2]);
        logging::Logger* lg = logging::getLogger("kbmod.search.psi_phi_array");
        const char* names[2] = {"psi", "phi"};
        const std::array<float, 3>* ranges[2] = {&psi_range, &phi_range};
        for (int i = 0; i < 2; ++i) {
            lg->info(std::string("Encoding ") + names[i] + " to " + std::to_string(bytes) + ": min=" +
                     std::to_string((*ranges[i])[0]) + ", max=" + std::to_string((*ranges[i])[1]) +
                     ", scale=" + std::to_string((*ranges[i])[2]));
        }
        if (bytes == 1) {
            set_encode_cpu_psi_phi_array<uint8_t>(result_data, psi_imgs, phi_imgs);
        } else {
            set_encode_cpu_psi_phi_array<uint16_t>(result_data, psi_imgs, phi_imgs);
        }
    }
    result_data.set_time_array(zeroed_times);
}

// psi_phi_array.cpp:374-410.  With a GPU the whole build (pixel preparation,
// both correlations, range scan, encoding) runs in HBM and stays there; without
// one the host loop of the reference is used.  force_cpu selects the host loop.
inline void fill_psi_phi_array_from_image_arrays(PsiPhiArray& result_data, int num_bytes,
                                                 std::vector<Image>& sci_imgs, std::vector<Image>& var_imgs,
                                                 std::vector<Image>& psf_kernels,
                                                 std::vector<double>& zeroed_times, bool force_cpu = false) {
    const uint64_t num_images = sci_imgs.size();
    if (num_images == 0) throw std::runtime_error("Trying to fill PsiPhi from empty vectors.");
    if (num_images != var_imgs.size()) {
        throw std::runtime_error("Number of images in sci and var do not match. Sci=" +
                                 std::to_string(num_images) + ", Var=" + std::to_string(var_imgs.size()));
    }
    if (num_images != psf_kernels.size()) {
        throw std::runtime_error("Number of images in sci and PSF kernels do not match. Sci=" +
                                 std::to_string(num_images) + ", PSF=" + std::to_string(psf_kernels.size()));
    }
    const int64_t height = sci_imgs[0].rows;
    const int64_t width = sci_imgs[0].cols;
    for (uint64_t i = 0; i < num_images; ++i) {
        check_same_dims(sci_imgs[i], var_imgs[i]);
        if (sci_imgs[i].rows != height || sci_imgs[i].cols != width) {
            throw std::runtime_error("All images in the stack must have the same dimensions.");
        }
    }
    logging::getLogger("kbmod.search.psi_phi_array")
            ->info("Building " + std::to_string(num_images * 2) + " temporary " + std::to_string(height) + " by " +
                   std::to_string(width) + " images, requiring " +
                   std::to_string(2 * height * width * num_images * sizeof(float)) + " bytes.");

    if (has_gpu() && !force_cpu) {
        if (result_data.get_cpu_array_ptr() != nullptr || result_data.device_resident()) return;
        if (num_bytes != -1 && num_bytes != 1 && num_bytes != 2 && num_bytes != 4) {
            throw std::runtime_error("Invalid setting of num_bytes. Must be (-1 [use default], 1, 2, or 4). Got " +
                                     std::to_string(num_bytes));
        }
        std::vector<const float*> sci_ptrs(num_images), var_ptrs(num_images);
        std::vector<int32_t> dims(num_images);
        std::vector<float> psf_packed;
        for (uint64_t i = 0; i < num_images; ++i) {
            sci_ptrs[i] = sci_imgs[i].data.data();
            var_ptrs[i] = var_imgs[i].data.data();
            if (psf_kernels[i].rows != psf_kernels[i].cols) throw std::runtime_error("PSF kernel must be square.");
            dims[i] = (int32_t)psf_kernels[i].rows;
            psf_packed.insert(psf_packed.end(), psf_kernels[i].data.begin(), psf_kernels[i].data.end());
        }
        kb_psi_phi_meta meta;
        void* dev = nullptr;
        check_status(kb_build_psi_phi_from_host(sci_ptrs.data(), var_ptrs.data(), psf_packed.data(), dims.data(),
                                                (int32_t)num_images, (int32_t)height, (int32_t)width, num_bytes,
                                                &meta, &dev));
        result_data.adopt_device_array(meta, dev);
        result_data.set_time_array(zeroed_times);
        return;
    }

    std::vector<Image> psi_images;
    std::vector<Image> phi_images;
    for (uint64_t i = 0; i < num_images; ++i) {
        psi_images.push_back(generate_psi_cpu(sci_imgs[i], var_imgs[i], psf_kernels[i]));
        phi_images.push_back(generate_phi_cpu(var_imgs[i], psf_kernels[i]));
    }
    fill_psi_phi_array(result_data, num_bytes, psi_images, phi_images, zeroed_times);
}

}  // namespace search
#endif
