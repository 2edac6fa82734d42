// Science / variance image stack resident in HBM for the post-search stages that read pixels at
// trajectory positions (stamp coadds: src/kbmod/filters/stamp_filters.py:72-168 over
// ImageStackPy.sci / .var).  Not part of the reference's compiled module: the reference does this
// stage on the host, one trajectory at a time.
#ifndef KB_HOST_DEVICE_STACK_H_
#define KB_HOST_DEVICE_STACK_H_

#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "kbmod_hip.h"

namespace search {
namespace py = pybind11;

// A block of HBM owned by a scope or an object (freed when it goes away, also on the error paths).
class HbmBlock {
public:
    HbmBlock() = default;
    HbmBlock(const void* host, uint64_t bytes) {
        check(kb_allocate_gpu_block(bytes == 0 ? 4 : bytes, &ptr_));
        if (host != nullptr && bytes != 0 && kb_copy_block_to_gpu(host, ptr_, bytes) != 0) {
            const std::string what = kb_last_error();
            reset();
            throw std::runtime_error(what);
        }
    }
    HbmBlock(const HbmBlock&) = delete;
    HbmBlock& operator=(const HbmBlock&) = delete;
    HbmBlock(HbmBlock&& o) noexcept : ptr_(o.ptr_), owned_(o.owned_) { o.ptr_ = nullptr; }
    HbmBlock& operator=(HbmBlock&& o) noexcept {
        if (this != &o) {
            reset();
            ptr_ = o.ptr_;
            owned_ = o.owned_;
            o.ptr_ = nullptr;
        }
        return *this;
    }
    ~HbmBlock() { reset(); }
    // A block somebody else owns (a torch tensor, ...): used, never freed here.
    static HbmBlock borrowed(void* dev) {
        HbmBlock b;
        b.ptr_ = dev;
        b.owned_ = false;
        return b;
    }
    void reset() {
        if (ptr_ != nullptr && owned_) (void)kb_free_gpu_block(ptr_);
        ptr_ = nullptr;
        owned_ = true;
    }
    template <typename T>
    T* as() const {
        return reinterpret_cast<T*>(ptr_);
    }
    explicit operator bool() const { return ptr_ != nullptr; }
    static void check(int rc) {
        if (rc != 0) throw std::runtime_error(kb_last_error());
    }

private:
    void* ptr_ = nullptr;
    bool owned_ = true;
};

class DeviceImageStack {
public:
    typedef py::array_t<float, py::array::c_style | py::array::forcecast> FloatArray;
    typedef py::array_t<int32_t, py::array::c_style | py::array::forcecast> IntArray;

    // sci / var: T x H x W arrays (or lists of T H x W arrays); var may be None
    DeviceImageStack(FloatArray sci, py::object var) {
        if (kb_device_count() == 0) throw std::runtime_error("GPU is not available for the image stack.");
        if (sci.ndim() != 3) throw std::runtime_error("expected a T x H x W science stack");
        T_ = (int)sci.shape(0);
        H_ = (int)sci.shape(1);
        W_ = (int)sci.shape(2);
        sci_ = HbmBlock(sci.data(), (uint64_t)sci.size() * sizeof(float));
        if (!var.is_none()) {
            FloatArray v = var.cast<FloatArray>();
            if (v.ndim() != 3 || v.shape(0) != sci.shape(0) || v.shape(1) != sci.shape(1) || v.shape(2) != sci.shape(2)) {
                throw std::runtime_error("science and variance stacks differ in shape");
            }
            var_ = HbmBlock(v.data(), (uint64_t)v.size() * sizeof(float));  // sci_ is released if this throws
        }
    }

    // Stacks that already are in the memory of the current device (what kbmod_amd.fits_ingest decodes a WorkUnit file
    // into): used where they lie; `owner` (the tensors, ...) is held for as long as this object lives.
    DeviceImageStack(uintptr_t sci_dev, uintptr_t var_dev, int T, int H, int W, py::object owner) : owner_(std::move(owner)) {
        if (kb_device_count() == 0) throw std::runtime_error("GPU is not available for the image stack.");
        if (sci_dev == 0 || T < 0 || H <= 0 || W <= 0) throw std::runtime_error("expected a T x H x W science stack on the device");
        T_ = T;
        H_ = H;
        W_ = W;
        sci_ = HbmBlock::borrowed(reinterpret_cast<void*>(sci_dev));
        if (var_dev != 0) var_ = HbmBlock::borrowed(reinterpret_cast<void*>(var_dev));
    }

    int num_times() const { return T_; }
    int height() const { return H_; }
    int width() const { return W_; }
    bool has_variance() const { return (bool)var_; }

    // xvals / yvals: N x T integer stamp centres; returns N x T x (2r+1) x (2r+1) float32 (append_all_stamps)
    py::array_t<float> all_stamps(IntArray xvals, IntArray yvals, int radius) {
        if (radius < 1) throw std::invalid_argument("Invalid stamp radius: " + std::to_string(radius));
        if (xvals.ndim() != 2 || yvals.ndim() != 2 || xvals.shape(0) != yvals.shape(0) || xvals.shape(1) != T_ ||
            yvals.shape(1) != T_) {
            throw std::invalid_argument("X and Y values must have the same length as the number of times.");
        }
        const uint64_t n = (uint64_t)xvals.shape(0);
        const int S = 2 * radius + 1;
        py::array_t<float> out({(py::ssize_t)n, (py::ssize_t)T_, (py::ssize_t)S, (py::ssize_t)S});
        if (n == 0 || T_ == 0) return out;
        const uint64_t nt = n * (uint64_t)T_, out_bytes = nt * (uint64_t)S * S * sizeof(float);
        HbmBlock x_dev(xvals.data(), nt * sizeof(int32_t)), y_dev(yvals.data(), nt * sizeof(int32_t)), out_dev(nullptr, out_bytes);
        HbmBlock::check(kb_extract_stamps(sci_.as<const float>(), T_, H_, W_, x_dev.as<const int32_t>(), y_dev.as<const int32_t>(),
                                          n, radius, out_dev.as<float>(), nullptr));
        HbmBlock::check(kb_copy_block_to_cpu(out.mutable_data(), out_dev.as<void>(), out_bytes));
        return out;
    }

    // xvals / yvals: N x T integer stamp centres; to_include: N x T bool (None = every epoch);
    // returns {type: N x (2r+1) x (2r+1) float32}
    std::map<std::string, py::array_t<float>> coadds(IntArray xvals, IntArray yvals, py::object to_include, int radius,
                                                     const std::vector<std::string>& coadd_types) {
        if (radius <= 0) throw std::invalid_argument("Invalid stamp radius " + std::to_string(radius));
        if (xvals.ndim() != 2 || yvals.ndim() != 2 || xvals.shape(0) != yvals.shape(0) || xvals.shape(1) != T_ ||
            yvals.shape(1) != T_) {
            throw std::invalid_argument("X and Y values must have the same length as the number of times.");
        }
        const uint64_t n = (uint64_t)xvals.shape(0);
        const uint64_t nt = n * (uint64_t)T_;
        const int S = 2 * radius + 1;
        std::map<std::string, py::array_t<float>> out;
        std::vector<int> types;
        for (const std::string& c : coadd_types) {
            if (c == "sum") types.push_back(KB_COADD_SUM);
            else if (c == "mean") types.push_back(KB_COADD_MEAN);
            else if (c == "median") types.push_back(KB_COADD_MEDIAN);
            else if (c == "weighted") types.push_back(KB_COADD_WEIGHTED);
            else throw std::invalid_argument("Unknown coadd type " + c);
            if (c == "weighted" && !var_) throw std::runtime_error("the weighted coadd needs the variance stack");
        }
        for (const std::string& c : coadd_types) {
            out[c] = py::array_t<float>({(py::ssize_t)n, (py::ssize_t)S, (py::ssize_t)S});
        }
        if (n == 0 || types.empty()) return out;
        py::array_t<uint8_t, py::array::c_style | py::array::forcecast> inc;
        if (!to_include.is_none()) {
            inc = to_include.cast<py::array_t<bool>>().cast<py::array_t<uint8_t, py::array::c_style | py::array::forcecast>>();
            if (inc.ndim() != 2 || (uint64_t)inc.shape(0) != n || inc.shape(1) != T_) {
                throw std::invalid_argument("Time mask must have the same length as the number of times.");
            }
        }
        const uint64_t out_bytes = n * (uint64_t)S * S * sizeof(float);
        HbmBlock x_dev(xvals.data(), nt * sizeof(int32_t)), y_dev(yvals.data(), nt * sizeof(int32_t)), out_dev(nullptr, out_bytes);
        HbmBlock inc_dev;
        if (!to_include.is_none()) inc_dev = HbmBlock(inc.data(), nt);
        for (size_t k = 0; k < types.size(); ++k) {
            HbmBlock::check(kb_coadd_stamps(sci_.as<const float>(), var_.as<const float>(), T_, H_, W_, x_dev.as<const int32_t>(),
                                            y_dev.as<const int32_t>(), inc_dev.as<const uint8_t>(), n, radius, types[k],
                                            out_dev.as<float>(), nullptr));
            HbmBlock::check(kb_copy_block_to_cpu(out[coadd_types[k]].mutable_data(), out_dev.as<void>(), out_bytes));
        }
        return out;
    }

private:
    py::object owner_;  // declared first: released last (what borrowed stacks belong to)
    HbmBlock sci_, var_;
    int T_ = 0, H_ = 0, W_ = 0;
};

}  // namespace search
#endif
