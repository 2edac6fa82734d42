// StackSearch of kbmod_amd.search: the object behind the reference's `kbmod.search.StackSearch`
// (stack_search.{h,cpp}:18-338; CPU search cpu_search_algorithms.cpp:20-124) with the same Python-visible
// methods, validation rules, error messages and log / timer labels.
//
// How it is put together here:
//  * every trajectory evaluation on the host -- the explicit CPU search (on_gpu=False), the
//    single-trajectory calls -- goes through kb::evaluate_trajectory_full of search_math.h, the one
//    evaluator the device kernels' epilogue also uses;
//  * on_gpu=True runs kb_device_search_filter of libkbmod_hip.so and filters / sorts in HBM; it never
//    falls back to the host: without a device it raises, as the reference does (stack_search.cpp:240);
//  * with more than one search device (set_search_devices) the candidate list is cut into contiguous
//    slices, one host thread per device searches its slice against its own replica of psi/phi
//    (kb_device_search_compact, 16-byte records), the records are copied to the first device and merged
//    there (kb_merge_compact) -- SURVEY 8(e): the fan-out lives behind search_all, the Python surface
//    does not change;
//  * psi/phi is built on the device, from a list of images (the reference's constructor) or from
//    contiguous [T][H][W] stacks without any per-image conversion (the ingest path).
#ifndef KBH_STACK_SEARCH_H_
#define KBH_STACK_SEARCH_H_

#include <algorithm>
#include <cfloat>
#include <cstring>
#include <sstream>
#include <chrono>
#include <thread>

#include "../search_math.h"
#include "common.h"
#include "image_utils.h"
#include "psi_phi_array.h"
#include "trajectory_list.h"

namespace search {

namespace detail {

inline void require(bool ok, const std::string& message) {
    if (!ok) throw std::runtime_error(message);
}

inline std::string bounds_text(const char* axis, int lo, int hi) {
    return std::string("Invalid search bounds for the ") + axis + " pixel [" + std::to_string(lo) + ", " +
           std::to_string(hi) + "]";
}

// A device block that is released when it goes out of scope.
struct DeviceBlock {
    void* ptr = nullptr;
    DeviceBlock() = default;
    explicit DeviceBlock(uint64_t bytes) { check_status(kb_allocate_gpu_block(std::max<uint64_t>(bytes, 1), &ptr)); }
    DeviceBlock(const DeviceBlock&) = delete;
    DeviceBlock& operator=(const DeviceBlock&) = delete;
    ~DeviceBlock() {
        if (ptr != nullptr) (void)kb_free_gpu_block(ptr);
    }
    template <typename T>
    T* as() const {
        return reinterpret_cast<T*>(ptr);
    }
    void* release() {  // the caller takes the block over
        void* p = ptr;
        ptr = nullptr;
        return p;
    }
};

}  // namespace detail

// cpu_search_algorithms.cpp:20-50: summed psi / phi, likelihood and flux of one trajectory on the host
// array -- kb::evaluate_trajectory_full with the sigma-G clip off (the CPU search has none).
inline void evaluate_trajectory_cpu(PsiPhiArray& psi_phi, Trajectory& candidate) {
    kb_search_params plain{};
    plain.do_sigmag_filter = 0;
    plain.min_observations = 0;
    kb_trajectory t;
    std::memcpy(&t, &candidate, sizeof(t));
    kb::evaluate_trajectory_full<1>(psi_phi.get_meta_data(), psi_phi.host_ptr(), psi_phi.get_cpu_time_array_ptr(), plain, &t,
                                    static_cast<const kb::SigmaGScratch<1>*>(nullptr));
    std::memcpy(&candidate, &t, sizeof(t));
}

// cpu_search_algorithms.cpp:57-124: per start pixel every candidate is evaluated, the list ordered by
// falling likelihood (candidates of equal likelihood stay in list order) and its head kept.
inline void search_cpu_only(PsiPhiArray& psi_phi_array, SearchParameters params, TrajectoryList& trj_to_search,
                            TrajectoryList& results) {
    const int64_t rows = (int64_t)params.y_start_max - params.y_start_min;
    const int64_t cols = (int64_t)params.x_start_max - params.x_start_min;
    detail::require(rows > 0 && cols > 0, "Invalid search bounds.");
    const std::vector<Trajectory>& cands = trj_to_search.get_list();
    const uint64_t keep = std::min<uint64_t>(cands.size(), params.results_per_pixel);
    results.resize(keep * (uint64_t)rows * (uint64_t)cols);
    results.reset_all();
    if (results.get_size() == 0) return;

    const kb_psi_phi_meta meta = psi_phi_array.get_meta_data();
    const void* array = psi_phi_array.host_ptr();  // brings the array to the host once, before the threads start
    const double* times = psi_phi_array.get_cpu_time_array_ptr();
    kb_search_params plain{};
    std::vector<Trajectory>& out = results.get_list();
#pragma omp parallel
    {
        std::vector<Trajectory> column(cands.size());
#pragma omp for schedule(dynamic, 64)
        for (int64_t pixel = 0; pixel < rows * cols; ++pixel) {
            const int y = (int)(pixel / cols + params.y_start_min), x = (int)(pixel % cols + params.x_start_min);
            for (size_t c = 0; c < cands.size(); ++c) {
                kb_trajectory t{};
                t.x = x;
                t.y = y;
                t.vx = cands[c].vx;
                t.vy = cands[c].vy;
                kb::evaluate_trajectory_full<1>(meta, array, times, plain, &t,
                                                static_cast<const kb::SigmaGScratch<1>*>(nullptr));
                std::memcpy(&column[c], &t, sizeof(t));
            }
            std::stable_sort(column.begin(), column.end(),
                             [](const Trajectory& a, const Trajectory& b) { return b.lh < a.lh; });
            std::copy_n(column.begin(), keep, out.begin() + pixel * keep);  // every pixel owns its slots (cf. :115)
        }
    }
}

// stack_search.cpp:22-39: psi in [0, T), phi in [T, 2T), invalid samples left at 0.
inline std::vector<float> extract_joint_psi_phi_curve(PsiPhiArray& psi_phi, const Trajectory& trj) {
    const unsigned int T = psi_phi.get_num_times();
    std::vector<float> curve(2 * T, 0.0f);
    for (unsigned int i = 0; i < T; ++i) {
        const double when = psi_phi.read_time(i);
        const PsiPhi sample = psi_phi.read_psi_phi(i, trj.get_y_index(when), trj.get_x_index(when));
        if (pixel_value_valid(sample.psi)) curve[i] = sample.psi;
        if (pixel_value_valid(sample.phi)) curve[i + T] = sample.phi;
    }
    return curve;
}

class StackSearch {
public:
    // wall-clock milliseconds of the last search_all on the host: search (tables + kernels), filter + sort in HBM, download of
    // the survivors, validity scan
    struct HostTimes {
        double search = 0, filter_sort = 0, download = 0, validate = 0, total = 0;
    };
    // stack_search.cpp:37-75
    StackSearch(std::vector<Image>& sci_imgs, std::vector<Image>& var_imgs, std::vector<Image>& psf_kernels,
                std::vector<double>& zeroed_times_in, int num_bytes = -1)
            : zeroed_times(zeroed_times_in), results(0) {
        rs_logger = logging::getLogger("kbmod.search.run_search");
        num_imgs = sci_imgs.size();
        detail::require(num_imgs != 0, "No images in the to process.");
        check_count("variance", "Variance", var_imgs.size());
        check_count("PSF kernel", "PSF Kernels", psf_kernels.size());
        check_count_times(zeroed_times.size());
        width = sci_imgs[0].cols;
        height = sci_imgs[0].rows;
        set_default_parameters(num_bytes);
        DebugTimer timer = DebugTimer("preparing Psi and Phi images", rs_logger);
        fill_psi_phi_array_from_image_arrays(psi_phi_array, num_bytes, sci_imgs, var_imgs, psf_kernels, zeroed_times);
        psi_phi_preloaded = false;
        timer.stop();
    }

    // The ingest form (work_unit.py:489-608 leaves the layers as arrays): contiguous
    // [T][H][W] float32 stacks go to the device builder as they are -- no per-image conversion, chunked
    // upload through pinned buffers overlapped with the correlation (kb_build_psi_phi_from_host_stack).
    StackSearch(const float* sci_stack, const float* var_stack, unsigned int T, unsigned int H, unsigned int W,
                std::vector<Image>& psf_kernels, std::vector<double>& zeroed_times_in, int num_bytes, uint32_t build_flags)
            : zeroed_times(zeroed_times_in), results(0) {
        rs_logger = logging::getLogger("kbmod.search.run_search");
        num_imgs = T;
        detail::require(num_imgs != 0 && H != 0 && W != 0, "No images in the to process.");
        check_count("PSF kernel", "PSF Kernels", psf_kernels.size());
        check_count_times(zeroed_times.size());
        detail::require(has_gpu(), "GPU is not available for the psi/phi build.");
        width = W;
        height = H;
        set_default_parameters(num_bytes);
        DebugTimer timer = DebugTimer("preparing Psi and Phi images", rs_logger);
        std::vector<int32_t> dims(T);
        std::vector<float> psf_packed;
        for (unsigned int i = 0; i < T; ++i) {
            detail::require(psf_kernels[i].rows == psf_kernels[i].cols, "PSF kernel must be square.");
            dims[i] = (int32_t)psf_kernels[i].rows;
            psf_packed.insert(psf_packed.end(), psf_kernels[i].data.begin(), psf_kernels[i].data.end());
        }
        kb_psi_phi_meta meta;
        void* dev = nullptr;
        check_status(kb_build_psi_phi_from_host_stack(sci_stack, var_stack, psf_packed.data(), dims.data(), (int32_t)T,
                                                      (int32_t)H, (int32_t)W, num_bytes, build_flags, &meta, &dev));
        psi_phi_array.adopt_device_array(meta, dev);
        psi_phi_array.set_time_array(zeroed_times);
        psi_phi_preloaded = false;
        timer.stop();
    }

    // The same build from stacks that are ALREADY on the current device (what kbmod_amd.fits_ingest leaves there after
    // decoding a WorkUnit file in HBM): kb_build_psi_phi_from_device_ex on the null stream, nothing crosses PCIe.
    struct DeviceStacks {};
    StackSearch(DeviceStacks, const float* sci_dev, const float* var_dev, unsigned int T, unsigned int H, unsigned int W,
                std::vector<Image>& psf_kernels, std::vector<double>& zeroed_times_in, int num_bytes, uint32_t build_flags)
            : zeroed_times(zeroed_times_in), results(0) {
        rs_logger = logging::getLogger("kbmod.search.run_search");
        num_imgs = T;
        detail::require(num_imgs != 0 && H != 0 && W != 0, "No images in the to process.");
        detail::require(sci_dev != nullptr && var_dev != nullptr, "Null device stack.");
        check_count("PSF kernel", "PSF Kernels", psf_kernels.size());
        check_count_times(zeroed_times.size());
        detail::require(has_gpu(), "GPU is not available for the psi/phi build.");
        width = W;
        height = H;
        set_default_parameters(num_bytes);
        DebugTimer timer = DebugTimer("preparing Psi and Phi images", rs_logger);
        std::vector<int32_t> dims(T);
        std::vector<float> psf_packed;
        for (unsigned int i = 0; i < T; ++i) {
            detail::require(psf_kernels[i].rows == psf_kernels[i].cols, "PSF kernel must be square.");
            dims[i] = (int32_t)psf_kernels[i].rows;
            psf_packed.insert(psf_packed.end(), psf_kernels[i].data.begin(), psf_kernels[i].data.end());
        }
        kb_psi_phi_meta meta;
        void* dev = nullptr;
        check_status(kb_build_psi_phi_from_device_ex(sci_dev, var_dev, psf_packed.data(), dims.data(), (int32_t)T, (int32_t)H,
                                                     (int32_t)W, num_bytes, build_flags, &meta, &dev, nullptr));
        check_status(kb_device_synchronize());
        psi_phi_array.adopt_device_array(meta, dev);
        psi_phi_array.set_time_array(zeroed_times);
        psi_phi_preloaded = false;
        timer.stop();
    }

    virtual ~StackSearch() {
        drop_replicas();
        psi_phi_array.clear();
    }

    unsigned int num_images() const { return num_imgs; }
    unsigned int get_image_width() const { return width; }
    unsigned int get_image_height() const { return height; }
    std::vector<double>& get_zeroed_times() { return zeroed_times; }
    PsiPhiArray& get_psi_phi_array() { return psi_phi_array; }
    const SearchParameters& get_params() const { return params; }
    const kb_search_stats& last_search_stats() const { return last_stats; }

    // stack_search.cpp:89-117
    void set_default_parameters(int num_bytes = -1) {
        detail::require(num_bytes == -1 || num_bytes == 1 || num_bytes == 2 || num_bytes == 4,
                        "Invalid encoding size. Must be -1, 1, 2 or 4. Got " + std::to_string(num_bytes));
        params = SearchParameters{};
        params.min_observations = 0;
        params.min_lh = 0.0;
        params.do_sigmag_filter = false;
        params.sgl_L = 0.25;
        params.sgl_H = 0.75;
        params.sigmag_coeff = -1.0;
        params.encode_num_bytes = (num_bytes == 1 || num_bytes == 2) ? num_bytes : -1;
        params.results_per_pixel = 8;
        params.x_start_min = 0;
        params.x_start_max = width;
        params.y_start_min = 0;
        params.y_start_max = height;
    }

    // stack_search.cpp:119-172
    void set_min_obs(int new_value) {
        detail::require(new_value >= 0, "min_obs must be >= 0. Got " + std::to_string(new_value));
        detail::require((unsigned int)new_value <= num_imgs,
                        "min_obs cannot be greater than the number of images. min_obs = " + std::to_string(new_value) +
                                ", num_imgs = " + std::to_string(num_imgs) + ".");
        params.min_observations = new_value;
    }
    void set_min_lh(float new_value) { params.min_lh = new_value; }
    void set_results_per_pixel(int new_value) {
        detail::require(new_value > 0, "Invalid results per pixel. Got " + std::to_string(new_value));
        params.results_per_pixel = new_value;
    }
    void enable_gpu_sigmag_filter(std::vector<float> percentiles, float sigmag_coeff, float min_lh) {
        detail::require(percentiles.size() == 2, "Invalid percentiles for sigma G filtering. Expected 2 values, got " +
                                                         std::to_string(percentiles.size()) + ".");
        const float lo = percentiles[0], hi = percentiles[1];
        detail::require(lo < hi && lo > 0.0 && hi < 1.0, "Invalid percentiles for sigma G filtering. Got [" +
                                                                 std::to_string(lo) + ", " + std::to_string(hi) + "].");
        detail::require(sigmag_coeff > 0.0,
                        "Invalid coefficient for sigma G filtering. Got " + std::to_string(sigmag_coeff) + ".");
        params.do_sigmag_filter = true;
        params.sgl_L = lo;
        params.sgl_H = hi;
        params.sigmag_coeff = sigmag_coeff;
        params.min_lh = min_lh;
    }
    void disable_gpu_sigmag_filter() { params.do_sigmag_filter = false; }
    void set_start_bounds_x(int x_min, int x_max) {
        detail::require(x_min < x_max, detail::bounds_text("x", x_min, x_max));
        params.x_start_min = x_min;
        params.x_start_max = x_max;
    }
    void set_start_bounds_y(int y_min, int y_max) {
        detail::require(y_min < y_max, detail::bounds_text("y", y_min, y_max));
        params.y_start_min = y_min;
        params.y_start_max = y_max;
    }

    // stack_search.cpp:174-186
    void preload_psi_phi_array() {
        if (psi_phi_array.on_gpu()) return;
        psi_phi_array.move_to_gpu();
        psi_phi_preloaded = true;
        resident_searched = false;
    }
    void unload_psi_phi_array() {
        if (!psi_phi_array.on_gpu()) return;
        psi_phi_array.clear_from_gpu();
        psi_phi_preloaded = false;
        resident_searched = false;
    }
    bool psi_phi_array_on_gpu() const { return psi_phi_array.on_gpu(); }

    // stack_search.cpp:193-219
    void evaluate_single_trajectory(Trajectory& trj, bool use_kernel) {
        if (!use_kernel) {
            evaluate_trajectory_cpu(psi_phi_array, trj);
            return;
        }
        detail::require(has_gpu(), "GPU is not available for kernel evaluation.");
        detail::require(psi_phi_array.get_num_times() <= MAX_NUM_IMAGES,
                        "Too many images to evaluate on GPU. Max = " + std::to_string(MAX_NUM_IMAGES));
        kb_trajectory t;
        std::memcpy(&t, &trj, sizeof(t));
        check_status(kb_evaluate_trajectory_host(&psi_phi_array.get_meta_data(), psi_phi_array.host_ptr(),
                                                 psi_phi_array.get_cpu_time_array_ptr(), params, &t));
        std::memcpy(&trj, &t, sizeof(t));
    }
    Trajectory search_linear_trajectory(int x, int y, float vx, float vy, bool use_kernel) {
        Trajectory trj = Trajectory::make_trajectory(x, y, vx, vy, 0.0f, 0.0f, 0);
        evaluate_single_trajectory(trj, use_kernel);
        return trj;
    }

    // Devices the GPU search fans out over (default: the current device alone).  Entries may repeat
    // (two slices on one device run one after the other): the tests exercise the fan-out that way on a
    // single GPU.
    void set_search_devices(const std::vector<int>& devices) {
        const int n = kb_device_count();
        for (int d : devices) {
            detail::require(d >= 0 && d < std::max(n, 1), "Invalid search device " + std::to_string(d));
        }
        drop_replicas();
        search_devices = devices;
    }
    const std::vector<int>& get_search_devices() const { return search_devices; }

    // stack_search.cpp:221-284
    void search_all(std::vector<Trajectory>& search_list, bool on_gpu) {
        TrajectoryList candidate_list(search_list);
        const uint64_t max_results = compute_max_results();
        DebugTimer core_timer = DebugTimer("Running batch search", rs_logger);
        std::stringstream logmsg;
        logmsg << "Searching X=[" << params.x_start_min << ", " << params.x_start_max << "] "
               << "Y=[" << params.y_start_min << ", " << params.y_start_max << "]\n"
               << "Allocating space for " << max_results << " results.";
        rs_logger->info(logmsg.str());

        DebugTimer search_timer = DebugTimer("Running search", rs_logger);
        using Clock = std::chrono::steady_clock;
        auto since = [](Clock::time_point t) { return std::chrono::duration<double, std::milli>(Clock::now() - t).count(); };
        const Clock::time_point t_call = Clock::now();
        host_ms = HostTimes{};
        int64_t validated_on_device = -1;  // -1: not checked there; 0: all valid; i + 1: result i is not
        if (on_gpu) {
            detail::require(has_gpu(), "GPU is not available for search.");
            detail::require(psi_phi_array.get_num_times() <= MAX_NUM_IMAGES,
                            "Number of images exceeds GPU maximum " + std::to_string(MAX_NUM_IMAGES));
            rs_logger->info("Moving all data to GPU.");
            if (!psi_phi_preloaded) psi_phi_array.move_to_gpu();
            struct ReleaseArray {  // the array leaves "in use on the device" however the search ends
                StackSearch* self;
                ~ReleaseArray() {
                    if (!self->psi_phi_preloaded) self->psi_phi_array.end_device_use();
                }
            } release{this};

            // The result slots are written by the kernels themselves; nothing is uploaded (the reference
            // uploads S*K*28 bytes of zeros here).
            // (both blocks stay with this object between searches, grown when a search needs more: two allocations and
            // two releases of S * K * 28 bytes were a third of the time search_all spent outside the search kernel)
            ResultBlocks& blocks = result_blocks;
            blocks.reserve(max_results * sizeof(Trajectory));
            detail::DeviceBlock &raw = blocks.raw, &kept = blocks.kept;
            int32_t counted = 0;  // the search wrote the per-pixel counts of its survivors (single device, likelihood threshold)
            bool fan_out = search_devices.size() > 1 && !search_list.empty();
            if (fan_out && params.results_per_pixel > 32) {
                // (the 16-byte exchange records and the merge kernels cover lists of up to 32 per pixel)
                rs_logger->warning("results_per_pixel = " + std::to_string(params.results_per_pixel) +
                                   " > 32: the search runs on one device instead of the " +
                                   std::to_string(search_devices.size()) + " set by set_search_devices.");
                fan_out = false;
            }
            if (fan_out) {
                // (what the parts send home has passed min_lh already -- the sparse exchange --, so the merged lists hold nothing
                // else and their lengths are the counts the filter below reads through)
                const bool want_counts = params.min_lh > -FLT_MAX && params.results_per_pixel <= 16;
                if (want_counts) blocks.reserve_counts(max_results / params.results_per_pixel);
                search_on_devices(candidate_list, raw.as<kb_trajectory>(), max_results,
                                  want_counts ? blocks.counts.as<uint8_t>() : nullptr, &counted);
            } else {
                candidate_list.move_to_gpu();
                // flag 256: this object owns the array and has not touched it since its last search on the device
                // (the library then reuses its padded copy when the frame geometry is the same)
                const uint32_t unchanged = (psi_phi_preloaded && resident_searched) ? 256u : 0u;
                // flag 1024: everything below min_lh is removed a few lines down (stack_search.cpp:266-270), so the kernels
                // need not insert it in the first place -- same survivors, no list work on pixels nothing reaches
                const uint32_t below_min_lh_dropped = 1024u;
                // With a likelihood threshold the search also writes, per start pixel, how many of its records pass it -- and
                // then leaves out the record runs of waves that keep nothing --, and the filter below reads through those
                // counts instead of walking S * K records (kb_device_search_filter_counted; include/kbmod_hip.h).
                if (params.min_lh > -FLT_MAX && params.results_per_pixel <= 32) {
                    blocks.reserve_counts(max_results / params.results_per_pixel);
                    check_status(kb_device_search_filter_counted(
                            &psi_phi_array.get_meta_data(), psi_phi_array.get_gpu_array_ptr(),
                            psi_phi_array.get_gpu_time_array_ptr(), params,
                            reinterpret_cast<const kb_trajectory*>(candidate_list.get_gpu_list_ptr()), candidate_list.get_size(),
                            raw.as<kb_trajectory>(), max_results, blocks.counts.as<uint8_t>(),
                            search_flags | unchanged | below_min_lh_dropped, nullptr, &last_stats, &counted));
                } else {
                    check_status(kb_device_search_filter(
                            &psi_phi_array.get_meta_data(), psi_phi_array.get_gpu_array_ptr(),
                            psi_phi_array.get_gpu_time_array_ptr(), params,
                            reinterpret_cast<const kb_trajectory*>(candidate_list.get_gpu_list_ptr()), candidate_list.get_size(),
                            raw.as<kb_trajectory>(), max_results, search_flags | unchanged | below_min_lh_dropped, nullptr, &last_stats));
                }
                resident_searched = psi_phi_preloaded;
                candidate_list.move_to_cpu();
            }
            search_timer.stop();
            host_ms.search = since(t_call);
            const Clock::time_point t_filter = Clock::now();
            // stack_search.cpp:266-277 (filter by lh, filter by obs_count, sort by lh) in HBM: only the
            // survivors cross PCIe.
            DebugTimer filter_timer = DebugTimer("Filtering results by LH and min_obs", rs_logger);
            uint64_t n_kept = 0;
            int64_t first_invalid = -1;  // (the validity scan of stack_search.cpp:280 rides on the gather of the survivors)
            if (counted != 0) {
                check_status(kb_filter_sort_results_counted(raw.as<kb_trajectory>(), max_results / params.results_per_pixel,
                                                            (int32_t)params.results_per_pixel, blocks.counts.as<uint8_t>(),
                                                            params.min_lh, params.min_observations, kept.as<kb_trajectory>(), &n_kept,
                                                            &first_invalid, nullptr));
            } else {
                check_status(kb_filter_sort_results_checked(raw.as<kb_trajectory>(), max_results, params.min_lh,
                                                            params.min_observations, kept.as<kb_trajectory>(), &n_kept, &first_invalid,
                                                            nullptr));
            }
            report_filtering(max_results, n_kept);
            filter_timer.stop();
            host_ms.filter_sort = since(t_filter);
            const Clock::time_point t_down = Clock::now();
            rs_logger->info("Clearing all data from GPU.");
            // (no resize(0) first: every kept slot is overwritten by the download, and value-initialising 58.7 MB of
            // trajectories a second time was a millisecond of cfg2's search_all)
            results.resize(n_kept);
            if (n_kept > 0) {
                check_status(kb_copy_block_to_cpu_locked(results.get_list().data(), kept.ptr, n_kept * sizeof(Trajectory)));
            }
            blocks.trim();
            host_ms.download = since(t_down);
            validated_on_device = first_invalid < 0 ? 0 : first_invalid + 1;
        } else {
            rs_logger->info("Running search on CPU.");
            results.resize(0);
            search_cpu_only(psi_phi_array, params, candidate_list, results);
            search_timer.stop();
            const uint64_t before = results.get_size();
            DebugTimer filter_timer = DebugTimer("Filtering results by LH and min_obs", rs_logger);
            results.filter_by_likelihood(params.min_lh);
            results.filter_by_obs_count(params.min_observations);
            report_filtering(before, results.get_size());
            filter_timer.stop();
            DebugTimer sort_timer = DebugTimer("Sorting results", rs_logger);
            results.sort_by_likelihood();
            sort_timer.stop();
        }
        const Clock::time_point t_valid = Clock::now();
        if (!on_gpu || validated_on_device < 0) {
            results.assert_valid();  // trajectory_list.cpp:152 / stack_search.cpp:280
        } else if (validated_on_device > 0) {
            const uint64_t at = (uint64_t)(validated_on_device - 1);
            throw std::runtime_error("Invalid trajectory detected at index " + std::to_string(at) + ": " +
                                     results.get_list()[at].to_string());
        }
        host_ms.validate = since(t_valid);
        host_ms.total = since(t_call);
        core_timer.stop();
    }

    // stack_search.cpp:286-300
    uint64_t compute_max_results() {
        detail::require(params.x_start_min < params.x_start_max,
                        detail::bounds_text("x", params.x_start_min, params.x_start_max));
        detail::require(params.y_start_min < params.y_start_max,
                        detail::bounds_text("y", params.y_start_min, params.y_start_max));
        return (uint64_t)(params.x_start_max - params.x_start_min) * (uint64_t)(params.y_start_max - params.y_start_min) *
               params.results_per_pixel;
    }

    // stack_search.cpp:302-318: (num_trj, 2*num_times) row-major.  With the array resident in HBM the
    // gather runs on the device (kb_psi_phi_curves); otherwise on the host threads.
    Image get_all_psi_phi_curves(const std::vector<Trajectory>& trajectories) {
        const int64_t num_trj = trajectories.size();
        Image out(num_trj, 2 * (int64_t)num_imgs);
        if (num_trj == 0) return out;
        if (has_gpu() && psi_phi_array.device_resident()) {
            psi_phi_array.ensure_device();  // times too
            detail::DeviceBlock trj_dev((uint64_t)num_trj * sizeof(Trajectory)), out_dev(out.data.size() * sizeof(float));
            check_status(kb_copy_block_to_gpu(trajectories.data(), trj_dev.ptr, (uint64_t)num_trj * sizeof(Trajectory)));
            check_status(kb_psi_phi_curves(&psi_phi_array.get_meta_data(), psi_phi_array.get_gpu_array_ptr(),
                                           psi_phi_array.get_gpu_time_array_ptr(), trj_dev.as<const kb_trajectory>(),
                                           (uint64_t)num_trj, out_dev.as<float>(), nullptr));
            check_status(kb_copy_block_to_cpu(out.data.data(), out_dev.ptr, out.data.size() * sizeof(float)));
            return out;
        }
        psi_phi_array.ensure_host();
#pragma omp parallel for schedule(dynamic, 64)
        for (int64_t i = 0; i < num_trj; ++i) {
            const std::vector<float> curve = extract_joint_psi_phi_curve(psi_phi_array, trajectories[i]);
            std::copy(curve.begin(), curve.end(), out.data.begin() + (size_t)i * 2 * num_imgs);
        }
        return out;
    }

    uint64_t get_number_total_results() { return results.get_size(); }
    std::vector<Trajectory> get_results(uint64_t start, uint64_t count) {  // :320-324
        rs_logger->debug("Reading results [" + std::to_string(start) + ", " + std::to_string(start + count) + ")");
        return results.get_batch(start, count);
    }
    std::vector<Trajectory>& get_all_results() { return results.get_list(); }
    void set_results(const std::vector<Trajectory>& new_results) { results.set_trajectories(new_results); }
    void clear_results() {
        if (results.on_gpu()) results.move_to_cpu();
        results.resize(0);
        result_blocks.release();  // (the device-side result buffers kept between searches go with them)
    }
    // kb_device_search_filter flags (include/kbmod_hip.h): kernel choice, self-check paths.
    void set_search_flags(uint32_t f) { search_flags = f; }
    const HostTimes& last_host_times() const { return host_ms; }

protected:
    void check_count(const char* what, const char* label, size_t n) const {
        detail::require(n == num_imgs, std::string("The number of science and ") + what + " images do not match. Science: " +
                                               std::to_string(num_imgs) + ", " + label + ": " + std::to_string(n));
    }
    void check_count_times(size_t n) const {
        detail::require(n == num_imgs, "The number of science images and zeroed times do not match. Science: " +
                                               std::to_string(num_imgs) + ", Zeroed Times: " + std::to_string(n));
    }
    void report_filtering(uint64_t before, uint64_t after) {
        rs_logger->debug("Core search returned " + std::to_string(before) + " results.\n");
        rs_logger->debug("After filtering by LH and min_obs " + std::to_string(after) + " results (" +
                         std::to_string(before - after) + " removed).\n");
    }

    // A device's own copy of the array and the epoch times (the home device uses the array itself).
    struct Replica {
        int device = -1;
        void* array = nullptr;
        void* times = nullptr;
    };
    void drop_replicas() {
        int prev = kb_get_device();
        for (Replica& r : replicas) {
            if (r.device < 0) continue;
            (void)kb_set_device(r.device);
            if (r.array != nullptr) (void)kb_free_gpu_block(r.array);
            if (r.times != nullptr) (void)kb_free_gpu_block(r.times);
        }
        replicas.clear();
        if (prev >= 0) (void)kb_set_device(prev);
    }
    const Replica& replica_on(int device, int home) {
        for (const Replica& r : replicas) {
            if (r.device == device) return r;
        }
        const uint64_t bytes = psi_phi_array.get_total_array_size(), tbytes = (uint64_t)num_imgs * sizeof(double);
        check_status(kb_set_device(device));
        // (DeviceBlock frees what was allocated if a later step throws; the replica takes the blocks over at the end)
        detail::DeviceBlock array(bytes), times(tbytes);
        check_status(kb_copy_block_between_gpus(array.ptr, device, psi_phi_array.get_gpu_array_ptr(), home, bytes));
        check_status(kb_copy_block_between_gpus(times.ptr, device, psi_phi_array.get_gpu_time_array_ptr(), home, tbytes));
        Replica r;
        r.device = device;
        r.array = array.release();
        r.times = times.release();
        replicas.push_back(r);
        return replicas.back();
    }

    // contiguous candidate slices, one host thread per slice on its device, 16-byte records
    // copied to the home device, per-pixel merge there.  `merged` (home device) receives the per-pixel lists.
    // Lists of up to 16 results take the tie-exact exchange: every device keeps 2 K records per pixel by stable
    // insertion (flag 512) and kb_merge_compact_exact replays the reference's insertion -- the result equals the
    // single-device search, ties included.  Longer lists (17 .. 32) take the plain merge (ties to the lower
    // candidate index: same likelihoods per slot, possibly another member of a tie), which is logged.
    // merged_counts (may be null): where the tie-exact merge leaves the number of merged records per start pixel -- it then
    // writes no slot for a wave nothing reaches (kb_merge_sparse_exact_counted); *counted says whether it did.
    void search_on_devices(TrajectoryList& candidates, kb_trajectory* merged, uint64_t max_results, uint8_t* merged_counts = nullptr,
                           int32_t* counted = nullptr) {
        if (counted != nullptr) *counted = 0;
        const int home = kb_get_device();
        const int n_parts = (int)search_devices.size();
        const std::vector<Trajectory>& cands = candidates.get_list();
        const uint64_t n = cands.size();
        const bool exact = params.results_per_pixel <= 16;
        if (!exact) {
            rs_logger->warning("results_per_pixel = " + std::to_string(params.results_per_pixel) +
                               " > 16: the multi-device merge breaks ties by candidate index (a tie may keep another "
                               "member than a single-device search).");
        }
        SearchParameters part_params = params;
        if (exact) part_params.results_per_pixel = 2 * params.results_per_pixel;
        const uint64_t part_results = max_results / params.results_per_pixel * part_params.results_per_pixel;
        const uint32_t part_flags = search_flags | (exact ? 512u : 0u) | 1024u;  // (1024: search_all drops lh < min_lh behind the merge)
        for (int d : search_devices) {  // replicas are made here, one after the other, before the threads start
            if (d != home) (void)replica_on(d, home);
        }
        check_status(kb_set_device(home));
        // What travels to the home device.  Tie-exact lists go in their sparse form (kb_sparsify_compact: one count byte per
        // pixel + the records search_all's post-filter would keep, a few MB instead of S x 2K x 16 bytes per device for a
        // search with a likelihood threshold); lists of 17 .. 32 as they are.
        const uint64_t n_pixels = max_results / params.results_per_pixel;
        const uint64_t header_bytes = kb_sparse_header_bytes(n_pixels);
        detail::DeviceBlock gathered(exact ? 1 : (uint64_t)n_parts * part_results * sizeof(kb_compact_result));
        detail::DeviceBlock headers(exact ? (uint64_t)n_parts * header_bytes : 1);
        std::vector<detail::DeviceBlock> packed_home(n_parts);  // (sized by what each part keeps)
        std::vector<const kb_compact_result*> packed_ptrs(n_parts, nullptr);
        detail::DeviceBlock all_cands(n * sizeof(Trajectory));
        check_status(kb_copy_block_to_gpu(cands.data(), all_cands.ptr, n * sizeof(Trajectory)));

        std::vector<std::string> errors(n_parts);
        std::vector<kb_search_stats> part_stats(n_parts);
        auto run_part = [&](int part) {
            try {
                const int device = search_devices[part];
                check_status(kb_set_device(device));
                const uint64_t lo = n * part / n_parts, hi = n * (part + 1) / n_parts;
                const void* array = psi_phi_array.get_gpu_array_ptr();
                const double* times = psi_phi_array.get_gpu_time_array_ptr();
                if (device != home) {
                    for (const Replica& r : replicas) {
                        if (r.device == device) {
                            array = r.array;
                            times = reinterpret_cast<const double*>(r.times);
                        }
                    }
                }
                detail::DeviceBlock slice((hi - lo) * sizeof(Trajectory)), records(part_results * sizeof(kb_compact_result));
                if (hi > lo) check_status(kb_copy_block_to_gpu(cands.data() + lo, slice.ptr, (hi - lo) * sizeof(Trajectory)));
                // Tie-exact parts: the search writes the count bytes of the sparse header itself where its kernel instance can
                // (and then leaves out the record runs of waves that keep nothing); kb_sparsify_compact counts otherwise.
                detail::DeviceBlock header(exact ? header_bytes : 1);
                int32_t counted = 0;
                if (exact) {
                    check_status(kb_device_search_counted(&psi_phi_array.get_meta_data(), array, times, part_params,
                                                          slice.as<const kb_trajectory>(), hi - lo, (int32_t)lo,
                                                          records.as<kb_compact_result>(), part_results, header.as<uint8_t>(),
                                                          part_flags, nullptr, &part_stats[part], &counted));
                } else {
                    check_status(kb_device_search_compact(&psi_phi_array.get_meta_data(), array, times, part_params,
                                                          slice.as<const kb_trajectory>(), hi - lo, (int32_t)lo,
                                                          records.as<kb_compact_result>(), part_results, part_flags, nullptr,
                                                          &part_stats[part]));
                }
                // (the stats path returns without a final stream synchronisation: a fault inside the kernels must
                // surface here, on the device it happened on, not in a later call)
                check_status(kb_device_synchronize());
                if (exact) {
                    uint64_t kept = 0, room = std::max<uint64_t>(1024, part_results / 16);
                    detail::DeviceBlock packed(room * sizeof(kb_compact_result));
                    auto sparsify = [&]() {
                        return counted ? kb_sparsify_counted(records.as<const kb_compact_result>(), n_pixels,
                                                             (int32_t)part_params.results_per_pixel, header.as<uint8_t>(),
                                                             packed.as<kb_compact_result>(), room, &kept, nullptr)
                                       : kb_sparsify_compact(records.as<const kb_compact_result>(), n_pixels,
                                                             (int32_t)part_params.results_per_pixel, params.min_lh, header.as<uint8_t>(),
                                                             packed.as<kb_compact_result>(), room, &kept, nullptr);
                    };
                    int rc = sparsify();
                    if (rc != 0 && kept > room) {  // more survivors than guessed: the count is known now
                        detail::DeviceBlock larger(kept * sizeof(kb_compact_result));
                        std::swap(packed.ptr, larger.ptr);
                        room = kept;
                        rc = sparsify();
                    }
                    check_status(rc);
                    check_status(kb_copy_block_between_gpus(headers.as<uint8_t>() + (uint64_t)part * header_bytes, home, header.ptr,
                                                            device, header_bytes));
                    if (kept > 0) {
                        check_status(kb_set_device(home));
                        detail::DeviceBlock at_home(kept * sizeof(kb_compact_result));
                        check_status(kb_set_device(device));
                        check_status(kb_copy_block_between_gpus(at_home.ptr, home, packed.ptr, device, kept * sizeof(kb_compact_result)));
                        std::swap(packed_home[part].ptr, at_home.ptr);
                        packed_ptrs[part] = packed_home[part].as<const kb_compact_result>();
                    }
                } else {
                    check_status(kb_copy_block_between_gpus(gathered.as<kb_compact_result>() + (uint64_t)part * part_results, home,
                                                            records.ptr, device, part_results * sizeof(kb_compact_result)));
                }
            } catch (const std::exception& e) {
                errors[part] = e.what();
            }
        };
        std::vector<std::thread> workers;
        for (int part = 1; part < n_parts; ++part) workers.emplace_back(run_part, part);
        run_part(0);
        for (std::thread& w : workers) w.join();
        check_status(kb_set_device(home));
        for (const std::string& e : errors) detail::require(e.empty(), e);
        if (exact) {
            check_status(kb_merge_sparse_exact_counted(headers.as<const uint8_t>(), header_bytes, packed_ptrs.data(), n_parts,
                                                       (int32_t)part_params.results_per_pixel, params,
                                                       all_cands.as<const kb_trajectory>(), n, merged, merged_counts, nullptr));
            if (counted != nullptr && merged_counts != nullptr) *counted = 1;
        } else {
            check_status(kb_merge_compact(gathered.as<const kb_compact_result>(), n_parts, params,
                                          all_cands.as<const kb_trajectory>(), n, merged, nullptr));
        }
        check_status(kb_device_synchronize());
        last_stats = part_stats[0];
        for (int part = 1; part < n_parts; ++part) {
            last_stats.num_evals += part_stats[part].num_evals;
            last_stats.algorithmic_bytes += part_stats[part].algorithmic_bytes;
            last_stats.search_kernel_ms = std::max(last_stats.search_kernel_ms, part_stats[part].search_kernel_ms);
            last_stats.num_search_launches += part_stats[part].num_search_launches;
        }
        // replicas are whole copies of the array: they go when the home array leaves the device with this search
        if (!psi_phi_preloaded) drop_replicas();
    }

    SearchParameters params;
    unsigned int height;
    unsigned int width;
    unsigned int num_imgs;
    std::vector<double> zeroed_times;
    bool psi_phi_preloaded;
    PsiPhiArray psi_phi_array;
    TrajectoryList results;
    logging::Logger* rs_logger;
    kb_search_stats last_stats{};
    uint32_t search_flags = 0;
    HostTimes host_ms;
    bool resident_searched = false;  // the resident array has been searched on the device since it was (re)loaded
    std::vector<int> search_devices;
    // result buffers of search_all on the home device (raw per-pixel lists, filtered + sorted survivors)
    struct ResultBlocks {
        detail::DeviceBlock raw, kept, counts;
        uint64_t bytes = 0, count_bytes = 0;
        int device = -1;  // the device the blocks live on: a later search from a thread whose current device differs gets its own
        // Blocks beyond this are returned once a search's results have reached the host (get_gpu_free_memory / validate_gpu
        // report them as taken between searches otherwise): 1 GiB or a sixteenth of the device's memory, whichever is larger
        // -- 18 GB on an MI355X; re-allocating 2 x 3.8 GB for every search of a 4096 x 4096 stack showed as an occasional
        // stall of seconds in hipMalloc.
        static uint64_t keep_bytes() {
            static const uint64_t k = std::max<uint64_t>(1ull << 30, (uint64_t)kb_gpu_total_memory() / 16);
            return k;
        }
        void reserve(uint64_t need) {
            const int now = kb_get_device();
            if (need <= bytes && raw.ptr != nullptr && kept.ptr != nullptr && device == now) return;
            release();
            check_status(kb_allocate_gpu_block(std::max<uint64_t>(need, 1), &raw.ptr));
            check_status(kb_allocate_gpu_block(std::max<uint64_t>(need, 1), &kept.ptr));
            bytes = need;
            device = now;
        }
        // (one byte per start pixel, on the device of the other two: call after reserve)
        void reserve_counts(uint64_t need) {
            if (need <= count_bytes && counts.ptr != nullptr) return;
            if (counts.ptr != nullptr) (void)kb_free_gpu_block(counts.release());
            check_status(kb_allocate_gpu_block(std::max<uint64_t>(need, 1), &counts.ptr));
            count_bytes = need;
        }
        void release() {
            const int now = kb_get_device();
            const bool elsewhere = device >= 0 && now >= 0 && device != now;
            if (elsewhere) (void)kb_set_device(device);  // a block is freed on the device it was allocated on
            if (raw.ptr != nullptr) (void)kb_free_gpu_block(raw.release());
            if (kept.ptr != nullptr) (void)kb_free_gpu_block(kept.release());
            if (counts.ptr != nullptr) (void)kb_free_gpu_block(counts.release());
            count_bytes = 0;
            if (elsewhere) (void)kb_set_device(now);
            bytes = 0;
            device = -1;
        }
        void trim() {
            if (bytes > keep_bytes()) release();
        }
    } result_blocks;
    std::vector<Replica> replicas;
};

}  // namespace search
#endif
