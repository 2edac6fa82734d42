// StackSearch and the explicit CPU search of kbmod_amd.search.
//
// Mirrors stack_search.{h,cpp}:18-338 and cpu_search_algorithms.cpp:20-124 of
// the reference: same methods, validation, error messages and log/timer labels.
// on_gpu=True dispatches to libkbmod_hip.so (kb_device_search_filter) and NEVER
// falls back to the host: without a device it raises, as the reference does
// (stack_search.cpp:240).  on_gpu=False is the reference's explicit CPU search.
#ifndef KBH_STACK_SEARCH_H_
#define KBH_STACK_SEARCH_H_

#include <sstream>

#include "common.h"
#include "image_utils.h"
#include "psi_phi_array.h"
#include "trajectory_list.h"

namespace search {

// cpu_search_algorithms.cpp:20-50
inline void evaluate_trajectory_cpu(PsiPhiArray& psi_phi, Trajectory& candidate) {
    const unsigned int num_times = psi_phi.get_num_times();
    float psi_sum = 0.0;
    float phi_sum = 0.0;
    candidate.obs_count = 0;
    candidate.lh = -1.0;
    candidate.flux = -1.0;
    int num_seen = 0;
    for (unsigned int i = 0; i < num_times; ++i) {
        double curr_time = psi_phi.read_time(i);
        int current_x = (int)(floor(candidate.x + candidate.vx * curr_time + 0.5f));
        int current_y = (int)(floor(candidate.y + candidate.vy * curr_time + 0.5f));
        PsiPhi pixel_vals = psi_phi.read_psi_phi(i, current_y, current_x);
        if (std::isfinite(pixel_vals.psi) && std::isfinite(pixel_vals.phi)) {
            psi_sum += pixel_vals.psi;
            phi_sum += pixel_vals.phi;
            num_seen += 1;
        }
    }
    candidate.obs_count = num_seen;
    candidate.lh = (phi_sum > 0) ? (psi_sum / std::sqrt(phi_sum)) : -1.0;
    candidate.flux = (phi_sum > 0) ? (psi_sum / phi_sum) : -1.0;
}

// cpu_search_algorithms.cpp:57-86: evaluate every candidate at (y, x), sort
// descending by lh, keep the first num_results (ties keep candidate order).
inline void evaluate_single_pixel(int y, int x, PsiPhiArray& psi_phi, const std::vector<Trajectory>& cands,
                                  int num_results, std::vector<Trajectory>& scratch, Trajectory* out) {
    const uint64_t num_candidates = cands.size();
    if ((uint64_t)num_results > num_candidates) {
        throw std::runtime_error("evaluate_single_pixel requesting more results than candidates.");
    }
    scratch.resize(num_candidates);
    for (uint64_t trj_idx = 0; trj_idx < num_candidates; ++trj_idx) {
        Trajectory& curr_trj = scratch[trj_idx];
        curr_trj.x = x;
        curr_trj.y = y;
        curr_trj.vx = cands[trj_idx].vx;
        curr_trj.vy = cands[trj_idx].vy;
        curr_trj.flux = 0.0;
        curr_trj.obs_count = 0;
        evaluate_trajectory_cpu(psi_phi, curr_trj);
    }
    std::stable_sort(scratch.begin(), scratch.end(),
                     [](const Trajectory& a, const Trajectory& b) { return b.lh < a.lh; });
    for (int i = 0; i < num_results; ++i) out[i] = scratch[i];
}

// cpu_search_algorithms.cpp:93-124
inline void search_cpu_only(PsiPhiArray& psi_phi_array, SearchParameters params, TrajectoryList& trj_to_search,
                            TrajectoryList& results) {
    const int64_t search_height = (int64_t)params.y_start_max - params.y_start_min;
    const int64_t search_width = (int64_t)params.x_start_max - params.x_start_min;
    if (search_height <= 0 || search_width <= 0) throw std::runtime_error("Invalid search bounds.");
    const uint64_t num_candidates = trj_to_search.get_size();
    const uint64_t results_per_test =
            (num_candidates < params.results_per_pixel) ? num_candidates : params.results_per_pixel;
    const uint64_t total_results = results_per_test * (uint64_t)search_height * (uint64_t)search_width;
    results.resize(total_results);
    results.reset_all();
    if (total_results == 0) return;
    psi_phi_array.ensure_host();  // once, outside the parallel region

    const std::vector<Trajectory>& cands = trj_to_search.get_list();
    std::vector<Trajectory>& out = results.get_list();
#pragma omp parallel
    {
        std::vector<Trajectory> scratch;
#pragma omp for collapse(2) schedule(dynamic, 16)
        for (int64_t y_i = 0; y_i < search_height; ++y_i) {
            for (int64_t x_i = 0; x_i < search_width; ++x_i) {
                // Each pixel owns its slots, so no critical section is needed (cf. :115).
                const uint64_t start_ind = ((uint64_t)y_i * (uint64_t)search_width + (uint64_t)x_i) * results_per_test;
                evaluate_single_pixel((int)(y_i + params.y_start_min), (int)(x_i + params.x_start_min), psi_phi_array,
                                      cands, (int)results_per_test, scratch, &out[start_ind]);
            }
        }
    }
}

// stack_search.cpp:22-39
inline std::vector<float> extract_joint_psi_phi_curve(PsiPhiArray& psi_phi, const Trajectory& trj) {
    const unsigned int num_times = psi_phi.get_num_times();
    std::vector<float> result(2 * num_times, 0.0);
    for (unsigned int i = 0; i < num_times; ++i) {
        double time = psi_phi.read_time(i);
        PsiPhi v = psi_phi.read_psi_phi(i, trj.get_y_index(time), trj.get_x_index(time));
        if (pixel_value_valid(v.psi)) result[i] = v.psi;
        if (pixel_value_valid(v.phi)) result[i + num_times] = v.phi;
    }
    return result;
}

class StackSearch {
public:
    // stack_search.cpp:37-75
    StackSearch(std::vector<Image>& sci_imgs, std::vector<Image>& var_imgs, std::vector<Image>& psf_kernels,
                std::vector<double>& zeroed_times_in, int num_bytes = -1)
            : zeroed_times(zeroed_times_in), results(0) {
        rs_logger = logging::getLogger("kbmod.search.run_search");
        num_imgs = sci_imgs.size();
        if (num_imgs == 0) throw std::runtime_error("No images in the to process.");
        if (sci_imgs.size() != var_imgs.size()) {
            throw std::runtime_error("The number of science and variance images do not match. Science: " +
                                     std::to_string(sci_imgs.size()) + ", Variance: " +
                                     std::to_string(var_imgs.size()));
        }
        if (sci_imgs.size() != psf_kernels.size()) {
            throw std::runtime_error("The number of science and PSF kernel images do not match. Science: " +
                                     std::to_string(sci_imgs.size()) + ", PSF Kernels: " +
                                     std::to_string(psf_kernels.size()));
        }
        if (sci_imgs.size() != zeroed_times.size()) {
            throw std::runtime_error("The number of science images and zeroed times do not match. Science: " +
                                     std::to_string(sci_imgs.size()) + ", Zeroed Times: " +
                                     std::to_string(zeroed_times.size()));
        }
        width = sci_imgs[0].cols;
        height = sci_imgs[0].rows;
        set_default_parameters(num_bytes);
        DebugTimer timer = DebugTimer("preparing Psi and Phi images", rs_logger);
        fill_psi_phi_array_from_image_arrays(psi_phi_array, num_bytes, sci_imgs, var_imgs, psf_kernels,
                                             zeroed_times);
        psi_phi_preloaded = false;
        timer.stop();
    }
    virtual ~StackSearch() { psi_phi_array.clear(); }

    unsigned int num_images() const { return num_imgs; }
    unsigned int get_image_width() const { return width; }
    unsigned int get_image_height() const { return height; }
    std::vector<double>& get_zeroed_times() { return zeroed_times; }
    PsiPhiArray& get_psi_phi_array() { return psi_phi_array; }
    const SearchParameters& get_params() const { return params; }
    const kb_search_stats& last_search_stats() const { return last_stats; }

    // stack_search.cpp:89-117
    void set_default_parameters(int num_bytes = -1) {
        params.min_observations = 0;
        params.min_lh = 0.0;
        params.do_sigmag_filter = false;
        params.sgl_L = 0.25;
        params.sgl_H = 0.75;
        params.sigmag_coeff = -1.0;
        if (num_bytes == 1 || num_bytes == 2) {
            params.encode_num_bytes = num_bytes;
        } else if (num_bytes == -1 || num_bytes == 4) {
            params.encode_num_bytes = -1;
        } else {
            throw std::runtime_error("Invalid encoding size. Must be -1, 1, 2 or 4. Got " +
                                     std::to_string(num_bytes));
        }
        params.results_per_pixel = 8;
        params.x_start_min = 0;
        params.x_start_max = width;
        params.y_start_min = 0;
        params.y_start_max = height;
    }
    // stack_search.cpp:119-172
    void set_min_obs(int new_value) {
        if (new_value < 0) throw std::runtime_error("min_obs must be >= 0. Got " + std::to_string(new_value));
        if ((unsigned int)new_value > num_imgs)
            throw std::runtime_error("min_obs cannot be greater than the number of images. min_obs = " +
                                     std::to_string(new_value) + ", num_imgs = " + std::to_string(num_imgs) + ".");
        params.min_observations = new_value;
    }
    void set_min_lh(float new_value) { params.min_lh = new_value; }
    void set_results_per_pixel(int new_value) {
        if (new_value <= 0) throw std::runtime_error("Invalid results per pixel. Got " + std::to_string(new_value));
        params.results_per_pixel = new_value;
    }
    void enable_gpu_sigmag_filter(std::vector<float> percentiles, float sigmag_coeff, float min_lh) {
        if (percentiles.size() != 2) {
            throw std::runtime_error("Invalid percentiles for sigma G filtering. Expected 2 values, got " +
                                     std::to_string(percentiles.size()) + ".");
        }
        if ((percentiles[0] >= percentiles[1]) || (percentiles[0] <= 0.0) || (percentiles[1] >= 1.0)) {
            throw std::runtime_error("Invalid percentiles for sigma G filtering. Got [" +
                                     std::to_string(percentiles[0]) + ", " + std::to_string(percentiles[1]) + "].");
        }
        if (sigmag_coeff <= 0.0) {
            throw std::runtime_error("Invalid coefficient for sigma G filtering. Got " +
                                     std::to_string(sigmag_coeff) + ".");
        }
        params.do_sigmag_filter = true;
        params.sgl_L = percentiles[0];
        params.sgl_H = percentiles[1];
        params.sigmag_coeff = sigmag_coeff;
        params.min_lh = min_lh;
    }
    void disable_gpu_sigmag_filter() { params.do_sigmag_filter = false; }
    void set_start_bounds_x(int x_min, int x_max) {
        if (x_min >= x_max) {
            throw std::runtime_error("Invalid search bounds for the x pixel [" + std::to_string(x_min) + ", " +
                                     std::to_string(x_max) + "]");
        }
        params.x_start_min = x_min;
        params.x_start_max = x_max;
    }
    void set_start_bounds_y(int y_min, int y_max) {
        if (y_min >= y_max) {
            throw std::runtime_error("Invalid search bounds for the y pixel [" + std::to_string(y_min) + ", " +
                                     std::to_string(y_max) + "]");
        }
        params.y_start_min = y_min;
        params.y_start_max = y_max;
    }

    // stack_search.cpp:174-186
    void preload_psi_phi_array() {
        if (!psi_phi_array.on_gpu()) {
            psi_phi_array.move_to_gpu();
            psi_phi_preloaded = true;
        }
    }
    void unload_psi_phi_array() {
        if (psi_phi_array.on_gpu()) {
            psi_phi_array.clear_from_gpu();
            psi_phi_preloaded = false;
        }
    }
    bool psi_phi_array_on_gpu() const { return psi_phi_array.on_gpu(); }

    // stack_search.cpp:193-207
    void evaluate_single_trajectory(Trajectory& trj, bool use_kernel) {
        if (!use_kernel) {
            evaluate_trajectory_cpu(psi_phi_array, trj);
        } else {
            if (!has_gpu()) throw std::runtime_error("GPU is not available for kernel evaluation.");
            if (psi_phi_array.get_num_times() > MAX_NUM_IMAGES) {
                throw std::runtime_error("Too many images to evaluate on GPU. Max = " +
                                         std::to_string(MAX_NUM_IMAGES));
            }
            kb_trajectory t;
            std::memcpy(&t, &trj, sizeof(t));
            check_status(kb_evaluate_trajectory_host(&psi_phi_array.get_meta_data(), psi_phi_array.host_ptr(),
                                                     psi_phi_array.get_cpu_time_array_ptr(), params, &t));
            std::memcpy(&trj, &t, sizeof(t));
        }
    }
    // stack_search.cpp:209-219
    Trajectory search_linear_trajectory(int x, int y, float vx, float vy, bool use_kernel) {
        Trajectory result;
        result.x = x;
        result.y = y;
        result.vx = vx;
        result.vy = vy;
        evaluate_single_trajectory(result, use_kernel);
        return result;
    }

    // stack_search.cpp:221-284
    void search_all(std::vector<Trajectory>& search_list, bool on_gpu) {
        TrajectoryList candidate_list(search_list);
        uint64_t max_results = compute_max_results();
        DebugTimer core_timer = DebugTimer("Running batch search", rs_logger);
        std::stringstream logmsg;
        logmsg << "Searching X=[" << params.x_start_min << ", " << params.x_start_max << "] "
               << "Y=[" << params.y_start_min << ", " << params.y_start_max << "]\n"
               << "Allocating space for " << max_results << " results.";
        rs_logger->info(logmsg.str());

        DebugTimer search_timer = DebugTimer("Running search", rs_logger);
        if (on_gpu) {
            if (!has_gpu()) throw std::runtime_error("GPU is not available for search.");
            if (psi_phi_array.get_num_times() > MAX_NUM_IMAGES) {
                throw std::runtime_error("Number of images exceeds GPU maximum " + std::to_string(MAX_NUM_IMAGES));
            }
            rs_logger->info("Moving all data to GPU.");
            if (!psi_phi_preloaded) psi_phi_array.move_to_gpu();
            candidate_list.move_to_gpu();
            // The result slots are initialised by the kernel itself; nothing is
            // uploaded (the reference uploads S*K*28 bytes of zeros here).
            void* results_dev = nullptr;
            void* sorted_dev = nullptr;
            const uint64_t n_alloc = std::max<uint64_t>(max_results, 1) * sizeof(Trajectory);
            check_status(kb_allocate_gpu_block(n_alloc, &results_dev));
            try {
                check_status(kb_allocate_gpu_block(n_alloc, &sorted_dev));
                check_status(kb_device_search_filter(
                        &psi_phi_array.get_meta_data(), psi_phi_array.get_gpu_array_ptr(),
                        psi_phi_array.get_gpu_time_array_ptr(), params,
                        reinterpret_cast<const kb_trajectory*>(candidate_list.get_gpu_list_ptr()),
                        candidate_list.get_size(), reinterpret_cast<kb_trajectory*>(results_dev), max_results,
                        search_flags, nullptr, &last_stats));
                search_timer.stop();
                // stack_search.cpp:266-277 (filter by lh, filter by obs_count, sort by lh) done in HBM:
                // only the survivors cross PCIe.
                DebugTimer filter_timer = DebugTimer("Filtering results by LH and min_obs", rs_logger);
                uint64_t kept = 0;
                check_status(kb_filter_sort_results(reinterpret_cast<const kb_trajectory*>(results_dev), max_results,
                                                    params.min_lh, params.min_observations,
                                                    reinterpret_cast<kb_trajectory*>(sorted_dev), &kept, nullptr));
                rs_logger->debug("Core search returned " + std::to_string(max_results) + " results.\n");
                rs_logger->debug("After filtering by LH and min_obs " + std::to_string(kept) + " results (" +
                                 std::to_string(max_results - kept) + " removed).\n");
                filter_timer.stop();
                rs_logger->info("Clearing all data from GPU.");
                results.resize(0);
                results.resize(kept);
                if (kept > 0) {
                    check_status(kb_copy_block_to_cpu(results.get_list().data(), sorted_dev, kept * sizeof(Trajectory)));
                }
            } catch (...) {
                (void)kb_free_gpu_block(results_dev);
                if (sorted_dev != nullptr) (void)kb_free_gpu_block(sorted_dev);
                if (!psi_phi_preloaded) psi_phi_array.end_device_use();
                throw;
            }
            (void)kb_free_gpu_block(results_dev);
            (void)kb_free_gpu_block(sorted_dev);
            candidate_list.move_to_cpu();
            if (!psi_phi_preloaded) psi_phi_array.end_device_use();
            results.assert_valid();  // trajectory_list.cpp:152 / stack_search.cpp:280
            core_timer.stop();
            return;
        } else {
            rs_logger->info("Running search on CPU.");
            results.resize(0);
            search_cpu_only(psi_phi_array, params, candidate_list, results);
        }
        search_timer.stop();

        uint64_t num_results = results.get_size();
        rs_logger->debug("Core search returned " + std::to_string(num_results) + " results.\n");
        DebugTimer filter_timer = DebugTimer("Filtering results by LH and min_obs", rs_logger);
        results.filter_by_likelihood(params.min_lh);
        results.filter_by_obs_count(params.min_observations);
        uint64_t new_num_results = results.get_size();
        rs_logger->debug("After filtering by LH and min_obs " + std::to_string(new_num_results) + " results (" +
                         std::to_string(num_results - new_num_results) + " removed).\n");
        filter_timer.stop();
        DebugTimer sort_timer = DebugTimer("Sorting results", rs_logger);
        results.sort_by_likelihood();
        sort_timer.stop();
        results.assert_valid();
        core_timer.stop();
    }

    // stack_search.cpp:286-300
    uint64_t compute_max_results() {
        if (params.x_start_min >= params.x_start_max)
            throw std::runtime_error("Invalid search bounds for the x pixel [" +
                                     std::to_string(params.x_start_min) + ", " +
                                     std::to_string(params.x_start_max) + "]");
        if (params.y_start_min >= params.y_start_max)
            throw std::runtime_error("Invalid search bounds for the y pixel [" +
                                     std::to_string(params.y_start_min) + ", " +
                                     std::to_string(params.y_start_max) + "]");
        uint64_t search_width = params.x_start_max - params.x_start_min;
        uint64_t search_height = params.y_start_max - params.y_start_min;
        return search_width * search_height * params.results_per_pixel;
    }

    // stack_search.cpp:302-318: (num_trj, 2*num_times) row-major.  With the array resident in HBM the
    // gather runs on the device (kb_psi_phi_curves); otherwise the reference's host loop.
    Image get_all_psi_phi_curves(const std::vector<Trajectory>& trajectories) {
        const int64_t num_trj = trajectories.size();
        Image out(num_trj, 2 * (int64_t)num_imgs);
        if (num_trj == 0) return out;
        if (has_gpu() && psi_phi_array.device_resident()) {
            psi_phi_array.ensure_device();  // times too
            void* trj_dev = nullptr;
            void* out_dev = nullptr;
            check_status(kb_allocate_gpu_block((uint64_t)num_trj * sizeof(Trajectory), &trj_dev));
            try {
                check_status(kb_allocate_gpu_block(out.data.size() * sizeof(float), &out_dev));
                check_status(kb_copy_block_to_gpu(trajectories.data(), trj_dev, (uint64_t)num_trj * sizeof(Trajectory)));
                check_status(kb_psi_phi_curves(&psi_phi_array.get_meta_data(), psi_phi_array.get_gpu_array_ptr(),
                                               psi_phi_array.get_gpu_time_array_ptr(),
                                               reinterpret_cast<const kb_trajectory*>(trj_dev), (uint64_t)num_trj,
                                               reinterpret_cast<float*>(out_dev), nullptr));
                check_status(kb_copy_block_to_cpu(out.data.data(), out_dev, out.data.size() * sizeof(float)));
            } catch (...) {
                (void)kb_free_gpu_block(trj_dev);
                if (out_dev != nullptr) (void)kb_free_gpu_block(out_dev);
                throw;
            }
            (void)kb_free_gpu_block(trj_dev);
            (void)kb_free_gpu_block(out_dev);
            return out;
        }
        psi_phi_array.ensure_host();
#pragma omp parallel for schedule(dynamic, 64)
        for (int64_t i = 0; i < num_trj; ++i) {
            std::vector<float> curve = extract_joint_psi_phi_curve(psi_phi_array, trajectories[i]);
            std::memcpy(&out.data[(size_t)i * 2 * num_imgs], curve.data(), curve.size() * sizeof(float));
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
    }
    // Debug/self-check hook: bit 0 forces the per-lane exact-position path.
    void set_search_flags(uint32_t f) { search_flags = f; }

protected:
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
};

}  // namespace search
#endif
