// TrajectoryList: host vector of Trajectory + an HBM mirror.
// Mirrors trajectory_list.{h,cpp}:18-239 and gpu_array.h:27-192 of the
// reference (state machine, bounds errors, filters, extract helpers).
#ifndef KBH_TRAJECTORY_LIST_H_
#define KBH_TRAJECTORY_LIST_H_

#include <algorithm>
#ifdef _OPENMP
#include <parallel/algorithm>
#endif

#include "common.h"
#include "../search_math.h"
#include "image_utils.h"

namespace search {

class TrajectoryList {
public:
    explicit TrajectoryList(uint64_t max_list_size) {  // trajectory_list.cpp:18-28
        max_size = max_list_size;
        cpu_list.resize(max_size);
        reset_all();
    }
    explicit TrajectoryList(const std::vector<Trajectory>& prev_list) {  // :30-40
        max_size = prev_list.size();
        cpu_list = prev_list;
        assert_valid();
    }
    virtual ~TrajectoryList() { free_device(); }
    TrajectoryList(const TrajectoryList&) = delete;
    TrajectoryList& operator=(const TrajectoryList&) = delete;

    inline uint64_t get_size() const { return max_size; }
    inline uint64_t get_memory() const { return max_size * sizeof(Trajectory); }
    static uint64_t estimate_memory(uint64_t num_elements) { return num_elements * sizeof(Trajectory); }

    inline Trajectory& get_trajectory(uint64_t index) {
        if (index >= max_size) throw std::runtime_error("Index out of bounds.");
        host_only();
        return cpu_list[index];
    }
    inline std::vector<Trajectory>& get_list() {
        host_only();
        return cpu_list;
    }
    void reset_all() {  // :62-67
        host_only();
        for (uint64_t i = 0; i < max_size; ++i) cpu_list[i].clear();
    }
    inline void set_trajectory(uint64_t index, const Trajectory& new_value) {
        if (index >= max_size) throw std::runtime_error("Index out of bounds.");
        host_only();
        cpu_list[index] = new_value;
    }
    void set_trajectories(const std::vector<Trajectory>& new_values) {  // :69-80
        host_only();
        const uint64_t new_size = new_values.size();
        resize(new_size);
        for (uint64_t i = 0; i < new_size; ++i) cpu_list[i] = new_values[i];
        assert_valid();
    }
    void resize(uint64_t new_size) {  // :48-60 (vector::resize value-initialises == clear())
        host_only();
        cpu_list.resize(new_size);
        max_size = new_size;
    }
    std::vector<Trajectory> get_batch(uint64_t start, uint64_t count) {  // :82-94
        host_only();
        if (count == 0) throw std::runtime_error("count must be greater than 0");
        if (start >= max_size) return std::vector<Trajectory>();
        if (start + count >= max_size) return std::vector<Trajectory>(cpu_list.begin() + start, cpu_list.end());
        return std::vector<Trajectory>(cpu_list.begin() + start, cpu_list.begin() + start + count);
    }

    // :96-107.  The reference's sort is unstable; a stable sort is one of the
    // orders it may produce and makes results reproducible.
    void sort_by_likelihood() {
        host_only();
        auto cmp = [](const Trajectory& a, const Trajectory& b) { return b.lh < a.lh; };
#ifdef _OPENMP
        __gnu_parallel::stable_sort(cpu_list.begin(), cpu_list.end(), cmp);
#else
        std::stable_sort(cpu_list.begin(), cpu_list.end(), cmp);
#endif
    }
    void filter_by_likelihood(float min_lh) {  // :109-116
        drop_where([min_lh](const Trajectory& t) { return t.lh < min_lh; });
    }
    void filter_by_obs_count(int min_obs_count) {  // :118-126
        drop_where([min_obs_count](const Trajectory& t) { return t.obs_count < min_obs_count; });
    }

    inline bool on_gpu() const { return data_on_gpu; }
    void move_to_gpu() {  // :128-138
        if (data_on_gpu) return;
        if (!has_gpu()) throw std::runtime_error("GPU not available for TrajectoryList");
        free_device();
        if (max_size > 0) {
            check_status(kb_allocate_gpu_block(get_memory(), &gpu_ptr));
            check_status(kb_copy_block_to_gpu(cpu_list.data(), gpu_ptr, get_memory()));
        }
        data_on_gpu = true;
    }
    void move_to_cpu() {  // :140-153
        if (!data_on_gpu) return;
        if (max_size > 0) check_status(kb_copy_block_to_cpu(cpu_list.data(), gpu_ptr, get_memory()));
        free_device();
        data_on_gpu = false;
        assert_valid();
    }
    inline Trajectory* get_gpu_list_ptr() { return reinterpret_cast<Trajectory*>(gpu_ptr); }

    void assert_valid() const {  // :155-164
        host_only();
        // (the scan runs over millions of results behind every search: all cores, the lowest invalid index reported
        // like the reference's serial loop would)
        const int64_t n = (int64_t)cpu_list.size();
        int64_t first_bad = n;
        // (a small team: waking every hardware thread of a 256-thread host for 2 ms of work costs more than it returns)
#pragma omp parallel for reduction(min : first_bad) schedule(static) num_threads(8) if (n > (1 << 16))
        for (int64_t i = 0; i < n; ++i) {
            if (!cpu_list[i].is_valid() && i < first_bad) first_bad = i;
        }
        if (first_bad < n) {
            throw std::runtime_error("Invalid trajectory detected at index " + std::to_string(first_bad) + ": " +
                                     cpu_list[first_bad].to_string());
        }
    }

private:
    // The host vector is the list only while the data is not on the device (gpu_array.h state machine).
    void host_only() const {
        if (data_on_gpu) throw std::runtime_error("Data on GPU");
    }
    template <typename Pred>
    void drop_where(Pred reject) {  // order-preserving removal
        host_only();
        cpu_list.erase(std::remove_if(cpu_list.begin(), cpu_list.end(), reject), cpu_list.end());
        max_size = cpu_list.size();
    }
    void free_device() {
        if (gpu_ptr != nullptr) {
            (void)kb_free_gpu_block(gpu_ptr);
            gpu_ptr = nullptr;
        }
    }
    uint64_t max_size = 0;
    bool data_on_gpu = false;
    std::vector<Trajectory> cpu_list;
    void* gpu_ptr = nullptr;
};

// trajectory_list.cpp:171-239
#define KBH_EXTRACT(NAME, TYPE, FIELD)                                                       \
    inline std::vector<TYPE> extract_all_trajectory_##NAME(const std::vector<Trajectory>& t) { \
        std::vector<TYPE> result(t.size());                                                  \
        for (size_t i = 0; i < t.size(); ++i) result[i] = t[i].FIELD;                        \
        return result;                                                                       \
    }
KBH_EXTRACT(x, int, x)
KBH_EXTRACT(y, int, y)
KBH_EXTRACT(vx, float, vx)
KBH_EXTRACT(vy, float, vy)
KBH_EXTRACT(lh, float, lh)
KBH_EXTRACT(flux, float, flux)
KBH_EXTRACT(obs_count, int, obs_count)
#undef KBH_EXTRACT

// Host twin of the device merge kernel kb_merge_topk (search_kernels.hip): lists
// = [n_lists][n_pixels][K], each per-pixel list sorted descending by lh; ties go
// to the lower list, then the lower slot.
inline std::vector<Trajectory> merge_topk_host(const Trajectory* lists, int n_lists, uint64_t n_pixels, int K) {
    std::vector<Trajectory> out(n_pixels * (uint64_t)K);
    const uint64_t stride = n_pixels * (uint64_t)K;
    std::vector<int> head(n_lists);
    for (uint64_t pix = 0; pix < n_pixels; ++pix) {
        std::fill(head.begin(), head.end(), 0);
        for (int s = 0; s < K; ++s) {
            int best = -1;
            float best_lh = 0.0f;
            for (int r = 0; r < n_lists; ++r) {
                if (head[r] >= K) continue;
                const float lh = lists[(uint64_t)r * stride + pix * K + head[r]].lh;
                if (best < 0 || lh > best_lh) {
                    best = r;
                    best_lh = lh;
                }
            }
            out[pix * K + s] = lists[(uint64_t)best * stride + pix * K + head[best]];
            head[best] += 1;
        }
    }
    return out;
}

// Host twin of kb_merge_compact: the same merge on the 16-byte exchange records
// (kb_compact_result), producing full trajectories.  sw = width of the search area.
inline std::vector<Trajectory> merge_compact_host(const kb_compact_result* lists, int n_lists, uint64_t n_pixels, int K,
                                                  int sw, int x_min, int y_min, const Trajectory* all_cands,
                                                  uint64_t n_all_cands) {
    std::vector<Trajectory> out(n_pixels * (uint64_t)K);
    const uint64_t stride = n_pixels * (uint64_t)K;
    std::vector<int> head(n_lists);
    for (uint64_t pix = 0; pix < n_pixels; ++pix) {
        std::fill(head.begin(), head.end(), 0);
        const int y_i = (int)(pix / (uint64_t)sw), x_i = (int)(pix % (uint64_t)sw);
        for (int s = 0; s < K; ++s) {
            int best = -1;
            float best_lh = 0.0f;
            for (int r = 0; r < n_lists; ++r) {
                if (head[r] >= K) continue;
                const float lh = lists[(uint64_t)r * stride + pix * K + head[r]].lh;
                if (best < 0 || lh > best_lh) {
                    best = r;
                    best_lh = lh;
                }
            }
            const kb_compact_result rec = lists[(uint64_t)best * stride + pix * K + head[best]];
            head[best] += 1;
            Trajectory t;
            t.x = x_i + x_min;
            t.y = y_i + y_min;
            t.vx = 0.0f;
            t.vy = 0.0f;
            t.lh = -FLT_MAX;
            t.flux = 0.0f;
            t.obs_count = 0;
            if (rec.cand >= 0 && (uint64_t)rec.cand < n_all_cands) {
                t.vx = all_cands[rec.cand].vx;
                t.vy = all_cands[rec.cand].vy;
                t.lh = rec.lh;
                t.flux = rec.flux;
                t.obs_count = rec.obs_count;
            }
            out[pix * K + s] = t;
        }
    }
    return out;
}

// Host twin of kb_merge_compact_exact: per-device lists of list_len records per pixel built by stable insertion
// (flag 512) -> the K results per pixel that ONE device produces on the whole candidate list, ties included
// (kb::merge_exact_pixel of search_math.h, the routine the device kernel runs).
inline std::vector<Trajectory> merge_compact_exact_host(const kb_compact_result* lists, int n_lists, uint64_t n_pixels,
                                                        int list_len, int K, int sw, int x_min, int y_min,
                                                        const Trajectory* all_cands, uint64_t n_all_cands) {
    if (K <= 0 || list_len < std::max(K, 2 * K - 1) || list_len > kb::MERGE_EXACT_MAX_K2) {
        throw std::runtime_error("merge_compact_exact: need 2 K - 1 <= list length <= 32");
    }
    std::vector<Trajectory> out(n_pixels * (uint64_t)K);
    const uint64_t stride = n_pixels * (uint64_t)list_len;
    std::vector<int> heads(n_lists);
    kb::MergedEntry merged[kb::MERGE_EXACT_MAX_K2];
    int slots[kb::MERGE_EXACT_MAX_K2];
    for (uint64_t pix = 0; pix < n_pixels; ++pix) {
        const kb_compact_result* mine = lists + pix * (uint64_t)list_len;
        auto read = [&](int r, int pos) { return mine[(uint64_t)r * stride + pos]; };
        const int n_out = kb::merge_exact_pixel(read, n_lists, list_len, K, merged, heads.data(), slots);
        const int y_i = (int)(pix / (uint64_t)sw), x_i = (int)(pix % (uint64_t)sw);
        for (int s = 0; s < K; ++s) {
            Trajectory t;
            t.x = x_i + x_min;
            t.y = y_i + y_min;
            t.lh = -FLT_MAX;
            if (s < n_out && slots[s] >= 0) {
                const uint32_t at = merged[slots[s]].at;
                const kb_compact_result rec = read((int)(at / (uint32_t)list_len), (int)(at % (uint32_t)list_len));
                if ((uint64_t)rec.cand < n_all_cands) {
                    t.vx = all_cands[rec.cand].vx;
                    t.vy = all_cands[rec.cand].vy;
                    t.lh = rec.lh;
                    t.flux = rec.flux;
                    t.obs_count = rec.obs_count;
                }
            }
            out[pix * K + s] = t;
        }
    }
    return out;
}

// Host twin of kb_merge_compact_repairable: per-device lists of K records per pixel, each the reference's insertion over
// that device's slice of the candidates (contiguous slices in ascending order, no stable lists), folded in candidate order
// (kb::merge_fold_pixel of search_math.h, the routine the device kernel runs) -> the K results per pixel of ONE device on
// the whole candidate list wherever the records decide it, and the numbers of the pixels where they do not (`hazards`: a
// dropped candidate of some slice may tie with the last slot); the slots of a hazard are placeholders until the caller
// re-makes them (kb_repair_pixels).
inline std::vector<Trajectory> merge_compact_repairable_host(const kb_compact_result* lists, int n_lists, uint64_t n_pixels, int K,
                                                             int sw, int x_min, int y_min, const Trajectory* all_cands,
                                                             uint64_t n_all_cands, std::vector<uint32_t>& hazards) {
    if (K <= 0 || K > kb::MERGE_EXACT_MAX_K2) throw std::runtime_error("merge_compact_repairable: lists of 1 to 32 records per pixel");
    std::vector<Trajectory> out(n_pixels * (uint64_t)K);
    const uint64_t stride = n_pixels * (uint64_t)K;
    kb_compact_result state[kb::MERGE_EXACT_MAX_K2], recs[kb::MERGE_EXACT_MAX_K2];
    hazards.clear();
    for (uint64_t pix = 0; pix < n_pixels; ++pix) {
        const kb_compact_result* mine = lists + pix * (uint64_t)K;
        auto read = [&](int r, int pos) { return mine[(uint64_t)r * stride + pos]; };
        uint64_t suspects = 0;
        const bool hazard = kb::merge_fold_pixel(read, n_lists, K, state, recs, &suspects);
        if (hazard) hazards.push_back((uint32_t)pix);
        const int y_i = (int)(pix / (uint64_t)sw), x_i = (int)(pix % (uint64_t)sw);
        for (int s = 0; s < K; ++s) {
            Trajectory t;
            t.x = x_i + x_min;
            t.y = y_i + y_min;
            t.lh = -FLT_MAX;
            const kb_compact_result rec = state[s];
            if (!hazard && rec.cand >= 0 && (uint64_t)rec.cand < n_all_cands) {
                t.vx = all_cands[rec.cand].vx;
                t.vy = all_cands[rec.cand].vy;
                t.lh = rec.lh;
                t.flux = rec.flux;
                t.obs_count = rec.obs_count;
            }
            out[pix * K + s] = t;
        }
    }
    return out;
}

// Host twins of kb_sparsify_compact / kb_merge_sparse_exact (csrc/exchange_kernels.hip): the sparse form of the
// exchange lists -- per pixel the number of records that survive the reference's post-filter on the likelihood
// (stack_search.cpp:266-270: lh < min_lh goes; empty slots carry cand = -1) + those records, pixel after pixel.
inline uint64_t sparse_header_bytes(uint64_t n_pixels) { return (n_pixels + 15) / 16 * 16 + 16; }

inline void sparsify_compact_host(const kb_compact_result* lists, uint64_t n_pixels, int list_len, float min_lh,
                                  std::vector<uint8_t>& header, std::vector<kb_compact_result>& packed) {
    if (list_len <= 0 || list_len > kb::MERGE_EXACT_MAX_K2) throw std::runtime_error("sparsify_compact: lists of 1 to 32 records per pixel");
    header.assign(sparse_header_bytes(n_pixels), 0);
    packed.clear();
    for (uint64_t pix = 0; pix < n_pixels; ++pix) {
        int kept = 0;
        for (int p = 0; p < list_len; ++p) {
            const kb_compact_result& r = lists[pix * (uint64_t)list_len + p];
            if (r.cand >= 0 && !(r.lh < min_lh)) {
                packed.push_back(r);
                kept += 1;
            }
        }
        header[pix] = (uint8_t)kept;
    }
    const uint64_t total = packed.size();
    std::memcpy(header.data() + (n_pixels + 15) / 16 * 16, &total, sizeof(uint64_t));
}

inline std::vector<Trajectory> merge_sparse_exact_host(const uint8_t* headers, uint64_t header_stride,
                                                       const std::vector<const kb_compact_result*>& packed, uint64_t n_pixels,
                                                       int list_len, int K, int sw, int x_min, int y_min,
                                                       const Trajectory* all_cands, uint64_t n_all_cands) {
    const int n_lists = (int)packed.size();
    if (K <= 0 || list_len < std::max(K, 2 * K - 1) || list_len > kb::MERGE_EXACT_MAX_K2) {
        throw std::runtime_error("merge_sparse_exact: need 2 K - 1 <= list length <= 32");
    }
    if (header_stride < sparse_header_bytes(n_pixels)) throw std::runtime_error("merge_sparse_exact: header stride shorter than a header");
    std::vector<Trajectory> out(n_pixels * (uint64_t)K);
    std::vector<uint64_t> at(n_lists, 0);  // first record of the current pixel in every list
    std::vector<int> heads(n_lists);
    kb::MergedEntry merged[kb::MERGE_EXACT_MAX_K2];
    int slots[kb::MERGE_EXACT_MAX_K2];
    for (uint64_t pix = 0; pix < n_pixels; ++pix) {
        auto read = [&](int r, int pos) {
            kb_compact_result rec{-FLT_MAX, 0.0f, -1, 0};
            if (pos < (int)headers[(uint64_t)r * header_stride + pix]) rec = packed[r][at[r] + (uint64_t)pos];
            return rec;
        };
        const int n_out = kb::merge_exact_pixel(read, n_lists, list_len, K, merged, heads.data(), slots);
        const int y_i = (int)(pix / (uint64_t)sw), x_i = (int)(pix % (uint64_t)sw);
        for (int s = 0; s < K; ++s) {
            Trajectory t;
            t.x = x_i + x_min;
            t.y = y_i + y_min;
            t.lh = -FLT_MAX;
            if (s < n_out && slots[s] >= 0) {
                const uint32_t where = merged[slots[s]].at;
                const kb_compact_result rec = read((int)(where / (uint32_t)list_len), (int)(where % (uint32_t)list_len));
                if ((uint64_t)rec.cand < n_all_cands) {
                    t.vx = all_cands[rec.cand].vx;
                    t.vy = all_cands[rec.cand].vy;
                    t.lh = rec.lh;
                    t.flux = rec.flux;
                    t.obs_count = rec.obs_count;
                }
            }
            out[pix * K + s] = t;
        }
        for (int r = 0; r < n_lists; ++r) at[r] += headers[(uint64_t)r * header_stride + pix];
    }
    return out;
}

}  // namespace search
#endif
