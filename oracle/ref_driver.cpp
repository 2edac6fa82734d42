/*
 * ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * C-ABI shims around the parts of the REAL reference that compile from their
 * own sources in this image with plain g++ (no Eigen, no CUDA, no stand-ins):
 *   common.h            Trajectory (get_*_pos / get_*_index / is_valid), sizeof
 *   psi_phi_array_ds.h  encode_uint_scalar / decode_uint_scalar (header inline)
 *   trajectory_list.cpp TrajectoryList sort / filter / get_batch
 *   kernel_helpers.cpp  has_gpu() == false without HAVE_CUDA
 * The reference files are #included from where they lie under /root/reference
 * (the Makefile passes -I); nothing of them is copied into this repository and
 * the built library lives only in oracle/_ref/ (git-ignored).
 *
 * NOT buildable here, hence not wrapped: psi_phi_array.cpp, image_utils_cpp.cpp and
 * stack_search.cpp include <Eigen/Core> (absent: include/eigen is an empty submodule);
 * cpu_search_algorithms.cpp parses on its own (g++ -fsyntax-only) but cannot link
 * without PsiPhiArray's members, which live in psi_phi_array.cpp; kernels/ *.cu need
 * the CUDA toolkit (absent).
 */
#include <cstdint>
#include <cstring>
#include <vector>

#include "logging.h"
#include "common.h"
#include "psi_phi_array_ds.h"
#include "kernel_helpers.cpp"
#include "trajectory_list.cpp"

using search::Trajectory;
using search::TrajectoryList;

extern "C" {

int ref_sizeof_trajectory() { return (int)sizeof(Trajectory); }
int ref_has_gpu() { return search::has_gpu() ? 1 : 0; }

float ref_encode_uint_scalar(float v, float mn, float mx, float sc) {
    return search::encode_uint_scalar(v, mn, mx, sc);
}
float ref_decode_uint_scalar(float v, float mn, float sc) { return search::decode_uint_scalar(v, mn, sc); }

float ref_get_x_pos(const Trajectory* t, double time, int centered) { return t->get_x_pos(time, centered != 0); }
float ref_get_y_pos(const Trajectory* t, double time, int centered) { return t->get_y_pos(time, centered != 0); }
int ref_get_x_index(const Trajectory* t, double time) { return t->get_x_index(time); }
int ref_get_y_index(const Trajectory* t, double time) { return t->get_y_index(time); }
int ref_is_valid(const Trajectory* t) { return t->is_valid() ? 1 : 0; }

/* filter_by_likelihood, filter_by_obs_count, sort_by_likelihood applied in the
 * order stack_search.cpp:268-277 uses; pass do_* = 0 to skip a stage.
 * Returns the new length, list written back in place. */
uint64_t ref_list_filter_sort(Trajectory* data, uint64_t n, int do_lh, float min_lh, int do_obs, int min_obs,
                              int do_sort) {
    std::vector<Trajectory> v(data, data + n);
    TrajectoryList lst(v);
    if (do_lh) lst.filter_by_likelihood(min_lh);
    if (do_obs) lst.filter_by_obs_count(min_obs);
    if (do_sort) lst.sort_by_likelihood();
    const uint64_t m = lst.get_size();
    if (m > 0) std::memcpy(data, lst.get_list().data(), m * sizeof(Trajectory));
    return m;
}

/* get_batch clamping (trajectory_list.cpp:85-94). Returns the batch length, or
 * -1 if the reference threw. */
int64_t ref_list_get_batch(const Trajectory* data, uint64_t n, uint64_t start, uint64_t count, Trajectory* out) {
    try {
        std::vector<Trajectory> v(data, data + n);
        TrajectoryList lst(v);
        std::vector<Trajectory> b = lst.get_batch(start, count);
        if (!b.empty()) std::memcpy(out, b.data(), b.size() * sizeof(Trajectory));
        return (int64_t)b.size();
    } catch (const std::runtime_error&) {
        return -1;
    }
}

}  // extern "C"
