/*
 * ref_kernels_driver.cpp -- TEST INFRASTRUCTURE ONLY, and CORROBORATION, not the formal pin of the oracle.
 *
 * The reference's device evaluator -- kernels/kernels.cu: predict_index, read_encoded_psi_phi, SigmaGFilteredIndicesCU,
 * evaluateTrajectory (:31-242) -- is written as `__host__ __device__` code.  Compiled as HIP source by `hipcc -x hip`
 * from where it lies under /root/reference, its HOST instantiation runs on this container's CPU (x86-64 baseline, no FMA:
 * the arithmetic of the reference's own host build).  Nothing of the reference is copied; the built library lives only in
 * oracle/_ref/ (git-ignored).
 *
 * What stands in for something the image lacks, said plainly: ONE macro.  kernels.cu:396 calls cudaDeviceSynchronize() in
 * deviceSearchFilter (a launcher this driver never calls); the CUDA runtime is absent, so the name is mapped onto the HIP
 * runtime's hipDeviceSynchronize below.  The `<<< >>>` launch and the __global__ kernel in the same file compile as HIP.
 * Because of that one stand-in this build does not upgrade the oracle's formal pin (tests/test_oracle_kat.py, the
 * reference's own known answers, does that); it is a fuzz cross-check: tests/golden/make_evaluator_fuzz.py runs 10^4 random
 * trajectories through it and stores inputs and outputs, tests/test_evaluator_fuzz.py holds the oracle to them bit for bit.
 */
#include <hip/hip_runtime.h>
#define cudaDeviceSynchronize hipDeviceSynchronize

#include <algorithm>
#include <cstdint>
#include <iostream>
#include <sstream>

#include "logging.h"
#include "kernels/kernels.cu"
#include "kernel_helpers.cpp"
#include "trajectory_list.cpp"

extern "C" {

int refk_sizes(int* out) {  /* layouts the Python side mirrors with ctypes */
    out[0] = (int)sizeof(search::PsiPhiArrayMeta);
    out[1] = (int)sizeof(search::SearchParameters);
    out[2] = (int)sizeof(search::Trajectory);
    out[3] = (int)search::MAX_NUM_IMAGES;
    return 0;
}

/* kernels.cu:154-242 over n trajectories (x, y, vx, vy read; lh, flux, obs_count written) */
void refk_evaluate(const search::PsiPhiArrayMeta* meta, void* psi_phi, double* times, const search::SearchParameters* params,
                   search::Trajectory* trjs, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) search::evaluateTrajectory(*meta, psi_phi, times, *params, &trjs[i]);
}

/* kernels.cu:77-147 */
void refk_sigmag(float* values, int n, float sgl0, float sgl1, float coeff, float width, int* idx, int* lo, int* hi) {
    search::SigmaGFilteredIndicesCU(values, n, sgl0, sgl1, coeff, width, idx, lo, hi);
}

}  // extern "C"
