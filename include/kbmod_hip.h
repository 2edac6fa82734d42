/*
 * kbmod_hip.h -- C ABI of libkbmod_hip.so, the MI355X (gfx950) device library
 * behind the shift-and-stack search path.
 *
 * This is the drop-in boundary for the reference's host<->device seam: every
 * entry point below names the reference interface it replaces (file:line
 * relative to /root/reference/src/kbmod/search/).  Plain pointers and sizes
 * only; no C++ types, no torch types.  All functions that can fail return a
 * status (0 = ok) and leave a message retrievable with kb_last_error(); the
 * host layer turns a non-zero status into std::runtime_error (the reference's
 * convention: kernels/kernel_memory.cu:93-136 throw std::runtime_error).
 *
 * Memory spaces are spelled out in the parameter names: *_host / *_dev.
 */
#ifndef KBMOD_HIP_H_
#define KBMOD_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The search refuses stacks deeper than this (the reference's device limit is
 * MAX_NUM_IMAGES = 200, common.h:31; its tests require an error at T = 1000,
 * tests/test_search.py:279-304; BASELINE config 5 needs T = 512). */
#define KB_MAX_NUM_IMAGES 999

/* common.h:55-68 -- 28-byte POD, same field order as search::Trajectory. */
typedef struct kb_trajectory {
    float vx, vy, lh, flux;
    int32_t x, y, obs_count;
} kb_trajectory;

/* common.h:119-143 -- same field order and types as search::SearchParameters. */
typedef struct kb_search_params {
    int32_t min_observations;
    float min_lh;
    uint8_t do_sigmag_filter; /* bool */
    float sgl_L, sgl_H, sigmag_coeff;
    int32_t encode_num_bytes; /* -1, 1 or 2 */
    int32_t x_start_min, x_start_max, y_start_min, y_start_max;
    uint32_t results_per_pixel;
    unsigned long long total_results;
} kb_search_params;

/* psi_phi_array_ds.h:50-69 -- same field order as search::PsiPhiArrayMeta. */
typedef struct kb_psi_phi_meta {
    uint64_t num_times, width, height, pixels_per_image, num_entries, block_size, total_array_size;
    int32_t num_bytes; /* 1, 2 or 4 */
    float psi_min_val, psi_max_val, psi_scale;
    float phi_min_val, phi_max_val, phi_scale;
} kb_psi_phi_meta;

/* Optional per-call measurements, filled when a non-NULL pointer is passed. */
typedef struct kb_search_stats {
    float search_kernel_ms;   /* HIP-event time of the search kernel(s) on the launch stream */
    float table_kernel_ms;    /* HIP-event time of the shift-table kernel */
    uint64_t num_evals;       /* S * N_c * T */
    uint64_t algorithmic_bytes; /* num_evals*2*bs + S*K*28 + N_c*28 + T*8 */
    int32_t kernel_variant;   /* which template instance ran */
    int32_t num_search_launches;
    uint64_t sigmag_work_items;   /* in-search sigma-G: (row of 64 start pixels, candidate) pairs with a trajectory to clip */
    uint64_t sigmag_trajectories; /* ... and the trajectories clipped */
    uint64_t lds_read_bytes;      /* kb_search_lds: bytes the sums read out of LDS (num_evals x staged pair size), else 0 */
    uint64_t sigmag_literal;      /* ... of the clipped trajectories, those that took the literal per-lane exchange sort
                                     (equal ratios from different (psi, phi) pairs, stacks deeper than 256 epochs) */
    char kernel_name[96];         /* the search kernel instance that ran, spelled as rocprofv3 prints it
                                     (e.g. "kb::kb_search_lds<8, 8, 16, 4, true, false, 3, 1>") */
    int32_t padded_copy_reused;   /* the padded copy of an earlier search was reused (an array the library built and nobody has
                                     written since, or flag 256): no decode-and-pad pass in this search */
    int32_t special_epochs;       /* kb_search_lds: (chunk of candidates, epoch) pairs summed per lane -- not staged, or a shift
                                     inside the guard band of a rounding boundary -- instead of by the uniform loops */
    int32_t edge_count_tables;    /* kb_search_lds: tables of epochs per shift were built, so that tiles at the image's edge of a
                                     stack without NO_DATA pixels take their observation counts from them instead of counting samples */
    int32_t env_overrides;        /* bit i: environment switch i was set while this search chose its kernels (they exist for tests
                                     and comparisons and change the kernel instance, never the result): 0 KBMOD_CHUNK (8 / 16 / 32
                                     candidates per staged slab; 32 asks for the instance the library otherwise takes only for
                                     arrays beyond the Infinity Cache with 192 epochs or more), 1 KBMOD_LIST_MODE,
                                     2 KBMOD_EDGE_COUNTS, 3 KBMOD_UNSTAGED_LIMIT, 4 KBMOD_SIGMAG_CAP, 5 KBMOD_DEBUG */
} kb_search_stats;

const char* kb_last_error(void);

/* ---- device / memory helpers: kernels/kernel_memory.cu:15-136 ------------ */
int kb_device_count(void);                  /* cuda_device_count      :15 */
void kb_print_stats(void);                  /* cuda_print_stats       :23 */
size_t kb_gpu_total_memory(void);           /* gpu_total_memory       :50 */
size_t kb_gpu_free_memory(void);            /* gpu_free_memory        :60 */
int kb_check_gpu(size_t req_memory);        /* cuda_check_gpu         :70  (1 = ok) */
int kb_allocate_gpu_block(uint64_t memory_size, void** out_dev);                     /* :93  */
int kb_free_gpu_block(void* ptr_dev);                                                /* :105 */
int kb_copy_block_to_gpu(const void* src_host, void* dst_dev, uint64_t memory_size); /* :112 */
int kb_copy_block_to_cpu(void* dst_host, const void* src_dev, uint64_t memory_size); /* :124 */
/* new: kb_copy_block_to_cpu for large result blocks -- the (pageable) destination is page-locked for the duration of the
 * copy, one DMA at the link's rate; falls back to the plain copy where the memory cannot be registered. */
int kb_copy_block_to_cpu_locked(void* dst_host, const void* src_dev, uint64_t memory_size);
int kb_device_synchronize(void);
/* new (multi-GPU fan-out inside one process): the calling thread's current device, and a copy between two devices'
 * HBM (over xGMI where the devices are peers).  Every entry point acts on the calling thread's current device; the
 * library keeps its workspaces per device. */
int kb_get_device(void);           /* -1 without a device */
int kb_set_device(int32_t device);
int kb_copy_block_between_gpus(void* dst_dev, int32_t dst_device, const void* src_dev, int32_t src_device,
                               uint64_t memory_size);
/* new: streaming device-to-device copy of `bytes`, `iters` timed passes; *gbps_out = (read + written bytes) / time.
 * The measured HBM peak of the roofline report (bench.py). */
int kb_measure_copy_bandwidth(uint64_t bytes, int32_t iters, void* stream, double* gbps_out);
/* new: read-only stream over a block of `bytes`; *gbps_out = bytes read / time.  With a block that fits the 256 MiB
 * Infinity Cache this is the rate at which that cache feeds the L2s (the ceiling of the cfg2 search, whose 134 MB
 * array stays resident there). */
int kb_measure_read_bandwidth(uint64_t bytes, int32_t iters, void* stream, double* gbps_out);
/* Aggregate LDS read rate of the device (new): every CU streams ds_read_b64 -- the read mix of kb_search_lds's summing
 * loop -- for `iters` rounds of eight reads per wave; *gbps_out = bytes read out of LDS / time.  The yardstick of the
 * search kernel's LDS traffic in bench.py's roofline block. */
int kb_measure_lds_bandwidth(int32_t iters, void* stream, double* gbps_out);

/* ---- PSF convolution: kernels/image_kernels.cu:68-108 (deviceConvolve) --- */
/* Host image in, host image out, one image.  empty_is_nan = 0 reproduces the
 * reference device kernel (empty PSF footprint -> 0.0, image_kernels.cu:61);
 * 1 gives the reference CPU value (NaN, image_utils_cpp.cpp:60-61). */
int kb_device_convolve(const float* src_host, float* dst_host, int width, int height, const float* psf_host,
                       int psf_radius, int empty_is_nan);

/* ---- psi/phi builder: replaces the per-image generate_psi / generate_phi /
 * deviceConvolve loop + fill_psi_phi_array of psi_phi_array.cpp:321-410 with
 * one fused device pass that leaves the encoded array resident in HBM. ------ */
/* sci_dev / var_dev: [T][H][W] float32 in device memory.  psf_host: the T PSF
 * kernels concatenated (kernel t is psf_dims[t] x psf_dims[t], row-major).
 * num_bytes: -1/4 float, 1 uint8, 2 uint16.  On success *meta_out is filled
 * (incl. min/max/scale) and *psi_phi_dev_out owns total_array_size bytes
 * (release with kb_free_gpu_block). */
int kb_build_psi_phi_from_device(const float* sci_dev, const float* var_dev, const float* psf_host,
                                 const int32_t* psf_dims, int32_t num_times, int32_t height, int32_t width,
                                 int32_t num_bytes, kb_psi_phi_meta* meta_out, void** psi_phi_dev_out,
                                 void* stream);
/* The same build with options (new).  build_flags:
 *   KB_BUILD_SEPARABLE      rank-1 PSF kernels (every Gaussian: core/psf.py:49-74) are correlated as a row pass and
 *                           a column pass over the masked image and the mask (kb_conv_sep_kernel).  Agrees with the
 *                           default 2-D kernel to rounding (1e-4 relative), not bit for bit; a stack with any kernel
 *                           that does not factor takes the 2-D kernel;
 *   KB_BUILD_EMPTY_IS_ZERO  a valid centre whose PSF footprint holds no valid pixel gives 0.0, as the reference's own
 *                           device builder does (image_kernels.cu:61); default NaN, the reference CPU value
 *                           (image_utils_cpp.cpp:60-61), which the CPU StackSearch and the parity tests assume.
 *   KB_BUILD_GENERAL_TILES  (diagnostic) always take the general 32 x 8 tile kernel (mixed kernel sizes, sizes beyond
 *                           9 x 9) instead of the strip kernel that stacks with one kernel size 3 .. 9 get; same bits.
 *   KB_BUILD_REGISTER_HOST  (kb_build_psi_phi_from_host_stack) page-lock the caller's stacks for the duration of the
 *                           build (hipHostRegister) so that every chunk is one DMA out of the caller's memory; stacks
 *                           that already are page-locked (pinned allocations) are sent that way without the flag,
 *                           pageable ones otherwise go through two internal pinned buffers. */
enum { KB_BUILD_SEPARABLE = 1, KB_BUILD_EMPTY_IS_ZERO = 2, KB_BUILD_GENERAL_TILES = 4, KB_BUILD_REGISTER_HOST = 8 };
/* HIP-event time (ms) of the correlation kernel launch of the calling thread's last kb_build_psi_phi_from_device[_ex]
 * (the measurement bench.py reports for the builder; 0 before the first build). */
float kb_last_build_kernel_ms(void);
int kb_build_psi_phi_from_device_ex(const float* sci_dev, const float* var_dev, const float* psf_host,
                                    const int32_t* psf_dims, int32_t num_times, int32_t height, int32_t width,
                                    int32_t num_bytes, uint32_t build_flags, kb_psi_phi_meta* meta_out,
                                    void** psi_phi_dev_out, void* stream);
/* From contiguous host stacks sci_host / var_host = [T][H][W] float32 (new; the ingest path, SURVEY 8(f4): what
 * WorkUnit.from_fits leaves in memory, work_unit.py:489-608): uploaded in chunks of epochs through pinned staging
 * buffers on a copy stream, each chunk correlated behind its arrival. */
int kb_build_psi_phi_from_host_stack(const float* sci_host, const float* var_host, const float* psf_host,
                                     const int32_t* psf_dims, int32_t num_times, int32_t height, int32_t width,
                                     int32_t num_bytes, uint32_t build_flags, kb_psi_phi_meta* meta_out,
                                     void** psi_phi_dev_out);
/* Same as kb_build_psi_phi_from_device, from separate host images (sci_host[t] / var_host[t] are H*W float32 each). */
int kb_build_psi_phi_from_host(const float* const* sci_host, const float* const* var_host,
                               const float* psf_host, const int32_t* psf_dims, int32_t num_times,
                               int32_t height, int32_t width, int32_t num_bytes, kb_psi_phi_meta* meta_out,
                               void** psi_phi_dev_out);
/* psi-only / phi-only images for one epoch (generate_psi / generate_phi,
 * image_utils_cpp.cpp:126-177), host in / host out. */
int kb_generate_psi_phi_host(const float* sci_host, const float* var_host, int width, int height,
                             const float* psf_host, int psf_dim, float* psi_out_host, float* phi_out_host);

/* ---- the search: kernels/kernels.cu:334-397 (deviceSearchFilter) --------- */
/* psi_phi_dev: encoded [t][row][col][psi,phi] array (meta->total_array_size
 * bytes); times_dev: double[T]; cands_dev: n_cands trajectories (vx, vy read);
 * results_dev: at least K*search_w*search_h trajectories, fully overwritten
 * (slot layout kernels.cu:286: ((y-y_min)*search_w + (x-x_min))*K + s).
 * flags (0 = let the library choose; results are identical for every combination):
 *   1  force the per-lane exact-position path of kb_search_direct (debug / self-check);
 *   2  use kb_search_direct (every sample a wave-wide load from the array);
 *   4  use kb_search_lds (slabs staged once per workgroup into LDS from a padded copy of the
 *      array) even for fewer than 8 candidates; the default from 8 candidates (one full chunk) on.  Falls back
 *      to kb_search_direct when more than 10 % of the (chunk, epoch) footprints cannot be staged,
 *      when K > 32, or when the apron of the padded copy would outweigh the image;
 *   8  always decode uint8/uint16 samples in double (skip the verified fp32-FMA form);
 *  16  keep an encoded array encoded in the padded copy (default: canonical floats when HBM
 *      has room, which makes the search as fast as on a float array);
 *  64 / 128  kb_search_lds with 64 x 16 / 64 x 8 start-pixel tiles whatever the search area (default:
 *      64 x 16 for lists of up to 8 results per pixel when that still gives >= 128 tiles).
 * 256  the caller vouches that the array at psi_phi_dev has not changed since its previous search on this
 *      device: the padded copy of that search is reused when array, meta data and frame geometry are the same
 *      (a StackSearch owns its array and sets this from its second search on; cfg4: 7 ms of 60 per search).
 * 512  (changes the result under ties) per-pixel lists by stable insertion -- the top K by (likelihood descending,
 *      candidate ascending) instead of the reference's swap-down order: the per-device half of the tie-exact
 *      multi-GPU exchange (kb_merge_compact_exact), not a search result of its own.
 *      An array the library built itself (kb_build_psi_phi_*: it allocated it) needs no vouching: its copy stands until
 *      something writes into the array or it is freed -- the canonical frame is made once per array, by its first search,
 *      not once per search.  The library's own writers (kb_copy_block_to_gpu, kb_copy_block_between_gpus) say so
 *      themselves; a caller that stores into such an array by ANY other means (its own kernel, hipMemcpy, a tensor view)
 *      must call kb_note_array_written(psi_phi_dev) afterwards, and must release the array with kb_free_gpu_block and
 *      nothing else -- the library knows the array by its address.
 * 2048 never reuse a padded copy (every search re-makes it: measurements of the pad pass, tests).
 * 1024 the caller drops every result below params.min_lh afterwards (the reference does, stack_search.cpp:266-270; the
 *      sparse exchange kb_sparsify_compact does): the kernels then need not insert such candidates.  The slots at or
 *      above min_lh are exactly those of the default -- the reference's insertion never lets a smaller likelihood touch
 *      the part of a list at or above a larger one --; which entries below min_lh a list still shows is unspecified.
 *      (With the sigma-G filter the kernel tests min_lh itself, kernels.cu:318-320; the flag changes nothing there.)
 * The library keeps its workspaces (shift tables, sigma-G scratch, padded copy) between
 * calls; kb_release_workspaces() returns them. */
int kb_device_search_filter(const kb_psi_phi_meta* meta, const void* psi_phi_dev, const double* times_dev,
                            kb_search_params params, const kb_trajectory* cands_dev, uint64_t n_cands,
                            kb_trajectory* results_dev, uint64_t n_results, uint32_t flags, void* stream,
                            kb_search_stats* stats_out);

int kb_release_workspaces(void);
/* new: the caller wrote into a device block by means the library cannot see.  If ptr_dev lies inside an array built by
 * kb_build_psi_phi_*, every padded copy made of that array (on any device) is stale from now on and the next search
 * re-makes it; any other pointer is ignored.  Returns 0. */
int kb_note_array_written(const void* ptr_dev);

/* ---- the same search with 16-byte result records (new; multi-GPU exchange format).  x / y follow from the
 * slot, vx / vy from the candidate: a record carries (lh, flux, candidate index, obs_count), 16 instead of
 * 28 bytes per slot on the wire.  cand holds cand_index_base + the index into cands_dev (the index into
 * the job-wide candidate list when each GPU searches a contiguous slice); empty slots carry
 * cand = -1, lh = -FLT_MAX.  results_per_pixel <= 32. */
typedef struct kb_compact_result {
    float lh, flux;
    int32_t cand, obs_count;
} kb_compact_result;
int kb_device_search_compact(const kb_psi_phi_meta* meta, const void* psi_phi_dev, const double* times_dev,
                             kb_search_params params, const kb_trajectory* cands_dev, uint64_t n_cands,
                             int32_t cand_index_base, kb_compact_result* results_dev, uint64_t n_results, uint32_t flags,
                             void* stream, kb_search_stats* stats_out);

/* ---- result post-processing in HBM: the filter_by_likelihood / filter_by_obs_count /
 * sort_by_likelihood sequence of stack_search.cpp:266-281 (trajectory_list.cpp:96-126).
 * results_dev: n trajectories (the slots written by kb_device_search_filter); out_dev: room for n.
 * Keeps entries with !(lh < min_lh) && !(obs_count < min_obs) in slot order, then sorts them by lh
 * descending (stable).  *n_out_host receives the number kept.  Synchronises the stream. */
int kb_filter_sort_results(const kb_trajectory* results_dev, uint64_t n, float min_lh, int32_t min_obs,
                           kb_trajectory* out_dev, uint64_t* n_out_host, void* stream);

/* The same, and the validity scan of TrajectoryList::assert_valid (trajectory_list.cpp:155-164, called behind every search:
 * stack_search.cpp:280) done while the survivors are gathered: *first_invalid_host = the lowest index in out_dev whose
 * record has a non-finite vx / vy / lh / flux or a negative obs_count (Trajectory::is_valid, common.h), -1 when all are
 * valid -- 2 M results: one more compare per record on the device instead of a pass over 58 MB on the host. */
int kb_filter_sort_results_checked(const kb_trajectory* results_dev, uint64_t n, float min_lh, int32_t min_obs,
                                   kb_trajectory* out_dev, uint64_t* n_out_host, int64_t* first_invalid_host, void* stream);

/* ---- psi/phi curves of result trajectories (StackSearch::get_all_psi_phi_curves,
 * stack_search.cpp:302-318): out_dev is [n][2*T] float32, psi in [0,T), phi in [T,2T), non-finite -> 0.
 * Synchronises the stream. */
int kb_psi_phi_curves(const kb_psi_phi_meta* meta, const void* psi_phi_dev, const double* times_dev,
                      const kb_trajectory* trjs_dev, uint64_t n, float* out_dev, void* stream);

/* ---- near-duplicate grid filter (apply_trajectory_grid_filter, src/kbmod/filters/clustering_grid.py:152-175,
 * called from run_search.py:294-301): trajectories that fall into the same (start bin, end bin at
 * max_time) of width bin_width are duplicates; the one with the largest lh survives (the earliest of
 * equals).  kept_idx_dev: room for n indices; receives the survivors' indices in the order in which
 * their bins first occur in the input; *n_kept_host their number.  Synchronises the stream. */
int kb_grid_filter(const kb_trajectory* trjs_dev, uint64_t n, double bin_width, double max_time, uint32_t* kept_idx_dev,
                   uint64_t* n_kept_host, void* stream);

/* ---- batched sigma-G clipping of likelihood curves (SigmaGClipping.compute_clipped_sigma_g_matrix,
 * src/kbmod/filters/sigma_g_filter.py:114-168; the step after get_all_psi_phi_curves in
 * SearchRunner.load_and_filter_results, run_search.py:251-337).  lh: [n_rows][n_cols] float32 (NaN =
 * masked point); valid: [n_rows][n_cols] bytes, 1 = inside median -+ n_sigma * coeff * (q_high - q_low),
 * quantiles by linear interpolation over the non-NaN (and, with clip_negative, positive) points.
 * low_pct / high_pct on the reference's [0, 100] scale; n_cols <= 4096.  Synchronises the stream. */
int kb_sigma_g_clip_matrix(const float* lh_dev, uint64_t n_rows, int32_t n_cols, float low_pct, float high_pct,
                           float n_sigma, float coeff, int32_t clip_negative, uint8_t* valid_dev, void* stream);
/* same on host buffers (upload, clip, download) */
int kb_sigma_g_clip_matrix_host(const float* lh_host, uint64_t n_rows, int32_t n_cols, float low_pct, float high_pct,
                                float n_sigma, float coeff, int32_t clip_negative, uint8_t* valid_host);

/* ---- stamp coadds of result trajectories (append_coadds, src/kbmod/filters/stamp_filters.py:72-168:
 * extract_stamp_stack + coadd_sum / coadd_mean / coadd_median / coadd_weighted of
 * src/kbmod/core/stamp_utils.py:16-84, 241-344, 352-397).  sci_dev / var_dev: [T][H][W] float32 image
 * stacks in HBM (var_dev only for KB_COADD_WEIGHTED); x_dev / y_dev: [n][T] integer stamp centres
 * (predict_pixel_locations, trajectory_utils.py:28-75); include_dev: [n][T] bytes, 0 = epoch not used
 * (obs_valid), or NULL for all epochs; out_dev: [n][2r+1][2r+1] float32.  Accumulates in double in
 * epoch order like the reference and rounds once.  Synchronises the stream. */
enum { KB_COADD_SUM = 0, KB_COADD_MEAN = 1, KB_COADD_MEDIAN = 2, KB_COADD_WEIGHTED = 3 };
int kb_coadd_stamps(const float* sci_dev, const float* var_dev, int32_t num_times, int32_t height, int32_t width,
                    const int32_t* x_dev, const int32_t* y_dev, const uint8_t* include_dev, uint64_t n, int32_t radius,
                    int32_t coadd_type, float* out_dev, void* stream);

/* All stamps of result trajectories (append_all_stamps, src/kbmod/filters/stamp_filters.py:171-211, a loop of
 * extract_stamp_stack over the results): out_dev is [n][T][2r+1][2r+1] float32, NaN outside the image.
 * Synchronises the stream. */
int kb_extract_stamps(const float* sci_dev, int32_t num_times, int32_t height, int32_t width, const int32_t* x_dev,
                      const int32_t* y_dev, uint64_t n, int32_t radius, float* out_dev, void* stream);

/* ---- multi-GPU: merge of per-rank top-K lists (new; the reference is single-GPU).
 * lists_dev: [n_lists][n_pixels][K] as gathered by one RCCL all_gather of each
 * rank's kb_device_search_filter output over its candidate slice; out_dev:
 * [n_pixels][K].  Ties go to the lower list, then the lower slot. */
int kb_merge_topk(const kb_trajectory* lists_dev, int32_t n_lists, uint64_t n_pixels, int32_t K,
                  kb_trajectory* out_dev, void* stream);
/* Same merge on the compact records of kb_device_search_compact, gathered to one GPU as
 * lists_dev = [n_lists][n_pixels][K]; writes full trajectories (x, y from the slot and the start bounds of
 * params, vx / vy from all_cands_dev, the job-wide candidate list the records index).  Where a pixel's
 * K + 1 best likelihoods are distinct the output equals what one GPU produces on the whole
 * candidate list; among EQUAL likelihoods it keeps the lower list = lower candidate index, whereas the
 * reference's swap-down insertion (kernels.cu:323-330) may keep another member of the tie
 * (kb_merge_compact_exact reproduces that too). */
int kb_merge_compact(const kb_compact_result* lists_dev, int32_t n_lists, kb_search_params params,
                     const kb_trajectory* all_cands_dev, uint64_t n_all_cands, kb_trajectory* out_dev, void* stream);
/* Tie-exact merge (new): the output equals what ONE device produces on the whole candidate list, ties included.
 * Every device searches its candidates with flag 512 (per-pixel lists by STABLE insertion: the top list_len by
 * likelihood descending, candidate ascending) and list_len = 2 * params.results_per_pixel slots per pixel
 * (params.results_per_pixel of ITS search = list_len); lists_dev = [n_lists][n_pixels][list_len] gathered on one
 * device, params.results_per_pixel here = the K of the job.  The lists merge exactly under that total order, and the
 * reference's insertion is then replayed per pixel over the only candidates its order-dependence can reach (all above
 * the K-th likelihood + the first K equal to it).  list_len <= 32, i.e. K <= 16; any candidate partition works
 * (contiguous slices or interleaved). */
int kb_merge_compact_exact(const kb_compact_result* lists_dev, int32_t n_lists, int32_t list_len, kb_search_params params,
                           const kb_trajectory* all_cands_dev, uint64_t n_all_cands, kb_trajectory* out_dev, void* stream);

/* new (round 6): the exchange with K records per device, repaired where the records do not decide.  lists_dev: n_lists
 * lists of [n_pixels][K] records, each what kb_device_search_compact leaves with list_len = K and WITHOUT flag 512 -- the
 * reference's insertion (kernels.cu:304-331) over that device's slice of the candidates, at the speed of the single-device
 * search; K = params.results_per_pixel.  out_dev [n_pixels][K] = the single-device search over the whole candidate list for
 * every pixel whose lists determine it; a pixel one of whose FULL lists ends at the pixel's K-th likelihood (a candidate
 * that ties with it may have fallen off that list) is appended to hazard_idx_dev (uint32 pixel numbers, room for n_pixels)
 * and counted in *n_hazard_host: its slots are to be re-made by kb_repair_pixels.  Synchronises the stream. */
int kb_merge_compact_repairable(const kb_compact_result* lists_dev, int32_t n_lists, kb_search_params params,
                                const kb_trajectory* all_cands_dev, uint64_t n_all_cands, kb_trajectory* out_dev,
                                uint32_t* hazard_idx_dev, uint64_t* n_hazard_host, void* stream);
/* ... the repair: the listed start pixels (row-major numbers inside params' bounds) searched again, one wavefront per pixel
 * -- lanes = candidates, evaluateTrajectory (kernels.cu:154-242) as every search kernel here runs it, the reference's
 * insertion in candidate order -- into their K slots of out_dev.  lists_dev = NULL: over the WHOLE candidate list.  With the
 * exchange's lists (lists_dev / n_lists as above, list r = the candidates [list_cand_begin_host[r], list_cand_begin_host[r + 1])
 * of the job-wide order, the ranges tiling [0, n_all_cands) ascending): only the slices of the SUSPECT lists -- full, ending
 * at or above the pixel's K-th likelihood -- are evaluated again, the other lists' K records stand for their slices (same
 * result; an eighth of the evaluations at eight ranks).  No in-search sigma-G. */
int kb_repair_pixels(const kb_psi_phi_meta* meta, const void* psi_phi_dev, const double* times_dev, kb_search_params params,
                     const kb_trajectory* all_cands_dev, uint64_t n_all_cands, const uint32_t* pixel_idx_dev,
                     uint64_t n_listed, const kb_compact_result* lists_dev, int32_t n_lists, const int32_t* list_cand_begin_host,
                     kb_trajectory* out_dev, void* stream);

/* ---- sparse form of the exchange (new; SURVEY 8(e): "shrink traffic by compacting lh >= min_lh first").
 * The reference removes results below min_lh after its kernel (stack_search.cpp:266-270); its swap-down insertion
 * (kernels.cu:323-330) never lets a smaller likelihood touch the part of a list at or above a larger one, so the entries
 * >= min_lh of a pixel's final list are what the same insertion yields over the candidates >= min_lh alone.  Dropping
 * the records below min_lh (and the empty slots) BEFORE the exchange therefore changes nothing that survives the
 * reference's own post-filter.
 * kb_sparsify_compact: dense lists [n_pixels][list_len] of kb_device_search_compact ->
 *   header_dev  kb_sparse_header_bytes(n_pixels) bytes: uint8 counts[n_pixels] (records kept per pixel: cand >= 0 and
 *               !(lh < min_lh), in list order), zero padding to a multiple of 16, then the uint64 total;
 *   packed_dev  the kept records, pixel after pixel (room for packed_capacity records; an error when more are kept --
 *               *total_out_host then still holds the number, and the header is complete).
 * min_lh = -INFINITY keeps every non-empty slot.  Synchronises the stream. */
#define KB_SPARSE_BLOCK 256
uint64_t kb_sparse_header_bytes(uint64_t n_pixels);
int kb_sparsify_compact(const kb_compact_result* lists_dev, uint64_t n_pixels, int32_t list_len, float min_lh,
                        uint8_t* header_dev, kb_compact_result* packed_dev, uint64_t packed_capacity,
                        uint64_t* total_out_host, void* stream);
/* The search itself can write the count bytes (new): kb_device_search_counted is kb_device_search_compact that also fills
 * counts_dev[n_pixels] -- records per start pixel with cand >= 0 and !(lh < params.min_lh), a prefix of the pixel's sorted
 * list -- and, where it does, does NOT write the record runs of waves (64 consecutive start pixels of a row) that keep nothing:
 * for a thresholded search nearly all of results_dev stays untouched.  *counts_written_out = 1 when the kernel instance that ran
 * did so (kb_search_lds with packed or pooled lists), 0 when it wrote every slot and no counts (then use kb_sparsify_compact).
 * kb_sparsify_counted: the rest of kb_sparsify_compact given such counts in header_dev[0 .. n_pixels) -- block sums, scan,
 * total, and a scatter that reads only the counted records.  Same header / packed layout; synchronises the stream. */
int kb_device_search_counted(const kb_psi_phi_meta* meta, const void* psi_phi_dev, const double* times_dev,
                             kb_search_params params, const kb_trajectory* cands_dev, uint64_t n_cands,
                             int32_t cand_index_base, kb_compact_result* results_dev, uint64_t n_results, uint8_t* counts_dev,
                             uint32_t flags, void* stream, kb_search_stats* stats_out, int32_t* counts_written_out);
/* The same for a single device's whole search (stack_search.cpp:221-284 with a likelihood threshold): the 28-byte trajectories
 * of kb_device_search_filter + the count bytes, and the filter / sort that reads through them -- only the counted records are
 * read, the working storage is sized by what survives, min_lh must be the search's own and above -FLT_MAX (the counts leave
 * out the placeholders of empty slots).  Otherwise kb_filter_sort_results_checked. */
int kb_device_search_filter_counted(const kb_psi_phi_meta* meta, const void* psi_phi_dev, const double* times_dev,
                                    kb_search_params params, const kb_trajectory* cands_dev, uint64_t n_cands,
                                    kb_trajectory* results_dev, uint64_t n_results, uint8_t* counts_dev, uint32_t flags, void* stream,
                                    kb_search_stats* stats_out, int32_t* counts_written_out);
int kb_filter_sort_results_counted(const kb_trajectory* results_dev, uint64_t n_pixels, int32_t list_len, const uint8_t* counts_dev,
                                   float min_lh, int32_t min_obs, kb_trajectory* out_dev, uint64_t* n_out_host,
                                   int64_t* first_invalid_host, void* stream);
int kb_sparsify_counted(const kb_compact_result* lists_dev, uint64_t n_pixels, int32_t list_len, uint8_t* header_dev,
                        kb_compact_result* packed_dev, uint64_t packed_capacity, uint64_t* total_out_host, void* stream);
/* kb_merge_compact_exact on sparse lists: headers_dev = n_lists headers header_stride bytes apart (what one gather of
 * the devices' headers leaves on the root), packed_ptrs_host[r] = device pointer to list r's records (may be NULL when
 * its total is 0).  out_dev: [n_pixels][K] trajectories; where a record survives the post-filter the output equals
 * kb_merge_compact_exact's on the dense lists (and hence the single-device search), every other slot is the
 * placeholder of kernels.cu:293-301.  Synchronises the stream. */
int kb_merge_sparse_exact(const uint8_t* headers_dev, uint64_t header_stride, const kb_compact_result* const* packed_ptrs_host,
                          int32_t n_lists, int32_t list_len, kb_search_params params, const kb_trajectory* all_cands_dev,
                          uint64_t n_all_cands, kb_trajectory* out_dev, void* stream);
/* ... with the number of merged records per start pixel to counts_out_dev[n_pixels] (they are a prefix of the pixel's K slots),
 * and NO slot written for a wave (64 consecutive start pixels) that nothing reaches: what kb_filter_sort_results_counted reads
 * through.  counts_out_dev = NULL: kb_merge_sparse_exact. */
int kb_merge_sparse_exact_counted(const uint8_t* headers_dev, uint64_t header_stride, const kb_compact_result* const* packed_ptrs_host,
                                  int32_t n_lists, int32_t list_len, kb_search_params params, const kb_trajectory* all_cands_dev,
                                  uint64_t n_all_cands, kb_trajectory* out_dev, uint8_t* counts_out_dev, void* stream);

/* ---- host instantiations of the device functions ------------------------- */
/* kernels.cu:154-242 evaluateTrajectory called with host pointers
 * (stack_search.cpp:203-204). */
int kb_evaluate_trajectory_host(const kb_psi_phi_meta* meta, const void* psi_phi_host, const double* times_host,
                                kb_search_params params, kb_trajectory* candidate);
/* kernels.cu:77-147 SigmaGFilteredIndicesCU (kernel_helpers.cpp:94). */
void kb_sigmag_filtered_indices(const float* values, int num_values, float sgl0, float sgl1, float sigmag_coeff,
                                float width, int* idx_array, int* min_keep_idx, int* max_keep_idx);

/* ---- FITS ingest on the device (new).  Replaces, for the image layers, what WorkUnit.from_fits /
 * from_sharded_fits / read_image_data_from_hdul do through astropy.io.fits (work_unit.py:489-608, 782-897, 1149-1200:
 * hdul["SCI_i"].data.astype(np.single), the same for VAR_i, sci[mask > 0] = var[mask > 0] = nan): the file's bytes are
 * uploaded as they are and decoded in HBM into the [T][H][W] float32 stacks kb_build_psi_phi_from_device[_ex] reads.
 * Header cards and table rows are parsed on the host (kbmod_amd/fits_ingest.py); these entry points take what that
 * parse yields.  All three are asynchronous on `stream`. ----------------------------------------------------------- */
/* One tile (= one row of the tiled-image table; the reference's writer makes one tile per image row) of a
 * tiled-compressed image HDU (FITS 4.0 section 10; work_unit.py:1108-1122 writes CompImageHDU(RICE_1, quantize_level
 * -0.01), quantize method NO_DITHER). */
enum { KB_FITS_TILE_SKIP = 0,   /* not decoded here (a GZIP_COMPRESSED_DATA fallback tile: the host patches it in) */
       KB_FITS_TILE_RICE = 1 }; /* COMPRESSED_DATA holds a RICE_1 stream */
typedef struct kb_fits_tile {
    uint64_t offset;    /* first byte of the tile's stream, relative to heap_dev */
    uint64_t out_index; /* index of the tile's first pixel in out_dev */
    double zscale;      /* value = (float)((double) integer * zscale + zzero): ZSCALE / ZZERO of the row (quantised */
    double zzero;       /* float images) or BSCALE / BZERO (integer images) */
    uint32_t nbytes;    /* length of the stream */
    int32_t mode;       /* KB_FITS_TILE_* */
} kb_fits_tile;
/* RICE_1 streams -> float32 pixels (cfitsio ricecomp.c fits_rdecomp behind astropy's CompImageHDU.data).  tile_len
 * pixels per tile; blocksize / bytepix: the ZVALn of BLOCKSIZE / BYTEPIX (32 / 4 in the reference's files); integers equal
 * to `blank` become NaN when has_blank (ZBLANK).  status_dev[0] = number of tiles whose stream ended before their pixels
 * did or holds a split code beyond FSMAX + 1 (their output is undefined), status_dev[1] = index + 1 of one of them; read it after synchronising.  heap_bytes (at
 * least 8) bounds what is read: the decoder fetches eight bytes at a time and may read past a tile's last byte into its
 * neighbour's, never past heap_dev + heap_bytes; every tile must lie inside the heap (the caller checks its table). */
int kb_fits_decode_rice(const uint8_t* heap_dev, uint64_t heap_bytes, const kb_fits_tile* tiles_dev, int32_t n_tiles,
                        int32_t tile_len, int32_t blocksize, int32_t bytepix, int32_t quantized, int32_t has_blank,
                        int32_t blank, float* out_dev, int32_t* status_dev, void* stream);
/* The data unit of a plain IMAGE HDU (big-endian, BITPIX 8 / 16 / 32 / -32 / -64; MSK_i and PSF_i of the reference's
 * files, every layer of an uncompressed one) -> float32: BZERO + BSCALE * array evaluated in double (astropy's
 * _ImageBaseHDU scaling) and rounded once, `.astype(np.single)`. */
int kb_fits_decode_image(const uint8_t* raw_dev, int32_t bitpix, double bscale, double bzero, uint64_t n_pixels,
                         float* out_dev, void* stream);
/* sci[mask > 0] = var[mask > 0] = NaN (work_unit.py:1187-1190; image_stack_py.py:379-383).  var_dev may be NULL. */
int kb_fits_apply_mask(float* sci_dev, float* var_dev, const float* mask_dev, uint64_t n_pixels, void* stream);

/* ---- self-test of the wavefront primitives behind the in-search sigma-G clip (new; tests only).
 * keys_dev: [n_waves][64] uint32 -> keys_out_dev ascending per wave, src_out_dev the lane each key came
 * from (DPP / permlane-swap sorting network).  values_dev: [n_waves][64] float32, bounds_dev:
 * [n_waves][2] (lo, hi) -> sums_dev[w] = ((0 + v[lo]) + v[lo+1]) + ... + v[hi] chained across the lanes.
 * Either half may be skipped with NULL inputs.  Synchronises the stream. */
int kb_debug_wave_ops(const uint32_t* keys_dev, uint32_t* keys_out_dev, uint32_t* src_out_dev, const float* values_dev,
                      const int32_t* bounds_dev, float* sums_dev, uint64_t n_waves, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KBMOD_HIP_H_ */
