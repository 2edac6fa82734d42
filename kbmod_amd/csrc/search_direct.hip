// Fallback search kernels of libkbmod_hip.so: kb_search_direct (every sample a wave-wide load from
// the array itself; few candidates, scattered candidate lists, shifts the table could not prove) and
// kb_search_large_k (results_per_pixel > 32).  See search_kernels.hip for the overall design.
#include <cstdio>

#include "search_device.h"

#pragma clang fp contract(off)

namespace kb {

template <int KS, int C, int NB, bool SIGMAG, bool RECORDS>
// second launch bound = waves per SIMD: 4 / 3 / 2 four-wave workgroups per CU for K <= 8 / 16 / 32
__global__ __launch_bounds__(DIRECT_ROWS * WAVE, (KS <= 8 ? 4 : (KS <= 16 ? 3 : 2))) void kb_search_direct(const SearchArgs a) {
    const TileCoords tc = tile_coords<DIRECT_ROWS>(a);
    if (!tc.row_active) return;  // whole wave (no barriers in this kernel)
    const int pix0 = tc.y * a.W + tc.x;
    // RECORDS (lists of up to 16, no sigma-G): whole result records in registers -- no re-evaluation of the
    // winners, which costs K x T exact samples per pixel: for a short candidate list more than the search itself.
    // Lists of up to 8 take the packed form (24 registers: always, when the candidate indices fit 16 bits); lists
    // of 9 .. 16 four words per slot when the candidate list is short against the stack depth (the host chooses).
    TopK<KS> top;
    TopKRecords<KS> rec;    // RECORDS, lists of 9 .. 16
    TopKPacked<KS> packed;  // RECORDS, lists of up to 8 (the host checks that candidate indices fit 16 bits)
    top.init();
    rec.init();
    packed.init();

    for (int chunk = a.chunk_lo; chunk < a.chunk_hi; ++chunk) {
        float ps[C], ph[C];
        int cnt[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            ps[c] = 0.0f;
            ph[c] = 0.0f;
            cnt[c] = 0;
        }
        const ChunkInfo ci = a.chunks[chunk];
        const bool exact = a.force_exact || ci.unsafe;
        const bool interior = (tc.tile_x0 + ci.dx_min >= 0) && (tc.tile_x0 + WAVE - 1 + ci.dx_max < a.W) &&
                              (tc.y + ci.dy_min >= 0) && (tc.y + ci.dy_max < a.H);
        if (exact) {
            accumulate_chunk_direct_all<C, NB, 2>(a, chunk, tc.x, tc.y, pix0, ps, ph, cnt);
        } else if (interior) {
            accumulate_chunk_direct_all<C, NB, 0>(a, chunk, tc.x, tc.y, pix0, ps, ph, cnt);
        } else {
            accumulate_chunk_direct_all<C, NB, 1>(a, chunk, tc.x, tc.y, pix0, ps, ph, cnt);
        }
        if constexpr (RECORDS && KS <= 8) {
            finish_chunk_packed<KS, C>(a, chunk * C, ps, ph, cnt, packed);
        } else if constexpr (RECORDS) {
            finish_chunk_records<KS, C>(a, chunk, ps, ph, cnt, rec);
        } else {
            finish_chunk<KS, C, SIGMAG>(a, tc, chunk, ps, ph, cnt, top);
        }
    }
    if constexpr (RECORDS && KS <= 8) {
        write_packed<KS>(a, tc, packed);
    } else if constexpr (RECORDS) {
        write_records<KS>(a, tc, rec);
    } else if constexpr (!SIGMAG) {
        write_results<KS>(a, tc, top);
    }
}

// ---------------------------------------------------------------------------
// large-K kernel (results_per_pixel > 32, e.g. TrajectoryExplorer's K up to 10 000)
// ---------------------------------------------------------------------------
// One lane per start pixel, candidates evaluated one at a time with exact
// per-lane positions, the K-slot list kept in the result array itself and
// updated with the reference's swap-down (kernels.cu:304-331).  This is the
// reference kernel's own structure; it is only used where the register top-K
// cannot hold the list (few start pixels x many results in practice).
template <bool SIGMAG>
__global__ __launch_bounds__(DIRECT_ROWS * WAVE) void kb_search_large_k(const SearchArgs a) {
    // A bounded grid walks the tiles (workgroup b takes tiles b, b + gridDim.x, ...), so that the
    // sigma-G scratch is sized by the waves of the launch and not by the search area.
    SigmaGScratch<WAVE> scratch = {};
    if constexpr (SIGMAG) {
        scratch = make_scratch(a.cold->sg_scratch, a.T, (size_t)blockIdx.x * DIRECT_ROWS + (threadIdx.x >> 6),
                               threadIdx.x & (WAVE - 1));
    }
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const TileCoords tc = tile_coords<DIRECT_ROWS>(a, tile);
        if (!tc.row_active || tc.x_i >= a.sw) continue;
        kb_trajectory* slots = a.cold->results.full + ((size_t)tc.y_i * a.sw + tc.x_i) * a.K;
        for (int s = 0; s < a.K; ++s) slots[s] = placeholder_result(tc.x, tc.y);  // kernels.cu:293-301
        for (int cand = 0; cand < a.n_cands; ++cand) {
            kb_trajectory cur;
            cur.x = tc.x;
            cur.y = tc.y;
            cur.vx = a.cold->cands[cand].vx;
            cur.vy = a.cold->cands[cand].vy;
            evaluate_trajectory_full<WAVE>(a.cold->meta, a.psi_phi, a.cold->times, a.cold->params, &cur, SIGMAG ? &scratch : nullptr);
            if ((cur.obs_count < a.min_obs) || (a.cold->params.do_sigmag_filter && cur.lh < a.min_lh))
                continue;  // kernels.cu:318-320
            if (!(cur.lh > slots[a.K - 1].lh)) continue;  // cannot displace anything
            bool placed = false;
            for (int s = 0; s < a.K; ++s) {  // kernels.cu:323-330 (stable lists: TopK::insert)
                const kb_trajectory t = slots[s];
                if (cur.lh > t.lh || (a.stable_lists != 0 && placed)) {
                    slots[s] = cur;
                    cur = t;
                    placed = true;
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------
// launchers (declared in search_common.h)
// ---------------------------------------------------------------------------
template <int KS, int NB>
static void launch_direct_fmt(const SearchArgs& a, bool sigmag, bool records, hipStream_t stream) {
    const dim3 grid(a.n_tiles), block(DIRECT_ROWS * WAVE);
    {
        const bool rec = !sigmag && records && KS <= 16;
        char name[96];
        std::snprintf(name, sizeof(name), "kb::kb_search_direct<%d, %d, %d, %s, %s>", sigmag ? 8 : KS, CHUNK, NB, sigmag ? "true" : "false",
                      rec ? "true" : "false");
        note_kernel_instance(name);
    }
    if (sigmag) {
        // the emitting instances keep no list: one set (KS = 8) serves every K
        if constexpr (KS == 8) hipLaunchKernelGGL((kb_search_direct<8, CHUNK, NB, true, false>), grid, block, 0, stream, a);
    } else if (records && KS <= 16) {
        if constexpr (KS <= 16) hipLaunchKernelGGL((kb_search_direct<KS, CHUNK, NB, false, true>), grid, block, 0, stream, a);
    } else {
        hipLaunchKernelGGL((kb_search_direct<KS, CHUNK, NB, false, false>), grid, block, 0, stream, a);
    }
}

template <int KS>
static void launch_direct_ks(const SearchArgs& a, int fmt, bool sigmag, bool records, hipStream_t stream) {
    switch (fmt) {
        case 1:
            launch_direct_fmt<KS, 1>(a, sigmag, records, stream);
            break;
        case 10:
            launch_direct_fmt<KS, 10>(a, sigmag, records, stream);
            break;
        case 2:
            launch_direct_fmt<KS, 2>(a, sigmag, records, stream);
            break;
        case 20:
            launch_direct_fmt<KS, 20>(a, sigmag, records, stream);
            break;
        default:
            launch_direct_fmt<KS, 4>(a, sigmag, records, stream);
            break;
    }
}

void launch_search_direct(const SearchArgs& a, int fmt, bool sigmag, bool records, hipStream_t stream) {
    if (sigmag || a.K <= 8) {
        launch_direct_ks<8>(a, fmt, sigmag, records, stream);
    } else if (a.K <= 16) {
        launch_direct_ks<16>(a, fmt, false, records, stream);
    } else {
        launch_direct_ks<32>(a, fmt, false, false, stream);
    }
}

void launch_search_large_k(const SearchArgs& a, bool sigmag, int blocks, hipStream_t stream) {
    note_kernel_instance(sigmag ? "kb::kb_search_large_k<true>" : "kb::kb_search_large_k<false>");
    if (sigmag) {
        hipLaunchKernelGGL((kb_search_large_k<true>), dim3(blocks), dim3(DIRECT_ROWS * WAVE), 0, stream, a);
    } else {
        hipLaunchKernelGGL((kb_search_large_k<false>), dim3(blocks), dim3(DIRECT_ROWS * WAVE), 0, stream, a);
    }
}

}  // namespace kb
