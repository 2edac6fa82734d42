// kb_search_lds staged as canonical floats (the default: float arrays, and encoded arrays decoded once
// per search into the padded copy).  The kernel never decodes in its loop, so one set of instances
// serves every array format.
#include "search_lds.h"

namespace kb {

template <int KS, int LM>
static void launch_canon(const SearchArgs& a, bool tall, hipStream_t stream) {
    if (tall) {
        launch_lds<KS, LDS_ROWS_TALL, 4, true, false, LM>(a, stream);
    } else {
        launch_lds<KS, LDS_ROWS_WIDE_K, 4, true, false, LM>(a, stream);
    }
}

// list_mode: ListMode; valid pairs are K <= 8 with registers, packed register records or records, K <= 16 with ids or records, K <= 32 with ids
// (the host chooses, search_kernels.hip)
void launch_search_lds_canon(const SearchArgs& a, int rows, bool sigmag, int list_mode, hipStream_t stream) {
    const bool tall = rows == LDS_ROWS_TALL;
    if (sigmag && a.chunk == WIDE_CHUNK) {
        if (tall) {
            launch_lds<8, LDS_ROWS_TALL, 4, true, true, LIST_REGISTERS, WIDE_CHUNK>(a, stream);
        } else {
            launch_lds<8, LDS_ROWS_WIDE_K, 4, true, true, LIST_REGISTERS, WIDE_CHUNK>(a, stream);
        }
    } else if (sigmag) {  // the emitting instances keep no list: KS is irrelevant
        if (tall) {
            launch_lds<8, LDS_ROWS_TALL, 4, true, true, LIST_REGISTERS>(a, stream);
        } else {
            launch_lds<8, LDS_ROWS_WIDE_K, 4, true, true, LIST_REGISTERS>(a, stream);
        }
    } else if (a.K <= 8 && a.chunk == XWIDE_CHUNK) {  // (arrays beyond the Infinity Cache: 32 candidates per staged slab, 64 x 16 tiles)
        launch_lds<8, LDS_ROWS_TALL, 4, true, false, LIST_REGISTER_RECORDS, XWIDE_CHUNK>(a, stream);
    } else if (a.K <= 8 && a.chunk == WIDE_CHUNK) {  // (the host pairs the wide chunks with this list mode, search_kernels.hip)
        if (tall) {
            launch_lds<8, LDS_ROWS_TALL, 4, true, false, LIST_REGISTER_RECORDS, WIDE_CHUNK>(a, stream);
        } else {
            launch_lds<8, LDS_ROWS_WIDE_K, 4, true, false, LIST_REGISTER_RECORDS, WIDE_CHUNK>(a, stream);
        }
    } else if (a.K <= 16 && a.chunk == WIDE_CHUNK) {  // (... and stable lists of 9 to 16 with the pooled store)
        if (tall) {
            launch_lds<16, LDS_ROWS_TALL, 4, true, false, LIST_STORE_POOLED, WIDE_CHUNK>(a, stream);
        } else {
            launch_lds<16, LDS_ROWS_WIDE_K, 4, true, false, LIST_STORE_POOLED, WIDE_CHUNK>(a, stream);
        }
    } else if (a.K <= 8) {
        if (list_mode == LIST_STORE_RECORDS) {
            launch_canon<8, LIST_STORE_RECORDS>(a, tall, stream);
        } else if (list_mode == LIST_REGISTER_RECORDS) {
            launch_canon<8, LIST_REGISTER_RECORDS>(a, tall, stream);
        } else {
            launch_canon<8, LIST_REGISTERS>(a, tall, stream);
        }
    } else if (a.K <= 16) {
        if (list_mode == LIST_STORE_RECORDS) {
            launch_canon<16, LIST_STORE_RECORDS>(a, tall, stream);
        } else {
            launch_canon<16, LIST_STORE_IDS>(a, tall, stream);
        }
    } else {
        launch_canon<32, LIST_STORE_IDS>(a, tall, stream);
    }
}

}  // namespace kb

#ifdef KB_EXP_PROFILE
extern "C" int kb_exp_read_profile(unsigned long long* out) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(kb::kb_exp_prof), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    return hipMemcpyToSymbol(HIP_SYMBOL(kb::kb_exp_prof), zero, sizeof(zero)) != hipSuccess;
}
#endif
