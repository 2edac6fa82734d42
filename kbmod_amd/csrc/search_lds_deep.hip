// kb_search_lds with two staged slabs in flight per wave (search_lds.h, STAGE_DEPTH = 2): the instances the host
// launches for deep stacks (from about a hundred epochs on, search_kernels.hip).  They need about 20 registers more than the
// one-deep ones, so they exist for the list modes that leave room: the sigma-G emit, lists of up to 8 as records in
// the HBM store, longer lists as (likelihood, candidate) pairs there.
#include "search_lds.h"

namespace kb {

template <int KS, bool SIGMAG, int LM>
static void launch_deep(const SearchArgs& a, bool tall, hipStream_t stream) {
    if (tall) {
        launch_lds<KS, LDS_ROWS_TALL, 4, true, SIGMAG, LM, 2>(a, stream);
    } else {
        launch_lds<KS, LDS_ROWS_WIDE_K, 4, true, SIGMAG, LM, 2>(a, stream);
    }
}

// list_mode: LIST_STORE_RECORDS for K <= 8, LIST_STORE_IDS beyond (the host sets it, search_kernels.hip)
void launch_search_lds_canon_deep(const SearchArgs& a, int rows, bool sigmag, hipStream_t stream) {
    const bool tall = rows == LDS_ROWS_TALL;
    if (sigmag) {
        launch_deep<8, true, LIST_REGISTERS>(a, tall, stream);
    } else if (a.K <= 8) {
        launch_deep<8, false, LIST_STORE_RECORDS>(a, tall, stream);
    } else if (a.K <= 16) {
        launch_deep<16, false, LIST_STORE_IDS>(a, tall, stream);
    } else {
        launch_deep<32, false, LIST_STORE_IDS>(a, tall, stream);
    }
}

}  // namespace kb
