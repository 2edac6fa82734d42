// kb_search_lds staged in the array's own encoding (flag 16 of kb_device_search_filter, or an HBM
// too full for the float copy): the samples are decoded in the loop.  results_per_pixel <= 8 (and the
// sigma-G emit), 64 x 8 tiles; larger lists with encoded staging take kb_search_direct.
#include "search_lds.h"

namespace kb {

template <int NB>
static void launch_encoded_fmt(const SearchArgs& a, bool sigmag, hipStream_t stream) {
    if (sigmag) {
        launch_lds<8, LDS_ROWS_WIDE_K, NB, false, true, LIST_REGISTERS>(a, stream);
    } else {
        launch_lds<8, LDS_ROWS_WIDE_K, NB, false, false, LIST_REGISTERS>(a, stream);
    }
}

void launch_search_lds_encoded(const SearchArgs& a, int rows, int fmt, bool sigmag, hipStream_t stream) {
    (void)rows;  // the host builds the tables for LDS_ROWS_WIDE_K when it stages the array encoded
    switch (fmt) {
        case 1:
            launch_encoded_fmt<1>(a, sigmag, stream);
            break;
        case 10:
            launch_encoded_fmt<10>(a, sigmag, stream);
            break;
        case 2:
            launch_encoded_fmt<2>(a, sigmag, stream);
            break;
        default:
            launch_encoded_fmt<20>(a, sigmag, stream);
            break;
    }
}

}  // namespace kb
