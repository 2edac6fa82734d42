// kb_search_lds staged as canonical floats (the default: float arrays, and encoded arrays decoded once
// per search into the padded copy).  The kernel never decodes in its loop, so one set of instances
// serves every array format.
#include "search_lds.h"

namespace kb {

void launch_search_lds_canon(const SearchArgs& a, int rows, bool sigmag, hipStream_t stream) {
    const bool tall = rows == LDS_ROWS_TALL;
    if (sigmag) {  // the emitting instances keep no list: KS is irrelevant
        if (tall) {
            launch_lds<8, LDS_ROWS_TALL, 4, true, true>(a, stream);
        } else {
            launch_lds<8, LDS_ROWS_WIDE_K, 4, true, true>(a, stream);
        }
    } else if (a.K <= 8) {
        if (tall) {
            launch_lds<8, LDS_ROWS_TALL, 4, true, false>(a, stream);
        } else {
            launch_lds<8, LDS_ROWS_WIDE_K, 4, true, false>(a, stream);
        }
    } else if (a.K <= 16) {
        if (tall) {
            launch_lds<16, LDS_ROWS_TALL, 4, true, false>(a, stream);
        } else {
            launch_lds<16, LDS_ROWS_WIDE_K, 4, true, false>(a, stream);
        }
    } else {
        if (tall) {
            launch_lds<32, LDS_ROWS_TALL, 4, true, false>(a, stream);
        } else {
            launch_lds<32, LDS_ROWS_WIDE_K, 4, true, false>(a, stream);
        }
    }
}

}  // namespace kb
