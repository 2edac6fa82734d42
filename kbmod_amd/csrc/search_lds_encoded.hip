// kb_search_lds staged in the array's own encoding (flag 16 of kb_device_search_filter, or an HBM
// too full for the float copy): the samples are decoded in the loop.  results_per_pixel <= 8 (and the
// sigma-G emit); larger lists with encoded staging take kb_search_direct.
#include "search_lds.h"

namespace kb {

template <int NB>
static void launch_encoded_fmt(const SearchArgs& a, bool sigmag, hipStream_t stream) {
    if (sigmag) {
        launch_lds<8, NB, false, true>(a, stream);
    } else {
        launch_lds<8, NB, false, false>(a, stream);
    }
}

bool launch_search_lds_encoded(const SearchArgs& a, int fmt, bool sigmag, hipStream_t stream) {
    if (!sigmag && a.K > 8) return false;
    switch (fmt) {
        case 1:
            launch_encoded_fmt<1>(a, sigmag, stream);
            return true;
        case 10:
            launch_encoded_fmt<10>(a, sigmag, stream);
            return true;
        case 2:
            launch_encoded_fmt<2>(a, sigmag, stream);
            return true;
        case 20:
            launch_encoded_fmt<20>(a, sigmag, stream);
            return true;
        default:
            return false;
    }
}

}  // namespace kb
