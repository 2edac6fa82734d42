// kb_search_lds staged as canonical floats (the default: float arrays, and encoded arrays decoded once
// per search into the padded copy).  The kernel never decodes in its loop, so one set of instances
// serves every array format.
#include "search_lds.h"

namespace kb {

void launch_search_lds_canon(const SearchArgs& a, bool sigmag, hipStream_t stream) {
    if (sigmag) {
        launch_lds<8, 4, true, true>(a, stream);  // the emitting instance keeps no list: KS is irrelevant
    } else if (a.K <= 8) {
        launch_lds<8, 4, true, false>(a, stream);
    } else if (a.K <= 16) {
        launch_lds<16, 4, true, false>(a, stream);
    } else {
        launch_lds<32, 4, true, false>(a, stream);
    }
}

}  // namespace kb
