// TEMPORARY: entry points not implemented yet return RQHIP_EUNSUPPORTED (replaced later this round).
#include "rqhip_common.h"
using namespace rqhip;
extern "C" int rqhip_gumbel_forward(const float *, int64_t, int, const float *, int, const float *, float, float,
                                    int64_t *, float *, float *, rqhip_stream_t) {
    set_error("rqhip_gumbel_forward: not implemented yet");
    return RQHIP_EUNSUPPORTED;
}
extern "C" size_t rqhip_gumbel_backward_workspace_bytes(int64_t, int, int) { return 16; }
extern "C" int rqhip_gumbel_backward(const float *, int64_t, int, const float *, int, const float *, float, float,
                                     const float *, const float *, float *, float *, void *, size_t,
                                     rqhip_stream_t) {
    set_error("rqhip_gumbel_backward: not implemented yet");
    return RQHIP_EUNSUPPORTED;
}
