// What the multi-GPU group (kl_group.hip) needs to see of a context (kl_api.cpp owns kl_ctx).
#pragma once

#include <hip/hip_runtime.h>

#include "Math/kl_render.h"

namespace kl {

struct CtxView {
    int device;
    hipStream_t stream;          // the context's stream: every render call ends ordered on it
    float4 *acc;                 // accumulation buffer (local rows x width), allocated if it was not yet
    uint32_t width, height, npl, local_rows;
    uint32_t rank, world, band_rows;
    uint32_t frame, max_frames, accum_mode, cus;
};
// fills `out`; allocates (and clears) the accumulation buffer if no frame has been rendered yet
kl_status ctx_view(kl_ctx *c, CtxView *out);

}  // namespace kl
