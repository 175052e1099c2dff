// Fused residual block (single kernel) -- see agx_resblock_forward in agx.h.
#include "common.hpp"

namespace agx {

bool resblock_fused_supported(const ConvPlan &) { return false; }

int launch_resblock_fused(const ConvPlan &, const float *, const float *, const float *, const float *,
                          const float *, float *, int, hipStream_t) {
    return fail(AGX_ERR_UNSUPPORTED, "resblock: no fused kernel for this shape");
}

}  // namespace agx
