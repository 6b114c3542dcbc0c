// Instantiates the MC_PT_MATH_FAST path tracer kernels (gfx950 hardware rcp/rsq/sqrt/sin/cos/exp/log).
// Split from the strict instantiations so the two halves compile in parallel.
#include "pathtrace_kernel.h"

namespace mc {
namespace pt {
int launch_fast(const PTArgs& a, int variant, int S, int prec, uint32_t tile_rows, hipStream_t s) {
    return launch_impl<true>(a, variant, S, prec, tile_rows, s);
}
}  // namespace pt
}  // namespace mc
