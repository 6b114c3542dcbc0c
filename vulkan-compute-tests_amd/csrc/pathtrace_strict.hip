// Instantiates the MC_PT_MATH_STRICT path tracer kernels (IEEE divide/sqrt + mc_math sin/cos/pow:
// bit-identical to the CPU oracle).  Split from the fast instantiations so both compile in parallel.
#include "pathtrace_kernel.h"
#include "pathtrace_pool.h"

namespace mc {
namespace pt {
int launch_strict(const PTArgs& a, int variant, int S, int prec, uint32_t tile_rows, hipStream_t s) {
    if (variant == 4) return launch_pool<0>(a, S, tile_rows, s);
    return launch_impl<0>(a, variant, S, prec, tile_rows, s);
}
}  // namespace pt
}  // namespace mc

#ifdef MC_PT_REGION_STATS
namespace mc { namespace pt { MC_PT_REGION_STATS_READER(region_stats_strict) } }   // (summed by mc_debug_pt_region_stats, pathtrace_fast.hip)
#endif
