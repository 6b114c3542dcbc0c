// Instantiates the MC_PT_MATH_FAST path tracer kernels (gfx950 hardware rcp/rsq/sqrt/sin/cos/exp/log).
// Split from the strict instantiations so the two halves compile in parallel.
#include "pathtrace_kernel.h"
#include "pathtrace_pq.h"

namespace mc {
namespace pt {
int launch_fast(const PTArgs& a, int variant, int S, int prec, uint32_t tile_rows, hipStream_t s) {
    if (variant == 2) {   // two-path-slots-per-lane scheduler (pathtrace_pq.h), slab scenes only
        launch_pq<true>(a, tile_rows, s);
        return MC_OK;
    }
    return launch_impl<true>(a, variant, S, prec, tile_rows, s);
}
}  // namespace pt
}  // namespace mc

#ifdef MC_PT_REGION_STATS
// Diagnostic build (make stats): read and reset the per-region execution / active-lane counters of the fast kernels.
extern "C" int mc_debug_pt_region_stats(unsigned long long* exec16, unsigned long long* lanes16) {
    unsigned long long zero[16] = {0};
    if (hipMemcpyFromSymbol(exec16, HIP_SYMBOL(mc::pt::g_region_exec), sizeof(zero)) != hipSuccess) return 3;
    if (hipMemcpyFromSymbol(lanes16, HIP_SYMBOL(mc::pt::g_region_lanes), sizeof(zero)) != hipSuccess) return 3;
    if (hipMemcpyToSymbol(HIP_SYMBOL(mc::pt::g_region_exec), zero, sizeof(zero)) != hipSuccess) return 3;
    if (hipMemcpyToSymbol(HIP_SYMBOL(mc::pt::g_region_lanes), zero, sizeof(zero)) != hipSuccess) return 3;
    return 0;
}
#endif
