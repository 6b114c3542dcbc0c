// Instantiates the MC_PT_MATH_FAST path tracer kernels (gfx950 hardware rcp/rsq/sqrt/sin/cos/exp/log).
// Split from the strict instantiations so the two halves compile in parallel.
//
// MC_PT_MATH_FAST is the toleranced mode.  Its bound was stated before any measurement and is asserted where it
// was stated — K2, 900x600, 500 spp, whole image, against the oracle with libm: RMSE <= 0.5 and 99.9-percentile per-pixel
// RGB L2 <= 4 (tests/test_gpu_fullsize.py::test_k2_fast_math_within_the_stated_tolerance; measured 0.22 / 3.86 with
// contraction everywhere, 0.15 / 1.74 without, profiles/r02a_fast_tolerance.log).  So this translation unit — and only this
// one — lets the compiler contract a*b+c into v_fmac/v_fma: 29.7 -> 25.4 ms at K2 (MC_PT_FAST_CONTRACT = 0 vs 2).
// The two-float / df64 primitives and the explicit-polynomial math are included FIRST, under the command line's
// -ffp-contract=off: their error-free transformations must never be contracted, in either mode.
#include <hip/hip_runtime.h>

#include "mc_internal.h"
#include "ds_arith.h"
#include "mc_math.h"
#ifndef MC_PT_FAST_CONTRACT
#define MC_PT_FAST_CONTRACT 2
#endif
#if MC_PT_FAST_CONTRACT >= 1
#pragma clang fp contract(fast)   // (reassociate(on) on top of this was tried: 697 vs 700 VALU instructions, not kept)
#endif
#if MC_PT_FAST_CONTRACT == 1
#define MC_PT_DECISION_FP _Pragma("clang fp contract(off)")
#endif
#include "pathtrace_kernel.h"
#include "pathtrace_pool.h"

namespace mc {
namespace pt {
int launch_fast(const PTArgs& a, int variant, int S, int prec, uint32_t tile_rows, hipStream_t s) {
    if (variant == 4) return launch_pool<1>(a, S, tile_rows, s);
    return launch_impl<1>(a, variant, S, prec, tile_rows, s);
}
}  // namespace pt
}  // namespace mc

#ifdef MC_PT_REGION_STATS
// Diagnostic build (make stats): read and reset the per-region execution / active-lane counters — of all three tiers' kernels.
namespace mc { namespace pt {
MC_PT_REGION_STATS_READER(region_stats_fast)
int region_stats_careful(unsigned long long*, unsigned long long*);
int region_stats_strict(unsigned long long*, unsigned long long*);
} }
extern "C" int mc_debug_pt_region_stats(unsigned long long* exec32, unsigned long long* lanes32) {   // (32 entries each)
    for (int i = 0; i < 32; i++) exec32[i] = lanes32[i] = 0;
    int rc = mc::pt::region_stats_fast(exec32, lanes32);
    if (!rc) rc = mc::pt::region_stats_careful(exec32, lanes32);
    if (!rc) rc = mc::pt::region_stats_strict(exec32, lanes32);
    return rc;
}
#endif
