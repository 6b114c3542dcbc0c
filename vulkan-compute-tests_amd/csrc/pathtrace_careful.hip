// Instantiates the CAREFUL tier of the MC_PT_MATH_FAST path tracer kernels (Fast = 2, csrc/mc_math.h): every kernel family of
// pathtrace_fast.hip, compiled WITHOUT contraction (this translation unit is built under the command line's -ffp-contract=off, like the
// strict one), with division, square root and reciprocal square root rounded as the reference rounds them (the strict mode's short
// forms, unguarded), the sphere discriminant in the reference's order and the sampled / refracted directions re-normalised; the
// fast-math identities that are free of side effects stay (wall bounce as a permutation, one division for the slab tests, closed box,
// shadow rays by squares, hardware sin / cos / exp / log).
//
// Why a second tier: a fast-math sample differs from the reference's by a rounding in almost every operation, and wherever a path
// runs through specular spheres such a difference is amplified bounce by bounce until a discrete decision (which object, shadowed or
// lit) goes the other way — the sample "forks" to another valid path.  The share of forked samples grows with the specular surface a path
// can run through, and a fork that a specular chain carries to a light moves its pixel by the light's whole emission: the fast tier keeps
// the stated tolerance (RMSE 0.5 / 99.9-percentile L2 4 at 500 spp) on boxes with up to three spheres, misses it on 2 of 44 random boxes with
// four, reads 3.2 at five and exceeds the bound from six on (p99.9 4.4 .. 5.6) — and misses it on 14 of 132 three-sphere rooms with a mirror or
// glass wall (up to 8.0).  Since round 6 the host switches at FOUR spheres and wherever a scene has more specular surface than the reference
// scene's (csrc/pathtrace.hip: kCarefulSpheres, kFastSpecularArea).  The census (tools/fork_census.py, profiles/r05_fork_census*.txt) shows
// what carries the forks: contraction and the hardware seeds' last bit, in equal parts — not any one shortcut.  With both removed the
// same scenes read 1.9 .. 3.0; with the two identities whose forks are one-sided (tools/fork_bias.py) in the reference's form as well,
// 0.65 .. 1.14 at 1.14 .. 1.43 of the fast tier's time (profiles/r05_fast_tiers.txt; the strict kernels: 2.35 x); the 170 promoted rooms of
// round 6 read at most 0.80 (profiles/r06_fast_tolerance_scenes*.txt).  The host selects this tier for an MC_PT_MATH_FAST request on such a
// scene (mc_pathtrace_scene_class bits 4 and 5), and for an explicit MC_PT_MATH_FAST_CAREFUL one.
#include "pathtrace_kernel.h"
#include "pathtrace_pool.h"

namespace mc {
namespace pt {
int launch_careful(const PTArgs& a, int variant, int S, int prec, uint32_t tile_rows, hipStream_t s) {
    if (variant == 4) return launch_pool<2>(a, S, tile_rows, s);
    return launch_impl<2>(a, variant, S, prec, tile_rows, s);
}
}  // namespace pt
}  // namespace mc

#ifdef MC_PT_REGION_STATS
namespace mc { namespace pt { MC_PT_REGION_STATS_READER(region_stats_careful) } }   // (summed by mc_debug_pt_region_stats, pathtrace_fast.hip)
#endif
