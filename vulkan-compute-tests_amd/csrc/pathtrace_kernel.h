// Path tracer device code for gfx950 (MI355X) — included by pathtrace_fast.hip / pathtrace_strict.hip.
//
// Replaces shaders/pathTracer.comp:343-458 AND the host-side sample loop that records `spp`
// dispatches (src/pathtracerApp.h:358-378): the whole [sample_begin, sample_end) range of a pixel
// runs inside one launch, the accumulator lives in registers, and the 16-B storage-buffer entry is
// written once (the reference re-reads and re-writes it every sample through host-visible memory).
//
// Parity rules:
//  * samples are accumulated in sample order, `acc += accrad / spp` per sample (pathTracer.comp:451-452);
//  * rand01 is the reference's stateless integer hash, bit-exact (pathTracer.comp:107-110);
//  * every fp32 expression keeps the GLSL source order, unfused (built -ffp-contract=off);
//  * strict mode uses IEEE divide/sqrt and the explicit mc_math.h algorithms for sin/cos/pow, which
//    makes the output buffer bit-identical to the CPU oracle; fast mode uses the gfx950 hardware
//    approximations and is compared under a stated tolerance.
//
// MI355X mapping:
//  * SAMPLE-PARALLEL LANES.  A wave64 owns 64/S pixels x S consecutive samples (S = 1,4,16).  The S
//    lanes of a pixel trace samples base+0..base+S-1 concurrently; after each round their radiances
//    are folded into the pixel accumulator IN SAMPLE ORDER through cross-lane reads, so the fp32 sum
//    is the same sequence of additions as the one-thread-per-pixel loop.  S is chosen on the host so
//    that even a 900x600 image yields >10^5 short waves: the 256 CUs stay full and the tail vanishes
//    (with S = 1 that image is only 8.4k long waves against ~7k resident wave slots).
//  * Scene constants arrive in the kernel-argument segment: uniform accesses in the intersection
//    loops become scalar loads / SGPR operands; the per-lane material fetch by hit index reads a copy
//    staged in LDS at kernel start.
//  * SLAB SPECIALISATION (exact).  When the host finds that every plane normal is +-1 along one axis
//    and 0 elsewhere (the reference scene is an axis-aligned box), dot(d,n) is exactly +-d[axis], at
//    most one of the two planes of an axis can face the ray, and the six plane tests collapse into
//    three — same quotients, same comparison order, bit-identical hit records (see intersect()).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mc_internal.h"
#include "ds_arith.h"
#include "mc_math.h"

// Fork census only (tools/fork_census.py, profiles/r05_fork_census.txt): the sampled / refracted directions re-normalised as the
// reference does (:413, :428, :441); the root form of the shadow test although the spheres are disjoint
#ifndef MC_PT_FAST_RENORMALISE
#define MC_PT_FAST_RENORMALISE 0
#endif
#ifndef MC_PT_FAST_NO_DISJOINT
#define MC_PT_FAST_NO_DISJOINT 0
#endif

namespace mc {
namespace pt {

constexpr int kMaxPlanes = 16;
constexpr int kMaxSpheres = 16;
constexpr int kMaxSlabSpheres = 8;   // the slab / closed-box / sample-pool kernels are instantiated for 1 .. 8 spheres

struct v3 {
    float x, y, z;
};

// Scene in the order the kernels consume it.  obj[] keeps the reference's 12-float records
// (planes first, then spheres) for the per-lane material fetch; r2[] caches radius*radius (the fp32
// product the shader computes at pathTracer.comp:318, evaluated once on the host in fp32).
struct SceneArgs {
    uint32_t n_planes, n_spheres;
    uint32_t emissive_mask;   // bit i set <=> dot(spheres[i].e, spheres[i].e) > 0  (pathTracer.comp:407)
    // Slab kernels: non-zero when the host proved that no plane can be the nearest hit of a shadow ray that reaches
    // its light (closed box, camera inside, every emissive sphere inside with a margin — pathtrace.hip,
    // lights_inside_box): the shadow `intersect` then skips the three slab tests.  Exact, see intersect().
    uint32_t nee_skip_planes;
    // Slab kernels: non-zero when every object's material code int(floor(m + 0.5)) is 1, 2 or 3, i.e. every bounce continues
    // from its hit point.  Any other code leaves the ray untouched (:400-448 match nothing) and the same ray is traced again.
    uint32_t materials_known;
    // Slab kernels: non-zero when the host proved that accmat stays finite (every colour component finite and in [0, 1], depth
    // limit small): only then is `accmat * e` of a non-emitting object a zero and the emission add skippable (see :391 below)
    uint32_t emit_skip_ok;
    // Fast math only: non-zero when the scene may take the closed-box kernel (Box = true): nee_skip_planes (closed box, camera
    // and lights inside), every material known, no wall of glass — no ray ever leaves the box, so every plane parameter is >= 0
    uint32_t box_ok;
    float obj[(kMaxPlanes + kMaxSpheres) * 12];
    float r2[kMaxSpheres];
    // slab form of axis-aligned planes (valid only for the Slab kernels): per axis the plane whose normal
    // is +e_axis ("pos") and -e_axis ("neg"): offset w and plane index, index < 0 when absent.
    float slab_w_pos[3], slab_w_neg[3];
    int32_t slab_id_pos[3], slab_id_neg[3];
    // Generic kernels (any scene): the 12-float records (planes, then spheres) and the list of emissive sphere
    // indices live in a device buffer and are staged into dynamic LDS by every block — obj[]/r2[]/emissive_mask
    // above are then unused.  Capacity is bounded by the 160 KB of LDS per CU, not by the kernel-argument segment.
    const float* d_obj;
    // the same records with the staged copy's derived slots ([7] = max colour component, [11] = floor(m + 0.5), stage_records)
    // already in place: read directly from memory by the kernels that leave the scene there (NP = NS = -2; large scenes)
    const float* d_obj_derived;
    const uint32_t* d_emissive;
    uint32_t n_emissive;
    // Fast math, slab scenes: non-zero when the spheres are pairwise disjoint with a margin (pathtrace.hip): the order of
    // two spheres along any ray is then the order of their centres' projections (shadow_visible_disjoint)
    uint32_t spheres_disjoint;
    uint32_t pad3, pad4;
};

struct PTArgs {
    uint32_t W, H, spp, sample_begin, sample_end, max_depth, row_begin, row_end, row_block, row_stride;
    // camera basis (pathTracer.comp:352-353,360), evaluated once on the host with the same IEEE ops
    v3 cam_o, cam_d, cx, cy, lc;
    // slab kernels: sphere centre - lc and its squared length for the camera ray's intersect (:317-:318 with o = lc), the same
    // IEEE operations evaluated once on the host — wave-uniform values that would otherwise sit in (spilled) vector registers
    float cam_oc[kMaxSlabSpheres][3], cam_occ[kMaxSlabSpheres];
    // fast mode: 1/W, 1/H, 1/spp (host: 1.0f / x) for the sensor position (:358-:359) and accrad / samps.y (:452) — uniform
    // reciprocals the kernel would otherwise form with v_rcp_f32 and carry in vector registers through every round
    float inv_W, inv_H, inv_spp;
    float4* __restrict__ out;   // tile-local storage rows
    uint32_t* status;           // context-owned device word: a kernel ORs a bit in when a scheduler bound trips (else untouched)
    SceneArgs scene;
};

__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return v3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return v3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ v3 operator*(v3 a, v3 b) { return v3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return v3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ v3 operator-(v3 a) { return v3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// GLSL cross: (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y)
__device__ __forceinline__ v3 cross(v3 a, v3 b) {
    return v3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
template <int Fast> __device__ __forceinline__ v3 normalize(v3 a) { return a * dm::inversesqrt<Fast>(dot(a, a)); }
// a / s, three numerators over one divisor.  Strict: the short division of mc_math.h inside its window (wave-wide test),
// the compiler's IEEE expansion outside.  divs_recip: the caller supplies y = RN(1/s) (a constant, a host-computed argument).
template <int Fast, bool HaveY> __device__ __forceinline__ v3 divs_impl(v3 a, float s, float y) {
    dm::div3<Fast, HaveY>(a.x, a.y, a.z, s, y);
    return a;
}
template <int Fast> __device__ __forceinline__ v3 divs(v3 a, float s) { return divs_impl<Fast, false>(a, s, 0.0f); }
template <int Fast> __device__ __forceinline__ v3 divs_recip(v3 a, float s, float y) { return divs_impl<Fast, true>(a, s, y); }
// First tangent of the orthonormal basis around w (:409, :427): normalize(cross(|w.x| > 0.1 ? (0,1,0) : (1,0,0), w)).
// Strict: the literal expression (its products with the axis' zeros decide the sign of zero components).
// Fast (toleranced): the cross product is (w.z, 0, -w.x) or (0, -w.z, w.y) — one select, one fused squared length, two
// scalings and three selects instead of two selects, nine cross-product operations, a 3-term dot and three scalings;
// the non-zero components are the values the literal form produces under contraction.
template <int Fast> __device__ __forceinline__ v3 tangent_u(v3 w) {
    const bool sel = __builtin_fabsf(w.x) > 0.1f;
    if constexpr (Fast) {
        const float q = sel ? w.x : w.y;
        const float inv = dm::inversesqrt<Fast>(__builtin_fmaf(q, q, w.z * w.z));
        const float zi = w.z * inv, qi = q * inv;
        return v3{sel ? zi : 0.0f, sel ? 0.0f : -zi, sel ? -qi : qi};
    } else {
        return normalize<false>(cross(sel ? v3{0, 1, 0} : v3{1, 0, 0}, w));
    }
}
// normalize() of a combination a*u + b*v + c*w of an orthonormal basis with a^2 + b^2 + c^2 = 1 (the sampled directions of
// :413 and :428).  Strict: the literal normalize.  Fast (toleranced): the vector is already of unit length to within the
// accuracy of v_sin/v_cos/v_sqrt (~1e-6), which is what the rescaling would remove; it is used as it is.
template <int Fast, bool UnitBasis> __device__ __forceinline__ v3 normalize_unit_combination(v3 a) {
    if constexpr (Fast == 1 && UnitBasis && !MC_PT_FAST_RENORMALISE) return a;
    else return normalize<Fast>(a);   // (the careful tier re-normalises as the reference does: profiles/r05_fork_bias_identities.txt)
}
// reflect(I,N) = I - 2*dot(N,I)*N
__device__ __forceinline__ v3 reflect(v3 I, v3 N) { return I - N * (2.0f * dot(N, I)); }
__device__ __forceinline__ v3 select(bool c, v3 a, v3 b) { return v3{c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z}; }
__device__ __forceinline__ float comp(v3 a, int axis) { return axis == 0 ? a.x : (axis == 1 ? a.y : a.z); }

// rand01 — pathTracer.comp:107-110 (float(0xffffffffU) rounds to 2^32: the scale is exactly 2^-32)
__device__ __forceinline__ v3 rand01(uint32_t x, uint32_t y, uint32_t z) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
        uint32_t nx = ((x >> 8) ^ y) * 1103515245u;
        uint32_t ny = ((y >> 8) ^ z) * 1103515245u;
        uint32_t nz = ((z >> 8) ^ x) * 1103515245u;
        x = nx; y = ny; z = nz;
    }
    const float s = 2.3283064365386963e-10f;   // 2^-32
    return v3{(float)x * s, (float)y * s, (float)z * s};
}

// Diagnostic build only (make stats -> lib/libmc_compute_stats.so, tools/pool_region_stats.py): how often each code
// region is executed by a wave and with how many active lanes — the divergence picture behind DESIGN.md §3.3.
#ifdef MC_PT_REGION_STATS
static __device__ unsigned long long g_region_exec[32];
static __device__ unsigned long long g_region_lanes[32];
#define MC_REGION(r)                                                                            \
    do {                                                                                        \
        unsigned long long m_ = __ballot(1);                                                    \
        if ((int)__lane_id() == __ffsll((long long)m_) - 1) {                                   \
            atomicAdd(&g_region_exec[r], 1ull);                                                 \
            atomicAdd(&g_region_lanes[r], (unsigned long long)__popcll(m_));                    \
        }                                                                                       \
    } while (0)
// The counters are per translation unit (fast / careful / strict tier): each defines a reader with this macro, and
// mc_debug_pt_region_stats (pathtrace_fast.hip) sums the three — a request the host renders with the careful or the strict tier
// (four or more spheres, an enclosed light, an explicit mode) is counted too.
#define MC_PT_REGION_STATS_READER(name)                                                                              \
    int name(unsigned long long* exec32, unsigned long long* lanes32) {                                              \
        unsigned long long e[32], l[32], zero[32] = {0};                                                             \
        if (hipMemcpyFromSymbol(e, HIP_SYMBOL(mc::pt::g_region_exec), sizeof(zero)) != hipSuccess) return 3;         \
        if (hipMemcpyFromSymbol(l, HIP_SYMBOL(mc::pt::g_region_lanes), sizeof(zero)) != hipSuccess) return 3;        \
        if (hipMemcpyToSymbol(HIP_SYMBOL(mc::pt::g_region_exec), zero, sizeof(zero)) != hipSuccess) return 3;        \
        if (hipMemcpyToSymbol(HIP_SYMBOL(mc::pt::g_region_lanes), zero, sizeof(zero)) != hipSuccess) return 3;       \
        for (int i = 0; i < 32; i++) { exec32[i] += e[i]; lanes32[i] += l[i]; }                                      \
        return 0;                                                                                                    \
    }
#else
#define MC_REGION(r) do { } while (0)
#define MC_PT_REGION_STATS_READER(name)
#endif

// Fast mode only: MC_PT_FAST_CONTRACT selects where the compiler may contract a*b+c (pathtrace_fast.hip):
//   0 = nowhere (hardware transcendentals only), 1 = everywhere except the code that feeds the discrete decisions of a
//   path (intersect(): nearest hit / det < 0 / denom > triEps; the glass block: total internal reflection, rnd.x < P),
//   2 = everywhere.  The strict translation unit is built -ffp-contract=off and never defines MC_PT_DECISION_FP.
#ifndef MC_PT_DECISION_FP
#define MC_PT_DECISION_FP
#endif
#ifndef MC_PT_FAST_PLANES_ONE_RCP   // fast math: one division for the three slab tests (intersect_slab)
#define MC_PT_FAST_PLANES_ONE_RCP 1
#endif
// Fast-math forms that can be switched off one by one for A/B measurements (`make exp EXP_NAME=x EXP_FLAGS=-D<macro>=0`; tools/fork_census.py and tools/fast_tolerance_k2.py take such libraries):
#ifndef MC_PT_FAST_TANGENT_ONE_RSQ      // light sample: sin_a / |tangent| as one A * rsq(A * B)
#define MC_PT_FAST_TANGENT_ONE_RSQ 1
#endif
#ifndef MC_PT_FAST_PLANE_ID_FROM_SIGN   // slab winner's id from constant selects and the sign of its d_a
#define MC_PT_FAST_PLANE_ID_FROM_SIGN 1
#endif
#ifndef MC_PT_FAST_FRAME_NO_CROSS       // light sample: a t1 + b t2 + c sw without the cross product
#define MC_PT_FAST_FRAME_NO_CROSS 1
#endif
#ifndef MC_PT_FAST_LIGHT_DET_ONLY       // shadow_visible_disjoint: the light's own root test is det >= 0
#define MC_PT_FAST_LIGHT_DET_ONLY 1
#endif
#ifndef MC_PT_FAST_COLOUR_OVER_P        // pool kernel: colour / p as a record row
#define MC_PT_FAST_COLOUR_OVER_P 1
#endif
#ifndef MC_PT_FAST_OCC_MINUS_R2         // pool kernel: |c - x|^2 - r^2 formed once per bounce
#define MC_PT_FAST_OCC_MINUS_R2 1
#endif
#ifndef MC_PT_GENERIC_PREFETCH          // generic kernel reading the scene from memory: next sphere record fetched ahead
#define MC_PT_GENERIC_PREFETCH 1
#endif

constexpr float kEps = 1e-4f, kTriEps = 1e-7f, kInf = 1e20f;   // pathTracer.comp:103-105
constexpr float kPi = 3.141592653589793f;                       // :102
constexpr float kInvPi = 1.0f / kPi;                            // RN(1 / kPi): the y of dm::div_step for :422's accmat / pi
static_assert(kInvPi == 0x1.45f306p-2f, "1/pi must be the correctly rounded fp32 quotient (numpy: float32(1)/float32(pi) = 0x3ea2f983)");

// ---- extended-precision sphere test (pathTracer.comp:132-256; every variant is compiled OUT in the reference's
// default build, emulateDouble.h.glsl:13-26).  Prec: 1 = USE_NATIVE_FP64 (:139-142), 2 = DS_f32_f32 (:151-204),
// 3 = DF64_F32_F32 (:221-255).  Used with the TEST_PRECISION_WITH_LARGE_SPHERE_WALLS scene (pathtracerApp.h:28-35).
__device__ __forceinline__ bool needs_precision(const float* sp, v3 o) {     // :134-137 / :146-149 / :216-219
    const float maxLen = 500.0f;
    v3 c{sp[0], sp[1], sp[2]};
    v3 co = c - o;
    return sp[3] > maxLen || dot(c, c) > maxLen * maxLen || dot(o, o) > maxLen * maxLen || dot(co, co) > maxLen * maxLen;
}
// Returns false for the shader's `continue` (det < 0); otherwise dd = the value the shader assigns to `d`.
template <int Fast, int Prec>
__device__ __forceinline__ bool sphere_extended(const float* sp, float r2, v3 o, v3 d, float& dd) {
    auto rsq = [](float x) { return dm::inversesqrt<Fast>(x); };
    if (Prec == 1) {
        double ocx = (double)sp[0] - (double)o.x, ocy = (double)sp[1] - (double)o.y, ocz = (double)sp[2] - (double)o.z;
        double dx = d.x, dy = d.y, dz = d.z;
        double b = (ocx * dx + ocy * dy) + ocz * dz;
        double det = (b * b - ((ocx * ocx + ocy * ocy) + ocz * ocz)) + (double)r2;   // geo.w*geo.w is an fp32 product (:140)
        if (det < 0) return false;
        det = __builtin_sqrt(det);
        dd = (float)(b - det);
        if (!(dd > kEps)) { dd = (float)(b + det); if (!(dd > kEps)) dd = kInf; }
        return true;
    } else if (Prec == 2) {
        ds2 ocX = ds_add(ds_set(sp[0]), ds_set(-o.x)), ocY = ds_add(ds_set(sp[1]), ds_set(-o.y)), ocZ = ds_add(ds_set(sp[2]), ds_set(-o.z));
        ds2 rdX = ds_set(d.x), rdY = ds_set(d.y), rdZ = ds_set(d.z);
        ds2 b = ds_dot3(ocX, ocY, ocZ, rdX, rdY, rdZ);
        ds2 w = ds_set(sp[3]);
        ds2 det = ds_add(ds_sub(ds_mul(b, b), ds_dot3(ocX, ocY, ocZ, ocX, ocY, ocZ)), ds_mul(w, w));
        if (ds_compare(det, ds_set(0.0f)) < 0.0f) return false;
        det = ds_sqrt(det, rsq);
        ds2 eps_ds = ds_set(kEps);
        ds2 bMinus = ds_sub(b, det), bPlus = ds_add(b, det);
        ds2 d_ds = bMinus;
        dd = d_ds.hi;
        if (ds_compare(d_ds, eps_ds) <= 0.0f) {
            d_ds = bPlus;
            dd = d_ds.hi;
            if (ds_compare(d_ds, eps_ds) <= 0.0f) dd = kInf;
        }
        return true;
    } else {
        ds2 ocX = df64_add(df64_from_f32(sp[0]), df64_from_f32(-o.x)), ocY = df64_add(df64_from_f32(sp[1]), df64_from_f32(-o.y)),
            ocZ = df64_add(df64_from_f32(sp[2]), df64_from_f32(-o.z));
        ds2 rdX = df64_from_f32(d.x), rdY = df64_from_f32(d.y), rdZ = df64_from_f32(d.z);
        ds2 b = df64_dot3(ocX, ocY, ocZ, rdX, rdY, rdZ);
        ds2 w = df64_from_f32(sp[3]);
        ds2 det = df64_add(df64_add(df64_mult(b, b), df64_mult(df64_dot3(ocX, ocY, ocZ, ocX, ocY, ocZ), df64_from_f32(-1.0f))),
                           df64_mult(w, w));
        if (df64_lt(det, df64_from_f32(0.0f))) return false;
        det = df64_sqrt(det, rsq);
        float bMinus = df64_add(b, df64_mult(det, df64_from_f32(-1.0f))).hi;
        float bPlus = df64_add(b, det).hi;
        dd = bMinus;
        if (!(dd > kEps)) { dd = bPlus; if (!(dd > kEps)) dd = kInf; }
        return true;
    }
}

// ---- hot constants of the slab scene, held in VECTOR registers ----------------------------------------------------------
// profiles/r02_valu_microbench2.txt: on gfx950 a v_add/v_sub/v_mul/v_fmac whose operand is an SGPR issues in ~4.1 cycles, with
// VGPR or literal operands in ~2.3-2.5.  The compiler keeps uniform values (kernel arguments, hoisted literals such as 1e20f)
// in SGPRs, so every `centre - o`, `w - o[a]`, `det + r2` of the intersection code paid the slow form.  The 6 + 3 object
// scene fits in 25 registers per lane; the slab kernels have room for them (36 of the 64 VGPRs of full occupancy in use).
// The asm is only an opaque move: the compiler cannot fold the value back into an SGPR operand.  Same values, same results.
__device__ __forceinline__ float to_vgpr(float uniform) {
    float v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(uniform));
    return v;
}
template <int NS> struct HotSlabN {
    static constexpr int ns = NS;   // spheres of the scene (the six planes are the slab form)
    float W_pos[3], W_negm[3];   // w of the +e_a plane; MINUS w of the -e_a plane (see intersect_slab)
    float c[NS][3], r2[NS];      // sphere centres, radius^2 (the fp32 product of pathTracer.comp:318)
    float eps, tri_eps, inf;     // kEps, kTriEps, kInf
    // InVgpr = false leaves the values to the compiler (SGPR operands): for kernels without the register headroom.
    // WPosInVgpr: only W_pos in vector registers — the select `d_a > 0 ? W_pos : W_negm` then is ONE v_cndmask (SGPR, VGPR, vcc)
    // instead of two v_mov and a v_cndmask (a VOP3 select reads its mask and one more scalar at most).
    template <bool InVgpr = true, bool WPosInVgpr = false> __device__ __forceinline__ void load(const SceneArgs& sc) {
        auto put = [](float u) { return InVgpr ? to_vgpr(u) : u; };
#pragma unroll
        for (int a = 0; a < 3; a++) { W_pos[a] = (InVgpr || WPosInVgpr) ? to_vgpr(sc.slab_w_pos[a]) : sc.slab_w_pos[a]; W_negm[a] = put(-sc.slab_w_neg[a]); }
#pragma unroll
        for (int i = 0; i < NS; i++) {
#pragma unroll
            for (int k = 0; k < 3; k++) c[i][k] = put(sc.obj[12 * (6 + i) + k]);
            r2[i] = put(sc.r2[i]);
        }
        eps = put(1e-4f); tri_eps = put(1e-7f); inf = put(1e20f);
    }
};

// intersect() for the slab scene (6 axis-aligned planes in canonical order + 3 spheres), constants in VGPRs.
// Plane of axis a facing the ray: "pos" (normal +e_a) iff d[a] > 0, else "neg".  The reference quotient of the neg plane,
// (w_neg + o[a]) / |d[a]|, equals ((-w_neg) - o[a]) / d[a] bit for bit: negation is exact, (-x) - y = -(x + y) under
// round-to-nearest, and (-n) / (-d) = n / d (IEEE divide; v_rcp_f32 is odd, tested).  So both cases are
// (W - o[a]) / d[a] with W = pos ? w_pos : -w_neg: one select and one subtraction instead of two additions and a select.
// Closed (fast math, SceneArgs::box_ok, origin inside the box): the nearest facing plane IS a hit — the |d_a| > 1e-7 / t < 1e20
// tests of :119 / :336 can only fail for a ray that runs along a wall it starts on to within 1e-7, or a NaN ray (which then
// gathers nothing: every later comparison with its NaN t is false) — so they and the final "anything hit?" select are dropped.
// HaveOc: the caller passes c_i - o and its squared length (a COMPILE-TIME fact: a run-time "pointer given?" test on a private
// array is not foldable on this target — address 0 is a valid stack address — and kept the 8-sphere array in scratch memory).
template <int Fast, bool Closed, bool OccR2, int StatsBase, bool HaveOc, int NS>   // OccR2: occ[i] holds |c_i - o|^2 - r_i^2 (fast math)
__device__ __forceinline__ int intersect_slab_impl(const HotSlabN<NS>& h, v3 o, v3 d, float& t_out, bool shadow_skip_planes,
                                                   const float* occ,       // occ[i] = dot(c_i - o, c_i - o) and
                                                   const v3* oc_at_o) {    // oc_at_o[i] = c_i - o if the caller has them
    MC_PT_DECISION_FP
    static_assert(HaveOc || !OccR2, "the r^2-reduced squared lengths come from the caller");
    float t = h.inf;
    int id = -1;
    if (Fast && MC_PT_FAST_PLANES_ONE_RCP && !shadow_skip_planes) {
        // Fast math: the nearest of the three facing planes is found BEFORE dividing — (W_a - o_a) / d_a < (W_b - o_b) / d_b is
        // compared as |W_a - o_a| |d_b| < |W_b - o_b| |d_a| (numerator and denominator of a facing plane have the same sign) —
        // and only the winner is divided: one v_rcp_f32 (8 issue cycles, profiles/r03_valu_microbench3.txt) instead of three,
        // and t is the same product the three-division form computes for that plane.  A plane the ray runs parallel to
        // (|d_a| <= 1e-7, :119) loses every comparison against a real candidate; decisions differ from the reference's only
        // at ties of two planes' parameters (the edges of the box).
        float num[3], den[3];
        int pid[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float da = comp(d, a), oa = comp(o, a);
            const bool pos = da > 0.0f;
            num[a] = (pos ? h.W_pos[a] : h.W_negm[a]) - oa; den[a] = da; pid[a] = pos ? 2 * a + 1 : 2 * a;
        }
        float bn = num[0], bd = den[0];
        int bid = MC_PT_FAST_PLANE_ID_FROM_SIGN ? 1 : pid[0];
#pragma unroll
        for (int a = 1; a < 3; a++) {
            const bool nearer = __builtin_fabsf(num[a]) * __builtin_fabsf(bd) < __builtin_fabsf(bn) * __builtin_fabsf(den[a]);
            bn = nearer ? num[a] : bn; bd = nearer ? den[a] : bd; bid = nearer ? (MC_PT_FAST_PLANE_ID_FROM_SIGN ? 2 * a + 1 : pid[a]) : bid;
        }
        // (the winner's id 2a + (d_a > 0): the odd id of its axis, less the sign bit of its d_a — two selects of constants and two
        // integer operations instead of three selects, two ORs and two selects of the ids)
        if (MC_PT_FAST_PLANE_ID_FROM_SIGN) bid ^= (int)(dm::as_uint(bd) >> 31);
        const float dd = dm::fdiv<Fast>(bn, bd);
        if constexpr (Closed) { t = dd; id = bid; }
        else if (__builtin_fabsf(bd) > h.tri_eps && dd < t) { t = dd; id = bid; }
    } else if (!shadow_skip_planes) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float da = comp(d, a), oa = comp(o, a);
            const bool pos = da > 0.0f;
            const float W = pos ? h.W_pos[a] : h.W_negm[a];
            const float dd = dm::fdiv<Fast>(W - oa, da);
            if (__builtin_fabsf(da) > h.tri_eps && dd < t) { t = dd; id = pos ? 2 * a + 1 : 2 * a; }
        }
    }
    // (Round 3 also tried ONE root evaluation for all three spheres — every lane's first candidate sphere selected, then the block; a
    // second / third pass for the lanes with more candidates, skipped per wave — on the reasoning that the three per-sphere blocks
    // each serve a handful of lanes.  Bit-identical, 10 instructions fewer on paper, and slower: 14.96 against 14.69 ms.  The
    // per-sphere blocks are skipped for the whole wave more often than the merged one, and the selects are paid by every lane.)
#pragma unroll
    for (int i = 0; i < NS; i++) {
        v3 oc;                                                               // :317
        if constexpr (HaveOc) oc = oc_at_o[i]; else oc = v3{h.c[i][0], h.c[i][1], h.c[i][2]} - o;
        float b = dot(oc, d);                                                // :318
        float occ_i;
        if constexpr (HaveOc) occ_i = occ[i]; else occ_i = dot(oc, oc);
        float det = OccR2 ? b * b - occ_i : (b * b - occ_i) + h.r2[i];
        if (!(det < 0.0f)) {                                                 // :319
            if constexpr (StatsBase >= 0) MC_REGION(StatsBase + i);          // (diagnostic build: root block of sphere i)
            float sq = dm::fsqrt<Fast>(det);
            float dd = b - sq;                                               // :322,324
            if (dd <= h.eps) {                                               // :325
                dd = b + sq;                                                 // :323,326
                if (dd <= h.eps) dd = h.inf;                                 // :327
            }
            if (dd < t) { t = dd; id = 6 + i; }                              // :333
        }
    }
    t_out = t;
    if constexpr (Closed) return id;
    return (t < h.inf) ? id : -1;                                            // :336
}
template <int Fast, bool Closed = false, bool OccR2 = false, int StatsBase = -1, int NS = 3>
__device__ __forceinline__ int intersect_slab(const HotSlabN<NS>& h, v3 o, v3 d, float& t_out, bool shadow_skip_planes) {
    return intersect_slab_impl<Fast, Closed, OccR2, StatsBase, false, NS>(h, o, d, t_out, shadow_skip_planes, nullptr, nullptr);
}
template <int Fast, bool Closed = false, bool OccR2 = false, int StatsBase = -1, int NS = 3>
__device__ __forceinline__ int intersect_slab(const HotSlabN<NS>& h, v3 o, v3 d, float& t_out, bool shadow_skip_planes,
                                              const float* occ, const v3* oc_at_o) {
    return intersect_slab_impl<Fast, Closed, OccR2, StatsBase, true, NS>(h, o, d, t_out, shadow_skip_planes, occ, oc_at_o);
}

// Shadow ray of next-event estimation against the three spheres only (slab scenes whose walls can never win, see
// intersect()): "is sphere `li` the nearest hit?" (pathTracer.comp:420) without carrying t and id through the loop.
// The reference keeps a later candidate only if it is STRICTLY nearer, so sphere li wins iff its root dd_li is finite,
// strictly below every root of the spheres before it and not above any root of the spheres after it.  dd_k is the value the
// loop assigns for sphere k (:319-327), 1e20 when there is none.  Same comparisons on the same values: exact.
template <int Fast, bool OccR2 = false, int NS = 3>   // OccR2: occ[i] holds |c_i - o|^2 - r_i^2 (the fast pool kernel's form)
__device__ __forceinline__ bool shadow_reaches_sphere(const HotSlabN<NS>& h, v3 o, v3 d, int li, v3 oc_li, const float* occ) {
    MC_PT_DECISION_FP
    float dd[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) {
        // centre - o of the light (:408) and all three squared lengths were formed by the caller: the same operations
        const v3 oc = i == li ? oc_li : v3{h.c[i][0], h.c[i][1], h.c[i][2]} - o;   // :317
        float b = dot(oc, d);                                                // :318
        float det = OccR2 ? b * b - occ[i] : (b * b - occ[i]) + h.r2[i];
        float r = h.inf;
        if (!(det < 0.0f)) {                                                 // :319
            float sq = dm::fsqrt<Fast>(det);
            r = b - sq;                                                      // :322,324
            if (r <= h.eps) {                                                // :325
                r = b + sq;                                                  // :323,326
                if (r <= h.eps) r = h.inf;                                   // :327
            }
        }
        dd[i] = r;
    }
    // li is a constant of the caller's unrolled light loop.  Strictly below every EARLIER sphere's root, not above any LATER one's
    // (:333 keeps a later candidate only if it is strictly nearer), and finite.  (Fast, the last sphere: dd[k] <= 1e20 makes the
    // finiteness test implied when there is an earlier sphere to beat.)
    bool nearest = true;
#pragma unroll
    for (int k = 0; k < NS; k++) {
        if (k < li) nearest = nearest && dd[li] < dd[k];
        else if (k > li) nearest = nearest && !(dd[k] < dd[li]);
    }
    if (Fast && li == NS - 1 && NS > 1) return nearest;
    return nearest && dd[li] < h.inf;
}

// The same question — "is sphere li the nearest thing the shadow ray hits?" (:420) — for pairwise DISJOINT spheres (host-proved,
// SceneArgs::spheres_disjoint) in fast math, WITHOUT a square root (a v_sqrt_f32 among other instructions costs ~12 issue cycles,
// profiles/r03_valu_microbench7.txt; the three root blocks were a fifth of the diffuse bounce):
//  * which root of sphere k the loop of :319-:327 would keep follows from comparing squares.  With e = b - eps: the near root
//    b - sqrt(det) exceeds eps iff e > 0 and e^2 > det; otherwise the far root b + sqrt(det) exceeds eps iff e > 0 or det > e^2.
//    So sphere k yields a hit iff det >= 0 and (e > 0 or det > e^2);
//  * the parameter intervals [b - sqrt(det), b + sqrt(det)] in which a line runs inside two disjoint spheres are disjoint, so the
//    kept root of one sphere lies before the kept root of another iff its b (the interval's midpoint) is the smaller one — also when
//    the origin lies inside one of them (its kept root is then the END of an interval around 0 and every other interval that counts
//    lies beyond it).
// Sphere li is the nearest hit iff it yields a hit and no other sphere that yields a hit has a smaller b.  In exact arithmetic these are
// the decisions of shadow_reaches_sphere; in fp32 they differ where a root lies within rounding of eps or a ray grazes a sphere.
template <bool OccR2, int NS>   // OccR2: occ[i] holds |c_i - x|^2 - r_i^2
__device__ __forceinline__ bool shadow_visible_disjoint(const HotSlabN<NS>& h, v3 d, int li, const v3* xoc, const float* occ) {
    MC_PT_DECISION_FP
    float b[NS];
    bool hit[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) {
        b[i] = dot(xoc[i], d);                                               // :318
        const float det = OccR2 ? b[i] * b[i] - occ[i] : (b[i] * b[i] - occ[i]) + h.r2[i];
        // (the ray is aimed INTO the light's cone, :408-:413: its line meets the light by construction and the root :319-:327 keep lies
        // ahead — only det >= 0 is tested for it: the rim of the cone, where rounding decides, and a NaN direction, which must not pass)
        if (MC_PT_FAST_LIGHT_DET_ONLY && i == li) { hit[i] = det >= 0.0f; continue; }
        const float e = b[i] - h.eps;
        hit[i] = !(det < 0.0f) && (e > 0.0f || det > e * e);
    }
    bool vis = hit[li];
#pragma unroll
    for (int i = 0; i < NS; i++)
        if (i != li) vis = vis && !(hit[i] && b[i] < b[li]);
    return vis;
}

// ---- closed-box kernels (fast math only, Box = true) ---------------------------------------------------------------------
// The host proved (SceneArgs::box_ok: closed box, camera and lights inside with a margin, every material 1..3, no wall of
// glass, colours in [0, 1]) what the general slab kernel tests at run time: materials_known, nee_skip_planes and emit_skip_ok
// are compile-time truths here, which removes their selects and one copy of the shadow-ray code (K2 19.37 -> 19.10 ms).
// (Round 3 also tried a branch-free "packed minimum" intersection for these kernels — candidates t - eps as unsigned bit
// patterns with the object id in the low mantissa bits, one v_min3_u32 tree, square roots taken unconditionally.  4 % fewer
// instructions, yet 20.1-21.0 ms and outside the tolerance (id bits in t: rmse 0.27 / p99.9 4.18): gfx950 issues every
// non-transcendental VALU instruction in ~2.4 cycles whatever its class, a transcendental blocks the SIMD for 8.2, and the
// compiler's exec-mask branches skip whole sphere blocks for a wave none of whose lanes can hit — profiles/r03_valu_microbench3.txt,
// profiles/r03_box_experiment_*.  Removed.)

// intersect — pathTracer.comp:112-131 + :316-341.  Returns the hit object id (planes 0..NP-1, spheres
// NP..NP+NS-1) or -1, and the ray parameter.  NP/NS < 0 select run-time counts; `obj` is the record array the
// loops read with wave-uniform indices: the kernel-argument copy (SGPR operands) for the specialised kernels,
// the LDS copy (broadcast ds_reads) for the generic ones.
//
// `shadow_skip_planes` (wave-uniform, slab kernels, shadow rays only): the caller uses nothing but "is the nearest hit
// sphere i".  All path vertices lie in the closed box (every ray that starts inside faces the wall it is heading to),
// the light lies inside it with a margin m, and a box is convex: a ray that hits the light does so at least m before
// it reaches any wall, orders of magnitude beyond fp32 rounding of either quotient, so no plane can win `dd < t`
// against that hit; when the ray misses the light the answer is "not sphere i" with or without the planes.  The
// sphere loop is unchanged, hence the same id among spheres.
template <int Fast, int NP, int NS, bool Slab, int Prec>
__device__ __forceinline__ int intersect(const SceneArgs& sc, const float* __restrict__ obj, v3 o, v3 d, float& t_out,
                                         bool shadow_skip_planes = false) {
    MC_PT_DECISION_FP
    constexpr bool LdsScene = NP < 0;
    const int np = NP >= 0 ? NP : (int)sc.n_planes;
    const int ns = NS >= 0 ? NS : (int)sc.n_spheres;
    float t = kInf;
    int id = -1;
    if (Slab && shadow_skip_planes) {
        // nothing: see above
    } else if (Slab) {
        // Axis-aligned planes.  For n = s*e_a (s = +-1, other components +-0) and a finite ray:
        //   denom = dot(d,n) = s*d[a] exactly;  t = (w - dot(o,n)) / denom = (w - s*o[a]) / (s*d[a]).
        // The "pos" plane faces the ray iff d[a] > triEps, the "neg" plane iff -d[a] > triEps: never both.
        // pos: (w_pos - o[a]) / d[a];  neg: (w_neg - (-o[a])) / (-d[a]) = (w_neg + o[a]) / |d[a]|.
        // Axes are visited in plane-index order (host-checked), so `dd < t` resolves ties as the
        // reference's loop over planes does.  A NaN ray fails every test in both formulations.
        // The host stores the six records in canonical slab order (x-,x+,y-,y+,z-,z+), so INSIDE the kernel the
        // id of the plane of axis a that can face the ray is the compile-time pattern 2a + (d[a] > 0); ids never
        // leave the kernel (they only index the record copy and tell planes from spheres).
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float da = comp(d, a), oa = comp(o, a);
            const bool pos = da > 0.0f;
            const float den = __builtin_fabsf(da);
            const float num = pos ? (sc.slab_w_pos[a] - oa) : (sc.slab_w_neg[a] + oa);
            const float dd = dm::fdiv<Fast>(num, den);
            if (den > kTriEps && dd < t) { t = dd; id = pos ? 2 * a + 1 : 2 * a; }
        }
    } else {
#pragma unroll
        for (int i = 0; i < np; i++) {
            const float* pl = obj + 12 * i;
            v3 n{pl[0], pl[1], pl[2]};
            float denom = dot(d, n);                                         // :118
            if (denom > kTriEps) {                                           // :119
                float dd = dm::fdiv<Fast>(pl[3] - dot(o, n), denom);         // :120
                if (dd < t) { t = dd; id = i; }                              // :121
            }
        }
    }
    // Scenes read from memory (NP = -2): the next sphere's (centre, radius) is fetched — one scalar load — before the current one is
    // tested, so its latency hides behind the ~15 instructions of a sphere test instead of stalling the wave at the top of every
    // iteration: 5-11 % at 1500 spheres.  (Measured the other way for the LDS copy — 5-10 % SLOWER at 64 and 512 spheres: its
    // broadcast reads are short enough already — so only here.)  Same values, same order.
    constexpr bool Prefetch = NP == -2 && MC_PT_GENERIC_PREFETCH;
    float4 nxt = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (Prefetch && ns > 0) nxt = *reinterpret_cast<const float4*>(obj + 12 * np);
#pragma unroll
    for (int i = 0; i < ns; i++) {
        const float* sp = obj + 12 * (np + i);
        float4 cur;
        if (Prefetch) {
            cur = nxt;
            const int j = (i + 1 < ns) ? i + 1 : i;
            nxt = *reinterpret_cast<const float4*>(obj + 12 * (np + j));
        } else {
            cur = make_float4(sp[0], sp[1], sp[2], sp[3]);
        }
        const float r2 = LdsScene ? cur.w * cur.w : sc.r2[i];   // the fp32 product of :318 either way
        bool extended = false;
        if (Prec != 0) {
            extended = needs_precision(sp, o);
            if (extended) {
                float dd;
                if (sphere_extended<Fast, Prec>(sp, r2, o, d, dd) && dd < t) { t = dd; id = np + i; }   // :333
            }
        }
        v3 oc = v3{cur.x, cur.y, cur.z} - o;                                 // :317
        float b = dot(oc, d);                                                // :318
        float det = (b * b - dot(oc, oc)) + r2;
        if (!extended && !(det < 0.0f)) {                                    // :319
            float sq = dm::fsqrt<Fast>(det);
            float dd = b - sq;                                               // :322,324
            if (dd <= kEps) {                                                // :325
                dd = b + sq;                                                 // :323,326
                if (dd <= kEps) dd = kInf;                                   // :327
            }
            if (dd < t) { t = dd; id = np + i; }                             // :333
        }
    }
    t_out = t;
    return (t < kInf) ? id : -1;                                             // :336
}

// Stages the 12-float records into LDS for the per-lane material fetch and replaces, in that copy only, three slots by
// values every bounce would otherwise re-derive: [7] (e.w, unused by the shader) = max(max(c.x, c.y), c.z) (:394), [11] =
// floor(m + 0.5) as a float (:378/:384), and — `emits_in_slot3`, slab kernels, whose intersection code never reads slot 3 of
// this copy — [3] = 1 if the object emits (any e component non-zero) else 0.  Call with all threads; ends with a barrier.
__device__ __forceinline__ void stage_records(float* lds_obj, const float* __restrict__ src, uint32_t n_obj, bool emits_in_slot3) {
    for (uint32_t i = threadIdx.x; i < n_obj * 12u; i += blockDim.x) lds_obj[i] = src[i];
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < n_obj; k += blockDim.x) {
        float* o = lds_obj + 12u * k;
        o[7] = dm::gmax(dm::gmax(o[8], o[9]), o[10]);
        o[11] = __builtin_floorf(o[11] + 0.5f);
        if (emits_in_slot3) o[3] = (o[4] != 0.0f || o[5] != 0.0f || o[6] != 0.0f) ? 1.0f : 0.0f;
    }
    __syncthreads();
}

// ---- pieces of one sample shared by the round-synchronous kernels (trace_sample) and the sample-pool kernel (pathtrace_pool.h) --
// Camera ray through the sensor sample of (pixel, samp) — pathTracer.comp:357-362; returns the direction (the origin is a.lc).
template <int Fast> __device__ __forceinline__ v3 camera_ray(const PTArgs& a, uint32_t gx, uint32_t gy, uint32_t samp) {
    v3 r0 = rand01(gx, gy, samp);
    float rnd2x = 2.0f * r0.x, rnd2y = 2.0f * r0.y;
    // :358-:359 tent filter: one sqrt of the selected argument instead of one per branch (the same value either way)
    const float sqx = dm::fsqrt<Fast>(rnd2x < 1.0f ? rnd2x : 2.0f - rnd2x), sqy = dm::fsqrt<Fast>(rnd2y < 1.0f ? rnd2y : 2.0f - rnd2y);
    float tentx = rnd2x < 1.0f ? sqx - 1.0f : 1.0f - sqx;
    float tenty = rnd2y < 1.0f ? sqy - 1.0f : 1.0f - sqy;
    float stratx = (float)((samp / 2u) % 2u), straty = (float)(samp % 2u);
    const float px = (float)gx + 0.5f * ((0.5f + stratx) + tentx), py = (float)gy + 0.5f * ((0.5f + straty) + tenty);
    float sx = ((Fast ? px * a.inv_W : dm::fdiv<Fast>(px, (float)a.W)) - 0.5f) * 0.036f;
    float sy = ((Fast ? py * a.inv_H : dm::fdiv<Fast>(py, (float)a.H)) - 0.5f) * 0.024f;
    v3 spos = (a.cam_o + a.cx * sx) + a.cy * sy;                          // :360
    return normalize<Fast>(a.lc - spos);                                  // :362
}
// Direction towards a point of the light's visible cap (:408-:413): xc = light centre - x, xcc = |xc|^2, lr2 = radius^2.
template <int Fast> __device__ __forceinline__ v3 light_sample_direction(v3 xc, float xcc, float lr2, v3 rnd, float& cos_a_max) {
    const float inv_len = dm::inversesqrt<Fast>(xcc);
    v3 sw = xc * inv_len;                                     // :409 normalize(xc)
    if constexpr (Fast && MC_PT_FAST_TANGENT_ONE_RSQ) {
        // The tangents are left UNNORMALISED — t1 = cross(axis, sw) (tangent_u before its scaling), t2 = cross(sw, t1), both of
        // length k = sqrt(q^2 + sw.z^2) — and sin_a / k is formed as one factor: sqrt(A / B) = A * rsq(A * B) with A = sin_a^2 =
        // 1 - cos_a^2, B = k^2: one transcendental where 1 / k and sin_a took two (each ~12 issue cycles among other instructions,
        // profiles/r03_valu_microbench7.txt).  A is kept above zero: rnd.x below ~1e-5 rounds cos_a to 1 and 0 * rsq(0) is a NaN.
        const bool sel = __builtin_fabsf(sw.x) > 0.1f;
        const float q = sel ? sw.x : sw.y;
        const float B = __builtin_fmaf(q, q, sw.z * sw.z);
        cos_a_max = dm::fsqrt<Fast>(1.0f - lr2 * (inv_len * inv_len));        // :410
        const float cos_a = (1.0f - rnd.x) + rnd.x * cos_a_max;               // :411
        const float A = __builtin_fmaxf(1.0f - cos_a * cos_a, 1e-30f);
        const float g = A * dm::inversesqrt<Fast>(A * B);
        float sphi, cphi;
        dm::sincos_angle<Fast>(0.0f, rnd.y, sphi, cphi);                      // :412
        if (MC_PT_FAST_FRAME_NO_CROSS) {
            // t1 = (z, 0, -x) or (0, -z, y) has a zero component and t2 = cross(sw, t1) = (-xy, B, -yz) or (B, -xy, -xz) carries B
            // itself, so a t1 + b t2 + c sw needs no cross product: with m = the axis component (y or x), q the other one,
            // cm = c - b m and sa = +-a:  special component = b B + c m,  the other two = sa z + q cm  and  -sa q + z cm.
            const float a = cphi * g, b = sphi * g, c = cos_a;
            const float m = sel ? sw.y : sw.x;
            const float cm = __builtin_fmaf(-b, m, c);
            const float sp = __builtin_fmaf(b, B, c * m);
            const float sa = sel ? a : -a;
            const float first = __builtin_fmaf(sa, sw.z, q * cm), lz = __builtin_fmaf(-sa, q, sw.z * cm);
            const v3 lv{sel ? first : sp, sel ? sp : first, lz};              // :413
            if constexpr (Fast == 2 || MC_PT_FAST_RENORMALISE != 0) return normalize<Fast>(lv);
            return lv;
        }
        const v3 t1{sel ? sw.z : 0.0f, sel ? 0.0f : -sw.z, sel ? -q : q};
        const v3 t2 = cross(sw, t1);
        const v3 lv = (t1 * (cphi * g) + t2 * (sphi * g)) + sw * cos_a;       // :413
        if constexpr (Fast == 2 || MC_PT_FAST_RENORMALISE != 0) return normalize<Fast>(lv);
        return lv;
    }
    v3 su = tangent_u<Fast>(sw);
    v3 sv = cross(sw, su);
    // :410; fast: 1 / |xc|^2 is the square of the 1 / |xc| above (one multiply instead of a v_rcp_f32)
    cos_a_max = dm::fsqrt<Fast>(1.0f - (Fast ? lr2 * (inv_len * inv_len) : dm::fdiv<Fast>(lr2, xcc)));
    float cos_a = (1.0f - rnd.x) + rnd.x * cos_a_max;         // :411
    float sin_a = dm::fsqrt<Fast>(1.0f - cos_a * cos_a);
    float phi = (2.0f * kPi) * rnd.y;                         // :412
    float sphi, cphi;
    dm::sincos_angle<Fast>(phi, rnd.y, sphi, cphi);
    // fast: the two scalar factors of a tangent are multiplied first (u*(c*s) for (u*c)*s: one product less per component)
    return Fast ? normalize_unit_combination<Fast, true>((su * (cphi * sin_a) + sv * (sphi * sin_a)) + sw * cos_a)
                : normalize_unit_combination<Fast, true>(((su * cphi) * sin_a + (sv * sphi) * sin_a) + sw * cos_a);   // :413
}
// Cosine-weighted bounce around w = nl (:426-:428).  Unit: w is of unit length (slab kernels: +-1 axis normals, normalised sphere
// normals); a generic scene's plane normal is used as given, and there :428's normalize is not an identity.
template <int Fast, bool Unit> __device__ __forceinline__ v3 cosine_bounce(v3 w, v3 rnd) {
    float r1 = (2.0f * kPi) * rnd.x, r2 = rnd.y, r2s = dm::fsqrt<Fast>(r2);   // :426
    v3 u = tangent_u<Fast>(w);                                    // :427
    v3 v = cross(w, u);
    float s1, c1;
    dm::sincos_angle<Fast>(r1, rnd.x, s1, c1);
    return Fast ? normalize_unit_combination<Fast, Unit>((u * (c1 * r2s) + v * (s1 * r2s)) + w * dm::fsqrt<Fast>(1.0f - r2))
                : normalize_unit_combination<Fast, Unit>(((u * c1) * r2s + (v * s1) * r2s) + w * dm::fsqrt<Fast>(1.0f - r2));   // :428
}
// The same bounce around the inward normal of an axis-aligned WALL of a slab scene: w = sigma * e_a, so the basis of :427 is two
// signed axis vectors — u = normalize(cross(axis, w)) has one component -+sigma and two zeros, its length factor is the (correctly
// rounded, or v_rsq_f32's) 1 / sqrt(1) = 1, v = cross(w, u) likewise — and :428's combination ((u*c1)*r2s + (v*s1)*r2s) + w*C is a
// signed permutation of (c1*r2s, s1*r2s, C = sqrt(1 - r2)): every component is the one non-zero term of that expression plus two
// products with exact zeros, i.e. the SAME VALUE (worked out from tangent_u / cross for the three axes below), without the basis:
// no reciprocal square root, no cross products, no 3 x 3 combination.  Strict mode then applies :428's normalize to it as before.
// (Only the sign of a component that is itself a zero — rnd.y = 1, or a sine / cosine of exactly 0 — can differ from the general
// form's; such a component multiplies t and is added to a non-zero coordinate, or fails the |d_a| > 1e-7 test of :119.)
// `id` = the wall's slab id 2a + (normal is +e_a); a hit wall faces the ray (:119: dot(d, n) > 0), so nl = -n and sigma is
// negative exactly for the odd ids.   a = 0: (sC, B, -sA)   a = 1: (B, sC, sA)   a = 2: (B, -sA, sC)   with sX = sigma * X.
template <int Fast> __device__ __forceinline__ v3 cosine_bounce_wall(int id, v3 rnd) {
    const float r1 = (2.0f * kPi) * rnd.x, r2 = rnd.y, r2s = dm::fsqrt<Fast>(r2);   // :426
    float s1, c1;
    dm::sincos_angle<Fast>(r1, rnd.x, s1, c1);
    const float A = c1 * r2s, B = s1 * r2s, C = dm::fsqrt<Fast>(1.0f - r2);
    const uint32_t sb = (uint32_t)id << 31;
    const float sC = dm::as_float(dm::as_uint(C) ^ sb), sA = dm::as_float(dm::as_uint(A) ^ sb);
    const float nsA = dm::as_float(dm::as_uint(sA) ^ 0x80000000u);
    const bool a0 = id < 2, a1 = id < 4;   // (a1 is read only where a0 is false)
    const v3 d{a0 ? sC : B, a0 ? B : (a1 ? sC : nsA), a0 ? nsA : (a1 ? sA : sC)};
    return normalize_unit_combination<Fast, true>(d);                                // :428
}

// Mirror / glass bounce in the fast slab form (:432-:447): every outcome is rd*alpha + n*beta — reflection (1, -2 dot(n, rd)),
// refraction (nnt, -k) — so the scalars are selected and ONE direction is formed; cos of the leaving ray = sqrt(cos2t), c^5
// through c^2.  mat is 2 or 3; dot_n_rd = dot(n, rd); rx = rnd.x; accmat receives :445's weight.
template <int Fast> __device__ __forceinline__ v3 specular_bounce_fast(int mat, v3 rd, v3 n, float dot_n_rd, float rx, v3& accmat) {
    MC_PT_DECISION_FP
    float alpha = 1.0f, beta = -2.0f * dot_n_rd;
    if (mat == 3) {
        MC_REGION(6);
        const bool into = (dm::as_uint(dot_n_rd) >> 31) != 0u;            // :438 (nl == n)
        const float nnt = into ? 1.0f / 1.5f : 1.5f;                      // :439
        const float a_dn = __builtin_fabsf(dot_n_rd);                      // = -dot(rd, nl)
        const float cos2t = 1.0f - (nnt * nnt) * (1.0f - a_dn * a_dn);    // :440
        if (cos2t >= 0.0f) {
            MC_REGION(7);
            const float sq2t = dm::fsqrt<Fast>(cos2t);
            const float k = (into ? 1.0f : -1.0f) * (sq2t - a_dn * nnt);  // :441
            const float c = 1.0f - (into ? a_dn : sq2t), c2 = c * c;
            const float Re = 0.04f + 0.96f * ((c2 * c2) * c);              // :442-:443, R0 = (0.5/2.5)^2
            const float P = 0.25f + 0.5f * Re;
            const bool pick_refl = rx < P;                                // :444
            accmat = accmat * dm::fdiv<Fast>(pick_refl ? Re : 1.0f - Re, pick_refl ? P : 1.0f - P);   // :445
            if (!pick_refl) { alpha = nnt; beta = -k; }
        }
    }
    const v3 out = rd * alpha + n * beta;
    // Careful tier: ONE direction is formed for all three outcomes, so the re-normalisation applies to all of them — the refracted ray,
    // which the reference normalises (:441), and the two reflections (:433, :446), which it leaves as reflect() returns them (unit to
    // within a rounding for unit rd and n: the extra normalize moves them by an ulp or so, inside the tier's tolerance).
    if constexpr (Fast == 2 || MC_PT_FAST_RENORMALISE != 0) return normalize<Fast>(out);
    return out;
}

// Mirror / glass bounce in the general form (:432-:447) — the strict kernels (literal operation order) and the fast generic
// kernel.  mat is 2 or 3; nl as :390; dot_n_rd = dot(n, rd) (fast only); rx = rnd.x; accmat receives :445's weight.
template <int Fast, bool Slab>
__device__ __forceinline__ v3 specular_bounce_general(int mat, v3 rd, v3 n, v3 nl, float dot_n_rd, float rx, v3& accmat) {
    MC_PT_DECISION_FP
    const v3 refl = reflect(rd, n);
    if (mat == 3) {
        MC_REGION(6);   // glass
        // fast: nl is n exactly when dot(n, rd) carries a sign bit, and dot(rd, nl) is then -|dot(n, rd)|
        bool into = Fast ? (dm::as_uint(dot_n_rd) >> 31) != 0u : (n.x == nl.x) && (n.y == nl.y) && (n.z == nl.z);  // :438
        const float nc = 1.0f, nt = 1.5f;
        float nnt = into ? dm::fdiv<Fast>(nc, nt) : dm::fdiv<Fast>(nt, nc);   // :439
        float ddn = Fast ? -__builtin_fabsf(dot_n_rd) : dot(rd, nl);
        float cos2t = 1.0f - (nnt * nnt) * (1.0f - ddn * ddn);        // :440
        if (cos2t >= 0.0f) {
            MC_REGION(7);   // glass: refraction branch
            const float sq2t = dm::fsqrt<Fast>(cos2t);
            float k = (into ? 1.0f : -1.0f) * (ddn * nnt + sq2t);
            v3 tdir = normalize_unit_combination<Fast, Slab>(rd * nnt - n * k);   // :441 (unit by Snell's law when rd, n are)
            float aa = nt - nc, bb = nt + nc;
            float R0 = dm::fdiv<Fast>(aa * aa, bb * bb);              // :442
            // fast: dot(tdir, n) of the leaving ray is sqrt(cos2t) in exact arithmetic (tdir = rd*nnt - n*k, n = -nl)
            float c = 1.0f - (into ? -ddn : (Fast && Slab ? sq2t : dot(tdir, n)));
            float Re;                                                 // :443 R0 + (1 - R0) c^5
            if constexpr (Fast) { const float c2 = c * c; Re = R0 + (1.0f - R0) * ((c2 * c2) * c); }   // c^5 through c^2, c^4
            else Re = R0 + (((((1.0f - R0) * c) * c) * c) * c) * c;
            float Tr = 1.0f - Re;
            float P = 0.25f + 0.5f * Re;
            bool pick_refl = rx < P;
            // :442's RP = Re / P and TP = Tr / (1 - P): only the one :445 uses is formed (the same quotient)
            const float weight = dm::fdiv<Fast>(pick_refl ? Re : Tr, pick_refl ? P : 1.0f - P);
            rd = select(pick_refl, refl, tdir);                       // :444
            accmat = accmat * weight;                                 // :445
        } else {
            rd = refl;                                                // :446
        }
    } else {
        rd = refl;                                                    // :433
    }
    return rd;
}

// Spheres of the slab kernels' VGPR-resident scene copy (1 for the generic kernels, which have none).
template <bool Slab, int NS> constexpr int slab_spheres() { return Slab && NS > 0 ? NS : 1; }

// One sample: returns accrad (pathTracer.comp:356-449).  Box (fast math, slab scenes with SceneArgs::box_ok): see above.
template <int Fast, int NP, int NS, bool Slab, int Prec, bool Box = false>
__device__ __forceinline__ v3 trace_sample(const PTArgs& a, const float* __restrict__ lds_obj,
                                           const uint32_t* __restrict__ lds_emissive, const HotSlabN<slab_spheres<Slab, NS>()>& hot,
                                           uint32_t gx, uint32_t gy, uint32_t samp) {
    constexpr int HS = slab_spheres<Slab, NS>();
    const SceneArgs& sc = a.scene;
    constexpr bool LdsScene = NP < 0;
    const float* __restrict__ uobj = LdsScene ? lds_obj : sc.obj;   // records read with wave-uniform indices
    const int np = NP >= 0 ? NP : (int)sc.n_planes;
    const int ns = NS >= 0 ? NS : (int)sc.n_spheres;
    // -- sample sensor (:357-362)
    v3 accrad{0.0f, 0.0f, 0.0f}, accmat{1.0f, 1.0f, 1.0f};               // :361
    v3 ro = a.lc, rd = camera_ray<Fast>(a, gx, gy, samp);                 // :362
    float emissive = 1.0f;                                                // :365
    // slab kernels: |c_i - ro|^2 of the three spheres travels with the ray origin.  Every material continues from the hit
    // point x (:429,:434,:447), where the shadow rays of next-event estimation start too, so the value :318 needs at the next
    // depth is the one formed at x in this one: once per bounce instead of twice (same operations, same operands).
    // The slab kernels' loop is rotated: the intersection of depth k + 1 is computed at the end of depth k, where c_i - x is still
    // in registers, so those nine subtractions are made once per bounce too.
    float occ[HS];
#pragma unroll
    for (int i = 0; i < HS; i++) occ[i] = 0.0f;
    float t = 0.0f;
    int id = -1;
    static_assert(!Box || (Fast && Slab), "the closed-box specialisation is a fast-math slab kernel");
    MC_REGION(0);   // ray generation done
    if constexpr (Slab) {
        MC_REGION(1);   // primary intersect (camera ray)
        v3 oc0[HS];
#pragma unroll
        for (int i = 0; i < HS; i++) { oc0[i] = v3{a.cam_oc[i][0], a.cam_oc[i][1], a.cam_oc[i][2]}; occ[i] = a.cam_occ[i]; }
        if (a.max_depth != 0u) id = intersect_slab<Fast, Box>(hot, ro, rd, t, false, occ, oc0);   // (Box: closed-box form, see intersect_slab)
    }
    for (uint32_t depth = 0; depth < a.max_depth; depth++) {              // :367
        if constexpr (!Slab) {
            MC_REGION(1);   // primary intersect
            id = intersect<Fast, NP, NS, Slab, Prec>(sc, uobj, ro, rd, t);
        }
        if (id < 0) break;   // :369 `continue` with an unchanged ray misses again at every later depth: no effect
        MC_REGION(2);   // bounce prologue
        v3 x = ro + rd * t;                                               // :374 (o + t*d: fp32 mul is commutative)
        v3 xoc[HS];                                                       // c_i - x (:317 at the next depth, :408 now)
        if constexpr (Slab) {
#pragma unroll
            for (int i = 0; i < HS; i++) { xoc[i] = v3{hot.c[i][0], hot.c[i][1], hot.c[i][2]} - x; occ[i] = dot(xoc[i], xoc[i]); }
        }
        const float* obj = lds_obj + 12 * id;                             // per-lane fetch from LDS
        const bool is_sphere = id >= np;
        v3 geo{obj[0], obj[1], obj[2]};
        v3 col{obj[8], obj[9], obj[10]};
        // per-object values derived once per block while the records are staged into LDS (stage_records): the same
        // fp32 operations on the same operands, so the same values as evaluating them here at every bounce
        const int mat = (int)obj[11];                                     // = int(floor(m + 0.5)), :378/:384
        const float p = obj[7];                                           // = max(max(c.x, c.y), c.z), :394
        // (fast mode keeps this normalize: a grazing hit's t = b - sqrt(det) cancels, x leaves the sphere by far more than an
        // ulp and (x - c) / r is visibly wrong: rmse 0.40 / p99.9 7.4 against the 0.5 / 4 bound, measured)
        v3 n = is_sphere ? normalize<Fast>(x - geo) : geo;                // :381/:387
        float dot_n_rd = 0.0f;                                            // fast: kept for the glass block
        v3 nl;                                                            // :390 nl = dot(n, rd) < 0 ? n : -n
        if constexpr (Fast) {   // the sign bit of the dot product, inverted, flips n: 5 two-cycle integer operations instead of a
                                // compare and three selects (differs from `<` only for a dot product of exactly -0)
            dot_n_rd = dot(n, rd);
            const uint32_t flip = ~dm::as_uint(dot_n_rd) & 0x80000000u;
            nl = v3{dm::as_float(dm::as_uint(n.x) ^ flip), dm::as_float(dm::as_uint(n.y) ^ flip), dm::as_float(dm::as_uint(n.z) ^ flip)};
        } else {
            nl = dot(n, rd) < 0.0f ? n : -n;
        }
        // :391 accrad += accmat * e * emissive.  For an object without emission (e = +-0) the product is a zero and
        // accrad (never -0: it starts at +0 and only receives sums) is unchanged, so the nine operations are skipped when no
        // lane of the wave hit an emitter — almost always (slab kernels; the flag sits in the record's unused slot 3).
        // (the host proves the premise — accmat finite: colours in [0, 1], emit_skip_ok — else inf * 0 = NaN must be formed)
        if (!Slab || !(Box || sc.emit_skip_ok) || __ballot(obj[3] != 0.0f) != 0ull) {
            v3 emi{obj[4], obj[5], obj[6]};
            accrad = accrad + (accmat * emi) * emissive;
        }
        accmat = accmat * col;                                            // :392
        v3 rnd = rand01(gx, gy, samp * a.max_depth + depth);              // :393
        if (depth > 5) {                                                  // :395
            if (rnd.z >= p) break;                                        // :396
            accmat = divs<Fast>(accmat, p);                               // :397
        }
        // every material 1..3 continues from x (:429,:434,:447): with all materials known (uniform) the assignment is made once,
        // ahead of the dispatch, so that no branch has to copy it at the merge
        if constexpr (Slab) { if (Box || sc.materials_known) ro = x; }
        if (mat == 1) {                                                   // :400 diffuse
            MC_REGION(3);   // diffuse: NEE set-up + shadow ray
            const int n_lights = LdsScene ? (int)sc.n_emissive : ns;
            // :422's accmat / pi.  Strict: formed here, once per diffuse bounce, not inside the `reached` block of every light —
            // the guarded short division's never-taken IEEE branch in that innermost block cost 3 ms at K2 (measured), here nothing;
            // the quotient of a light that is not reached is simply not used.
            v3 accmat_over_pi{0.0f, 0.0f, 0.0f};
            if constexpr (!Fast) accmat_over_pi = divs_recip<Fast>(accmat, kPi, kInvPi);
#pragma unroll
            for (int k = 0; k < (Slab ? HS : n_lights); k++) {            // :403 (slab: unrolled, the index is a constant)
                int i = k;
                if (LdsScene) i = (int)lds_emissive[k];                   // host-built list of the spheres passing :407
                else if (!((sc.emissive_mask >> i) & 1u)) continue;       // :407 (uniform)
                const float* ls = uobj + 12 * (np + i);
                float lr2;
                v3 lc;
                if constexpr (Slab) { lr2 = hot.r2[i]; lc = v3{hot.c[i][0], hot.c[i][1], hot.c[i][2]}; }   // the VGPR copies
                else { lr2 = LdsScene ? ls[3] * ls[3] : sc.r2[i]; lc = v3{ls[0], ls[1], ls[2]}; }
                v3 le{ls[4], ls[5], ls[6]};
                v3 xc = Slab ? xoc[Slab ? i : 0] : lc - x;                // :408
                const float xcc = Slab ? occ[Slab ? i : 0] : dot(xc, xc);
                float cos_a_max;
                v3 l = light_sample_direction<Fast>(xc, xcc, lr2, rnd, cos_a_max);   // :409-:413
                float tne;
                bool reached;                                             // :420 shadow ray: is the nearest hit sphere i?
                if constexpr (Slab) {
                    // (closed-box kernel, disjoint spheres — a uniform scene fact: the root-free test of the sample-pool kernel)
                    if (Box && sc.spheres_disjoint != 0u && !MC_PT_FAST_NO_DISJOINT) reached = shadow_visible_disjoint<false>(hot, l, i, xoc, occ);
                    else if (Box || sc.nee_skip_planes != 0u) reached = shadow_reaches_sphere<Fast>(hot, x, l, i, xc, occ);
                    else reached = intersect_slab<Fast>(hot, x, l, tne, false) == np + i;
                } else {
                    reached = intersect<Fast, NP, NS, Slab, Prec>(sc, uobj, x, l, tne, false) == np + i;
                }
                if (reached) {
                    MC_REGION(4);   // shadow ray reached the light
                    float omega = (2.0f * kPi) * (1.0f - cos_a_max);      // :421
                    if constexpr (Fast) {   // the scalar factors of :422 are multiplied first; omega / pi = 2 (1 - cos_a_max)
                        const float scale = __builtin_fmaxf(dot(l, nl), 0.0f) * (2.0f - (cos_a_max + cos_a_max));
                        accrad = accrad + (accmat * le) * scale;
                    } else {
                        accrad = accrad + ((accmat_over_pi * dm::gmax(dot(l, nl), 0.0f)) * le) * omega;   // :422
                    }
                }
            }
            MC_REGION(8);   // diffuse bounce direction
            // :426-:428.  Slab kernels: a bounce off a wall (axis-aligned, facing the ray: nl = -n) needs no tangent basis — the
            // same values, see cosine_bounce_wall; the general form only when some lane of the wave bounces off a diffuse sphere.
            if (Slab && __ballot(is_sphere) == 0ull) rd = cosine_bounce_wall<Fast>(id, rnd);
            else rd = cosine_bounce<Fast, Slab>(nl, rnd);
            if (!Slab || (!Box && !sc.materials_known)) ro = x;   // (slab scenes of known materials: moved ahead of the dispatch)
            emissive = 0.0f;                                              // :429
        } else if (mat == 2 || mat == 3) {                                // :432 mirror, :437 glass
            // one block for both specular materials: the glass branch needs reflect(rd, n) (:444, :446) — the mirror's whole
            // bounce (:433) — so a wave that holds lanes of both kinds evaluates it once
            MC_PT_DECISION_FP
            MC_REGION(5);   // mirror direction = the glass block's reflected direction
            if constexpr (Fast && Slab) {
                rd = specular_bounce_fast<Fast>(mat, rd, n, dot_n_rd, rnd.x, accmat);   // (the fast slab form, see there)
            } else {
            rd = specular_bounce_general<Fast, Slab>(mat, rd, n, nl, dot_n_rd, rnd.x, accmat);
            }   // (general form)
            if (!Slab || (!Box && !sc.materials_known)) ro = x;   // (slab scenes of known materials: moved ahead of the dispatch)
            emissive = 1.0f;                                              // :447
        }
        if constexpr (Slab) {
            if (depth + 1u < a.max_depth) {                               // (uniform) the intersection of the next depth
                MC_REGION(1);
                if (!Box && !sc.materials_known) {   // (uniform, cold) an unknown material kept its ray: c_i - o must follow ro, not x
#pragma unroll
                    for (int i = 0; i < HS; i++) { xoc[i] = v3{hot.c[i][0], hot.c[i][1], hot.c[i][2]} - ro; occ[i] = dot(xoc[i], xoc[i]); }
                }
                id = intersect_slab<Fast, Box>(hot, ro, rd, t, false, occ, xoc);
            }
        }
    }
    return accrad;
}

// Wave tile of 64/S pixels: width x height.
template <int S> struct WaveTile;
template <> struct WaveTile<1> { static constexpr uint32_t w = 8, h = 8; };
template <> struct WaveTile<4> { static constexpr uint32_t w = 4, h = 4; };
template <> struct WaveTile<16> { static constexpr uint32_t w = 2, h = 2; };
template <> struct WaveTile<64> { static constexpr uint32_t w = 1, h = 1; };

// Pixels covered by one 256-thread block (2 x 2 wave tiles).
template <int S> constexpr uint32_t block_w() { return 2u * WaveTile<S>::w; }
template <int S> constexpr uint32_t block_h() { return 2u * WaveTile<S>::h; }

// Waves per SIMD the register budget is set for: 6 / 5 (fast / strict) for the generic kernels and the reference's three spheres;
// a slab scene with more spheres keeps five more values per sphere live across a bounce (and leaves its constants to the scalar file).
template <int Fast, bool Slab, int NS> constexpr int rounds_waves() {
    return (!Slab || NS <= 3) ? (Fast ? 6 : 5) : NS <= 5 ? (Fast ? 5 : 4) : (Fast ? 4 : 3);
}

template <int Fast, int NP, int NS, bool Slab, int S, int Prec, bool Box = false>
__global__ void __launch_bounds__(256, (rounds_waves<Fast, Slab, NS>())) pathtrace_kernel(PTArgs a) {
    // dynamic LDS (no static __shared__ in front: the base stays 16-B aligned): [records | emissive list]
    extern __shared__ float lds_dyn[];
    float* lds_obj = lds_dyn;
    const uint32_t count = (a.scene.n_planes + a.scene.n_spheres) * 12u;
    uint32_t* lds_emissive = reinterpret_cast<uint32_t*>(lds_dyn + count);
    if (NP == -2) {   // generic, scene left in memory: the intersection loops read the records with wave-uniform indices
                      // (scalar loads through the constant cache), the per-lane material fetch is a vector load; no LDS at all,
                      // so a scene of thousands of objects does not cost occupancy (one 144 KB block per CU otherwise)
        lds_obj = const_cast<float*>(a.scene.d_obj_derived);
        lds_emissive = const_cast<uint32_t*>(a.scene.d_emissive);
    } else if (NP < 0) {   // generic: stage from the device buffer (its intersection loops read this copy too: slot 3 stays)
        for (uint32_t i = threadIdx.x; i < a.scene.n_emissive; i += blockDim.x) lds_emissive[i] = a.scene.d_emissive[i];
        stage_records(lds_obj, a.scene.d_obj, a.scene.n_planes + a.scene.n_spheres, false);
    } else {        // specialised: stage the kernel-argument copy for the per-lane material fetch
        stage_records(lds_obj, a.scene.obj, a.scene.n_planes + a.scene.n_spheres, Slab);
    }
    constexpr uint32_t TW = WaveTile<S>::w, TH = WaveTile<S>::h;
    // Pixel / sample slot of a lane.  Derived afresh from the thread id wherever it is needed — before the loop (accumulator
    // load), at the head of every round, after the loop (store) — through an opaque copy the compiler cannot merge with the
    // previous one: the ten-odd derived values are then dead inside the bounce loop instead of being carried through it
    // (the 80-VGPR budget of 6 waves per SIMD spilled them to scratch: 0.7 GB of HBM writes per launch, r02 profile).
    struct LaneCoords { uint32_t j, group_base, gx, gy; size_t idx; bool valid; };
    auto lane_coords = [&]() {
        uint32_t tid = threadIdx.x, row_block = a.row_block;
        asm volatile("" : "+v"(tid));
        asm volatile("" : "+s"(row_block));          // (the division's reciprocal of this uniform is not worth a register either)
        const uint32_t lane = tid & 63u, wave = tid >> 6;
        LaneCoords c;
        c.j = lane % (uint32_t)S;                    // sample slot of this lane within its pixel
        const uint32_t pix = lane / (uint32_t)S;     // pixel of this lane within the wave tile
        c.group_base = lane - c.j;                   // first lane of this pixel's group
        c.gx = blockIdx.x * (2u * TW) + (wave & 1u) * TW + (pix % TW);
        const uint32_t ty = blockIdx.y * (2u * TH) + (wave >> 1) * TH + (pix / TW);   // tile-local storage row
        const uint32_t r = tile_row_to_storage(ty, a.row_begin, row_block, a.row_stride);
        c.valid = c.gx < a.W && r < a.row_end;                                  // pathTracer.comp:348
        c.gy = a.H - 1u - (c.valid ? r : 0u);                                   // :349 gid = (H-1-y)*W + x
        c.idx = c.valid ? (size_t)ty * a.W + c.gx : 0;
        return c;
    };
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (a.sample_begin > 0) {   // progressive continuation (samps.x protocol); s==0 resets (:451)
        const LaneCoords c = lane_coords();
        if (c.valid) acc = a.out[c.idx];
    }
    const float fspp = (float)a.spp;
    HotSlabN<slab_spheres<Slab, NS>()> hot;
    if constexpr (Slab) hot.template load<(NS <= 3)>(a.scene);   // (the 25 constants of 6 + 3 objects in VGPRs; larger scenes: SGPR operands)
    for (uint32_t base = a.sample_begin; base < a.sample_end; base += (uint32_t)S) {
        const LaneCoords c = lane_coords();
        const uint32_t s = base + c.j;
        v3 q{0.0f, 0.0f, 0.0f};
        if (c.valid && s < a.sample_end) {
            v3 rad = trace_sample<Fast, NP, NS, Slab, Prec, Box>(a, lds_obj, lds_emissive, hot, c.gx, c.gy, s);
            q = Fast ? rad * a.inv_spp : divs_recip<Fast>(rad, fspp, a.inv_spp);      // :452 accrad / samps.y (inv_spp = RN(1/spp), host)
        }
        // fold the round's S samples into the accumulator in sample order (every lane of the group
        // performs the same additions, so all S copies of acc stay identical)
        const uint32_t count = min((uint32_t)S, a.sample_end - base);           // wave-uniform
        if (S == 1) {
            acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += 0.0f;
        } else {
            uint32_t tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const uint32_t first = (tid & 63u) - (tid & 63u) % (uint32_t)S;     // = group_base, derived after the bounce loop
            auto from_lane = [](float v, uint32_t src) {                        // __shfl without its own lane-id bookkeeping
                return __int_as_float(__builtin_amdgcn_ds_bpermute((int)(src << 2), __float_as_int(v)));
            };
            for (uint32_t k = 0; k < count; k++) {
                const uint32_t src = first + k;
                acc.x += from_lane(q.x, src); acc.y += from_lane(q.y, src); acc.z += from_lane(q.z, src); acc.w += 0.0f;
            }
        }
    }
    const LaneCoords c = lane_coords();
    const bool valid = c.valid;
    const uint32_t j = c.j;
    const size_t idx = c.idx;
    if (a.sample_end == a.spp) {                                                // :453 after sample spp-1
        acc.x = dm::fpow<Fast>(dm::gmin(dm::gmax(acc.x, 0.0f), 1.0f), 0.45f) * 255.0f + 0.5f;
        acc.y = dm::fpow<Fast>(dm::gmin(dm::gmax(acc.y, 0.0f), 1.0f), 0.45f) * 255.0f + 0.5f;
        acc.z = dm::fpow<Fast>(dm::gmin(dm::gmax(acc.z, 0.0f), 1.0f), 0.45f) * 255.0f + 0.5f;
    }
    if (valid && j == 0) a.out[idx] = acc;
}

// Launch helpers implemented once per math mode (pathtrace_fast.hip / pathtrace_strict.hip).
// variant: 0 = generic (run-time object counts), 1 = slab-specialised 6 planes + 3 spheres.
// prec: 0 = fp32 sphere test (the reference's default build); 1/2/3 = native fp64 / DS / DF64 branch (generic kernel,
// S in {1,16} only).
// variant 4 = the sample-pool kernels (pathtrace_pool.h; closed-box slab scenes), variant 5 = generic, scene read from memory.
// variant 3 = the closed-box fast kernel (slab scenes with SceneArgs::box_ok, fast math only).
int launch_fast(const PTArgs& a, int variant, int S, int prec, uint32_t tile_rows, hipStream_t s);
int launch_careful(const PTArgs& a, int variant, int S, int prec, uint32_t tile_rows, hipStream_t s);   // pathtrace_careful.hip: tier 2
int launch_strict(const PTArgs& a, int variant, int S, int prec, uint32_t tile_rows, hipStream_t s);

inline size_t scene_lds_bytes(const PTArgs& a) {
    return ((size_t)(a.scene.n_planes + a.scene.n_spheres) * 12u + a.scene.n_emissive) * sizeof(float);
}
constexpr size_t kMaxSceneLdsBytes = 144u * 1024u;   // of the 160 KB per CU: 3072 objects
// Generic scenes larger than this are read from memory instead of being staged into LDS by every block (pathtrace.hip).  Measured
// (tools/bench_widened.py, per object and ray): LDS 0.22 at 6 blocks per CU (512 spheres, 25 KB), 0.55 at 2 (1500 spheres, 72 KB);
// memory 0.27-0.30 whatever the size — the LDS copy wins while four blocks fit a CU's 160 KB.
constexpr size_t kSceneLdsAutoBytes = 36u * 1024u;

template <int Fast, int NP, int NS, bool Slab, int S, int Prec, bool Box = false>
inline int launch_one(const PTArgs& a, uint32_t tile_rows, hipStream_t s) {
    dim3 grid((a.W + block_w<S>() - 1u) / block_w<S>(), (tile_rows + block_h<S>() - 1u) / block_h<S>());
    size_t lds = NP == -2 ? 0u : scene_lds_bytes(a);
    auto kern = pathtrace_kernel<Fast, NP, NS, Slab, S, Prec, Box>;
    if (lds > 48u * 1024u) {   // beyond the default dynamic-LDS window: opt in (gfx950 has 160 KB per CU)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error_detail(std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(e));
            return MC_ERR_HIP;
        }
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    return MC_OK;
}

// The slab kernels (6 axis-aligned planes + NS spheres, NS = 1 .. kMaxSlabSpheres: one instantiation per count, the sphere loops
// unrolled over VGPR-resident centres) at width S.
template <int Fast, bool Box, int NS>
inline int launch_slab_width(const PTArgs& a, int S, uint32_t tile_rows, hipStream_t s) {
    if (S == 1) return launch_one<Fast, 6, NS, true, 1, 0, Box>(a, tile_rows, s);
    if (S == 4) return launch_one<Fast, 6, NS, true, 4, 0, Box>(a, tile_rows, s);
    if (S == 16) return launch_one<Fast, 6, NS, true, 16, 0, Box>(a, tile_rows, s);
    return MC_ERR_INVALID_ARGUMENT;
}
template <int Fast, bool Box>
inline int launch_slab(const PTArgs& a, int S, uint32_t tile_rows, hipStream_t s) {
    switch (a.scene.n_spheres) {
        case 1: return launch_slab_width<Fast, Box, 1>(a, S, tile_rows, s);
        case 2: return launch_slab_width<Fast, Box, 2>(a, S, tile_rows, s);
        case 3: return launch_slab_width<Fast, Box, 3>(a, S, tile_rows, s);
        case 4: return launch_slab_width<Fast, Box, 4>(a, S, tile_rows, s);
        case 5: return launch_slab_width<Fast, Box, 5>(a, S, tile_rows, s);
        case 6: return launch_slab_width<Fast, Box, 6>(a, S, tile_rows, s);
        case 7: return launch_slab_width<Fast, Box, 7>(a, S, tile_rows, s);
        case 8: return launch_slab_width<Fast, Box, 8>(a, S, tile_rows, s);
        default: return MC_ERR_INVALID_ARGUMENT;
    }
}

template <int Fast>
inline int launch_impl(const PTArgs& a, int variant, int S, int prec, uint32_t tile_rows, hipStream_t s) {
    if (prec == 0) {
        if constexpr (Fast) {
            if (variant == 3) return launch_slab<Fast, true>(a, S, tile_rows, s);
        }
        if (variant == 3) return MC_ERR_INVALID_ARGUMENT;
        if (variant == 1) {
            return launch_slab<Fast, false>(a, S, tile_rows, s);
        } else if (variant == 5) {   // generic, scene read from memory
            if (S == 1) return launch_one<Fast, -2, -2, false, 1, 0>(a, tile_rows, s);
            if (S == 4) return launch_one<Fast, -2, -2, false, 4, 0>(a, tile_rows, s);
            if (S == 16) return launch_one<Fast, -2, -2, false, 16, 0>(a, tile_rows, s);
        } else {
            if (S == 1) return launch_one<Fast, -1, -1, false, 1, 0>(a, tile_rows, s);
            if (S == 4) return launch_one<Fast, -1, -1, false, 4, 0>(a, tile_rows, s);
            if (S == 16) return launch_one<Fast, -1, -1, false, 16, 0>(a, tile_rows, s);
        }
        return MC_ERR_INVALID_ARGUMENT;
    }
    if (S != 1 && S != 16) return MC_ERR_INVALID_ARGUMENT;
    switch (prec) {
        case 1: return S == 1 ? launch_one<Fast, -1, -1, false, 1, 1>(a, tile_rows, s) : launch_one<Fast, -1, -1, false, 16, 1>(a, tile_rows, s);
        case 2: return S == 1 ? launch_one<Fast, -1, -1, false, 1, 2>(a, tile_rows, s) : launch_one<Fast, -1, -1, false, 16, 2>(a, tile_rows, s);
        case 3: return S == 1 ? launch_one<Fast, -1, -1, false, 1, 3>(a, tile_rows, s) : launch_one<Fast, -1, -1, false, 16, 3>(a, tile_rows, s);
        default: return MC_ERR_INVALID_ARGUMENT;
    }
}

}  // namespace pt
}  // namespace mc
