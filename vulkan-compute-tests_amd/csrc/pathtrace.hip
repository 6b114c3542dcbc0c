// Host side of the path tracer launch: argument validation, camera basis, scene analysis (slab
// specialisation, emissive mask), choice of the sample-parallel width S.  Device code: pathtrace_kernel.h.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pathtrace_kernel.h"

namespace mc {

using pt::PTArgs;
using pt::v3;

namespace {

// ---- host-side IEEE fp32 helpers for the camera basis (same operation order as the kernels) ------
inline v3 h_add(v3 a, v3 b) { return v3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline v3 h_muls(v3 a, float s) { return v3{a.x * s, a.y * s, a.z * s}; }
inline float h_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline v3 h_cross(v3 a, v3 b) { return v3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline v3 h_normalize(v3 a) { return h_muls(a, 1.0f / sqrtf(h_dot(a, a))); }

// Slab specialisation is exact only when (see pathtrace_kernel.h, intersect()):
//   * there are exactly 6 planes and 1 .. 8 spheres (the shapes the specialised kernels are instantiated for: the reference scene
//     has 3, pathtracerApp.h:36-38; pathTracer.comp:127,403 loop over spheres.length()),
//   * every plane normal is +-1 along one axis and +-0 along the others, with a non-zero offset w,
//   * no two planes share (axis, sign), and
//   * plane indices are grouped by axis in x, y, z order, so visiting the axes in that order compares
//     candidates in the reference's plane-index order (ties on `d < t` keep the earlier plane).
bool analyse_slabs(const float* planes, uint32_t n_planes, uint32_t n_spheres, pt::SceneArgs& sc) {
    for (int a = 0; a < 3; a++) { sc.slab_id_pos[a] = sc.slab_id_neg[a] = -1; sc.slab_w_pos[a] = sc.slab_w_neg[a] = 0.0f; }
    if (n_planes != 6 || n_spheres < 1 || n_spheres > (uint32_t)pt::kMaxSlabSpheres) return false;
    int axis_of[6];
    for (uint32_t i = 0; i < n_planes; i++) {
        const float* pl = planes + 12 * i;
        int axis = -1;
        for (int a = 0; a < 3; a++) {
            if (pl[a] == 1.0f || pl[a] == -1.0f) { if (axis >= 0) return false; axis = a; }
            else if (pl[a] != 0.0f) return false;   // also rejects NaN
        }
        if (axis < 0 || pl[3] == 0.0f || !std::isfinite(pl[3])) return false;
        axis_of[i] = axis;
        if (pl[axis] > 0.0f) { if (sc.slab_id_pos[axis] >= 0) return false; sc.slab_id_pos[axis] = (int)i; sc.slab_w_pos[axis] = pl[3]; }
        else { if (sc.slab_id_neg[axis] >= 0) return false; sc.slab_id_neg[axis] = (int)i; sc.slab_w_neg[axis] = pl[3]; }
    }
    for (uint32_t i = 1; i < n_planes; i++)
        if (axis_of[i] < axis_of[i - 1]) return false;   // indices must be grouped x, then y, then z
    return true;
}

// True when no plane can be the nearest hit of a shadow ray that reaches its light (SceneArgs::nee_skip_planes): the six
// slabs close a box with positive extent, the camera (sensor centre and pinhole) lies inside it, and every emissive
// sphere lies inside it with a margin above the fp32 error of the two ray parameters being compared.
// Error bound behind the margin: the wall parameter (w - o[a]) / d[a] is good to a few ulp, but the sphere root
// b - sqrt(b*b - |oc|^2 + r^2) of a GRAZING shadow ray is not: det carries an absolute error of a few ulp(b*b)
// <= 4 eps |oc|^2, and sqrt turns that into up to 2 sqrt(eps) |oc| ~ 7e-4 |oc|, with |oc| <= 2 sqrt(3) scale for
// points of the box.  kMarginFactor * sqrt(eps) * scale = 5.5e-3 * scale leaves a factor >2 over that bound
// (tests/test_gpu_scenes.py renders scenes just above and just below it against the oracle).
bool lights_inside_box(const pt::SceneArgs& sc, const float* spheres, uint32_t n_spheres, v3 cam_o, v3 cam_lc) {
    float lo[3], hi[3], scale = 1.0f;
    for (int a = 0; a < 3; a++) {
        if (sc.slab_id_pos[a] < 0 || sc.slab_id_neg[a] < 0) return false;
        lo[a] = -sc.slab_w_neg[a]; hi[a] = sc.slab_w_pos[a];
        if (!(lo[a] < hi[a])) return false;
        scale = std::fmax(scale, std::fmax(std::fabs(lo[a]), std::fabs(hi[a])));
    }
    constexpr float kMarginFactor = 16.0f;
    const float margin = kMarginFactor * sqrtf(1.1920929e-7f) * scale;
    const float cam[2][3] = {{cam_o.x, cam_o.y, cam_o.z}, {cam_lc.x, cam_lc.y, cam_lc.z}};
    for (auto& c : cam)
        for (int a = 0; a < 3; a++)
            if (!(c[a] - 0.1f > lo[a] && c[a] + 0.1f < hi[a])) return false;   // sensor half-diagonal 0.022 + pinhole 0.035
    bool any = false;
    for (uint32_t i = 0; i < n_spheres; i++) {
        const float* sp = spheres + 12 * i;
        v3 e{sp[4], sp[5], sp[6]};
        if (!(h_dot(e, e) > 0.0f)) continue;
        any = true;
        if (!(sp[3] > 0.0f) || !std::isfinite(sp[3])) return false;
        for (int a = 0; a < 3; a++)
            if (!(sp[a] - sp[3] - margin > lo[a] && sp[a] + sp[3] + margin < hi[a])) return false;
    }
    return any;
}

// camera — pathTracer.comp:352-353,360, evaluated once on the host with the shader's fp32 operations
void set_camera(PTArgs& a) {
    a.cam_o = v3{0.0f, 0.52f, 7.4f};
    a.cam_d = h_normalize(v3{0.0f, -0.06f, -1.0f});
    v3 up = (fabsf(a.cam_d.y) < 0.9f) ? v3{0, 1, 0} : v3{0, 0, 1};
    a.cx = h_normalize(h_cross(a.cam_d, up));
    a.cy = h_cross(a.cx, a.cam_d);
    a.lc = h_add(a.cam_o, h_muls(a.cam_d, 0.035f));
}

// Sample-parallel width: enough waves to keep 256 CUs x ~28 wave slots busy with a short tail.
int choose_S(uint64_t pixels, uint32_t samples) {
    const uint64_t target_waves = 65536;
    if (pixels / 64 >= target_waves || samples < 4) return 1;
    if (pixels * 4 / 64 >= target_waves || samples < 16) return 4;
    return 16;
}

}  // namespace

// The spheres of a slab scene pairwise disjoint, with a margin far above fp32 rounding of the kernel's squared distances
// (the fast sample-pool kernel orders the spheres a shadow ray meets by their centres' projections, shadow_visible_disjoint).
static bool spheres_disjoint(const float* spheres, uint32_t n_spheres) {
    for (uint32_t i = 0; i < n_spheres; i++)
        for (uint32_t j = i + 1; j < n_spheres; j++) {
            const float* si = spheres + 12 * i;
            const float* sj = spheres + 12 * j;
            const double dx = (double)si[0] - sj[0], dy = (double)si[1] - sj[1], dz = (double)si[2] - sj[2];
            const double radii = std::fabs((double)si[3]) + std::fabs((double)sj[3]);
            if (!(std::sqrt(dx * dx + dy * dy + dz * dz) - radii > 1e-3 * (1.0 + radii))) return false;   // also NaN
        }
    return true;
}

// A light that INTERSECTS a diffuse sphere, or is (all but) enclosed by an opaque one — found by tools/fuzz_fast.py: 0.01 of a light poking
// out of a diffuse sphere.  Next-event estimation from the sphere's surface next to the intersection circle samples the light at point-blank
// range through rays that graze the sphere they start on: whether :325-:327 keep such a ray's far root decides samples worth a firefly of
// the light's full emission, and the fast kernels' contracted discriminants decide differently from the reference arithmetic far more
// often than the tolerance allows — RMSE 3.6 / p99.9 77 against the 0.5 / 4 bound at 300 x 200 x 256 in the case found, still 0.7 / 14
// with the light three quarters outside — while the oracle's own two evaluations stay inside it (0.17 / 0.48).  Fast math cannot hold
// its tolerance there, so it does not run there: the host classifies the scene (mc_pathtrace_scene_class bit 3) and renders an
// MC_PT_MATH_FAST request with the strict kernels.  Criterion, calibrated by tools/enclosed_light_sweep.py
// (profiles/r04_enclosed_light_sweep.txt: the fast kernels against the oracle with the light moved from deep inside the sphere to well
// clear of it), with out = |c_i - c_j| + r_i - r_j = how far emissive sphere i pokes out of the non-emissive sphere j (2 r_i: they touch):
//   * j diffuse (material 1):  out < (2 + kLightGapMargin) r_i — the spheres intersect or are closer than kLightGapMargin light radii
//     (measured: inside the bound again from the touching point on, with the margin on top);
//   * j a mirror (material 2): out < kMirrorMargin r_i — all but enclosed (a mirror takes no next-event estimation, :432: measured inside
//     the bound for every position but the light's last sliver);
//   * j of glass: never (next-event estimation does not pass glass, :420; the light is seen through refraction only), two lights: never.
constexpr double kLightGapMargin = 1.5, kMirrorMargin = 0.25;
// Where an MC_PT_MATH_FAST request is rendered by the careful tier instead of the fast one.  A forked sample — one that takes another
// path than the reference arithmetic's — moves its pixel by little while paths end on diffuse surfaces (next-event estimation, bounded) and
// by a light's whole emission when a specular chain carries it to the light (:391 with emissive = 1 after :432 / :447).  Every census of rounds 5
// and 6 has its worst scenes where there is the most specular surface; none without any reads above 3.1 of the bound 4 (300 x 200 x 500,
// 99.9-percentile L2).  So the fast tier renders the scenes that have NO MORE SPECULAR SURFACE THAN THE REFERENCE SCENE (pathtracerApp.h:14-39:
// diffuse walls, one mirror and one glass sphere of r = 0.8, one light), kind by kind, and the careful tier the rest:
//   * FOUR or more spheres (round 5 set five on sixteen boxes; of 44 boxes with four, two are outside: 5.5 with three specular spheres, 6.3 with
//     two lights; 36 boxes with three stay inside, at most 3.3)                                                       MC_PT_SCENE_MANY_SPHERES
//   * a mirror or glass WALL; mirror spheres, or glass spheres, whose squared radii sum to more than kFastSpecularArea   MC_PT_SCENE_SPECULAR
//     276 jittered three-sphere rooms rendered in the fast tier (tools/fast_tolerance_scenes.py seeds 7 .. 11; profiles/r06_fast_tolerance_scenes*.txt)
//     hold eighteen outside the bound: fourteen of the 132 with a specular wall (8.0, 7.0, 6.5, 6.4, 5.0 behind a glass wall, 5.8 .. 4.1 with a
//     mirror wall), two of the 38 with diffuse walls and mirror spheres beyond 0.65 (4.1: one of r = 0.88; 4.0), and two of the other 106 (4.2).
//     The mirror half of the rule was set on seed 7, the wall half on seed 8 (which refuted a mirrors-only rule: 7.0 without any mirror); the
//     192 rooms of seeds 9 .. 11 were drawn afterwards, the rule left as it was: 76 in the fast tier, 74 of them at most 3.76 and two at 4.21
//     and 4.19 (a glass sphere of r = 0.77 beside a diffuse one; a mirror and a glass sphere SMALLER than the reference scene's); the 116
//     promoted ones at most 0.80 (fast tier forced: nine outside).
// Of the 388 scenes of all censuses 174 stay in the fast tier: TWO outside the bound (4.21, 4.19), six above 3.5, median 1.8; the reference scene
// (K2: 2.49) is one of them by construction.  So the fast tier meets the bound where it was stated and on 99 % of a measured neighbourhood —
// 2 of 174 bounds the share of scenes outside below 3.6 % at 95 % confidence — and no rule on the scene tables will make that a guarantee:
// a caller who needs the margin rather than the speed asks for MC_PT_MATH_FAST_CAREFUL (worst of all censuses: 1.8, a generic room of 40
// spheres; the 170 promoted rooms: 0.8).
constexpr uint32_t kCarefulSpheres = 4;
constexpr double kFastSpecularArea = 0.65;   // sum of r^2 over the mirror (the glass) spheres; the reference scene's is 0.64 of each
static bool specular_beyond_reference(const float* planes, uint32_t n_planes, const float* spheres, uint32_t n_spheres) {
    auto material = [](const float* o) { return floorf(o[11] + 0.5f); };   // pathTracer.comp:378/:384 (a light reflects, too: :391 then :432)
    for (uint32_t i = 0; i < n_planes; i++) {
        const float m = material(planes + 12 * i);
        if (m == 2.0f || m == 3.0f) return true;
    }
    double mirror = 0.0, glass = 0.0;
    for (uint32_t i = 0; i < n_spheres; i++) {
        const float m = material(spheres + 12 * i);
        const double r2 = (double)spheres[12 * i + 3] * spheres[12 * i + 3];
        if (m == 2.0f) mirror += r2;
        else if (m == 3.0f) glass += r2;
    }
    return !(mirror <= kFastSpecularArea && glass <= kFastSpecularArea);   // (also NaN)
}
// careful tier for a fast request?  (the scan only where the sphere count has not decided already: at most three sphere records)
static bool beyond_fast_tier(const float* planes, uint32_t n_planes, const float* spheres, uint32_t n_spheres, uint32_t* bits) {
    const uint32_t b = (n_spheres >= kCarefulSpheres ? MC_PT_SCENE_MANY_SPHERES : 0u) |
                       (n_spheres < kCarefulSpheres && specular_beyond_reference(planes, n_planes, spheres, n_spheres) ? MC_PT_SCENE_SPECULAR : 0u);
    if (bits) *bits = b;
    return b != 0u;
}
static bool light_nearly_enclosed_scan(const float* spheres, uint32_t n_spheres) {
    // the emissive spheres first (one pass; almost every scene has a handful), then each of them against the others
    std::vector<uint32_t> lights;
    for (uint32_t i = 0; i < n_spheres; i++) {
        const float* si = spheres + 12 * i;
        if (h_dot(v3{si[4], si[5], si[6]}, v3{si[4], si[5], si[6]}) > 0.0f) lights.push_back(i);   // :407
    }
    for (uint32_t i : lights) {
        const float* si = spheres + 12 * i;
        for (uint32_t j = 0; j < n_spheres; j++) {
            const float* sj = spheres + 12 * j;
            if (j == i || h_dot(v3{sj[4], sj[5], sj[6]}, v3{sj[4], sj[5], sj[6]}) > 0.0f) continue;
            const float mat = floorf(sj[11] + 0.5f);
            if (mat != 1.0f && mat != 2.0f) continue;                                       // glass; unknown codes keep their ray (no surface)
            const double dx = (double)si[0] - sj[0], dy = (double)si[1] - sj[1], dz = (double)si[2] - sj[2];
            const double ri = std::fabs((double)si[3]), rj = std::fabs((double)sj[3]);
            const double out = std::sqrt(dx * dx + dy * dy + dz * dz) + ri - rj;           // how far the light pokes out of sphere j
            const double limit = mat == 1.0f ? (2.0 + kLightGapMargin) * ri : kMirrorMargin * ri;
            if (!(out >= limit)) return true;                                               // (also NaN)
        }
    }
    return false;
}
// The scan is O(lights x spheres) and runs for every fast-math launch, scene-class and kernel query (generic scenes are accepted up to
// 2^20 objects): the answer for the LAST table seen by this thread is kept, keyed by the table's length and a 64-bit hash of its bytes
// (one multiply-xor per 8 bytes: ~1 ms per 10^5 spheres), so a render loop over one scene pays the scan once.
bool light_nearly_enclosed(const float* spheres, uint32_t n_spheres) {
    if (n_spheres < 2u) return false;
    uint64_t h = 0x9E3779B97F4A7C15ull ^ n_spheres;
    const size_t words = (size_t)n_spheres * 6u;                 // 12 floats = 6 x 8 bytes per record
    for (size_t k = 0; k < words; k++) {
        uint64_t w;
        std::memcpy(&w, reinterpret_cast<const char*>(spheres) + 8 * k, 8);
        h = (h ^ w) * 0xFF51AFD7ED558CCDull;
        h ^= h >> 32;
    }
    thread_local uint64_t cached_hash = 0;
    thread_local uint32_t cached_n = 0;
    thread_local bool cached_result = false;
    if (cached_n == n_spheres && cached_hash == h) return cached_result;
    cached_result = light_nearly_enclosed_scan(spheres, n_spheres);
    cached_n = n_spheres; cached_hash = h;
    return cached_result;
}

// Host-side scene analysis behind mc_pathtrace_scene_class (no device involved): bit 0 = the scene takes the slab
// kernels, bit 1 = its shadow rays skip the plane tests, bit 2 = its three spheres are pairwise disjoint, bit 3 = a light is
// (all but) enclosed by an opaque sphere (any scene: fast math is then rendered by the strict kernels), bits 4 and 5 = four or more
// spheres / more specular surface than the reference scene's (any scene: fast math is then rendered by the careful tier).
uint32_t pathtrace_scene_class(const float* planes, uint32_t n_planes, const float* spheres, uint32_t n_spheres) {
    PTArgs a;
    std::memset(&a, 0, sizeof(a));
    set_camera(a);
    uint32_t tier_bits = 0;
    beyond_fast_tier(planes, n_planes, spheres, n_spheres, &tier_bits);
    const uint32_t ill = (light_nearly_enclosed(spheres, n_spheres) ? MC_PT_SCENE_LIGHT_ENCLOSED : 0u) | tier_bits;
    if (!analyse_slabs(planes, n_planes, n_spheres, a.scene)) return ill;
    return 1u | (lights_inside_box(a.scene, spheres, n_spheres, a.cam_o, a.lc) ? 2u : 0u) | (spheres_disjoint(spheres, n_spheres) ? 4u : 0u) | ill;
}

namespace {

// What pathtrace_launch will run for a request: decided on the host from the parameters and the scene alone (no device state), so
// that mc_pathtrace_select_kernel can tell a caller — an N-GPU or progressive one wants the SAME kernel for every tile and range.
struct PTPlan {
    int variant = 0;          // MC_PT_KERNEL_*: 0 generic (scene staged in LDS), 1 slab, 3 closed box, 4 sample pool, 5 generic (scene in memory)
    int S = 1;                // sample-parallel width of the (first) launch
    int tail_S = 0;           // round-synchronous kernels, ragged sample count: width of the second launch (0: one launch)
    int prec = 0;
    bool slab = false;
    uint32_t math_mode = MC_PT_MATH_STRICT;   // the mode that RUNS (a fast request may be rendered strict: light_nearly_enclosed)
    uint32_t n_emissive = 0;  // generic scenes
};

// Validates the request, fills `a` (everything but the generic scenes' device pointers) and chooses the kernel.
int pathtrace_plan(const mc_pathtrace_params* p, const float* planes, uint32_t n_planes, const float* spheres, uint32_t n_spheres,
                   PTArgs& a, PTPlan& plan) {
    if (!p || (!planes && n_planes) || (!spheres && n_spheres)) return MC_ERR_INVALID_ARGUMENT;
    if (!p->width || !p->height || !p->spp || p->row_end > p->height || p->row_begin >= p->row_end ||
        p->sample_end > p->spp || p->sample_begin >= p->sample_end)   // an empty range would re-apply the epilogue (:453)
        return MC_ERR_INVALID_ARGUMENT;
    if (p->math_mode != MC_PT_MATH_STRICT && p->math_mode != MC_PT_MATH_FAST && p->math_mode != MC_PT_MATH_FAST_CAREFUL) return MC_ERR_INVALID_ARGUMENT;
    if (p->row_stride && (!p->row_block || p->row_block > p->row_stride)) return MC_ERR_INVALID_ARGUMENT;
    // flags: bits 0, 2-6 diagnostics, bits 8-15 MC_PT_FORCE_S, bits 16-19 MC_PT_PRECISION; everything else is reserved (bit 1 was the
    // removed lane-regrouping experiment) and refused, so that a stray bit never selects a kernel silently
    constexpr uint32_t kKnownFlags = MC_PT_GENERIC_KERNEL | MC_PT_NO_BOX_KERNEL | MC_PT_NO_POOL_KERNEL | MC_PT_SCENE_IN_LDS |
                                     MC_PT_SCENE_IN_MEMORY | MC_PT_NO_FAST_GUARD | 0xff00u | 0xf0000u;
    if (p->flags & ~kKnownFlags) {
        set_error_detail("mc_pathtrace_params.flags: reserved bits set");
        return MC_ERR_INVALID_ARGUMENT;
    }
    const int prec = (int)((p->flags >> 16) & 0xfu);   // MC_PT_PRECISION(x)
    if (prec > 3) return MC_ERR_INVALID_ARGUMENT;
    plan.prec = prec;
    // A scene beyond the LDS-resident store (about 3000 objects) is read from memory — by the fp32 kernels; the extended-precision
    // sphere branches exist for LDS-resident scenes only, as does the forced MC_PT_SCENE_IN_LDS.
    const bool beyond_lds = ((size_t)n_planes + n_spheres) * 48u + (size_t)n_spheres * 4u > pt::kMaxSceneLdsBytes;
    if (beyond_lds && (prec != 0 || (p->flags & MC_PT_SCENE_IN_LDS))) {
        set_error_detail("scene exceeds the LDS-resident scene store (about 3000 objects): fp32 sphere test from memory only");
        return MC_ERR_UNSUPPORTED;
    }
    if (((size_t)n_planes + n_spheres) > (1u << 20)) {
        set_error_detail("more than 2^20 objects");
        return MC_ERR_UNSUPPORTED;
    }
    // Fast math does not run where it was measured not to hold its tolerance: a scene with a light all but enclosed by an opaque sphere
    // (light_nearly_enclosed) is rendered strict; a scene with kCarefulSpheres or more spheres, or with more specular surface than the
    // reference scene's, by the careful tier (pathtrace_careful.hip; beyond_fast_tier above has the measurements).
    plan.math_mode = p->math_mode;
    if (p->math_mode != MC_PT_MATH_STRICT && !(p->flags & MC_PT_NO_FAST_GUARD)) {
        if (light_nearly_enclosed(spheres, n_spheres)) plan.math_mode = MC_PT_MATH_STRICT;
        else if (beyond_fast_tier(planes, n_planes, spheres, n_spheres, nullptr)) plan.math_mode = MC_PT_MATH_FAST_CAREFUL;
    }
    const bool fast = plan.math_mode != MC_PT_MATH_STRICT;
    std::memset(&a, 0, sizeof(a));
    a.W = p->width; a.H = p->height; a.spp = p->spp;
    a.sample_begin = p->sample_begin; a.sample_end = p->sample_end;
    a.max_depth = p->max_depth; a.row_begin = p->row_begin; a.row_end = p->row_end;
    a.row_block = p->row_stride ? p->row_block : 0u; a.row_stride = p->row_stride;
    set_camera(a);
    a.inv_W = 1.0f / (float)p->width; a.inv_H = 1.0f / (float)p->height; a.inv_spp = 1.0f / (float)p->spp;
    a.scene.n_planes = n_planes; a.scene.n_spheres = n_spheres;
    const bool slab = prec == 0 && analyse_slabs(planes, n_planes, n_spheres, a.scene) && !(p->flags & MC_PT_GENERIC_KERNEL);
    plan.slab = slab;
    if (slab) {   // 6 planes + 1 .. 8 spheres: the records travel in the kernel-argument segment, planes in canonical
                  // slab order (x-,x+,y-,y+,z-,z+) so that the kernel's plane id is 2*axis + (d[axis] > 0)
        for (int ax = 0; ax < 3; ax++) {
            std::memcpy(a.scene.obj + 12 * (2 * ax), planes + 12 * a.scene.slab_id_neg[ax], sizeof(float) * 12);
            std::memcpy(a.scene.obj + 12 * (2 * ax + 1), planes + 12 * a.scene.slab_id_pos[ax], sizeof(float) * 12);
        }
        std::memcpy(a.scene.obj + 12 * n_planes, spheres, sizeof(float) * 12 * n_spheres);
        for (uint32_t i = 0; i < n_spheres; i++) {
            const float* sp = spheres + 12 * i;
            a.scene.r2[i] = sp[3] * sp[3];
            v3 e{sp[4], sp[5], sp[6]};
            if (h_dot(e, e) > 0.0f) a.scene.emissive_mask |= 1u << i;
        }
        a.scene.nee_skip_planes = lights_inside_box(a.scene, spheres, n_spheres, a.cam_o, a.lc) ? 1u : 0u;
        for (uint32_t i = 0; i < n_spheres; i++) {   // c_i - lc and its squared length, as dot() associates: (x*x + y*y) + z*z
            const float* sp = spheres + 12 * i;
            const float ox = sp[0] - a.lc.x, oy = sp[1] - a.lc.y, oz = sp[2] - a.lc.z;
            a.cam_oc[i][0] = ox; a.cam_oc[i][1] = oy; a.cam_oc[i][2] = oz;
            a.cam_occ[i] = (ox * ox + oy * oy) + oz * oz;
        }
        a.scene.materials_known = 1u;
        bool glass_wall = false;
        for (uint32_t i = 0; i < n_planes + n_spheres; i++) {
            const float m = floorf(a.scene.obj[12 * i + 11] + 0.5f);          // the kernel's int(floor(m + 0.5)), :378/:384
            if (!(m == 1.0f || m == 2.0f || m == 3.0f)) a.scene.materials_known = 0u;
            if (i < n_planes && m == 3.0f) glass_wall = true;
        }
        // accmat stays finite — so that `accmat * e` of a non-emitting object IS a zero and :391's add can be skipped — when every
        // colour component is finite and in [0, 1] (accmat * c, accmat / max(c) <= 1 per step; the glass weights Re/P, Tr/(1-P)
        // are < 4/3) and the depth limit is small enough for 4/3 per bounce to stay far from overflow.
        a.scene.emit_skip_ok = p->max_depth <= 64u ? 1u : 0u;
        for (uint32_t i = 0; i < n_planes + n_spheres; i++)
            for (int k = 8; k < 11; k++) {
                const float c = a.scene.obj[12 * i + k];
                if (!(c >= 0.0f && c <= 1.0f)) a.scene.emit_skip_ok = 0u;     // also rejects NaN
            }
        a.scene.spheres_disjoint = spheres_disjoint(spheres, n_spheres) ? 1u : 0u;
        // closed-box fast kernel: no ray may ever leave the box (pathtrace_kernel.h, intersect_box)
        a.scene.box_ok = (a.scene.nee_skip_planes && a.scene.materials_known && !glass_wall && a.scene.emit_skip_ok) ? 1u : 0u;
    } else {
        for (uint32_t i = 0; i < n_spheres; i++) {
            const float* sp = spheres + 12 * i;
            if (h_dot(v3{sp[4], sp[5], sp[6]}, v3{sp[4], sp[5], sp[6]}) > 0.0f) plan.n_emissive++;   // pathTracer.comp:407
        }
    }
    const uint32_t rows = tile_rows(p->row_begin, p->row_end, a.row_block, a.row_stride);
    int S = (int)((p->flags >> 8) & 0xffu);   // MC_PT_FORCE_S(s)
    const bool auto_width = S == 0;
    if (S == 0) S = choose_S((uint64_t)rows * p->width, p->sample_end - p->sample_begin);
    if (S != 1 && S != 4 && S != 16) return MC_ERR_INVALID_ARGUMENT;
    if (prec != 0 && S == 4) S = (p->sample_end - p->sample_begin) >= 16 ? 16 : 1;   // precision variants exist for S = 1, 16
    int variant = slab ? 1 : 0;
    // Generic scenes (fp32 sphere test): the records are staged into LDS by every block while that leaves room for a full set of
    // blocks per CU, else read from memory (MC_PT_SCENE_IN_LDS / MC_PT_SCENE_IN_MEMORY force one or the other; same results)
    if (!slab && prec == 0) {
        const size_t lds = ((size_t)(n_planes + n_spheres) * 12u + plan.n_emissive) * sizeof(float);
        bool in_memory = lds > pt::kSceneLdsAutoBytes;
        if (p->flags & MC_PT_SCENE_IN_LDS) in_memory = false;
        if (p->flags & MC_PT_SCENE_IN_MEMORY) in_memory = true;
        if (in_memory) variant = 5;
    }
    if (slab && a.scene.box_ok && !(p->flags & MC_PT_NO_BOX_KERNEL)) {
        if (fast) variant = 3;   // the closed-box round-synchronous kernels (scene facts at compile time)
        // The sample-pool kernels (pathtrace_pool.h): the automatic width; 16 lanes per pixel and batch (2 x 2 pixels per wave) for every image size — never a function of the tile.
        // Fast math adds a pixel's radiance in an order that depends on the wave's schedule, so it is selected only for tiles whose
        // wave tiles are those of the whole image — a wave's pixels, hence its schedule, are then the same for every tiling:
        // N-GPU output == 1-GPU output, bit for bit.  The strict variant adds in sample order whatever the tile.
        const uint32_t th = 2u;   // WaveTile<16>::h
        // (any sample range: strict continues the ordered sum from the stored accumulator exactly as the round-synchronous kernels do;
        // fast adds the range's share to it — since round 4; before, fast ranges fell back to the round-synchronous kernel)
        const bool aligned = p->row_begin % th == 0u && (a.row_block == 0u || (a.row_block % th == 0u && a.row_stride % th == 0u)) &&
                             (p->row_end % th == 0u || p->row_end == p->height);
        const bool fits = p->max_depth >= 1u && (uint64_t)p->spp * p->max_depth < (1ull << 32) && p->width < (1u << 24);
        // (fast, pairwise disjoint spheres: shadow rays decided without square roots; overlapping spheres: the pool kernel's root form)
        if (auto_width && !(p->flags & MC_PT_NO_POOL_KERNEL) && (aligned || !fast) && fits)
            variant = 4;
    }
    plan.variant = variant;
    plan.S = variant == 4 ? 16 : S;
    // Ragged sample count (K2: 500 = 31 x 16 + 4) in the round-synchronous kernels: the last round of the S-wide kernel would run
    // with S - r of every S lanes idle.  Render the full rounds, then the r remaining samples as a progressive continuation (the same
    // mechanism a caller uses through sample_begin/sample_end: the fp32 accumulator round-trips through the storage buffer
    // unchanged, so the sum — and its order — is the same) with a narrower sample-parallel width.  (The pool kernels handle it inside.)
    const uint32_t n_samples = p->sample_end - p->sample_begin;
    const uint32_t rest = n_samples % (uint32_t)S;
    if (variant != 4 && auto_width && prec == 0 && S > 1 && rest != 0u && n_samples > (uint32_t)S) plan.tail_S = rest >= 4u ? 4 : 1;
    return MC_OK;
}

}  // namespace

int pathtrace_select_kernel(const mc_pathtrace_params* p, const float* planes, uint32_t n_planes, const float* spheres,
                            uint32_t n_spheres, mc_pathtrace_kernel_info* out) {
    if (!out) return MC_ERR_INVALID_ARGUMENT;
    PTArgs a;
    PTPlan plan;
    int rc = pathtrace_plan(p, planes, n_planes, spheres, n_spheres, a, plan);
    if (rc) return rc;
    out->kernel = (uint32_t)plan.variant;
    out->lanes_per_pixel = (uint32_t)plan.S;
    out->math_mode = plan.math_mode;
    out->launches = plan.tail_S ? 2u : 1u;
    return MC_OK;
}

int pathtrace_launch(mc_context* ctx, const mc_pathtrace_params* p, const float* planes, uint32_t n_planes,
                     const float* spheres, uint32_t n_spheres, void* d_rgba, hipStream_t s) {
    if (!ctx || !d_rgba) return MC_ERR_INVALID_ARGUMENT;
    PTArgs a;
    PTPlan plan;
    int rc = pathtrace_plan(p, planes, n_planes, spheres, n_spheres, a, plan);
    if (rc) return rc;
    a.out = (float4*)d_rgba;
    if (!plan.slab) {   // any other scene: device buffer [records | emissive sphere indices | records with the derived slots], staged
                        // into LDS by the kernel — or, for large scenes, read where they lie (the third part)
        const size_t n_rec = (size_t)(n_planes + n_spheres) * 12;
        std::vector<float> host(n_rec + n_spheres);
        if (n_planes) std::memcpy(host.data(), planes, sizeof(float) * 12 * n_planes);
        if (n_spheres) std::memcpy(host.data() + 12 * (size_t)n_planes, spheres, sizeof(float) * 12 * n_spheres);
        uint32_t* em = reinterpret_cast<uint32_t*>(host.data() + (size_t)(n_planes + n_spheres) * 12);
        uint32_t n_em = 0;
        for (uint32_t i = 0; i < n_spheres; i++) {
            const float* sp = spheres + 12 * i;
            v3 e{sp[4], sp[5], sp[6]};
            if (h_dot(e, e) > 0.0f) em[n_em++] = i;                      // pathTracer.comp:407
        }
        host.resize(n_rec + n_em);
        const size_t derived_at = n_rec + ((n_em + 3u) & ~(size_t)3u);   // 16-byte aligned like the records: the kernels read (centre, radius) as one float4
        host.resize(derived_at + n_rec);   // the staged copy's derived slots (pathtrace_kernel.h, stage_records): the same fp32 operations
        for (size_t k = 0; k < (size_t)(n_planes + n_spheres); k++) {
            float* o = host.data() + derived_at + 12 * k;
            std::memcpy(o, host.data() + 12 * k, 12 * sizeof(float));
            const float m01 = (o[8] < o[9]) ? o[9] : o[8];        // dm::gmax
            o[7] = (m01 < o[10]) ? o[10] : m01;
            o[11] = floorf(o[11] + 0.5f);
        }
        if (host != ctx->scene_host || !ctx->scene_buf.ptr) {            // upload only when the scene changed
            // Earlier launches of THIS context may still read the old copy: wait for the streams it has launched on
            // (never the whole device — other contexts and streams keep running), then upload in stream order.
            rc = ctx->drain_launch_streams();
            if (rc) return rc;
            if ((rc = ctx->scene_buf.reserve(host.size() * sizeof(float) + 16))) return rc;
            ctx->scene_host.clear();   // the cache key is only valid once the upload has succeeded
            MC_HIP_TRY(hipMemcpyAsync(ctx->scene_buf.ptr, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice, s));
            MC_HIP_TRY(hipStreamSynchronize(s));   // pageable source `host` (a local): staged before the call returns
            ctx->scene_host = std::move(host);
        }
        a.scene.d_obj = (const float*)ctx->scene_buf.ptr;
        a.scene.d_emissive = (const uint32_t*)((const float*)ctx->scene_buf.ptr + n_rec);
        a.scene.d_obj_derived = (const float*)ctx->scene_buf.ptr + derived_at;
        a.scene.n_emissive = n_em;
    }
    const uint32_t rows = tile_rows(p->row_begin, p->row_end, a.row_block, a.row_stride);
    // The sample-pool kernels bound their scheduling loop; a tripped bound (a scheduling defect, never seen) sets a bit of the
    // context's status word instead of storing an incomplete image silently: the blocking entry points and
    // mc_context_synchronize() then return MC_ERR_HIP.
    if (plan.variant == 4) {
        rc = ctx->ensure_status();
        if (rc) return rc;
        a.status = (uint32_t*)ctx->status.ptr;
    }
    auto launch = [&](const PTArgs& args, int width) {
        return plan.math_mode == MC_PT_MATH_FAST           ? pt::launch_fast(args, plan.variant, width, plan.prec, rows, s)
               : plan.math_mode == MC_PT_MATH_FAST_CAREFUL ? pt::launch_careful(args, plan.variant, width, plan.prec, rows, s)
                                                           : pt::launch_strict(args, plan.variant, width, plan.prec, rows, s);
    };
    if (plan.tail_S) {
        const uint32_t rest = (p->sample_end - p->sample_begin) % (uint32_t)plan.S;
        PTArgs head = a, tail = a;
        head.sample_end = tail.sample_begin = a.sample_end - rest;
        if ((rc = launch(head, plan.S))) return rc;
        if ((rc = launch(tail, plan.tail_S))) return rc;
    } else {
        if ((rc = launch(a, plan.S))) return rc;
    }
    MC_HIP_TRY(hipGetLastError());
    return plan.slab ? MC_OK : ctx->note_launch(s);   // only the generic kernels read a cached device table
}

}  // namespace mc
