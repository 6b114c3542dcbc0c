// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product.
//
// CPU restatement of the two compute-shader hot paths of pjhusky/vulkan-compute-tests, literal in
// operation order, IEEE fp32, no contraction (build with -ffp-contract=off -fno-fast-math).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this code; the
// product (vulkan-compute-tests_amd/) never links, imports or executes anything under oracle/.
//
// Pinning status (see DESIGN.md "Oracle"): the reference has NO tests and NO golden vectors.
//   * Integer/bit-exact parts (rand01, ds_* primitives, Mandelbrot iteration counts, LUT bytes) are
//     pinned by closed-form facts derivable from the reference source and checked in
//     tests/test_oracle.py (known LUT bytes, exact 2^-32 scaling, double-precision cross-checks).
//   * The path tracer is pinned statistically against the reference's only artefact,
//     imageForReadme.png (900x600), and is otherwise "parity unpinned" at the bit level because GLSL
//     transcendental precision is implementation-defined.
//
// Everything below cites the reference file:line it follows (paths relative to /root/reference).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace oracle {

// ------------------------------------------------------------------------------------------------
// Scalar policy.  `Real` is either plain float (fast path) or Counted (per-op counters used to
// freeze "algorithmic flops per sample").  All arithmetic goes through these so the
// two instantiations execute the identical IEEE operation sequence.
// ------------------------------------------------------------------------------------------------
struct OpCounts {
    uint64_t add = 0;    // fp32 add/sub
    uint64_t mul = 0;    // fp32 mul
    uint64_t div = 0;    // fp32 divide
    uint64_t sqrt = 0;   // sqrt / inversesqrt
    uint64_t trig = 0;   // sin / cos
    uint64_t pow_ = 0;   // pow
    uint64_t cmp = 0;    // fp32 compares / min / max / select-producing ops
    uint64_t iop = 0;    // uint32 ops in rand01 (shift/xor/mul)
    uint64_t cvt = 0;    // int<->float conversions
    uint64_t intersect_calls = 0;
    uint64_t bounces = 0;
    uint64_t samples = 0;
    void operator+=(const OpCounts& o) {
        add += o.add; mul += o.mul; div += o.div; sqrt += o.sqrt; trig += o.trig; pow_ += o.pow_;
        cmp += o.cmp; iop += o.iop; cvt += o.cvt; intersect_calls += o.intersect_calls;
        bounces += o.bounces; samples += o.samples;
    }
};

inline OpCounts*& tls_counts() {
    static thread_local OpCounts* p = nullptr;
    return p;
}

// Math back-end selector: which implementation stands in for the GLSL built-ins whose precision the
// GLSL spec leaves implementation-defined (sin, cos, pow).  sqrt and '/' are always IEEE-correct.
enum MathMode : int {
    MATH_LIBM = 0,   // glibc sinf/cosf/powf (≈ correctly rounded) — the "infinitely precise" yardstick
    MATH_MC = 1      // the explicit fp32 algorithms specified in DESIGN.md §"mc math"; the HIP kernels'
                     // strict mode implements the same operation sequence, so results are bit-identical
};

// ---- explicit "mc math" algorithms: every operation is a single IEEE fp32 op or fmaf ---
namespace mcmath {

inline float as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline uint32_t as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

// Cody–Waite split of pi/2 (2x the cephes DP1..DP3 constants), then cephes sinf/cosf minimax kernels.
inline void sincos_reduce(float x, float& r, int& k) {
    const float TWO_OVER_PI = 0.636619772367581343f;
    const float PIO2_HI = 1.5703125f;
    const float PIO2_MID = 4.837512969970703125e-4f;
    const float PIO2_LO = 7.54978995489188216e-8f;
    float q = std::nearbyintf(x * TWO_OVER_PI);   // RTE, = v_rndne_f32 on the device
    r = std::fmaf(q, -PIO2_HI, x);
    r = std::fmaf(q, -PIO2_MID, r);
    r = std::fmaf(q, -PIO2_LO, r);
    k = (int)q;
}
inline float sin_kernel(float r) {
    float z = r * r;
    float p = std::fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    p = std::fmaf(p, z, -1.6666654611e-1f);
    return std::fmaf(p * z, r, r);
}
inline float cos_kernel(float r) {
    float z = r * r;
    float p = std::fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    p = std::fmaf(p, z, 4.166664568298827e-2f);
    float t = std::fmaf(-0.5f, z, 1.0f);
    return std::fmaf(p * z, z, t);
}
inline float mc_sin(float x) {
    float r; int k; sincos_reduce(x, r, k);
    float s = (k & 1) ? cos_kernel(r) : sin_kernel(r);
    return (k & 2) ? -s : s;
}
inline float mc_cos(float x) {
    float r; int k; sincos_reduce(x, r, k);
    float c = (k & 1) ? sin_kernel(r) : cos_kernel(r);
    return ((k + 1) & 2) ? -c : c;
}

// log2 for x >= 0 (x == 0 -> -inf); exp2 for y <= 0 (y < -125 -> 0).  Used only as
// pow(x, 0.45) = exp2(0.45 * log2(x)) with x in [0,1] (pathTracer.comp:453).
inline float mc_log2(float x) {
    if (x == 0.0f) return -INFINITY;
    int e_adj = 0;
    if (x < 1.17549435e-38f) { x = x * 16777216.0f; e_adj = -24; }   // subnormal -> normal
    uint32_t u = as_uint(x);
    int e = (int)(u >> 23) - 127;
    float m = as_float((u & 0x007fffffu) | 0x3f800000u);   // [1,2)
    if (m > 1.41421356237f) { m = m * 0.5f; e += 1; }     // [sqrt(1/2), sqrt(2))
    float t = m - 1.0f;
    // cephes logf polynomial: log(1+t) = t - t^2/2 + t^3 * P(t)
    float z = t * t;
    float p = std::fmaf(7.0376836292e-2f, t, -1.1514610310e-1f);
    p = std::fmaf(p, t, 1.1676998740e-1f);
    p = std::fmaf(p, t, -1.2420140846e-1f);
    p = std::fmaf(p, t, 1.4249322787e-1f);
    p = std::fmaf(p, t, -1.6668057665e-1f);
    p = std::fmaf(p, t, 2.0000714765e-1f);
    p = std::fmaf(p, t, -2.4999993993e-1f);
    p = std::fmaf(p, t, 3.3333331174e-1f);
    float ln = std::fmaf(t * z, p, std::fmaf(-0.5f, z, t));
    // log2(x) = e + ln * log2(e), log2(e) split hi/lo
    const float LOG2E_HI = 1.44269502162933349609375f;
    const float LOG2E_LO = 1.92596299112661746e-8f;
    float r = std::fmaf(ln, LOG2E_LO, 0.0f);
    r = std::fmaf(ln, LOG2E_HI, r);
    return r + (float)(e + e_adj);
}
inline float mc_exp2(float y) {
    if (!(y >= -125.0f)) return 0.0f;     // also catches -inf and NaN
    if (y > 127.0f) return INFINITY;
    float n = std::nearbyintf(y);
    float f = y - n;                       // [-0.5, 0.5], exact
    // 2^f = exp(f ln2): degree-6 polynomial in f (Taylor/minimax coefficients of 2^f)
    float p = std::fmaf(1.535336188319500e-4f, f, 1.339887440266574e-3f);
    p = std::fmaf(p, f, 9.618437357674640e-3f);
    p = std::fmaf(p, f, 5.550357105498874e-2f);
    p = std::fmaf(p, f, 2.402264791363012e-1f);
    p = std::fmaf(p, f, 6.931472028550421e-1f);
    p = std::fmaf(p, f, 1.0f);
    int e = (int)n + 127;                  // in [2, 254]
    return p * as_float((uint32_t)e << 23);
}
inline float mc_pow(float x, float y) { return mc_exp2(y * mc_log2(x)); }

}  // namespace mcmath

struct PlainPolicy {
    using R = float;
    static inline float add(float a, float b) { return a + b; }
    static inline float sub(float a, float b) { return a - b; }
    static inline float mul(float a, float b) { return a * b; }
    static inline float div(float a, float b) { return a / b; }
    static inline float sqrt(float a) { return std::sqrt(a); }
    static inline void c_trig() {}
    static inline void c_pow() {}
    static inline void c_cmp(int = 1) {}
    static inline void c_iop(int) {}
    static inline void c_cvt(int = 1) {}
    static inline void c_intersect() {}
    static inline void c_bounce() {}
    static inline void c_material(int) {}   // hook: the material branch a surviving bounce takes (tools/sched_sim.cpp)
    static inline void c_hit(int) {}        // hook: the kind of object a bounce hit, 0 plane / 1 sphere (tools/sched_sim.cpp)
    static inline void c_sample() {}
};
struct CountPolicy {
    using R = float;
    static inline float add(float a, float b) { tls_counts()->add++; return a + b; }
    static inline float sub(float a, float b) { tls_counts()->add++; return a - b; }
    static inline float mul(float a, float b) { tls_counts()->mul++; return a * b; }
    static inline float div(float a, float b) { tls_counts()->div++; return a / b; }
    static inline float sqrt(float a) { tls_counts()->sqrt++; return std::sqrt(a); }
    static inline void c_trig() { tls_counts()->trig++; }
    static inline void c_pow() { tls_counts()->pow_++; }
    static inline void c_cmp(int n = 1) { tls_counts()->cmp += n; }
    static inline void c_iop(int n) { tls_counts()->iop += n; }
    static inline void c_cvt(int n = 1) { tls_counts()->cvt += n; }
    static inline void c_intersect() { tls_counts()->intersect_calls++; }
    static inline void c_bounce() { tls_counts()->bounces++; }
    static inline void c_material(int) {}
    static inline void c_hit(int) {}
    static inline void c_sample() { tls_counts()->samples++; }
};

// ================================================================================================
// Mandelbrot, fp32 — shaders/mandelbrot.comp:27-46
// ================================================================================================
struct MandelView {
    // c = centre + (uv - 0.5) * scale, uv = (float(gx)/float(W), float(gy)/float(H))
    // reference: centre (-0.445, 0), scale 2.0+1.7*0.2 -> 2.34f on both axes (mandelbrot.comp:38)
    float cx_hi, cx_lo, cy_hi, cy_lo;   // lo words used only by the ds (two-float) variant
    float sx_hi, sx_lo, sy_hi, sy_lo;
};

// Returns n in [0, maxIter]: the number of non-escaping iterations (mandelbrot.comp:40-46).
// The executed-loop-body count (the unit of "pixel-iters") is n+1 if n<maxIter else
// maxIter.
inline uint32_t mandel_f32_pixel(uint32_t gx, uint32_t gy, uint32_t W, uint32_t H, uint32_t maxIter,
                                 const MandelView& v) {
    float x = (float)gx / (float)W;            // :30
    float y = (float)gy / (float)H;            // :31
    float cx = v.cx_hi + (x - 0.5f) * v.sx_hi; // :38
    float cy = v.cy_hi + (y - 0.5f) * v.sy_hi;
    float zx = 0.0f, zy = 0.0f;
    uint32_t n = 0;
    for (uint32_t i = 0; i < maxIter; i++) {   // :41
        float nzx = (zx * zx - zy * zy) + cx;  // :43
        float nzy = ((2.0f * zx) * zy) + cy;
        zx = nzx; zy = nzy;
        if (zx * zx + zy * zy > 2.0f) break;   // :44  dot(z,z) > 2
        n++;                                   // :45
    }
    return n;
}

// ================================================================================================
// ds ("double-single") primitives — shaders/emulateDouble.h.glsl:59-139
// ================================================================================================
struct ds2 { float x, y; };   // value = x (hi) + y (lo)

inline ds2 ds_set(float a) { return ds2{a, 0.0f}; }                       // :59-64
inline ds2 ds_add(ds2 a, ds2 b) {                                          // :71-83
    float t1 = a.x + b.x;
    float e = t1 - a.x;
    float t2 = ((b.x - e) + (a.x - (t1 - e))) + a.y + b.y;
    ds2 c;
    c.x = t1 + t2;
    c.y = t2 - (c.x - t1);
    return c;
}
inline ds2 ds_sub(ds2 a, ds2 b) {                                          // :86-97
    float t1 = a.x - b.x;
    float e = t1 - a.x;
    float t2 = ((-b.x - e) + (a.x - (t1 - e))) + a.y - b.y;
    ds2 c;
    c.x = t1 + t2;
    c.y = t2 - (c.x - t1);
    return c;
}
inline float ds_compare(ds2 a, ds2 b) {                                    // :102-111
    if (a.x < b.x) return -1.0f;
    else if (a.x == b.x) {
        if (a.y < b.y) return -1.0f;
        else if (a.y == b.y) return 0.0f;
        else return 1.0f;
    } else return 1.0f;
}
inline ds2 ds_mul(ds2 a, ds2 b) {                                          // :114-139 (split = 8193)
    const float split = 8193.0f;
    float cona = a.x * split;
    float conb = b.x * split;
    float a1 = cona - (cona - a.x);
    float b1 = conb - (conb - b.x);
    float a2 = a.x - a1;
    float b2 = b.x - b1;
    float c11 = a.x * b.x;
    float c21 = a2 * b2 + (a2 * b1 + (a1 * b2 + (a1 * b1 - c11)));
    float c2 = a.x * b.y + a.y * b.x;
    float t1 = c11 + c2;
    float e = t1 - c11;
    float t2 = a.y * b.y + ((c2 - e) + (c11 - (t1 - e))) + c21;
    ds2 c;
    c.x = t1 + t2;
    c.y = t2 - (c.x - t1);
    return c;
}

// ---- remaining DS_f32_f32 helpers used by the path tracer's large-sphere branch ------------------------
// inversesqrt(x) := 1.0f / sqrt(x) (IEEE), the canonical choice of DESIGN.md
inline float inversesqrt_f32(float x) { return 1.0f / std::sqrt(x); }
inline ds2 ds_split(float a) {                                             // emulateDouble.h.glsl:181-187 (4097)
    const float split = 4097.0f;
    float t = a * split;
    float a_hi = t - (t - a);
    float a_lo = a - a_hi;
    return ds2{a_hi, a_lo};
}
inline ds2 ds_twoProd(float a, float b) {                                  // :189-197
    float p = a * b;
    ds2 aS = ds_split(a), bS = ds_split(b);
    float err = ((aS.x * bS.x - p) + aS.x * bS.y + aS.y * bS.x) + aS.y * bS.y;
    return ds2{p, err};
}
inline ds2 ds_sqrt(ds2 a) {                                                // :199-210
    float xn = inversesqrt_f32(a.x);
    float yn = a.x * xn;
    ds2 yn_ds = ds_set(yn);
    ds2 ynsqr = ds_mul(yn_ds, yn_ds);
    float diff = ds_sub(a, ynsqr).x;
    ds2 prod = ds_twoProd(xn, diff);
    prod.x *= 0.5f; prod.y *= 0.5f;
    return ds_add(ds_set(yn), prod);
}
inline ds2 ds_dot3(ds2 ax, ds2 ay, ds2 az, ds2 bx, ds2 by, ds2 bz) {       // :213-223
    return ds_add(ds_add(ds_mul(ax, bx), ds_mul(ay, by)), ds_mul(az, bz));
}

// ds_div — emulateDouble.h.glsl:143-178 (hand-typed in the reference, "may contain typos", :142; restated as written)
inline ds2 ds_div(ds2 a, ds2 b) {
    const float split = 8193.0f;
    float s1 = a.x / b.x;
    float cona = s1 * split;
    float conb = b.x * split;
    float a1 = cona - (cona - s1);
    float b1 = conb - (conb - b.x);
    float a2 = s1 - a1;
    float b2 = b.x - b1;
    float c11 = s1 * b.x;
    float c21 = (((a1 * b1 - c11) + a1 * b2) + a2 * b1) + a2 * b2;
    float c2 = s1 * b.y;
    float t1 = c11 + c2;
    float e = t1 - c11;
    float t2 = ((c2 - e) + (c11 - (t1 - e))) + c21;
    float t12 = t1 + t2;
    float t22 = t2 - (t12 - t1);
    float t11 = a.x - t12;
    e = t11 - a.x;
    float t21 = ((-t12 - e) + (a.x - (t11 - e))) + a.y - t22;
    float s2 = (t11 + t21) / b.x;
    ds2 c;
    c.x = s1 + s2;
    c.y = s2 - (c.x - s1);
    return c;
}
inline ds2 twoDiff(float a, float b) {                                     // :272-277
    float s = a - b;
    float v = s - a;
    float e = (a - (s - v)) - (b + v);
    return ds2{s, e};
}
inline bool df64_eq(ds2 a, ds2 b) { return a.x == b.x && a.y == b.y; }     // :243-246
inline bool df64_neq(ds2 a, ds2 b) { return a.x != b.x || a.y != b.y; }    // :248-251

// ---- DF64_F32_F32 package — emulateDouble.h.glsl:225-356 (A. Thall's df64) ---------------------------------
inline ds2 df64_from_f32(float v) { return ds2{v, 0.0f}; }                 // :232-235
inline bool df64_lt(ds2 a, ds2 b) { return a.x < b.x || (a.x == b.x && a.y < b.y); }   // :253-255
inline ds2 quickTwoSum(float a, float b) {                                 // :259-263
    float s = a + b;
    float e = b - (s - a);
    return ds2{s, e};
}
inline ds2 twoSum(float a, float b) {                                      // :265-270
    float s = a + b;
    float v = s - a;
    float e = (a - (s - v)) + (b - v);
    return ds2{s, e};
}
inline ds2 df64_add(ds2 a, ds2 b) {                                        // :279-288
    ds2 s = twoSum(a.x, b.x);
    ds2 t = twoSum(a.y, b.y);
    s.y += t.x;
    s = quickTwoSum(s.x, s.y);
    s.y += t.y;
    s = quickTwoSum(s.x, s.y);
    return s;
}
inline ds2 df64_split(float a) {                                           // :292-311 (4097)
    const float split = 4097.0f;
    float t = a * split;
    float a_hi = t - (t - a);
    float a_lo = a - a_hi;
    return ds2{a_hi, a_lo};
}
inline ds2 df64_twoProd(float a, float b) {                                // :313-321
    float p = a * b;
    ds2 aS = df64_split(a), bS = df64_split(b);
    float err = ((aS.x * bS.x - p) + aS.x * bS.y + aS.y * bS.x) + aS.y * bS.y;
    return ds2{p, err};
}
inline ds2 df64_mult(ds2 a, ds2 b) {                                       // :323-329
    ds2 p = df64_twoProd(a.x, b.x);
    p.y += a.x * b.y;
    p.y += a.y * b.x;
    p = quickTwoSum(p.x, p.y);
    return p;
}
inline ds2 df64_sqrt(ds2 a) {                                              // :331-342
    float xn = inversesqrt_f32(a.x);
    float yn = a.x * xn;
    ds2 yn_df = df64_from_f32(yn);
    ds2 ynsqr = df64_mult(yn_df, yn_df);
    float diff = df64_add(a, df64_mult(ynsqr, df64_from_f32(-1.0f))).x;
    ds2 prod = df64_twoProd(xn, diff);
    prod.x *= 0.5f; prod.y *= 0.5f;
    return df64_add(df64_from_f32(yn), prod);
}
inline ds2 df64_dot3(ds2 ax, ds2 ay, ds2 az, ds2 bx, ds2 by, ds2 bz) {     // :346-356
    return df64_add(df64_add(df64_mult(ax, bx), df64_mult(ay, by)), df64_mult(az, bz));
}

// Two-float Mandelbrot: the composition SURVEY.md §8a row M3 / DESIGN.md defines (the reference has
// no df64 Mandelbrot; only these primitives).  Same loop structure as mandelbrot.comp:40-46.
inline uint32_t mandel_ds_pixel(uint32_t gx, uint32_t gy, uint32_t W, uint32_t H, uint32_t maxIter,
                                const MandelView& v) {
    float x = (float)gx / (float)W;
    float y = (float)gy / (float)H;
    ds2 cx = ds_add(ds2{v.cx_hi, v.cx_lo}, ds_mul(ds_set(x - 0.5f), ds2{v.sx_hi, v.sx_lo}));
    ds2 cy = ds_add(ds2{v.cy_hi, v.cy_lo}, ds_mul(ds_set(y - 0.5f), ds2{v.sy_hi, v.sy_lo}));
    ds2 zx = ds_set(0.0f), zy = ds_set(0.0f);
    const ds2 two = ds_set(2.0f);
    uint32_t n = 0;
    for (uint32_t i = 0; i < maxIter; i++) {
        ds2 zx2 = ds_mul(zx, zx);
        ds2 zy2 = ds_mul(zy, zy);
        ds2 zxy = ds_mul(zx, zy);
        ds2 twoxy = ds2{2.0f * zxy.x, 2.0f * zxy.y};        // exact doubling of both words
        ds2 nzx = ds_add(ds_sub(zx2, zy2), cx);
        ds2 nzy = ds_add(twoxy, cy);
        zx = nzx; zy = nzy;
        ds2 mag = ds_add(ds_mul(zx, zx), ds_mul(zy, zy));
        if (ds_compare(mag, two) > 0.0f) break;
        n++;
    }
    return n;
}

// ================================================================================================
// Colour map + host conversion — mandelbrot.comp:50-59, mandelbrotApp.h:139-141,149-170
// ================================================================================================
// colour(n) = d + e*cos(6.28318*(f*t+g)), t = n/M, evaluated in fp32 in source order with libm cosf.
inline void mandel_colour(uint32_t n, uint32_t maxIter, const float kColor[4], float rgba[4]) {
    float t = (float)n / (float)maxIter;                                   // :50
    const float e[3] = {-0.2f, -0.3f, -0.5f};                              // :53
    const float f[3] = {2.1f, 2.0f, 3.0f};                                 // :54
    const float g[3] = {0.0f, 0.1f, 0.0f};                                 // :55
    for (int c = 0; c < 3; c++) {
        float arg = 6.28318f * (f[c] * t + g[c]);
        rgba[c] = kColor[c] + e[c] * std::cos(arg);                        // :56
    }
    rgba[3] = 1.0f;
}
// static_cast<uint8_t>(scale * v) as compiled by g++/clang on x86-64: cvttss2si to int32, low byte.
// (mandelbrotApp.h:162-164, pathtracerApp.h:215-217; formally UB out of range — made explicit here.)
inline uint8_t x86_float_to_u8(float v) {
    if (!(v > -2147483648.0f && v < 2147483648.0f)) return 0;   // cvttss2si "indefinite" 0x80000000 -> low byte 0
    int32_t i = (int32_t)v;   // trunc toward zero, in range
    return (uint8_t)(i & 0xff);
}

// ================================================================================================
// Path tracer — shaders/pathTracer.comp
// ================================================================================================
struct v3 { float x, y, z; };

// rand01 — pathTracer.comp:107-110.  Pure uint32; float(0xffffffffU) rounds to 2^32 so the scale is 2^-32.
template <class P>
inline v3 rand01(uint32_t x, uint32_t y, uint32_t z) {
    for (int i = 3; i-- > 0;) {
        uint32_t nx = ((x >> 8) ^ y) * 1103515245u;
        uint32_t ny = ((y >> 8) ^ z) * 1103515245u;
        uint32_t nz = ((z >> 8) ^ x) * 1103515245u;
        x = nx; y = ny; z = nz;
    }
    P::c_iop(27); P::c_cvt(3);
    const float s = 1.0f / (float)0xffffffffu;   // == 2^-32
    return v3{P::mul((float)x, s), P::mul((float)y, s), P::mul((float)z, s)};
}

template <class P>
struct PT {
    static inline v3 add(v3 a, v3 b) { return v3{P::add(a.x, b.x), P::add(a.y, b.y), P::add(a.z, b.z)}; }
    static inline v3 sub(v3 a, v3 b) { return v3{P::sub(a.x, b.x), P::sub(a.y, b.y), P::sub(a.z, b.z)}; }
    static inline v3 mul(v3 a, v3 b) { return v3{P::mul(a.x, b.x), P::mul(a.y, b.y), P::mul(a.z, b.z)}; }
    static inline v3 muls(v3 a, float s) { return v3{P::mul(a.x, s), P::mul(a.y, s), P::mul(a.z, s)}; }
    static inline v3 divs(v3 a, float s) { return v3{P::div(a.x, s), P::div(a.y, s), P::div(a.z, s)}; }
    static inline v3 neg(v3 a) { return v3{-a.x, -a.y, -a.z}; }
    // GLSL dot: x*x' + y*y' + z*z', left to right, unfused
    static inline float dot(v3 a, v3 b) { return P::add(P::add(P::mul(a.x, b.x), P::mul(a.y, b.y)), P::mul(a.z, b.z)); }
    // GLSL cross (spec): (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y)
    static inline v3 cross(v3 a, v3 b) {
        return v3{P::sub(P::mul(a.y, b.z), P::mul(b.y, a.z)), P::sub(P::mul(a.z, b.x), P::mul(b.z, a.x)),
                  P::sub(P::mul(a.x, b.y), P::mul(b.x, a.y))};
    }
    // inversesqrt(x) := 1.0f / sqrt(x), both IEEE-correct (canonical choice)
    static inline float inversesqrt(float x) { return P::div(1.0f, P::sqrt(x)); }
    static inline v3 normalize(v3 a) { return muls(a, inversesqrt(dot(a, a))); }
    // reflect(I,N) = I - 2*dot(N,I)*N
    static inline v3 reflect(v3 I, v3 N) { return sub(I, muls(N, P::mul(2.0f, dot(N, I)))); }
    // GLSL max/min definitions (NaN behaviour differs from fmaxf): max(x,y) = x<y ? y : x
    static inline float gmax(float x, float y) { P::c_cmp(); return (x < y) ? y : x; }
    static inline float gmin(float x, float y) { P::c_cmp(); return (y < x) ? y : x; }
    static inline float clamp01(float x) { return gmin(gmax(x, 0.0f), 1.0f); }

    static inline float fsin(float x, int mode) { P::c_trig(); return mode == MATH_MC ? mcmath::mc_sin(x) : std::sin(x); }
    static inline float fcos(float x, int mode) { P::c_trig(); return mode == MATH_MC ? mcmath::mc_cos(x) : std::cos(x); }
    static inline float fpow(float x, float y, int mode) { P::c_pow(); return mode == MATH_MC ? mcmath::mc_pow(x, y) : std::pow(x, y); }

    struct Ray { v3 o, d; };
    struct HitInfo { float rayT; int objType; int objIdx; };
    enum { ePlane = 0, eSphere = 1 };

    const float* planes; uint32_t nPlanes;     // 12 floats each: equation.xyzw | e.xyzw | c.xyzw   (pathTracer.comp:59)
    const float* spheres; uint32_t nSpheres;   // 12 floats each: geo.xyzw      | e.xyzw | c.xyzw   (pathTracer.comp:60)
    int mathMode;
    // Which `#if` branch of the sphere test is compiled in (emulateDouble.h.glsl:13-26; all FALSE in the
    // reference's default build): 0 fp32 only, 1 USE_NATIVE_FP64, 2 DS_f32_f32, 3 DF64_F32_F32.
    int precMode = 0;

    // pathTracer.comp:134-137 / :146-149 / :216-219: does this sphere need the extended-precision test?
    static inline bool needs_precision(const float* sp, v3 o) {
        const float maxLen = 500.0f;
        v3 c{sp[0], sp[1], sp[2]};
        v3 co{c.x - o.x, c.y - o.y, c.z - o.z};
        auto d3 = [](v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; };
        return sp[3] > maxLen || d3(c, c) > maxLen * maxLen || d3(o, o) > maxLen * maxLen || d3(co, co) > maxLen * maxLen;
    }
    // Extended-precision sphere distance (the value assigned to `d`); returns false for `continue` (det < 0).
    bool sphere_extended(const float* sp, const Ray& ray, float& d) const {
        const float eps = 1e-4f, inf = 1e20f;
        if (precMode == 1) {                                               // :139-142 native fp64
            double ocx = (double)sp[0] - (double)ray.o.x, ocy = (double)sp[1] - (double)ray.o.y, ocz = (double)sp[2] - (double)ray.o.z;
            double dx = ray.d.x, dy = ray.d.y, dz = ray.d.z;
            double b = (ocx * dx + ocy * dy) + ocz * dz;
            double det = (b * b - ((ocx * ocx + ocy * ocy) + ocz * ocz)) + (double)(sp[3] * sp[3]);
            if (det < 0) return false;
            det = std::sqrt(det);
            d = (float)(b - det);
            if (!(d > eps)) { d = (float)(b + det); if (!(d > eps)) d = inf; }
            return true;
        }
        if (precMode == 2) {                                               // :151-204 DS_f32_f32
            ds2 ocX = ds_add(ds_set(sp[0]), ds_set(-ray.o.x)), ocY = ds_add(ds_set(sp[1]), ds_set(-ray.o.y)),
                ocZ = ds_add(ds_set(sp[2]), ds_set(-ray.o.z));
            ds2 rdX = ds_set(ray.d.x), rdY = ds_set(ray.d.y), rdZ = ds_set(ray.d.z);
            ds2 b = ds_dot3(ocX, ocY, ocZ, rdX, rdY, rdZ);
            ds2 w = ds_set(sp[3]);
            ds2 det = ds_add(ds_sub(ds_mul(b, b), ds_dot3(ocX, ocY, ocZ, ocX, ocY, ocZ)), ds_mul(w, w));
            if (ds_compare(det, ds_set(0.0f)) < 0.0f) return false;
            det = ds_sqrt(det);
            ds2 eps_ds = ds_set(eps);
            ds2 bMinus = ds_sub(b, det), bPlus = ds_add(b, det);
            ds2 d_ds = bMinus; d = d_ds.x;
            if (ds_compare(d_ds, eps_ds) <= 0.0f) {
                d_ds = bPlus; d = d_ds.x;
                if (ds_compare(d_ds, eps_ds) <= 0.0f) d = inf;
            }
            return true;
        }
        // :221-255 DF64_F32_F32
        ds2 ocX = df64_add(df64_from_f32(sp[0]), df64_from_f32(-ray.o.x)), ocY = df64_add(df64_from_f32(sp[1]), df64_from_f32(-ray.o.y)),
            ocZ = df64_add(df64_from_f32(sp[2]), df64_from_f32(-ray.o.z));
        ds2 rdX = df64_from_f32(ray.d.x), rdY = df64_from_f32(ray.d.y), rdZ = df64_from_f32(ray.d.z);
        ds2 b = df64_dot3(ocX, ocY, ocZ, rdX, rdY, rdZ);
        ds2 w = df64_from_f32(sp[3]);
        ds2 det = df64_add(df64_add(df64_mult(b, b), df64_mult(df64_dot3(ocX, ocY, ocZ, ocX, ocY, ocZ), df64_from_f32(-1.0f))),
                           df64_mult(w, w));
        if (df64_lt(det, df64_from_f32(0.0f))) return false;
        det = df64_sqrt(det);
        float bMinus = df64_add(b, df64_mult(det, df64_from_f32(-1.0f))).x;
        float bPlus = df64_add(b, det).x;
        d = bMinus;
        if (!(d > eps)) { d = bPlus; if (!(d > eps)) d = inf; }
        return true;
    }

    // intersect — pathTracer.comp:112-131 + :316-341 (fp32 branch; all #if variants compiled out by
    // emulateDouble.h.glsl:13-26)
    bool intersect(const Ray& ray, HitInfo& hit) const {
        P::c_intersect();
        const float eps = 1e-4f, triEps = 1e-7f, inf = 1e20f;            // :103-105
        float d;
        float t = inf;
        for (uint32_t i = 0; i < nPlanes; i++) {                           // :116
            const float* pl = planes + 12 * i;
            v3 n{pl[0], pl[1], pl[2]};
            float denom = dot(ray.d, n);                                   // :118
            P::c_cmp();
            if (denom > triEps) {                                          // :119
                d = P::div(P::sub(pl[3], dot(ray.o, n)), denom);           // :120
                P::c_cmp();
                if (d < t) { t = d; hit.objType = ePlane; hit.objIdx = (int)i; }   // :121-123
            }
        }
        for (uint32_t i = 0; i < nSpheres; i++) {                          // :127
            const float* sp = spheres + 12 * i;
            if (precMode != 0 && needs_precision(sp, ray.o)) {             // :132-315 (compiled out by default)
                if (!sphere_extended(sp, ray, d)) continue;
                P::c_cmp();
                if (d < t) { t = d; hit.objType = eSphere; hit.objIdx = (int)i; }   // :333
                continue;
            }
            v3 oc = sub(v3{sp[0], sp[1], sp[2]}, ray.o);                   // :317
            float b = dot(oc, ray.d);                                      // :318
            float det = P::add(P::sub(P::mul(b, b), dot(oc, oc)), P::mul(sp[3], sp[3]));
            P::c_cmp();
            if (det < 0.0f) continue; else det = P::sqrt(det);             // :319
            float bMinusDet = P::sub(b, det);                              // :322
            float bPlusDet = P::add(b, det);                               // :323
            d = bMinusDet;                                                 // :324
            P::c_cmp();
            if (d <= eps) {                                                // :325
                d = bPlusDet;
                P::c_cmp();
                if (d <= eps) d = inf;                                     // :327
            }
            P::c_cmp();
            if (d < t) { t = d; hit.objType = eSphere; hit.objIdx = (int)i; }   // :333
        }
        P::c_cmp();
        if (t < inf) { hit.rayT = t; return true; }                        // :336-339
        return false;
    }

    // One sample of one pixel: returns accrad (pathTracer.comp:343-449, everything before :451).
    v3 sample(uint32_t gx, uint32_t gy, uint32_t W, uint32_t H, uint32_t samp, uint32_t maxDepth) const {
        P::c_sample();
        const float pi = 3.141592653589793f;                               // :102
        // -- camera (:352-354)
        Ray cam{v3{0.0f, 0.52f, 7.4f}, normalize(v3{0.0f, -0.06f, -1.0f})};
        v3 up = (std::fabs(cam.d.y) < 0.9f) ? v3{0, 1, 0} : v3{0, 0, 1};
        v3 cx = normalize(cross(cam.d, up));
        v3 cy = cross(cx, cam.d);
        const float sdimx = 0.036f, sdimy = 0.024f;
        // -- sample sensor (:357-362)
        v3 r0 = rand01<P>(gx, gy, samp);
        float rnd2x = P::mul(2.0f, r0.x), rnd2y = P::mul(2.0f, r0.y);
        P::c_cmp(2);
        float tentx = rnd2x < 1.0f ? P::sub(P::sqrt(rnd2x), 1.0f) : P::sub(1.0f, P::sqrt(P::sub(2.0f, rnd2x)));
        float tenty = rnd2y < 1.0f ? P::sub(P::sqrt(rnd2y), 1.0f) : P::sub(1.0f, P::sqrt(P::sub(2.0f, rnd2y)));
        float stratx = (float)((samp / 2u) % 2u), straty = (float)(samp % 2u);
        P::c_cvt(6);
        float sx = P::mul(P::sub(P::div(P::add((float)gx, P::mul(0.5f, P::add(P::add(0.5f, stratx), tentx))), (float)W), 0.5f), sdimx);
        float sy = P::mul(P::sub(P::div(P::add((float)gy, P::mul(0.5f, P::add(P::add(0.5f, straty), tenty))), (float)H), 0.5f), sdimy);
        v3 spos = add(add(cam.o, muls(cx, sx)), muls(cy, sy));             // :360
        v3 lc = add(cam.o, muls(cam.d, 0.035f));
        v3 accrad{0, 0, 0}, accmat{1, 1, 1};                                // :361
        Ray ray{lc, normalize(sub(lc, spos))};                             // :362
        float emissive = 1.0f;                                             // :365
        for (uint32_t depth = 0; depth < maxDepth; depth++) {              // :367
            HitInfo hit;
            if (!intersect(ray, hit)) continue;                            // :369
            P::c_bounce();
            P::c_hit(hit.objType);
            v3 x = add(ray.o, muls(ray.d, hit.rayT));                      // :374  o + t*d
            const float* obj = (hit.objType == ePlane) ? planes + 12 * hit.objIdx : spheres + 12 * hit.objIdx;
            int mat = (int)std::floor(P::add(obj[11], 0.5f));              // :378/:384
            P::c_cvt();
            v3 col{obj[8], obj[9], obj[10]};
            v3 emi{obj[4], obj[5], obj[6]};
            v3 n;
            if (hit.objType == ePlane) n = v3{obj[0], obj[1], obj[2]};    // :381
            else n = normalize(sub(x, v3{obj[0], obj[1], obj[2]}));        // :387
            P::c_cmp();
            v3 nl = dot(n, ray.d) < 0.0f ? n : neg(n);                     // :390
            accrad = add(accrad, muls(mul(accmat, emi), emissive));        // :391
            accmat = mul(accmat, col);                                     // :392
            v3 rnd = rand01<P>(gx, gy, samp * maxDepth + depth);           // :393
            float p = gmax(gmax(col.x, col.y), col.z);                     // :394
            if (depth > 5) {                                               // :395
                P::c_cmp();
                if (rnd.z >= p) break;                                     // :396
                else accmat = divs(accmat, p);                             // :397
            }
            P::c_material(mat);
            if (mat == 1) {                                                // :400 diffuse
                for (uint32_t i = 0; i < nSpheres; i++) {                  // :403
                    const float* ls = spheres + 12 * i;
                    v3 le{ls[4], ls[5], ls[6]};
                    P::c_cmp();
                    if (dot(le, le) <= 0.0f) continue;                     // :407
                    v3 xc = sub(v3{ls[0], ls[1], ls[2]}, x);               // :408
                    v3 sw = normalize(xc);                                 // :409
                    P::c_cmp();
                    v3 su = normalize(cross((std::fabs(sw.x) > 0.1f ? v3{0, 1, 0} : v3{1, 0, 0}), sw));
                    v3 sv = cross(sw, su);
                    float cos_a_max = P::sqrt(P::sub(1.0f, P::div(P::mul(ls[3], ls[3]), dot(xc, xc))));   // :410
                    float cos_a = P::add(P::sub(1.0f, rnd.x), P::mul(rnd.x, cos_a_max));                  // :411
                    float sin_a = P::sqrt(P::sub(1.0f, P::mul(cos_a, cos_a)));
                    float phi = P::mul(2.0f * pi, rnd.y);                  // :412
                    v3 l = normalize(add(add(muls(muls(su, fcos(phi, mathMode)), sin_a),
                                             muls(muls(sv, fsin(phi, mathMode)), sin_a)),
                                         muls(sw, cos_a)));                // :413
                    HitInfo hne;
                    if (intersect(Ray{x, l}, hne) && hne.objType == eSphere && hne.objIdx == (int)i) {    // :420
                        float omega = P::mul(2.0f * pi, P::sub(1.0f, cos_a_max));                         // :421
                        v3 contrib = muls(mul(muls(divs(accmat, pi), gmax(dot(l, nl), 0.0f)), le), omega);
                        accrad = add(accrad, contrib);                     // :422
                    }
                }
                float r1 = P::mul(2.0f * pi, rnd.x), r2 = rnd.y, r2s = P::sqrt(r2);                       // :426
                v3 w = nl;
                P::c_cmp();
                v3 u = normalize(cross((std::fabs(w.x) > 0.1f ? v3{0, 1, 0} : v3{1, 0, 0}), w));          // :427
                v3 v = cross(w, u);
                v3 dir = normalize(add(add(muls(muls(u, fcos(r1, mathMode)), r2s),
                                           muls(muls(v, fsin(r1, mathMode)), r2s)),
                                       muls(w, P::sqrt(P::sub(1.0f, r2)))));                               // :428
                ray = Ray{x, dir};
                emissive = 0.0f;                                           // :429
            } else if (mat == 2) {                                         // :432 mirror
                ray = Ray{x, reflect(ray.d, n)};
                emissive = 1.0f;
            } else if (mat == 3) {                                         // :437 glass
                bool into = (n.x == nl.x && n.y == nl.y && n.z == nl.z);   // :438
                P::c_cmp(3);
                const float nc = 1.0f, nt = 1.5f;
                float nnt = into ? P::div(nc, nt) : P::div(nt, nc);        // :439
                float ddn = dot(ray.d, nl);
                float cos2t = P::sub(1.0f, P::mul(P::mul(nnt, nnt), P::sub(1.0f, P::mul(ddn, ddn))));     // :440
                P::c_cmp();
                if (cos2t >= 0.0f) {
                    float k = P::mul(into ? 1.0f : -1.0f, P::add(P::mul(ddn, nnt), P::sqrt(cos2t)));
                    v3 tdir = normalize(sub(muls(ray.d, nnt), muls(n, k)));                               // :441
                    float a = P::sub(nt, nc), b = P::add(nt, nc);
                    float R0 = P::div(P::mul(a, a), P::mul(b, b));         // :442
                    float c = P::sub(1.0f, into ? -ddn : dot(tdir, n));
                    float Re = P::add(R0, P::mul(P::mul(P::mul(P::mul(P::mul(P::sub(1.0f, R0), c), c), c), c), c));   // :443
                    float Tr = P::sub(1.0f, Re);
                    float Pr = P::add(0.25f, P::mul(0.5f, Re));
                    float RP = P::div(Re, Pr), TP = P::div(Tr, P::sub(1.0f, Pr));
                    P::c_cmp();
                    bool refl = rnd.x < Pr;
                    ray = Ray{x, refl ? reflect(ray.d, n) : tdir};         // :444
                    accmat = muls(accmat, refl ? RP : TP);                 // :445
                } else {
                    ray = Ray{x, reflect(ray.d, n)};                       // :446
                }
                emissive = 1.0f;                                           // :447
            }
        }
        return accrad;
    }

    // Full pixel: the spp-dispatch accumulation of pathtracerApp.h:358-378 serialised in sample order
    // (pathTracer.comp:451-453).  out = the vec4 the storage buffer holds after the last dispatch.
    // [sBegin,sEnd) allows progressive ranges; acc_in is the accumulator carried between ranges.
    void pixel(uint32_t gx, uint32_t gy, uint32_t W, uint32_t H, uint32_t spp, uint32_t sBegin, uint32_t sEnd,
               uint32_t maxDepth, float acc[4]) const {
        for (uint32_t s = sBegin; s < sEnd; s++) {
            v3 r = sample(gx, gy, W, H, s, maxDepth);
            if (s == 0) { acc[0] = acc[1] = acc[2] = acc[3] = 0.0f; }      // :451
            P::c_cvt();
            v3 q = divs(r, (float)spp);                                    // :452
            acc[0] = P::add(acc[0], q.x); acc[1] = P::add(acc[1], q.y); acc[2] = P::add(acc[2], q.z);
            acc[3] = P::add(acc[3], 0.0f);
            if (s == spp - 1) {                                            // :453
                for (int c = 0; c < 3; c++)
                    acc[c] = P::add(P::mul(fpow(clamp01(acc[c]), 0.45f, mathMode), 255.0f), 0.5f);
            }
        }
    }
};

}  // namespace oracle
