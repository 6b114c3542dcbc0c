// Device math for the path tracer, in two flavours selected at compile time by `Fast`:
//
//  strict (Fast = 0): every operation is one IEEE-754 fp32 operation (correctly rounded +,-,*,/,
//     sqrt, fma) in a FIXED order, so results are bit-identical to a CPU evaluating the same sequence
//     (DESIGN.md §"mc math").  GLSL leaves sin/cos/pow/inversesqrt precision implementation-defined
//     (SURVEY.md H5); these algorithms are this build's canonical choice for them:
//        inversesqrt(x) := 1.0f / sqrt(x)
//        sin/cos        := Cody–Waite pi/2 reduction + cephes sinf/cosf kernels (fmaf form)
//        pow(x, y)      := exp2(y * log2(x)) with cephes-style logf/exp2f kernels (fmaf form)
//  fast (Fast = 1): gfx950 hardware approximations — v_rcp_f32, v_rsq_f32, v_sqrt_f32,
//     v_sin_f32/v_cos_f32 (input in revolutions), v_exp_f32/v_log_f32.  ~1 ulp each; toleranced parity.
//  careful (Fast = 2; the fast mode's second tier, include/mc_compute.h MC_PT_MATH_FAST_CAREFUL): division, square root and reciprocal
//     square root ROUNDED AS THE REFERENCE ROUNDS THEM — the strict mode's short forms (one hardware seed + 3 .. 7 ordinary
//     instructions) without their window tests; hardware sine / cosine / exp / log as in the fast tier.  Every fast-math shortcut
//     that is an identity of exact arithmetic stays; the translation unit that instantiates it does not contract.  A path then differs
//     from the reference's by far fewer roundings, and forks correspondingly less often (profiles/r05_fork_census*.txt).
//  `Fast` is an int so that `if (Fast)` keeps meaning "not strict".
//
// Requires -ffp-contract=off (explicit __builtin_fmaf calls are the only fused operations).
#pragma once
#include <hip/hip_runtime.h>

namespace mc {
namespace dm {

__device__ __forceinline__ float as_float(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t as_uint(float f) { return __float_as_uint(f); }

// ---- IEEE primitives ------------------------------------------------------------------------------
// hipcc's default for HIP is correctly-rounded fp32 divide and sqrt (-fhip-fp32-correctly-rounded-divide-sqrt);
// the parity tests (tests/test_gpu_parity.py::test_mc_math_bit_exact) verify both against the host bit for bit.
__device__ __forceinline__ float ieee_div(float a, float b) { return a / b; }
__device__ __forceinline__ float ieee_sqrt(float a) { return __builtin_sqrtf(a); }

// Short forms of the correctly rounded sqrt and reciprocal for arguments in [2^-100, 2^100): no operand scaling, no
// special-case fix-up, so 5 and 3 instructions (one transcendental each) instead of the compiler's 16 and 11.  Correct rounding is a property of the
// gfx950 v_sqrt/v_rsq/v_rcp seeds, established by enumeration, not by argument: tools/exact_math_exhaustive.hip and
// tests/test_gpu_parity.py::test_short_forms_exhaustive compare EVERY fp32 bit pattern with the IEEE expansions
// (sqrt clean for 2^-102 <= x < 2^128, 1/x for 2^-126 <= x < 2^126; profiles/r02_exact_math_exhaustive.txt).
// Outside the window — zero, denormals, inf, NaN, negatives — the wave takes the compiler's expansion, so the
// functions are correctly rounded everywhere.  The test is wave-wide (one compare, one scalar branch).
__device__ __forceinline__ bool in_short_window(float x) { return (as_uint(x) - 0x0D800000u) < 0x64000000u; }   // 2^-100 <= x < 2^100
__device__ __forceinline__ bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0ull; }            // over the active lanes
__device__ __forceinline__ float sqrt_short(float x) {
    float y = __builtin_amdgcn_rsqf(x);                      // the one transcendental
    float g = x * y;                                         // sqrt(x) within ~1.5 ulp
    float d = __builtin_fmaf(-g, g, x);                      // exact residual
    return __builtin_fmaf(d, 0.5f * y, g);
}
__device__ __forceinline__ float rcp_short(float b) {
    float r = __builtin_amdgcn_rcpf(b);
    float e = __builtin_fmaf(-b, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

// inversesqrt(x) := RN(1 / RN(sqrt(x))) (two roundings: the definition above) with ONE transcendental: the correctly rounded root s as in
// sqrt_short, then one Newton step on 1 / s seeded with the v_rsq_f32 value already at hand instead of a second seed from v_rcp_f32
// (a transcendental costs 13 issue cycles among other instructions, profiles/r04_no_trans_pmc.txt).  The step lands on the correctly
// rounded quotient everywhere in the window except where s = 2 - ulp (x = 4 - 1 ulp and 4 - 2 ulp of every second binade): there 1 / s =
// 0.5 + 2^-25 + ... sits just above a round-to-even tie that no step from below leaves (profiles/r02_exact_math_exhaustive.txt, "rsq only,
// 1 step on 1/s from y": exactly these two patterns per binade pair) — those two get their ulp added by an integer test on x's bits.
// Established like the other short forms: EVERY fp32 pattern of the window compared with the IEEE expansion on the device
// (tests/test_gpu_parity.py::test_short_forms_exhaustive, fn 5; profiles/r04_exact_rsqrt_one_trans.txt).
#ifndef MC_MATH_RSQRT_ONE_TRANS
#define MC_MATH_RSQRT_ONE_TRANS 1
#endif
__device__ __forceinline__ float rsqrt_short(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y;
    const float d = __builtin_fmaf(-g, g, x);
    const float s = __builtin_fmaf(d, 0.5f * y, g);           // = sqrt_short(x)
    const float e = __builtin_fmaf(-s, y, 1.0f);
    const float r = __builtin_fmaf(e, y, y);
    const uint32_t tie = ((as_uint(x) & 0x00fffffeu) == 0x007ffffeu) ? 1u : 0u;   // even biased exponent, mantissa 0x7ffffe / 0x7fffff
    return as_float(as_uint(r) + tie);
}

// Short division for numerators that share a positive divisor s (a colour divided by a probability, by pi, by the sample
// count): with y = RN(1/s),  q0 = a*y; r = fma(-s, q0, a); q = fma(r, y, q0)  IS the correctly rounded a/s for every pair of
// normal operands while nothing over- or underflows — all 2^23 x 2^23 mantissa pairs enumerated against the IEEE expansion,
// tools/exact_div_exhaustive.hip, profiles/r02_exact_div_exhaustive.txt: 7.04e13 divisions, 0 mismatches — and for a = +0.
// Window: s in [2^-60, 2^60), every numerator +0 or in [2^-60, 2^60) (quotient, residual and reciprocal stay normal); anything
// else — negative, -0, denormal, inf, NaN — is redone by the compiler's expansion (per lane; no render has such a lane).
// 3 instructions per numerator instead of 11, after one reciprocal (rcp_short, or a constant / host-computed RN(1/s)).
// 0x21800000 = 2^-60, 0x5D800000 = 2^60 as bit patterns; the window test is three unsigned compares (div3 below).
__device__ __forceinline__ float div_step(float a, float s, float y) {
    const float q0 = a * y;
    const float r = __builtin_fmaf(-s, q0, a);
    return __builtin_fmaf(r, y, q0);
}

// Fork census (tools/fork_census.py; measurement builds only, `make exp EXP_FLAGS=-D...=1`): the fast mode's hardware seeds replaced by
// the correctly rounded operations (MC_PT_FAST_IEEE: v_rcp / v_rsq / v_sqrt -> IEEE divide, sqrt, 1/sqrt) or its hardware sine / cosine
// by the strict mode's Cody-Waite + cephes pair (MC_PT_FAST_ACCURATE_SINCOS), everything else of fast math left as it is — which
// approximation carries how many of the forked samples (profiles/r05_fork_census.txt).
#ifndef MC_PT_FAST_IEEE
#define MC_PT_FAST_IEEE 0
#endif
#ifndef MC_PT_FAST_ACCURATE_SINCOS
#define MC_PT_FAST_ACCURATE_SINCOS 0
#endif
// The careful tier (Fast == 2; MC_PT_FAST_SHORT = 1 gives the fast tier the same forms, for the census): the short forms of the strict
// mode (correct rounding enumerated over every fp32 pattern of their windows, see sqrt_short / rcp_short / div_step / rsqrt_short below)
// without the strict mode's window tests: outside the windows (denormal, infinite, NaN arguments — no finite path produces one) the
// result is merely inaccurate, and the one argument a path does produce there, an exact zero under a square root, is selected per lane.
#ifndef MC_PT_FAST_SHORT
#define MC_PT_FAST_SHORT 0
#endif
#ifdef MC_EXPERIMENT_NO_TRANS
// MEASUREMENT BUILD ONLY (make exp EXP_FLAGS=-DMC_EXPERIMENT_NO_TRANS; profiled with round 4's tools/pmc_libs.sh, since removed: profiles/r04_no_trans_pmc.txt): the fast kernels' hardware transcendentals
// replaced by ordinary-instruction approximations (bit-pattern seeds + two Newton steps; a parabola pair for sin / cos) good to ~1e-3 —
// the paths stay statistically the same, the image is NOT the product's.  Answers one question: what does the SIMD pay per VALU
// instruction when no transcendental is in the stream (profiles/r04_no_trans_pmc.txt)?
__device__ __forceinline__ float nt_rcp(float x) {
    const float ax = __builtin_fabsf(x);
    float r = as_float(0x7EF311C7u - as_uint(ax));
    r = r * (2.0f - ax * r); r = r * (2.0f - ax * r);
    return __builtin_copysignf(r, x);
}
__device__ __forceinline__ float nt_rsq(float x) {
    float r = as_float(0x5F375A86u - (as_uint(x) >> 1));
    const float h = 0.5f * x;
    r = r * (1.5f - h * r * r); r = r * (1.5f - h * r * r);
    return r;
}
__device__ __forceinline__ float nt_sin_rev(float u) {      // sin(2 pi u), u in revolutions
    const float x = u - __builtin_rintf(u);                 // [-0.5, 0.5]
    const float y = 8.0f * x * (1.0f - 2.0f * __builtin_fabsf(x));
    return y * (0.775f + 0.225f * __builtin_fabsf(y));
}
#endif
template <int Fast> __device__ __forceinline__ float fdiv(float a, float b) {
#ifdef MC_EXPERIMENT_NO_TRANS
    if (Fast) return a * nt_rcp(b);
#endif
    if (Fast == 2 || (Fast && MC_PT_FAST_SHORT)) return div_step(a, b, rcp_short(b));
    if (Fast && !MC_PT_FAST_IEEE) return a * __builtin_amdgcn_rcpf(b);
    return ieee_div(a, b);
}
// (a0, a1, a2) / s.  Strict: the short division inside its window (wave-wide test), the compiler's IEEE expansion outside.
// HaveY: the caller supplies y = RN(1/s) — a constant or a host-computed kernel argument — instead of rcp_short(s).
template <int Fast, bool HaveY> __device__ __forceinline__ void div3(float& a0, float& a1, float& a2, float s, float y) {
    if constexpr (!Fast) {
        // the short form for every lane; the lanes outside the window (never seen in a render) redo it the long way
        const uint32_t b0 = as_uint(a0), b1 = as_uint(a1), b2 = as_uint(a2);
        uint32_t hi = b0 > b1 ? b0 : b1; hi = hi > b2 ? hi : b2;                           // v_max3_u32
        const uint32_t l0 = b0 - 1u, l1 = b1 - 1u, l2 = b2 - 1u;                           // +0 wraps to 0xffffffff: passes
        uint32_t lo = l0 < l1 ? l0 : l1; lo = lo < l2 ? lo : l2;                           // v_min3_u32
        const bool outside = ((as_uint(s) - 0x21800000u) >= 0x3C000000u) | (hi >= 0x5D800000u) | (lo < 0x21800000u - 1u);
        if constexpr (!HaveY) y = rcp_short(s);
        const float q0 = div_step(a0, s, y), q1 = div_step(a1, s, y), q2 = div_step(a2, s, y);
        if (__builtin_expect(outside, 0)) { a0 = ieee_div(a0, s); a1 = ieee_div(a1, s); a2 = ieee_div(a2, s); }
        else { a0 = q0; a1 = q1; a2 = q2; }
        return;
    }
    a0 = fdiv<Fast>(a0, s); a1 = fdiv<Fast>(a1, s); a2 = fdiv<Fast>(a2, s);
}
template <int Fast> __device__ __forceinline__ float fsqrt(float a) {
#ifdef MC_EXPERIMENT_NO_TRANS
    if (Fast) return a * nt_rsq(a);
#endif
    // Careful tier (and the MC_PT_FAST_SHORT experiment): the strict mode's short form WITHOUT its window test.  Outside the window it is
    // merely inaccurate, with two exceptions stated here: -0 maps to +0 (the reference keeps -0; no decision downstream
    // reads the sign of a zero root), and a DENORMAL argument gives NaN (rsq = inf, inf - inf) — in the intersectors a sphere whose
    // discriminant is a denormal then fails every comparison and is a miss where the reference reports a grazing hit.  A discriminant
    // below 2^-126 at scene scale (radii 0.2 - 1e5, |b| >= 1e-4) does not occur: 254 566 fuzzed scenes, 0 non-finite pixels.
    if (Fast == 2 || (Fast && MC_PT_FAST_SHORT)) { const float r = sqrt_short(a); return a == 0.0f ? 0.0f : r; }
    if (Fast && !MC_PT_FAST_IEEE) return __builtin_amdgcn_sqrtf(a);
    if (Fast) return ieee_sqrt(a);
#ifdef MC_EXPERIMENT_NO_WINDOW_GUARD   // measurement only: what the per-call window tests cost (NOT exact outside the window)
    return sqrt_short(a);
#endif
    if (__builtin_expect(wave_all(in_short_window(a)), 1)) return sqrt_short(a);
    return ieee_sqrt(a);
}
template <int Fast> __device__ __forceinline__ float inversesqrt(float a) {
#ifdef MC_EXPERIMENT_NO_TRANS
    if (Fast) return nt_rsq(a);
#endif
    if (Fast == 2 || (Fast && MC_PT_FAST_SHORT)) return rsqrt_short(a);
    if (Fast && !MC_PT_FAST_IEEE) return __builtin_amdgcn_rsqf(a);
    if (Fast) return ieee_div(1.0f, ieee_sqrt(a));
#ifdef MC_EXPERIMENT_NO_WINDOW_GUARD
    return rsqrt_short(a);
#endif
    if (__builtin_expect(wave_all(in_short_window(a)), 1))
        return MC_MATH_RSQRT_ONE_TRANS ? rsqrt_short(a) : rcp_short(sqrt_short(a));   // (sqrt in [2^-50, 2^50): inside rcp_short's window)
    return ieee_div(1.0f, ieee_sqrt(a));
}

// ---- strict sin / cos ------------------------------------------------------------------------------
__device__ __forceinline__ void sincos_reduce(float x, float& r, int& k) {
    const float TWO_OVER_PI = 0.636619772367581343f;
    const float PIO2_HI = 1.5703125f;
    const float PIO2_MID = 4.837512969970703125e-4f;
    const float PIO2_LO = 7.54978995489188216e-8f;
    float q = __builtin_rintf(x * TWO_OVER_PI);   // v_rndne_f32 (round to nearest even)
    r = __builtin_fmaf(q, -PIO2_HI, x);
    r = __builtin_fmaf(q, -PIO2_MID, r);
    r = __builtin_fmaf(q, -PIO2_LO, r);
    k = (int)q;
}
__device__ __forceinline__ float sin_kernel(float r) {
    float z = r * r;
    float p = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    p = __builtin_fmaf(p, z, -1.6666654611e-1f);
    return __builtin_fmaf(p * z, r, r);
}
__device__ __forceinline__ float cos_kernel(float r) {
    float z = r * r;
    float p = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    p = __builtin_fmaf(p, z, 4.166664568298827e-2f);
    float t = __builtin_fmaf(-0.5f, z, 1.0f);
    return __builtin_fmaf(p * z, z, t);
}
__device__ __forceinline__ float mc_sin(float x) {
    float r; int k; sincos_reduce(x, r, k);
    float s = (k & 1) ? cos_kernel(r) : sin_kernel(r);
    return (k & 2) ? -s : s;
}
__device__ __forceinline__ float mc_cos(float x) {
    float r; int k; sincos_reduce(x, r, k);
    float c = (k & 1) ? sin_kernel(r) : cos_kernel(r);
    return ((k + 1) & 2) ? -c : c;
}
// both at once (shares the reduction and both kernels; bit-identical to mc_sin/mc_cos)
__device__ __forceinline__ void mc_sincos(float x, float& s, float& c) {
    float r; int k; sincos_reduce(x, r, k);
    float sk = sin_kernel(r), ck = cos_kernel(r);
    float sv = (k & 1) ? ck : sk;
    float cv = (k & 1) ? sk : ck;
    // conditional negation = flipping the sign bit (also of a zero or a NaN, as the unary minus does): bit 1 of k, moved to
    // bit 31, xor-ed in — three 2-cycle integer operations instead of a compare and a select of the 4-cycle class
    s = as_float(as_uint(sv) ^ (((uint32_t)k << 30) & 0x80000000u));
    c = as_float(as_uint(cv) ^ (((uint32_t)(k + 1) << 30) & 0x80000000u));
}

// sin/cos of angle = two_pi_f32 * u where `angle` is the already-rounded fp32 product the shader
// computes (pathTracer.comp:412,426) and `u` the random number it came from.
template <int Fast> __device__ __forceinline__ void sincos_angle(float angle, float u, float& s, float& c) {
#ifdef MC_EXPERIMENT_NO_TRANS
    if (Fast) { s = nt_sin_rev(u); c = nt_sin_rev(u + 0.25f); return; }
#endif
    if (Fast && MC_PT_FAST_ACCURATE_SINCOS) {   // (census build; a caller of the fast form may pass angle = 0: formed here)
        mc_sincos(6.283185307179586f * u, s, c);
    } else if (Fast) {
        // v_sin_f32 / v_cos_f32 take revolutions: sin(2*pi*u).  u in [0,1].
        s = __builtin_amdgcn_sinf(u);
        c = __builtin_amdgcn_cosf(u);
    } else {
        mc_sincos(angle, s, c);
    }
}

// ---- strict log2 / exp2 / pow ------------------------------------------------------------------------
__device__ __forceinline__ float mc_log2(float x) {
    if (x == 0.0f) return -__builtin_inff();
    int e_adj = 0;
    if (x < 1.17549435e-38f) { x = x * 16777216.0f; e_adj = -24; }
    uint32_t u = as_uint(x);
    int e = (int)(u >> 23) - 127;
    float m = as_float((u & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356237f) { m = m * 0.5f; e += 1; }
    float t = m - 1.0f;
    float z = t * t;
    float p = __builtin_fmaf(7.0376836292e-2f, t, -1.1514610310e-1f);
    p = __builtin_fmaf(p, t, 1.1676998740e-1f);
    p = __builtin_fmaf(p, t, -1.2420140846e-1f);
    p = __builtin_fmaf(p, t, 1.4249322787e-1f);
    p = __builtin_fmaf(p, t, -1.6668057665e-1f);
    p = __builtin_fmaf(p, t, 2.0000714765e-1f);
    p = __builtin_fmaf(p, t, -2.4999993993e-1f);
    p = __builtin_fmaf(p, t, 3.3333331174e-1f);
    float ln = __builtin_fmaf(t * z, p, __builtin_fmaf(-0.5f, z, t));
    const float LOG2E_HI = 1.44269502162933349609375f;
    const float LOG2E_LO = 1.92596299112661746e-8f;
    float r = __builtin_fmaf(ln, LOG2E_LO, 0.0f);
    r = __builtin_fmaf(ln, LOG2E_HI, r);
    return r + (float)(e + e_adj);
}
__device__ __forceinline__ float mc_exp2(float y) {
    if (!(y >= -125.0f)) return 0.0f;
    if (y > 127.0f) return __builtin_inff();
    float n = __builtin_rintf(y);
    float f = y - n;
    float p = __builtin_fmaf(1.535336188319500e-4f, f, 1.339887440266574e-3f);
    p = __builtin_fmaf(p, f, 9.618437357674640e-3f);
    p = __builtin_fmaf(p, f, 5.550357105498874e-2f);
    p = __builtin_fmaf(p, f, 2.402264791363012e-1f);
    p = __builtin_fmaf(p, f, 6.931472028550421e-1f);
    p = __builtin_fmaf(p, f, 1.0f);
    int e = (int)n + 127;
    return p * as_float((uint32_t)e << 23);
}
template <int Fast> __device__ __forceinline__ float fpow(float x, float y) {
    if (Fast) return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));   // v_exp_f32(y * v_log_f32(x))
    return mc_exp2(y * mc_log2(x));
}

// GLSL max/min (NaN behaviour differs from v_max_f32): max(x,y) = x<y ? y : x ; min(x,y) = y<x ? y : x
__device__ __forceinline__ float gmax(float x, float y) { return (x < y) ? y : x; }
__device__ __forceinline__ float gmin(float x, float y) { return (y < x) ? y : x; }

}  // namespace dm
}  // namespace mc
