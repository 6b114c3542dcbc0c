// Two-float ("double-single") arithmetic for gfx950, value = hi + lo.
//
// Device restatement of the reference's DS_f32_f32 package (shaders/emulateDouble.h.glsl:59-139,
// DSFUN90 via H. Thasler's GLSL Mandelbrot).  The error-free transforms depend on the exact order of
// the fp32 operations: this file MUST be compiled with -ffp-contract=off and without fast-math (the
// Makefile does; the static_assert-like #error below guards against -ffast-math).
#pragma once
#include <hip/hip_runtime.h>

#if defined(__FAST_MATH__)
#error "ds_arith.h requires IEEE fp32 semantics: do not build with -ffast-math"
#endif

namespace mc {

struct ds2 {
    float hi, lo;
};

// ds_set / ds_add / ds_mul are also evaluated on the host (the per-column / per-row `c` tables of the two-float
// Mandelbrot, mandelbrot.hip): the host pass is compiled with the same -ffp-contract=off, so both sides execute
// the identical IEEE operation sequence.
__host__ __device__ __forceinline__ ds2 ds_set(float a) { return ds2{a, 0.0f}; }   // emulateDouble.h.glsl:59-64

__host__ __device__ __forceinline__ ds2 ds_add(ds2 a, ds2 b) {                      // :71-83
    float t1 = a.hi + b.hi;
    float e = t1 - a.hi;
    float t2 = ((b.hi - e) + (a.hi - (t1 - e))) + a.lo + b.lo;
    ds2 c;
    c.hi = t1 + t2;
    c.lo = t2 - (c.hi - t1);
    return c;
}

__device__ __forceinline__ ds2 ds_sub(ds2 a, ds2 b) {                      // :86-97
    float t1 = a.hi - b.hi;
    float e = t1 - a.hi;
    float t2 = ((-b.hi - e) + (a.hi - (t1 - e))) + a.lo - b.lo;
    ds2 c;
    c.hi = t1 + t2;
    c.lo = t2 - (c.hi - t1);
    return c;
}

// -1 / 0 / +1, lexicographic on (hi, lo)                                  // :102-111
__device__ __forceinline__ float ds_compare(ds2 a, ds2 b) {
    if (a.hi < b.hi) return -1.0f;
    if (a.hi == b.hi) {
        if (a.lo < b.lo) return -1.0f;
        if (a.lo == b.lo) return 0.0f;
        return 1.0f;
    }
    return 1.0f;
}
// ds_compare(a, b) > 0 without materialising the float
__device__ __forceinline__ bool ds_greater(ds2 a, ds2 b) {
    // ds_compare returns +1 exactly when !(a.hi < b.hi) && !(a.hi == b.hi && a.lo <= b.lo);
    // NaN hi -> both tests false -> +1, as in the reference's if/else chain.
    if (a.hi < b.hi) return false;
    if (a.hi == b.hi) return !(a.lo < b.lo) && !(a.lo == b.lo);
    return true;
}

__host__ __device__ __forceinline__ ds2 ds_mul(ds2 a, ds2 b) {                      // :114-139, split = 8193 (H2)
    const float split = 8193.0f;
    float cona = a.hi * split;
    float conb = b.hi * split;
    float a1 = cona - (cona - a.hi);
    float b1 = conb - (conb - b.hi);
    float a2 = a.hi - a1;
    float b2 = b.hi - b1;
    float c11 = a.hi * b.hi;
    float c21 = a2 * b2 + (a2 * b1 + (a1 * b2 + (a1 * b1 - c11)));
    float c2 = a.hi * b.lo + a.lo * b.hi;
    float t1 = c11 + c2;
    float e = t1 - c11;
    float t2 = a.lo * b.lo + ((c2 - e) + (c11 - (t1 - e))) + c21;
    ds2 c;
    c.hi = t1 + t2;
    c.lo = t2 - (c.hi - t1);
    return c;
}

// ds_mul with the Dekker error term c21 (8 splitting + 8 product/sum operations) replaced by one fma.  Dekker's sequence
// is an error-free transformation: with split 2^13+1 the four partial products and every partial sum are exact in
// fp32, so c21 == a.hi*b.hi - fl(a.hi*b.hi) == fmaf(a.hi, b.hi, -c11) bit for bit (up to the sign of an exact zero,
// which no later operation can turn into a different value) PROVIDED the error term is representable: ulp(a.hi) *
// ulp(b.hi) >= 2^-149 and no overflow of a.hi*8193.  tools/dekker_vs_fma.c checks 4e8 random + 7e6 adversarial
// pairs in that range.  The caller owns the precondition (mandelbrot.hip: |operands| >= 2^-50, else the literal
// ds_mul runs).
__device__ __forceinline__ ds2 ds_mul_fma(ds2 a, ds2 b) {
    float c11 = a.hi * b.hi;
    float c21 = __builtin_fmaf(a.hi, b.hi, -c11);
    float c2 = a.hi * b.lo + a.lo * b.hi;
    float t1 = c11 + c2;
    float e = t1 - c11;
    float t2 = a.lo * b.lo + ((c2 - e) + (c11 - (t1 - e))) + c21;
    ds2 c;
    c.hi = t1 + t2;
    c.lo = t2 - (c.hi - t1);
    return c;
}

// ds_mul(a, a): same operation sequence with b == a (the compiler CSEs the duplicated split).
__device__ __forceinline__ ds2 ds_sqr(ds2 a) { return ds_mul(a, a); }

// ---- helpers of the path tracer's large-sphere branch (pathTracer.comp:144-213) ---------------------------
// `RSQ` supplies inversesqrt: IEEE 1/sqrt in strict builds, v_rsq_f32 in fast builds.
__device__ __forceinline__ ds2 ds_split(float a) {                         // emulateDouble.h.glsl:181-187 (4097)
    const float split = 4097.0f;
    float t = a * split;
    float a_hi = t - (t - a);
    float a_lo = a - a_hi;
    return ds2{a_hi, a_lo};
}
__device__ __forceinline__ ds2 ds_twoProd(float a, float b) {              // :189-197
    float p = a * b;
    ds2 aS = ds_split(a), bS = ds_split(b);
    float err = ((aS.hi * bS.hi - p) + aS.hi * bS.lo + aS.lo * bS.hi) + aS.lo * bS.lo;
    return ds2{p, err};
}
template <class RSQ>
__device__ __forceinline__ ds2 ds_sqrt(ds2 a, RSQ rsq) {                   // :199-210
    float xn = rsq(a.hi);
    float yn = a.hi * xn;
    ds2 yn_ds = ds_set(yn);
    ds2 ynsqr = ds_mul(yn_ds, yn_ds);
    float diff = ds_sub(a, ynsqr).hi;
    ds2 prod = ds_twoProd(xn, diff);
    prod.hi *= 0.5f; prod.lo *= 0.5f;
    return ds_add(ds_set(yn), prod);
}
__device__ __forceinline__ ds2 ds_dot3(ds2 ax, ds2 ay, ds2 az, ds2 bx, ds2 by, ds2 bz) {   // :213-223
    return ds_add(ds_add(ds_mul(ax, bx), ds_mul(ay, by)), ds_mul(az, bz));
}

// ds_div — emulateDouble.h.glsl:143-178.  Not used by any shader path (the reference notes it was hand-typed and may
// contain typos, :142); restated as written so the DS package is complete.  Both divisions are IEEE.
__device__ __forceinline__ ds2 ds_div(ds2 a, ds2 b) {
    const float split = 8193.0f;
    float s1 = a.hi / b.hi;
    float cona = s1 * split;
    float conb = b.hi * split;
    float a1 = cona - (cona - s1);
    float b1 = conb - (conb - b.hi);
    float a2 = s1 - a1;
    float b2 = b.hi - b1;
    float c11 = s1 * b.hi;
    float c21 = (((a1 * b1 - c11) + a1 * b2) + a2 * b1) + a2 * b2;
    float c2 = s1 * b.lo;
    float t1 = c11 + c2;
    float e = t1 - c11;
    float t2 = ((c2 - e) + (c11 - (t1 - e))) + c21;
    float t12 = t1 + t2;
    float t22 = t2 - (t12 - t1);
    float t11 = a.hi - t12;
    e = t11 - a.hi;
    float t21 = ((-t12 - e) + (a.hi - (t11 - e))) + a.lo - t22;
    float s2 = (t11 + t21) / b.hi;
    ds2 c;
    c.hi = s1 + s2;
    c.lo = s2 - (c.hi - s1);
    return c;
}
__device__ __forceinline__ ds2 twoDiff(float a, float b) {                 // :272-277
    float s = a - b;
    float v = s - a;
    float e = (a - (s - v)) - (b + v);
    return ds2{s, e};
}
__device__ __forceinline__ bool df64_eq(ds2 a, ds2 b) { return a.hi == b.hi && a.lo == b.lo; }    // :243-246
__device__ __forceinline__ bool df64_neq(ds2 a, ds2 b) { return a.hi != b.hi || a.lo != b.lo; }   // :248-251

// ---- DF64_F32_F32 package — emulateDouble.h.glsl:225-356 (pathTracer.comp:214-256) --------------------------
__device__ __forceinline__ ds2 df64_from_f32(float v) { return ds2{v, 0.0f}; }                     // :232-235
__device__ __forceinline__ bool df64_lt(ds2 a, ds2 b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }   // :253-255
__device__ __forceinline__ ds2 quickTwoSum(float a, float b) {             // :259-263
    float s = a + b;
    float e = b - (s - a);
    return ds2{s, e};
}
__device__ __forceinline__ ds2 twoSum(float a, float b) {                  // :265-270
    float s = a + b;
    float v = s - a;
    float e = (a - (s - v)) + (b - v);
    return ds2{s, e};
}
__device__ __forceinline__ ds2 df64_add(ds2 a, ds2 b) {                    // :279-288
    ds2 s = twoSum(a.hi, b.hi);
    ds2 t = twoSum(a.lo, b.lo);
    s.lo += t.hi;
    s = quickTwoSum(s.hi, s.lo);
    s.lo += t.lo;
    s = quickTwoSum(s.hi, s.lo);
    return s;
}
__device__ __forceinline__ ds2 df64_twoProd(float a, float b) {            // :313-321 (split 4097, :292-311)
    float p = a * b;
    ds2 aS = ds_split(a), bS = ds_split(b);
    float err = ((aS.hi * bS.hi - p) + aS.hi * bS.lo + aS.lo * bS.hi) + aS.lo * bS.lo;
    return ds2{p, err};
}
__device__ __forceinline__ ds2 df64_mult(ds2 a, ds2 b) {                   // :323-329
    ds2 p = df64_twoProd(a.hi, b.hi);
    p.lo += a.hi * b.lo;
    p.lo += a.lo * b.hi;
    p = quickTwoSum(p.hi, p.lo);
    return p;
}
template <class RSQ>
__device__ __forceinline__ ds2 df64_sqrt(ds2 a, RSQ rsq) {                 // :331-342
    float xn = rsq(a.hi);
    float yn = a.hi * xn;
    ds2 yn_df = df64_from_f32(yn);
    ds2 ynsqr = df64_mult(yn_df, yn_df);
    float diff = df64_add(a, df64_mult(ynsqr, df64_from_f32(-1.0f))).hi;
    ds2 prod = df64_twoProd(xn, diff);
    prod.hi *= 0.5f; prod.lo *= 0.5f;
    return df64_add(df64_from_f32(yn), prod);
}
__device__ __forceinline__ ds2 df64_dot3(ds2 ax, ds2 ay, ds2 az, ds2 bx, ds2 by, ds2 bz) {   // :346-356
    return df64_add(df64_add(df64_mult(ax, bx), df64_mult(ay, by)), df64_mult(az, bz));
}

}  // namespace mc
