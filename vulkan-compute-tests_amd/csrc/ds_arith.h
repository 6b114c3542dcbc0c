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

__device__ __forceinline__ ds2 ds_set(float a) { return ds2{a, 0.0f}; }   // emulateDouble.h.glsl:59-64

__device__ __forceinline__ ds2 ds_add(ds2 a, ds2 b) {                      // :71-83
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

__device__ __forceinline__ ds2 ds_mul(ds2 a, ds2 b) {                      // :114-139, split = 8193 (H2)
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

// ds_mul(a, a): same operation sequence with b == a (the compiler CSEs the duplicated split).
__device__ __forceinline__ ds2 ds_sqr(ds2 a) { return ds_mul(a, a); }

}  // namespace mc
