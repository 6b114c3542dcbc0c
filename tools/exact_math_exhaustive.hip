// Exhaustive check of the guarded short forms of correctly-rounded fp32 sqrt and reciprocal used by the strict path tracer
// (csrc/mc_math.h) against the compiler's IEEE expansions (which tests/test_gpu_parity.py pins to the host's sqrtf and /).
// Every fp32 bit pattern in [lo, hi) is evaluated with both; mismatches are counted per binary exponent.
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o tools/bin/exact_math_exhaustive tools/exact_math_exhaustive.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ float sqrt_short(float x) {          // Markstein: faithful g, then one fused correction
    float y = __builtin_amdgcn_rsqf(x);
    float g = x * y, h = 0.5f * y;
    float e = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, e, g);
    h = __builtin_fmaf(h, e, h);
    float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float sqrt_short2(float x) {         // from v_sqrt_f32: one residual + correction with rsq
    float g = __builtin_amdgcn_sqrtf(x);
    float h = 0.5f * __builtin_amdgcn_rsqf(x);
    float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float rcp_short2(float b) {          // two Newton steps
    float r = __builtin_amdgcn_rcpf(b);
    float e = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-b, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float rcp_short3(float b) {          // the compiler's sequence without scaling / fix-up
    float r = __builtin_amdgcn_rcpf(b);
    float e = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = r;
    float e1 = __builtin_fmaf(-b, q, 1.0f);
    q = __builtin_fmaf(e1, r, q);
    float e2 = __builtin_fmaf(-b, q, 1.0f);
    return __builtin_fmaf(e2, r, q);
}
__device__ __forceinline__ float rsqrt_a(float x, int steps) {   // v_sqrt + rsq correction, then Newton on 1/s from y = rsq(x)
    float g = __builtin_amdgcn_sqrtf(x), y = __builtin_amdgcn_rsqf(x);
    float d = __builtin_fmaf(-g, g, x);
    float s = __builtin_fmaf(d, 0.5f * y, g);
    float r = y;
    for (int i = 0; i < steps; i++) { float e = __builtin_fmaf(-s, r, 1.0f); r = __builtin_fmaf(e, r, r); }
    return r;
}
__device__ __forceinline__ float rsqrt_b(float x) {              // Markstein sqrt, reciprocal seeded with 2h
    float y = __builtin_amdgcn_rsqf(x);
    float g = x * y, h = 0.5f * y;
    float e = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, e, g);
    h = __builtin_fmaf(h, e, h);
    float d = __builtin_fmaf(-g, g, x);
    float s = __builtin_fmaf(d, h, g);
    float r = h + h;
    e = __builtin_fmaf(-s, r, 1.0f); r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-s, r, 1.0f); return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float rcp_short1(float b) {
    float r = __builtin_amdgcn_rcpf(b);
    float e = __builtin_fmaf(-b, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float sqrt_short3(float x) {         // rsq only: g = x*y, one fused correction
    float y = __builtin_amdgcn_rsqf(x);
    float g = x * y;
    float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, 0.5f * y, g);
}
__device__ __forceinline__ float rsqrt_c(float x) {              // sqrt_short3, then one Newton step on 1/s from y
    float y = __builtin_amdgcn_rsqf(x);
    float g = x * y;
    float d = __builtin_fmaf(-g, g, x);
    float s = __builtin_fmaf(d, 0.5f * y, g);
    float e = __builtin_fmaf(-s, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}
__device__ __forceinline__ float rsqrt_d(float x) {              // as rsqrt_c with the second-order term (breaks the tie at s = 2 - ulp)
    float y = __builtin_amdgcn_rsqf(x);
    float g = x * y;
    float d = __builtin_fmaf(-g, g, x);
    float s = __builtin_fmaf(d, 0.5f * y, g);
    float e = __builtin_fmaf(-s, y, 1.0f);
    e = __builtin_fmaf(e, e, e);
    return __builtin_fmaf(e, y, y);
}
// which: 0 sqrt_short, 1 sqrt_short2, 2 rcp_short2, 3 rcp_short3, 4 rsqrt = rcp_short2(sqrt_short) vs 1/sqrt
__global__ void sweep(uint32_t lo, uint64_t n, int which, unsigned long long* bad_per_exp, uint32_t* first_bad) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t u = lo + (uint32_t)i;
        float x = __uint_as_float(u), ref, got;
        switch (which) {
            case 0: ref = __builtin_sqrtf(x); got = sqrt_short(x); break;
            case 1: ref = __builtin_sqrtf(x); got = sqrt_short2(x); break;
            case 2: ref = 1.0f / x; got = rcp_short2(x); break;
            case 3: ref = 1.0f / x; got = rcp_short3(x); break;
            case 4: ref = 1.0f / __builtin_sqrtf(x); got = rcp_short2(sqrt_short(x)); break;
            case 5: ref = 1.0f / __builtin_sqrtf(x); got = rsqrt_a(x, 2); break;
            case 6: ref = 1.0f / __builtin_sqrtf(x); got = rsqrt_a(x, 1); break;
            case 7: ref = 1.0f / __builtin_sqrtf(x); got = rsqrt_a(x, 3); break;
            case 8: ref = 1.0f / __builtin_sqrtf(x); got = rsqrt_b(x); break;
            case 9: ref = 1.0f / x; got = rcp_short1(x); break;
            case 10: ref = 1.0f / __builtin_sqrtf(x); got = rcp_short1(sqrt_short2(x)); break;
            case 11: ref = __builtin_sqrtf(x); got = sqrt_short3(x); break;
            case 12: ref = 1.0f / __builtin_sqrtf(x); got = rsqrt_c(x); break;
            default: ref = 1.0f / __builtin_sqrtf(x); got = rsqrt_d(x); break;
        }
        if (__float_as_uint(ref) != __float_as_uint(got)) {
            atomicAdd(&bad_per_exp[(u >> 23) & 0xff], 1ull);
            atomicMin(first_bad, u);
            if (((u >> 23) & 0xff) == 128 || ((u >> 23) & 0xff) == 127) printf("  which %d: x = 0x%08x  ref 0x%08x got 0x%08x\n", which, u, __float_as_uint(ref), __float_as_uint(got));
        }
    }
}
int main(int argc, char** argv) {
    CHECK(hipSetDevice(0));
    unsigned long long* bad; uint32_t* first;
    CHECK(hipMalloc(&bad, 256 * 8)); CHECK(hipMalloc(&first, 4));
    const char* names[] = {"sqrt: rsq + Markstein (8 ops)", "sqrt: v_sqrt + rsq correction (5 ops)", "1/x: rcp + 2 Newton steps",
                           "1/x: rcp + 3 steps (compiler sequence unscaled)", "1/sqrt(x): composition of the two short forms",
                           "1/sqrt(x): short sqrt, 1/s by 2 steps from rsq(x)", "1/sqrt(x): short sqrt, 1/s by 1 step from rsq(x)",
                           "1/sqrt(x): short sqrt, 1/s by 3 steps from rsq(x)", "1/sqrt(x): Markstein sqrt, 1/s by 2 steps from 2h",
                           "1/x: rcp + 1 Newton step", "1/sqrt(x): short sqrt (5 ops), 1/s by rcp + 1 step",
                           "sqrt: rsq only, g = x*y + one correction (5 ops, 1 transcendental)", "1/sqrt(x): rsq only, 1 step on 1/s from y",
                           "1/sqrt(x): rsq only, 1 second-order step on 1/s from y"};
    // all positive normal numbers: bit patterns 0x00800000 .. 0x7f7fffff
    const uint32_t lo = 0x00800000u; const uint64_t n = 0x7f800000ull - lo;
    for (int which = 0; which < 14; which++) {
        CHECK(hipMemset(bad, 0, 256 * 8)); CHECK(hipMemset(first, 0xff, 4));
        sweep<<<4096, 256>>>(lo, n, which, bad, first);
        CHECK(hipDeviceSynchronize());
        unsigned long long h[256]; uint32_t f;
        CHECK(hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost));
        unsigned long long total = 0; int emin = 999, emax = -999;      // the clean exponent window around 2^0
        for (int e = 0; e < 256; e++) total += h[e];
        int elo = 127, ehi = 127;
        while (elo > 1 && h[elo - 1] == 0) elo--;
        while (ehi < 254 && h[ehi + 1] == 0) ehi++;
        (void)emin; (void)emax;
        printf("%-52s inputs %llu  mismatches %llu  first 0x%08x  clean for 2^%d <= x < 2^%d%s\n", names[which], (unsigned long long)n,
               total, f, elo - 127, ehi - 127 + 1, h[127] ? "  (NOT clean at 2^0)" : "");
        if (total) {
            printf("   per exponent:"); int shown = 0;
            for (int e = 0; e < 256; e++) if (h[e] && shown++ < 24) printf(" 2^%d:%llu", e - 127, h[e]);
            if (shown > 24) printf(" ... (%d exponents)", shown);
            printf("\n");
        }
    }
    return 0;
}
