// Evidence for DESIGN.md §3.2 (two-float Mandelbrot fast path): for fp32 operands whose product's error term
// is representable (no underflow/overflow), the error term of the reference's Dekker product with split constant
// 8193 (shaders/emulateDouble.h.glsl:114-139: c21 = a2*b2 + (a2*b1 + (a1*b2 + (a1*b1 - c11)))) is EXACTLY
// a*b - fl(a*b), i.e. bit-identical to fmaf(a, b, -c11).  The kernel's fast path uses the one-instruction form
// under that precondition and falls back to the literal sequence otherwise.
//
//   gcc -O2 -ffp-contract=off -mfma -fopenmp tools/dekker_vs_fma.c -o /tmp/dekker_vs_fma -lm && /tmp/dekker_vs_fma
//
// Test population: (1) uniformly random bit patterns with exponents restricted to the precondition
// |a|,|b| in [2^-50, 2^60); (2) adversarial mantissas: all-ones, single bits, half-way patterns around the split
// point (bits 11..13), combined pairwise; (3) squares a*a of both populations.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline float bits_to_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f_to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static inline float dekker_err(float a, float b, float* c11_out) {
    const float split = 8193.0f;
    float cona = a * split, conb = b * split;
    float a1 = cona - (cona - a), b1 = conb - (conb - b);
    float a2 = a - a1, b2 = b - b1;
    float c11 = a * b;
    *c11_out = c11;
    return a2 * b2 + (a2 * b1 + (a1 * b2 + (a1 * b1 - c11)));
}

static inline uint64_t splitmix(uint64_t* s) {
    uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

static inline float make(uint32_t mant, int e, int neg) {   // value = 1.mant * 2^e
    return bits_to_f(((uint32_t)neg << 31) | ((uint32_t)(e + 127) << 23) | (mant & 0x7fffffu));
}

static int check(float a, float b, unsigned long long* bad) {
    float c11;
    float d = dekker_err(a, b, &c11);
    float f = fmaf(a, b, -c11);
    if (f_to_bits(d) != f_to_bits(f) && !(d == 0.0f && f == 0.0f)) {
        if (*bad < 10) printf("MISMATCH a=%a b=%a dekker=%a fma=%a\n", a, b, d, f);
        (*bad)++;
        return 1;
    }
    return 0;
}

int main(void) {
    unsigned long long bad = 0, total = 0, signed_zero_diff = 0;
    // (2) adversarial mantissas
    uint32_t pats[512];
    int np = 0;
    pats[np++] = 0; pats[np++] = 0x7fffff; pats[np++] = 0x400000; pats[np++] = 0x3fffff; pats[np++] = 1;
    for (int k = 0; k < 23; k++) { pats[np++] = 1u << k; pats[np++] = (1u << k) - 1; pats[np++] = 0x7fffffu ^ (1u << k); }
    for (uint32_t m = 0; m < 64; m++) pats[np++] = (m << 9) | 0x1ff;     // around the split point
    for (uint32_t m = 0; m < 64; m++) pats[np++] = (m << 9);
    for (uint32_t m = 0; m < 64; m++) pats[np++] = (m << 9) | 0x100;
    for (int i = 0; i < np; i++)
        for (int j = 0; j < np; j++)
            for (int ea = -50; ea <= 59; ea += 109 / 4)
                for (int eb = -50; eb <= 59; eb += 109 / 4)
                    for (int s = 0; s < 4; s++) { check(make(pats[i], ea, s & 1), make(pats[j], eb, s >> 1), &bad); total++; }
    printf("adversarial: %llu pairs, %llu mismatches\n", total, bad);
    // (1)+(3) random
    const long long N = 400000000ll;
    unsigned long long bad_r = 0;
#pragma omp parallel for reduction(+ : bad_r, signed_zero_diff) schedule(static)
    for (int t = 0; t < 64; t++) {
        uint64_t s = 0x1234567ull * (t + 1);
        for (long long i = 0; i < N / 64; i++) {
            uint64_t r = splitmix(&s), q = splitmix(&s);
            float a = make((uint32_t)r, (int)((r >> 32) % 110) - 50, (r >> 60) & 1);
            float b = make((uint32_t)q, (int)((q >> 32) % 110) - 50, (q >> 60) & 1);
            if (fabsf(a) * fabsf(b) >= 0x1p120f) continue;   // overflow is outside the precondition
            float c11, d, f;
            d = dekker_err(a, b, &c11); f = fmaf(a, b, -c11);
            if (f_to_bits(d) != f_to_bits(f)) { if (d == 0.0f && f == 0.0f) signed_zero_diff++; else bad_r++; }
            d = dekker_err(a, a, &c11); f = fmaf(a, a, -c11);
            if (f_to_bits(d) != f_to_bits(f)) { if (d == 0.0f && f == 0.0f) signed_zero_diff++; else bad_r++; }
        }
    }
    printf("random: %lld pairs + squares, %llu mismatches, %llu differ only in the sign of a zero error term\n", N, bad_r,
           signed_zero_diff);
    return (bad || bad_r) ? 1 : 0;
}
