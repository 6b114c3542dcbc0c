// Proof by enumeration for the strict path tracer's short division (csrc/mc_math.h, div_short): with y = RN(1/s) — the value
// rcp_short(s) returns, tools/exact_math_exhaustive.hip — is  q0 = a*y; r = fma(-s, q0, a); q = fma(r, y, q0)  the correctly
// rounded a/s?  All operations are scale-invariant while nothing overflows or underflows (the guard's window), so the
// 2^23 x 2^23 mantissa pairs a, s in [1, 2) decide it for every pair of normal operands in the window: 7.04e13 divisions,
// compared with the compiler's IEEE expansion (which tests/test_gpu_parity.py pins to the host's '/').
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o tools/bin/exact_div_exhaustive tools/exact_div_exhaustive.hip
// Usage: exact_div_exhaustive [slices=64] [first_slice=0] [n_slices=all]   (one launch per slice of the s mantissas)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Result { unsigned long long bad; unsigned int first_s, first_a; unsigned long long bad_s_allones; };

// one thread per (s, a-block): s = 1.m_s, a runs over a block of 2^13 mantissas
__global__ void sweep(uint32_t s_first, uint32_t s_count, Result* res) {
    const uint32_t si = blockIdx.y * 64u + (threadIdx.x & 63u);           // 64 values of s per block row
    const uint32_t ablk = blockIdx.x * (blockDim.x / 64u) + (threadIdx.x >> 6);   // 2^10 blocks of 2^13 a's
    if (si >= s_count) return;
    const uint32_t sb = 0x3f800000u | (s_first + si);
    const float s = __uint_as_float(sb);
    const float y = 1.0f / s;                                             // RN(1/s)
    unsigned long long bad = 0;
    uint32_t first_a = 0xffffffffu;
    for (uint32_t k = 0; k < (1u << 13); k++) {
        const uint32_t ab = 0x3f800000u | (ablk << 13) | k;
        const float a = __uint_as_float(ab);
        const float q0 = a * y;
        const float r = __builtin_fmaf(-s, q0, a);
        const float q = __builtin_fmaf(r, y, q0);
        const float ref = a / s;
        if (__float_as_uint(q) != __float_as_uint(ref)) { bad++; if (ab < first_a) first_a = ab; }
    }
    if (bad) {
        atomicAdd(&res->bad, bad);
        if ((sb & 0x7fffffu) == 0x7fffffu) atomicAdd(&res->bad_s_allones, bad);
        unsigned int old = atomicMin(&res->first_s, sb);
        if (sb <= old) atomicMin(&res->first_a, first_a);
    }
}

int main(int argc, char** argv) {
    const int slices = argc > 1 ? atoi(argv[1]) : 64;
    const int first = argc > 2 ? atoi(argv[2]) : 0;
    const int count = argc > 3 ? atoi(argv[3]) : slices - first;
    CHECK(hipSetDevice(0));
    Result* d; CHECK(hipMalloc(&d, sizeof(Result)));
    const uint32_t per = (1u << 23) / (uint32_t)slices;
    unsigned long long total_bad = 0, total = 0;
    for (int sl = first; sl < first + count; sl++) {
        Result h{0ull, 0xffffffffu, 0xffffffffu, 0ull};
        CHECK(hipMemcpy(d, &h, sizeof h, hipMemcpyHostToDevice));
        dim3 grid((1u << 10) / 4u, (per + 63u) / 64u);                     // x: a-blocks (4 per workgroup), y: s rows
        sweep<<<grid, 256>>>(sl * per, per, d);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost));
        total_bad += h.bad; total += (unsigned long long)per << 23;
        printf("slice %3d/%d  s mantissas [%u, %u)  pairs %llu  mismatches %llu (of which s = 2 - ulp: %llu)  first s 0x%08x a 0x%08x\n", sl, slices,
               sl * per, (sl + 1) * per, (unsigned long long)per << 23, h.bad, h.bad_s_allones, h.first_s, h.first_a);
        fflush(stdout);
    }
    printf("TOTAL pairs %llu mismatches %llu\n", total, total_bad);
    return 0;
}
