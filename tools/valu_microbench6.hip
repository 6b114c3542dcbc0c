// Sixth VALU survey for gfx950: packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, two lanes' worth of work per
// instruction and register pair).  Alone they issue at ~4.2 cycles (valu_microbench.hip) — the "slow class" that
// valu_microbench3 found to overlap completely with add / mul / fma.  Do they?  If a packed instruction costs one 2.4-cycle
// issue slot inside a mixed stream, pairing the path tracer's 3-vector arithmetic would remove a sixth of its instructions.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/valu_microbench6 tools/valu_microbench6.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kIters = 2048;
typedef float float2v __attribute__((ext_vector_type(2)));
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(b), "v"(pb)
#define F(k) "v_add_f32_e32 %" #k ", %" #k ", %8\n\t"
#define M(k) "v_fmac_f32_e32 %" #k ", %8, %8\n\t"
#define P(k) "v_pk_fma_f32 %" #k ", %" #k ", %9, %9\n\t"
#define Q(k) "v_pk_mul_f32 %" #k ", %" #k ", %9\n\t"
#define R(k) "v_pk_add_f32 %" #k ", %" #k ", %9\n\t"
#define P_F   F(0) F(1) F(2) F(3) F(0) F(1) F(2) F(3) F(0) F(1) F(2) F(3) F(0) F(1) F(2) F(3)
#define P_P   P(4) P(5) P(6) P(7) P(4) P(5) P(6) P(7) P(4) P(5) P(6) P(7) P(4) P(5) P(6) P(7)
#define P_Q   Q(4) Q(5) Q(6) Q(7) Q(4) Q(5) Q(6) Q(7) Q(4) Q(5) Q(6) Q(7) Q(4) Q(5) Q(6) Q(7)
#define P_R   R(4) R(5) R(6) R(7) R(4) R(5) R(6) R(7) R(4) R(5) R(6) R(7) R(4) R(5) R(6) R(7)
#define P_FP  F(0) P(4) F(1) P(5) F(2) P(6) F(3) P(7) F(0) P(4) F(1) P(5) F(2) P(6) F(3) P(7)
#define P_FQ  F(0) Q(4) F(1) Q(5) F(2) Q(6) F(3) Q(7) F(0) Q(4) F(1) Q(5) F(2) Q(6) F(3) Q(7)
#define P_MR  M(0) R(4) M(1) R(5) M(2) R(6) M(3) R(7) M(0) R(4) M(1) R(5) M(2) R(6) M(3) R(7)
#define P_FFP F(0) F(1) P(4) F(2) F(3) P(5) F(0) F(1) P(6) F(2) F(3) P(7) F(0) F(1) P(4) F(2)
#define P_FPP F(0) P(4) P(5) F(1) P(6) P(7) F(2) P(4) P(5) F(3) P(6) P(7) F(0) P(4) P(5) F(1)
#define KERNEL(NAME, PAT)                                                                           \
    __global__ void __launch_bounds__(512) NAME(float* out, float seed) {                            \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, b = seed * 0.5f + 1.0f;        \
        float2v p0 = {seed, seed + 1}, p1 = {seed + 2, seed + 3}, p2 = {seed + 4, seed + 5}, p3 = {seed + 6, seed + 7}, pb = {b, b + 1}; \
        for (int i = 0; i < kIters; i++) asm volatile(PAT OPS);                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p1.y + p2.x + p3.y;  \
    }
KERNEL(k_f, P_F) KERNEL(k_p, P_P) KERNEL(k_q, P_Q) KERNEL(k_r, P_R) KERNEL(k_fp, P_FP) KERNEL(k_fq, P_FQ) KERNEL(k_mr, P_MR)
KERNEL(k_ffp, P_FFP) KERNEL(k_fpp, P_FPP)
struct Entry { const char* name; void (*fn)(float*, float); double ops; };
int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* out; CHECK(hipMalloc(&out, sizeof(float) * 512 * cus * 8));
    std::vector<Entry> es = {{"v_add_f32 x16", k_f, 16}, {"v_pk_fma_f32 x16", k_p, 32}, {"v_pk_mul_f32 x16", k_q, 32}, {"v_pk_add_f32 x16", k_r, 32},
        {"F P F P (add, pk_fma)", k_fp, 24}, {"F Q F Q (add, pk_mul)", k_fq, 24}, {"M R M R (fmac, pk_add)", k_mr, 24},
        {"F F P ...", k_ffp, 21}, {"F P P ...", k_fpp, 26}};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps : {2, 6}) {
        printf("---- %d waves per SIMD\n%-28s %10s %s\n", wps, "stream", "ms", "cycles per wave-instruction | per lane-operation (pk = 2), per SIMD @2.4 GHz");
        for (auto& e : es) {
            const int blocks = cus * (wps / 2);
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(512), 0, 0, out, 1.0f); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 5; r++) hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(512), 0, 0, out, 1.0f);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
            const double winst = (double)blocks * 8 * kIters * 16, wops = (double)blocks * 8 * kIters * e.ops;
            const double cyc = (ms * 1e-3) * 2.4e9 * (cus * 4.0);
            printf("%-28s %10.4f %8.3f | %6.3f\n", e.name, ms, cyc / winst, cyc / wops);
        }
    }
    return 0;
}
