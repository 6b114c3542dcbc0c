// Eighth VALU survey for gfx950 (round 4): which transcendental opcodes cost what?  The fast path tracer spends 24 % of its time on
// 4.9 % of its instructions (profiles/r04_no_trans_pmc.txt: 13.2 cycles per transcendental among other instructions).  Rounds 1-3
// only timed v_rcp_f32 / v_sqrt_f32; this one times every opcode the kernels use or could use — v_rcp, v_rsq, v_sqrt, v_sin, v_cos,
// v_exp, v_log in f32 and the f16 forms — alone (16 per trip) and as 4 among 12 v_fmac_f32 (the in-kernel situation), at 6 waves per SIMD.
// If an f16 form or exp / log were markedly cheaper, a seed + Newton step could replace an f32 transcendental.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/valu_microbench8 tools/valu_microbench8.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kIters = 2048;
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)
#define F(k) "v_fmac_f32_e32 %" #k ", %8, %9\n\t"
#define ALONE(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define MIXED(OP) F(0) F(1) F(2) OP(3) F(4) F(5) F(6) OP(7) F(1) F(2) F(3) OP(0) F(5) F(6) F(7) OP(4)
#define KERNEL(NAME, PAT)                                                                           \
    __global__ void __launch_bounds__(512) NAME(float* out, float seed) {                            \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
        float b = seed * 1e-6f, c = seed * 0.25f + 2.0f;                                             \
        for (int i = 0; i < kIters; i++) asm volatile(PAT OPS);                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;          \
    }
#define DEF(tag, insn)                                                              \
    _Pragma("clang diagnostic push")                                                \
    KERNEL(k_##tag##_alone, ALONE(T_##tag)) KERNEL(k_##tag##_mixed, MIXED(T_##tag)) \
    _Pragma("clang diagnostic pop")
#define T_rcp32(k) "v_rcp_f32_e32 %" #k ", %" #k "\n\t"
#define T_rsq32(k) "v_rsq_f32_e32 %" #k ", %" #k "\n\t"
#define T_sqrt32(k) "v_sqrt_f32_e32 %" #k ", %" #k "\n\t"
#define T_sin32(k) "v_sin_f32_e32 %" #k ", %" #k "\n\t"
#define T_cos32(k) "v_cos_f32_e32 %" #k ", %" #k "\n\t"
#define T_exp32(k) "v_exp_f32_e32 %" #k ", %" #k "\n\t"
#define T_log32(k) "v_log_f32_e32 %" #k ", %" #k "\n\t"
#define T_rcp16(k) "v_rcp_f16_e32 %" #k ", %" #k "\n\t"
#define T_rsq16(k) "v_rsq_f16_e32 %" #k ", %" #k "\n\t"
#define T_sqrt16(k) "v_sqrt_f16_e32 %" #k ", %" #k "\n\t"
#define T_sin16(k) "v_sin_f16_e32 %" #k ", %" #k "\n\t"
#define T_exp16(k) "v_exp_f16_e32 %" #k ", %" #k "\n\t"
#define T_fma(k) "v_fmac_f32_e32 %" #k ", %8, %9\n\t"
#define T_cvt(k) "v_cvt_f16_f32_e32 %" #k ", %" #k "\n\t"
DEF(rcp32, ) DEF(rsq32, ) DEF(sqrt32, ) DEF(sin32, ) DEF(cos32, ) DEF(exp32, ) DEF(log32, )
DEF(rcp16, ) DEF(rsq16, ) DEF(sqrt16, ) DEF(sin16, ) DEF(exp16, ) DEF(fma, ) DEF(cvt, )
struct Entry { const char* name; void (*alone)(float*, float); void (*mixed)(float*, float); };
int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* out; CHECK(hipMalloc(&out, sizeof(float) * 512 * cus * 8));
#define E(tag) {#tag, k_##tag##_alone, k_##tag##_mixed}
    std::vector<Entry> es = {E(fma), E(cvt), E(rcp32), E(rsq32), E(sqrt32), E(sin32), E(cos32), E(exp32), E(log32), E(rcp16), E(rsq16), E(sqrt16), E(sin16), E(exp16)};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int wps = 6, blocks = cus * (wps / 2);
    auto time = [&](void (*fn)(float*, float)) {
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(512), 0, 0, out, 1.0f); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(fn, dim3(blocks), dim3(512), 0, 0, out, 1.0f);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        return (double)ms / 5;
    };
    const double winst = (double)blocks * 8 * kIters * 16;
    const double fma_alone = time(k_fma_alone) * 1e-3 * 2.4e9 * (cus * 4.0) / winst;
    printf("%d waves per SIMD; cycles per wave-instruction per SIMD @2.4 GHz.  'in a mix' = cost of ONE such instruction among v_fmac_f32 (4 per 12 fmac),\n"
           "i.e. (16 x mixed - 12 x fmac) / 4\n%-10s %12s %12s\n", wps, "opcode", "alone", "in a mix");
    for (auto& e : es) {
        const double a = time(e.alone) * 1e-3 * 2.4e9 * (cus * 4.0) / winst, m = time(e.mixed) * 1e-3 * 2.4e9 * (cus * 4.0) / winst;
        printf("%-10s %12.3f %12.3f\n", e.name, a, (16.0 * m - 12.0 * fma_alone) / 4.0);
    }
    return 0;
}
