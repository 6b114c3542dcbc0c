// Fourth VALU survey for gfx950: do special VALUES cost time?  v_sqrt_f32 of negative inputs (NaN results), v_add_f32 /
// v_min_u32 on NaN operands, v_rcp_f32 of zero — the closed-box intersection of round 3 took square roots of negative
// discriminants unconditionally and ran slower than its instruction count predicts.  Also: dependent chains (ILP 1, 2, 4, 8)
// at 6 waves per SIMD, the occupancy of the path tracer.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/valu_microbench4 tools/valu_microbench4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kIters = 2048;
// sqrt of a value that stays what it is: r = sqrt(a) is written to a scratch register, a is never overwritten
#define SQ(k) "v_sqrt_f32_e32 %8, %" #k "\n\t"
#define RC(k) "v_rcp_f32_e32 %8, %" #k "\n\t"
#define AD(k) "v_add_f32_e32 %8, %" #k ", %9\n\t"
#define MN(k) "v_min_u32_e32 %8, %" #k ", %9\n\t"
#define R16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define KV(NAME, PAT)                                                                               \
    __global__ void __launch_bounds__(512) NAME(float* out, float v) {                               \
        float a0 = v, a1 = v * 2, a2 = v * 3, a3 = v * 4, a4 = v * 5, a5 = v * 6, a6 = v * 7, a7 = v * 8, r = 0.0f, c = v; \
        for (int i = 0; i < kIters; i++)                                                             \
            asm volatile(PAT : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(r) : "v"(c)); \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r + a0;                                         \
    }
KV(k_sqrt, R16(SQ)) KV(k_rcp, R16(RC)) KV(k_add, R16(AD)) KV(k_min, R16(MN))
// dependent chains of v_add_f32 / v_max_f32 / v_sqrt_f32: 16 instructions per trip over `ILP` independent registers
#define D1(I) I " %0, %0, %8\n\t"
#define CH(I, k) I " %" #k ", %" #k ", %8\n\t"
#define CHAIN1(I) CH(I,0) CH(I,0) CH(I,0) CH(I,0) CH(I,0) CH(I,0) CH(I,0) CH(I,0) CH(I,0) CH(I,0) CH(I,0) CH(I,0) CH(I,0) CH(I,0) CH(I,0) CH(I,0)
#define CHAIN2(I) CH(I,0) CH(I,1) CH(I,0) CH(I,1) CH(I,0) CH(I,1) CH(I,0) CH(I,1) CH(I,0) CH(I,1) CH(I,0) CH(I,1) CH(I,0) CH(I,1) CH(I,0) CH(I,1)
#define CHAIN4(I) CH(I,0) CH(I,1) CH(I,2) CH(I,3) CH(I,0) CH(I,1) CH(I,2) CH(I,3) CH(I,0) CH(I,1) CH(I,2) CH(I,3) CH(I,0) CH(I,1) CH(I,2) CH(I,3)
#define KC(NAME, PAT)                                                                               \
    __global__ void __launch_bounds__(512) NAME(float* out, float v) {                               \
        float a0 = v, a1 = v * 2, a2 = v * 3, a3 = v * 4, a4 = v * 5, a5 = v * 6, a6 = v * 7, a7 = v * 8, b = 1.0f, c = v; \
        for (int i = 0; i < kIters; i++)                                                             \
            asm volatile(PAT : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b) : "v"(c)); \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;                              \
    }
KC(k_add1, CHAIN1("v_add_f32_e32")) KC(k_add2, CHAIN2("v_add_f32_e32")) KC(k_add4, CHAIN4("v_add_f32_e32"))
KC(k_max1, CHAIN1("v_max_f32_e32")) KC(k_max2, CHAIN2("v_max_f32_e32")) KC(k_max4, CHAIN4("v_max_f32_e32"))
#define TU(k) "v_sqrt_f32_e32 %" #k ", %" #k "\n\t"
#define TCH1 TU(0) TU(0) TU(0) TU(0) TU(0) TU(0) TU(0) TU(0) TU(0) TU(0) TU(0) TU(0) TU(0) TU(0) TU(0) TU(0)
#define TCH2 TU(0) TU(1) TU(0) TU(1) TU(0) TU(1) TU(0) TU(1) TU(0) TU(1) TU(0) TU(1) TU(0) TU(1) TU(0) TU(1)
KC(k_sq1, TCH1) KC(k_sq2, TCH2)
// the sphere-candidate pattern: trans then three dependent fast/slow instructions, ILP 1 and 3
#define CAND(k) "v_sqrt_f32_e32 %8, %" #k "\n\tv_sub_f32_e32 %" #k ", %9, %8\n\tv_add_f32_e32 %8, %9, %8\n\tv_min_u32_e32 %" #k ", %" #k ", %8\n\t"
KC(k_cand1, CAND(0) CAND(0) CAND(0) CAND(0))
struct Entry { const char* name; void (*fn)(float*, float); float v; };
int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* out; CHECK(hipMalloc(&out, sizeof(float) * 512 * cus * 8));
    const float nan = __builtin_nanf("");
    std::vector<Entry> es = {
        {"v_sqrt_f32 of positive", k_sqrt, 3.0f}, {"v_sqrt_f32 of negative (NaN out)", k_sqrt, -3.0f}, {"v_sqrt_f32 of NaN", k_sqrt, nan},
        {"v_sqrt_f32 of denormal", k_sqrt, 1e-40f}, {"v_rcp_f32 of positive", k_rcp, 3.0f}, {"v_rcp_f32 of zero", k_rcp, 0.0f},
        {"v_add_f32 normal", k_add, 3.0f}, {"v_add_f32 NaN operand", k_add, nan}, {"v_add_f32 denormal operands", k_add, 1e-40f},
        {"v_min_u32 normal", k_min, 3.0f}, {"v_min_u32 NaN bits", k_min, nan},
        {"v_add_f32 dependent chain ILP 1", k_add1, 1.0f}, {"v_add_f32 ILP 2", k_add2, 1.0f}, {"v_add_f32 ILP 4", k_add4, 1.0f},
        {"v_max_f32 dependent chain ILP 1", k_max1, 1.0f}, {"v_max_f32 ILP 2", k_max2, 1.0f}, {"v_max_f32 ILP 4", k_max4, 1.0f},
        {"v_sqrt_f32 dependent chain ILP 1", k_sq1, 3.0f}, {"v_sqrt_f32 ILP 2", k_sq2, 3.0f},
        {"sqrt -> sub, add -> min_u32 (serial)", k_cand1, 3.0f},
    };
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps : {2, 6}) {
        printf("---- %d waves per SIMD\n%-42s %10s %s\n", wps, "stream", "ms", "cycles per wave-instruction per SIMD @2.4 GHz");
        for (auto& e : es) {
            const int blocks = cus * (wps / 2);
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(512), 0, 0, out, e.v); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 5; r++) hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(512), 0, 0, out, e.v);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
            const double winst = (double)blocks * 8 * kIters * 16;
            printf("%-42s %10.4f %8.3f\n", e.name, ms, (ms * 1e-3) * 2.4e9 * (cus * 4.0) / winst);
        }
    }
    return 0;
}
