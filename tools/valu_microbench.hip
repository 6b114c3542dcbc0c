// VALU issue-rate microbenchmark for gfx950: measures wave-instructions per cycle per SIMD for the
// instruction classes the Mandelbrot / path-tracer kernels are made of, at several occupancies.
// Used to fix the roofline denominators in DESIGN.md (is v_pk_*_f32 2 flops/lane/issue or half rate?
// what do v_rcp/v_rsq/v_sqrt/v_sin/v_mul_lo_u32/v_cmp cost relative to v_add_f32?).
//
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_microbench tools/valu_microbench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                     \
    do {                                                                             \
        hipError_t e = (x);                                                          \
        if (e != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                   \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

constexpr int kIters = 4096;   // loop trips
constexpr int kUnroll = 16;    // instructions per trip (8 independent chains x 2)

// 8 independent accumulator chains so dependent-issue latency never limits throughput.
#define BODY1(INS)                                                                                        \
    asm volatile(INS " %0, %0, %8\n\t" INS " %1, %1, %8\n\t" INS " %2, %2, %8\n\t" INS " %3, %3, %8\n\t"  \
                 INS " %4, %4, %8\n\t" INS " %5, %5, %8\n\t" INS " %6, %6, %8\n\t" INS " %7, %7, %8\n\t"  \
                 INS " %0, %0, %8\n\t" INS " %1, %1, %8\n\t" INS " %2, %2, %8\n\t" INS " %3, %3, %8\n\t"  \
                 INS " %4, %4, %8\n\t" INS " %5, %5, %8\n\t" INS " %6, %6, %8\n\t" INS " %7, %7, %8\n\t"  \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)         \
                 : "v"(b))
#define BODY_UN(INS)                                                                                      \
    asm volatile(INS " %0, %0\n\t" INS " %1, %1\n\t" INS " %2, %2\n\t" INS " %3, %3\n\t"                  \
                 INS " %4, %4\n\t" INS " %5, %5\n\t" INS " %6, %6\n\t" INS " %7, %7\n\t"                  \
                 INS " %0, %0\n\t" INS " %1, %1\n\t" INS " %2, %2\n\t" INS " %3, %3\n\t"                  \
                 INS " %4, %4\n\t" INS " %5, %5\n\t" INS " %6, %6\n\t" INS " %7, %7\n\t"                  \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))
#define BODY_FMA(INS)                                                                                                     \
    asm volatile(INS " %0, %0, %8, %8\n\t" INS " %1, %1, %8, %8\n\t" INS " %2, %2, %8, %8\n\t" INS " %3, %3, %8, %8\n\t"  \
                 INS " %4, %4, %8, %8\n\t" INS " %5, %5, %8, %8\n\t" INS " %6, %6, %8, %8\n\t" INS " %7, %7, %8, %8\n\t"  \
                 INS " %0, %0, %8, %8\n\t" INS " %1, %1, %8, %8\n\t" INS " %2, %2, %8, %8\n\t" INS " %3, %3, %8, %8\n\t"  \
                 INS " %4, %4, %8, %8\n\t" INS " %5, %5, %8, %8\n\t" INS " %6, %6, %8, %8\n\t" INS " %7, %7, %8, %8\n\t"  \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                         \
                 : "v"(b))
#define BODY_CMP(INS)                                                                                                     \
    asm volatile(INS " vcc, %0, %8\n\t" INS " vcc, %1, %8\n\t" INS " vcc, %2, %8\n\t" INS " vcc, %3, %8\n\t"              \
                 INS " vcc, %4, %8\n\t" INS " vcc, %5, %8\n\t" INS " vcc, %6, %8\n\t" INS " vcc, %7, %8\n\t"              \
                 INS " vcc, %0, %8\n\t" INS " vcc, %1, %8\n\t" INS " vcc, %2, %8\n\t" INS " vcc, %3, %8\n\t"              \
                 INS " vcc, %4, %8\n\t" INS " vcc, %5, %8\n\t" INS " vcc, %6, %8\n\t" INS " vcc, %7, %8\n\t"              \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                         \
                 : "v"(b)                                                                                                 \
                 : "vcc")

#define KERNEL32(NAME, BODY)                                                     \
    __global__ void __launch_bounds__(256) NAME(float* out, float seed) {        \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;            \
        float a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;        \
        float b = seed * 0.5f + 1.0f;                                            \
        for (int i = 0; i < kIters; i++) { BODY; }                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; \
    }
typedef float float2v __attribute__((ext_vector_type(2)));
#define KERNEL64(NAME, BODY)                                                     \
    __global__ void __launch_bounds__(256) NAME(float* out, float seed) {        \
        float2v a0 = {seed, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;  \
        float2v a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;      \
        float2v b = a0 * 0.5f + 1.0f;                                            \
        for (int i = 0; i < kIters; i++) { BODY; }                               \
        float2v r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r.x + r.y;                  \
    }
#define KERNELU32(NAME, BODY)                                                    \
    __global__ void __launch_bounds__(256) NAME(float* out, float seed) {        \
        unsigned a0 = (unsigned)seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;     \
        unsigned a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;             \
        unsigned b = a0 * 3u + 12345u;                                           \
        for (int i = 0; i < kIters; i++) { BODY; }                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7); \
    }

KERNEL32(k_add, BODY1("v_add_f32"))
KERNEL32(k_mul, BODY1("v_mul_f32"))
KERNEL32(k_fma, BODY_FMA("v_fma_f32"))
KERNEL32(k_max, BODY1("v_max_f32"))
KERNEL32(k_cmp, BODY_CMP("v_cmp_gt_f32"))
KERNEL32(k_rcp, BODY_UN("v_rcp_f32"))
KERNEL32(k_rsq, BODY_UN("v_rsq_f32"))
KERNEL32(k_sqrt, BODY_UN("v_sqrt_f32"))
KERNEL32(k_sin, BODY_UN("v_sin_f32"))
KERNEL32(k_exp, BODY_UN("v_exp_f32"))
KERNEL32(k_log, BODY_UN("v_log_f32"))
KERNEL32(k_cvt, BODY_UN("v_cvt_f32_u32"))
KERNEL32(k_rndne, BODY_UN("v_rndne_f32"))
KERNEL64(k_pk_add, BODY1("v_pk_add_f32"))
KERNEL64(k_pk_mul, BODY1("v_pk_mul_f32"))
KERNEL64(k_pk_fma, BODY_FMA("v_pk_fma_f32"))
KERNELU32(k_mullo, BODY1("v_mul_lo_u32"))
KERNELU32(k_xor, BODY1("v_xor_b32"))
KERNELU32(k_lshr, BODY1("v_lshrrev_b32"))
KERNELU32(k_addu, BODY1("v_add_u32"))
// second series (round 1, after the fast-math contraction result): the "other" class of the issue model
#define BODY_CNDMASK                                                                                       \
    asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\t" \
                 "v_cndmask_b32 %3, %3, %8, vcc\n\tv_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\t" \
                 "v_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc\n\tv_cndmask_b32 %0, %0, %8, vcc\n\t" \
                 "v_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\t" \
                 "v_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\t" \
                 "v_cndmask_b32 %7, %7, %8, vcc\n\t"                                                                    \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                         \
                 : "v"(b)                                                                                                 \
                 : "vcc")
KERNELU32(k_mov, BODY_UN("v_mov_b32"))
KERNELU32(k_cndmask, BODY_CNDMASK)
KERNELU32(k_and, BODY1("v_and_b32"))
KERNEL32(k_fmac, BODY1("v_fmac_f32"))
KERNEL32(k_sub, BODY1("v_sub_f32"))

struct Entry {
    const char* name;
    void (*fn)(float*, float);
    int flops_per_lane;   // fp32 flops per lane per instruction (0 for non-fp)
};

int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s  CUs %d  clock %d kHz\n", prop.name, cus, prop.clockRate);
    float* out;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * cus * 64));
    std::vector<Entry> es = {
        {"v_add_f32", k_add, 1},     {"v_mul_f32", k_mul, 1},       {"v_fma_f32", k_fma, 2},     {"v_max_f32", k_max, 1},
        {"v_cmp_gt_f32", k_cmp, 0},  {"v_pk_add_f32", k_pk_add, 2}, {"v_pk_mul_f32", k_pk_mul, 2}, {"v_pk_fma_f32", k_pk_fma, 4},
        {"v_rcp_f32", k_rcp, 1},     {"v_rsq_f32", k_rsq, 1},       {"v_sqrt_f32", k_sqrt, 1},   {"v_sin_f32", k_sin, 1},
        {"v_exp_f32", k_exp, 1},     {"v_log_f32", k_log, 1},       {"v_cvt_f32_u32", k_cvt, 0}, {"v_rndne_f32", k_rndne, 0},
        {"v_mul_lo_u32", k_mullo, 0}, {"v_xor_b32", k_xor, 0},      {"v_lshrrev_b32", k_lshr, 0}, {"v_add_u32", k_addu, 0},
        {"v_mov_b32", k_mov, 0},     {"v_cndmask_b32", k_cndmask, 0}, {"v_and_b32", k_and, 0},   {"v_fmac_f32", k_fmac, 2},
        {"v_sub_f32", k_sub, 1},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%-16s %5s %10s %14s %12s %10s\n", "instr", "w/SIMD", "ms", "Gwaveinst/s", "cyc/inst@2.4", "TFLOP/s");
    for (auto& e : es) {
        for (int wps : {1, 2, 4, 8}) {
            // wps waves per SIMD: blocks of 256 threads = 4 waves = 1 wave per SIMD; launch wps blocks per CU
            int blocks = cus * wps;
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0f);   // warm-up
            CHECK(hipDeviceSynchronize());
            const int reps = 5;
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++) hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            ms /= reps;
            double winst = (double)blocks * 4 * kIters * kUnroll;           // wave-instructions
            double rate = winst / (ms * 1e-3);                              // per second, whole chip
            double per_simd_cycle = rate / (cus * 4.0) / 2.4e9;             // wave-inst per SIMD per cycle @2.4GHz
            double tflops = rate * 64.0 * e.flops_per_lane / 1e12;
            printf("%-16s %5d %10.4f %14.2f %12.3f %10.2f\n", e.name, wps, ms, rate / 1e9, 1.0 / per_simd_cycle, tflops);
        }
    }
    CHECK(hipFree(out));
    return 0;
}
