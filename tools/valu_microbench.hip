// VALU issue-cost surveys of gfx950 (MI355X) — ONE program with four modes (rounds 1-4 kept them as four files):
//   valu_microbench classes   every opcode class in isolation (profiles/r01_valu_microbench.txt)
//   valu_microbench mix       how the issue classes combine; instructions under EXEC = 0 (profiles/r03_valu_microbench3.txt)
//   valu_microbench among     a transcendental among other instructions (profiles/r03_valu_microbench7.txt)
//   valu_microbench opcodes   every transcendental opcode, f32 and f16, alone and in a mix (profiles/r04_valu_microbench8.txt)
// These are the measurements the path tracer's design rests on: add / mul / fmac / mov / logic issue in ~2.3 cycles
// per wave64 instruction per SIMD, compare / select / min / max / convert / three-operand integer in ~4.2 alone but ~2.4 in a mix,
// transcendentals in ~8.2 wherever they stand.  Each mode is the former file's code, unchanged, in a namespace of its own.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/valu_microbench tools/valu_microbench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// ======================================================================================================================
// mode "classes" — formerly tools/valu_microbench.hip
// VALU issue-rate microbenchmark for gfx950: measures wave-instructions per cycle per SIMD for the
// instruction classes the Mandelbrot / path-tracer kernels are made of, at several occupancies.
// Used to fix the roofline denominators in DESIGN.md (is v_pk_*_f32 2 flops/lane/issue or half rate?
// what do v_rcp/v_rsq/v_sqrt/v_sin/v_mul_lo_u32/v_cmp cost relative to v_add_f32?).
//
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_microbench tools/valu_microbench.hip
namespace classes {
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

static int run() {
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
}  // namespace classes
#undef CHECK
#undef BODY1
#undef BODY_UN
#undef BODY_FMA
#undef BODY_CMP
#undef KERNEL32
#undef KERNEL64
#undef KERNELU32
#undef BODY_CNDMASK

// ======================================================================================================================
// mode "mix" — formerly tools/valu_microbench3.hip
// Third VALU survey for gfx950: how the issue classes COMBINE.  valu_microbench{,2}.hip priced each opcode in isolation
// (add/mul/fmac/mov/logic ~2.3 cycles per wave64 instruction per SIMD, compare/select/min/max/convert/3-operand integer ~4.2,
// transcendental ~8.2).  The path tracer's measured time is well under the sum of those prices, and a rewrite that removed
// 4 % of its instructions made it slower — so: do classes overlap when they come from different waves, from one wave, and
// what does an instruction cost whose EXEC mask is empty?
namespace mix {
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kIters = 2048;
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(sm)
#define F(k) "v_add_f32_e32 %" #k ", %" #k ", %8\n\t"
#define S(k) "v_max_f32_e32 %" #k ", %" #k ", %8\n\t"
#define C(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, %10\n\t"
#define M(k) "v_cmp_lt_f32_e64 s[20:21], %" #k ", %8\n\t"
#define T(k) "v_rcp_f32_e32 %" #k ", %" #k "\n\t"
#define I(k) "v_min_u32_e32 %" #k ", %" #k ", %8\n\t"
#define A(k) "v_and_or_b32 %" #k ", %" #k ", %8, %9\n\t"
// 16 instructions per trip on 8 independent chains
#define P_FAST   F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
#define P_SLOW   S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define P_SEL    C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7)
#define P_TRANS  T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
#define P_FS     F(0) S(1) F(2) S(3) F(4) S(5) F(6) S(7) F(1) S(0) F(3) S(2) F(5) S(4) F(7) S(6)
#define P_FC     F(0) C(1) F(2) C(3) F(4) C(5) F(6) C(7) F(1) C(0) F(3) C(2) F(5) C(4) F(7) C(6)
#define P_FM     F(0) M(1) F(2) M(3) F(4) M(5) F(6) M(7) F(1) M(0) F(3) M(2) F(5) M(4) F(7) M(6)
#define P_FFFT   F(0) F(1) F(2) T(3) F(4) F(5) F(6) T(7) F(1) F(2) F(3) T(0) F(5) F(6) F(7) T(4)
#define P_FFS    F(0) F(1) S(2) F(3) F(4) S(5) F(6) F(7) S(0) F(1) F(2) S(3) F(4) F(5) S(6) F(7)
#define P_FIA    F(0) I(1) F(2) A(3) F(4) I(5) F(6) A(7) F(1) I(0) F(3) A(2) F(5) I(4) F(7) A(6)
#define KERNEL(NAME, PAT)                                                                           \
    __global__ void __launch_bounds__(512) NAME(float* out, float seed, int mode) {                  \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
        float b = seed * 0.5f + 1.0f, c = seed * 0.25f + 2.0f;                                      \
        unsigned long long sm = 0x5555aaaa3333ccccull;                                               \
        for (int i = 0; i < kIters; i++) asm volatile(PAT OPS : "vcc", "s20", "s21");                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;          \
    }
KERNEL(k_fast, P_FAST) KERNEL(k_slow, P_SLOW) KERNEL(k_sel, P_SEL) KERNEL(k_trans, P_TRANS) KERNEL(k_fs, P_FS) KERNEL(k_fc, P_FC)
KERNEL(k_fm, P_FM) KERNEL(k_ffft, P_FFFT) KERNEL(k_ffs, P_FFS) KERNEL(k_fia, P_FIA)
// two programs on one SIMD: waves 0-3 of a 512-thread block run pattern X, waves 4-7 pattern Y (a block's waves go to the
// SIMDs cyclically, so wave w and wave w + 4 share one)
#define KERNEL2(NAME, PX, PY)                                                                        \
    __global__ void __launch_bounds__(512) NAME(float* out, float seed, int mode) {                   \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
        float b = seed * 0.5f + 1.0f, c = seed * 0.25f + 2.0f;                                       \
        unsigned long long sm = 0x5555aaaa3333ccccull;                                                \
        if ((threadIdx.x >> 8) == 0) { for (int i = 0; i < kIters; i++) asm volatile(PX OPS : "vcc", "s20", "s21"); } \
        else { for (int i = 0; i < kIters; i++) asm volatile(PY OPS : "vcc", "s20", "s21"); }        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;           \
    }
KERNEL2(k2_fast_slow, P_FAST, P_SLOW) KERNEL2(k2_fast_trans, P_FAST, P_TRANS) KERNEL2(k2_slow_trans, P_SLOW, P_TRANS)
KERNEL2(k2_fast_sel, P_FAST, P_SEL)
// EXEC = 0: the same streams issued with an empty mask
#define KERNEL0(NAME, PAT)                                                                          \
    __global__ void __launch_bounds__(512) NAME(float* out, float seed, int mode) {                  \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
        float b = seed * 0.5f + 1.0f, c = seed * 0.25f + 2.0f;                                      \
        unsigned long long sm = 0x5555aaaa3333ccccull;                                               \
        for (int i = 0; i < kIters; i++)                                                             \
            asm volatile("s_mov_b64 s[22:23], exec\n\ts_mov_b64 exec, 0\n\t" PAT "s_mov_b64 exec, s[22:23]\n\t" OPS : "vcc", "s20", "s21", "s22", "s23"); \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;          \
    }
KERNEL0(k0_fast, P_FAST) KERNEL0(k0_slow, P_SLOW) KERNEL0(k0_sel, P_SEL) KERNEL0(k0_trans, P_TRANS)
struct Entry { const char* name; void (*fn)(float*, float, int); };
static int run() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* out; CHECK(hipMalloc(&out, sizeof(float) * 512 * cus * 8));
    std::vector<Entry> es = {
        {"fast  (v_add_f32) x16", k_fast}, {"slow  (v_max_f32) x16", k_slow}, {"sel   (v_cndmask sgpr) x16", k_sel}, {"trans (v_rcp_f32) x16", k_trans},
        {"one wave: F S F S ...", k_fs}, {"one wave: F C F C ... (cndmask)", k_fc}, {"one wave: F M F M ... (v_cmp->sgpr)", k_fm},
        {"one wave: F F F T ...", k_ffft}, {"one wave: F F S ...", k_ffs}, {"one wave: F min_u32 F and_or ...", k_fia},
        {"two waves/SIMD: fast | slow", k2_fast_slow}, {"two waves/SIMD: fast | trans", k2_fast_trans},
        {"two waves/SIMD: slow | trans", k2_slow_trans}, {"two waves/SIMD: fast | sel", k2_fast_sel},
        {"EXEC=0 fast x16", k0_fast}, {"EXEC=0 slow x16", k0_slow}, {"EXEC=0 sel x16", k0_sel}, {"EXEC=0 trans x16", k0_trans},
    };
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps : {2, 4, 6}) {   // waves per SIMD: 512-thread blocks, 1 / 2 / 3 blocks per CU
        printf("---- %d waves per SIMD\n%-40s %10s %s\n", wps, "stream", "ms", "cycles per wave-instruction per SIMD @2.4 GHz");
        for (auto& e : es) {
            const int blocks = cus * (wps / 2);
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(512), 0, 0, out, 1.0f, 0); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 5; r++) hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(512), 0, 0, out, 1.0f, 0);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
            const double winst = (double)blocks * 8 * kIters * 16;
            printf("%-40s %10.4f %8.3f\n", e.name, ms, (ms * 1e-3) * 2.4e9 * (cus * 4.0) / winst);
        }
    }
    return 0;
}
}  // namespace mix
#undef CHECK
#undef OPS
#undef F
#undef S
#undef C
#undef M
#undef T
#undef I
#undef A
#undef P_FAST
#undef P_SLOW
#undef P_SEL
#undef P_TRANS
#undef P_FS
#undef P_FC
#undef P_FM
#undef P_FFFT
#undef P_FFS
#undef P_FIA
#undef KERNEL
#undef KERNEL2
#undef KERNEL0

// ======================================================================================================================
// mode "among" — formerly tools/valu_microbench7.hip
// Seventh VALU survey for gfx950: does it matter WHERE the transcendentals of a stream sit?  valu_microbench3 measured a lone
// v_rcp_f32 among adds at ~11.6 cycles (F F F T: 4.7 per instruction) against 8.2 in a stream of its own — is the difference a
// price per SWITCH between the two pipes (then grouping the transcendentals of a block back to back pays), and does the
// dependency of the next instruction on the transcendental's result matter?
namespace among {
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kIters = 2048;
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(sm)
#define F(k) "v_add_f32_e32 %" #k ", %" #k ", %8\n\t"
#define S(k) "v_max_f32_e32 %" #k ", %" #k ", %8\n\t"
#define C(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, %10\n\t"
#define M(k) "v_cmp_lt_f32_e64 s[20:21], %" #k ", %8\n\t"
#define T(k) "v_rcp_f32_e32 %" #k ", %" #k "\n\t"
#define I(k) "v_min_u32_e32 %" #k ", %" #k ", %8\n\t"
#define A(k) "v_and_or_b32 %" #k ", %" #k ", %8, %9\n\t"
#define Q(k) "v_sqrt_f32_e32 %" #k ", %" #k "\n\t"
// 16 instructions per trip: 12 adds + 4 transcendentals, differently placed (8 independent chains)
#define P_FFFT   F(0) F(1) F(2) T(3) F(4) F(5) F(6) T(7) F(1) F(2) F(3) T(0) F(5) F(6) F(7) T(4)
#define P_PAIR   F(0) F(1) F(2) F(3) F(4) F(5) T(6) T(7) F(0) F(1) F(2) F(3) F(6) F(7) T(4) T(5)
#define P_QUAD   F(0) F(1) F(2) F(3) F(0) F(1) F(2) F(3) F(0) F(1) F(2) F(3) T(4) T(5) T(6) T(7)
#define P_QUADQ  F(0) F(1) F(2) F(3) F(0) F(1) F(2) F(3) F(0) F(1) F(2) F(3) T(4) Q(5) T(6) Q(7)
// the instruction after a transcendental uses its result at once / the transcendental uses the result of the add before it
#define P_DEP    F(0) F(1) T(2) F(2) F(4) F(5) T(6) F(6) F(1) F(0) T(3) F(3) F(5) F(4) T(7) F(7)
#define P_DEPIN  F(0) F(1) F(2) T(2) F(4) F(5) F(6) T(6) F(1) F(0) F(3) T(3) F(5) F(4) F(7) T(7)
// 8 + 8, 14 + 2
#define P_FT     F(0) T(1) F(2) T(3) F(4) T(5) F(6) T(7) F(1) T(0) F(3) T(2) F(5) T(4) F(7) T(6)
#define P_F7T    F(0) F(1) F(2) F(3) F(4) F(5) F(6) T(7) F(0) F(1) F(2) F(3) F(4) F(5) F(7) T(6)
#define P_F14TT  F(0) F(1) F(2) F(3) F(4) F(5) F(0) F(1) F(2) F(3) F(4) F(5) F(0) F(1) T(6) T(7)
// slow-class neighbours: S S S T
#define P_SSST   S(0) S(1) S(2) T(3) S(4) S(5) S(6) T(7) S(1) S(2) S(3) T(0) S(5) S(6) S(7) T(4)
#define P_FAST   F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
#define P_TRANS  T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
#define KERNEL(NAME, PAT)                                                                           \
    __global__ void __launch_bounds__(512) NAME(float* out, float seed, int mode) {                  \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
        float b = seed * 0.5f + 1.0f, c = seed * 0.25f + 2.0f;                                      \
        unsigned long long sm = 0x5555aaaa3333ccccull;                                               \
        for (int i = 0; i < kIters; i++) asm volatile(PAT OPS : "vcc", "s20", "s21");                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;          \
    }
KERNEL(k_fast, P_FAST) KERNEL(k_trans, P_TRANS) KERNEL(k_ffft, P_FFFT) KERNEL(k_pair, P_PAIR) KERNEL(k_quad, P_QUAD) KERNEL(k_quadq, P_QUADQ)
KERNEL(k_dep, P_DEP) KERNEL(k_depin, P_DEPIN) KERNEL(k_ft, P_FT) KERNEL(k_f7t, P_F7T) KERNEL(k_f14tt, P_F14TT) KERNEL(k_ssst, P_SSST)
struct Entry { const char* name; void (*fn)(float*, float, int); };
static int run() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* out; CHECK(hipMalloc(&out, sizeof(float) * 512 * cus * 8));
    std::vector<Entry> es = {
        {"F x16", k_fast}, {"T x16", k_trans}, {"(F F F T) x4        12 F + 4 T, lone", k_ffft}, {"(F x6 T T) x2       12 F + 4 T, pairs", k_pair},
        {"F x12 T x4          12 F + 4 T, one group", k_quad}, {"F x12 rcp sqrt rcp sqrt", k_quadq},
        {"F F T F(dep) ...    result used at once", k_dep}, {"F F F(dep) T ...    operand just made", k_depin},
        {"(F T) x8", k_ft}, {"(F x7 T) x2", k_f7t}, {"F x14 T T", k_f14tt}, {"(S S S T) x4", k_ssst},
    };
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps : {2, 4, 6}) {   // waves per SIMD: 512-thread blocks, 1 / 2 / 3 blocks per CU
        printf("---- %d waves per SIMD\n%-40s %10s %s\n", wps, "stream", "ms", "cycles per wave-instruction per SIMD @2.4 GHz");
        for (auto& e : es) {
            const int blocks = cus * (wps / 2);
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(512), 0, 0, out, 1.0f, 0); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 5; r++) hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(512), 0, 0, out, 1.0f, 0);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
            const double winst = (double)blocks * 8 * kIters * 16;
            printf("%-40s %10.4f %8.3f\n", e.name, ms, (ms * 1e-3) * 2.4e9 * (cus * 4.0) / winst);
        }
    }
    return 0;
}
}  // namespace among
#undef CHECK
#undef OPS
#undef F
#undef S
#undef C
#undef M
#undef T
#undef I
#undef A
#undef Q
#undef P_FFFT
#undef P_PAIR
#undef P_QUAD
#undef P_QUADQ
#undef P_DEP
#undef P_DEPIN
#undef P_FT
#undef P_F7T
#undef P_F14TT
#undef P_SSST
#undef P_FAST
#undef P_TRANS
#undef KERNEL

// ======================================================================================================================
// mode "opcodes" — formerly tools/valu_microbench8.hip
// Eighth VALU survey for gfx950 (round 4): which transcendental opcodes cost what?  The fast path tracer spends 24 % of its time on
// 4.9 % of its instructions (profiles/r04_no_trans_pmc.txt: 13.2 cycles per transcendental among other instructions).  Rounds 1-3
// only timed v_rcp_f32 / v_sqrt_f32; this one times every opcode the kernels use or could use — v_rcp, v_rsq, v_sqrt, v_sin, v_cos,
// v_exp, v_log in f32 and the f16 forms — alone (16 per trip) and as 4 among 12 v_fmac_f32 (the in-kernel situation), at 6 waves per SIMD.
// If an f16 form or exp / log were markedly cheaper, a seed + Newton step could replace an f32 transcendental.
namespace opcodes {
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
static int run() {
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
}  // namespace opcodes
#undef CHECK
#undef OPS
#undef F
#undef ALONE
#undef MIXED
#undef KERNEL
#undef DEF
#undef T_rcp32
#undef T_rsq32
#undef T_sqrt32
#undef T_sin32
#undef T_cos32
#undef T_exp32
#undef T_log32
#undef T_rcp16
#undef T_rsq16
#undef T_sqrt16
#undef T_sin16
#undef T_exp16
#undef T_fma
#undef T_cvt
#undef E

int main(int argc, char** argv) {
    const std::string m = argc > 1 ? argv[1] : "";
    if (m == "classes") return classes::run();
    if (m == "mix") return mix::run();
    if (m == "among") return among::run();
    if (m == "opcodes") return opcodes::run();
    fprintf(stderr, "usage: valu_microbench classes | mix | among | opcodes\n");
    return 2;
}
