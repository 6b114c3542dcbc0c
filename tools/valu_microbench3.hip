// Third VALU survey for gfx950: how the issue classes COMBINE.  valu_microbench{,2}.hip priced each opcode in isolation
// (add/mul/fmac/mov/logic ~2.3 cycles per wave64 instruction per SIMD, compare/select/min/max/convert/3-operand integer ~4.2,
// transcendental ~8.2).  The path tracer's measured time is well under the sum of those prices, and a rewrite that removed
// 4 % of its instructions made it slower — so: do classes overlap when they come from different waves, from one wave, and
// what does an instruction cost whose EXEC mask is empty?
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/valu_microbench3 tools/valu_microbench3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
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
int main() {
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
