// Seventh VALU survey for gfx950: does it matter WHERE the transcendentals of a stream sit?  valu_microbench3 measured a lone
// v_rcp_f32 among adds at ~11.6 cycles (F F F T: 4.7 per instruction) against 8.2 in a stream of its own — is the difference a
// price per SWITCH between the two pipes (then grouping the transcendentals of a block back to back pays), and does the
// dependency of the next instruction on the transcendental's result matter?
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/valu_microbench7 tools/valu_microbench7.hip
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
int main() {
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
