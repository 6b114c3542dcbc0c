// Second VALU issue-rate survey for gfx950: the "other" class of the path tracer's instruction mix (28 % of its VALU
// instructions): selects, compares, min/max/med3, conversions, 3-operand integer ops, bit-field ops, cross-lane reads.
// Same method as valu_microbench.hip (8 independent chains, 16 instructions per trip, 4 waves per SIMD).
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/valu_microbench2 tools/valu_microbench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kIters = 4096;
#define R8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(sm)
// each macro emits one instruction on chain k (operands %k, %8 = b, %9 = c, %10 = 64-bit SGPR mask)
#define I_CNDMASK_VCC(k) "v_cndmask_b32_e32 %" #k ", %" #k ", %8, vcc\n\t"
#define I_CNDMASK_SGPR(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, %10\n\t"
#define I_CMP_F32_VCC(k) "v_cmp_lt_f32_e32 vcc, %" #k ", %8\n\t"
#define I_CMP_F32_SGPR(k) "v_cmp_lt_f32_e64 s[20:21], %" #k ", %8\n\t"
#define I_CMP_U32_VCC(k) "v_cmp_lt_u32_e32 vcc, %" #k ", %8\n\t"
#define I_CMP_EQ_U32(k) "v_cmp_eq_u32_e32 vcc, %" #k ", %8\n\t"
#define I_MAX_F32(k) "v_max_f32_e32 %" #k ", %" #k ", %8\n\t"
#define I_MIN_U32(k) "v_min_u32_e32 %" #k ", %" #k ", %8\n\t"
#define I_MED3(k) "v_med3_f32 %" #k ", %" #k ", %8, %9\n\t"
#define I_MAX3(k) "v_max3_f32 %" #k ", %" #k ", %8, %9\n\t"
#define I_BFI(k) "v_bfi_b32 %" #k ", %8, %" #k ", %9\n\t"
#define I_AND_OR(k) "v_and_or_b32 %" #k ", %" #k ", %8, %9\n\t"
#define I_LSHL_ADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 2, %8\n\t"
#define I_ADD3(k) "v_add3_u32 %" #k ", %" #k ", %8, %9\n\t"
#define I_MAD_U24(k) "v_mad_u32_u24 %" #k ", %" #k ", %8, %9\n\t"
#define I_MUL_U24(k) "v_mul_u32_u24_e32 %" #k ", %" #k ", %8\n\t"
#define I_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 4, 8\n\t"
#define I_CVT_U32_F32(k) "v_cvt_u32_f32_e32 %" #k ", %" #k "\n\t"
#define I_FLOOR(k) "v_floor_f32_e32 %" #k ", %" #k "\n\t"
#define I_FRACT(k) "v_fract_f32_e32 %" #k ", %" #k "\n\t"
#define I_MBCNT(k) "v_mbcnt_lo_u32_b32 %" #k ", %8, %" #k "\n\t"
#define I_FMA_SAME(k) "v_fma_f32 %" #k ", %" #k ", %8, %" #k "\n\t"
#define I_FMA_NEG(k) "v_fma_f32 %" #k ", %" #k ", %" #k ", -%8\n\t"
#define I_MUL_ABS(k) "v_mul_f32_e64 %" #k ", |%" #k "|, %8\n\t"
#define I_ADD_NEG(k) "v_add_f32_e64 %" #k ", -%" #k ", %8\n\t"
#define I_SUBREV(k) "v_subrev_f32_e32 %" #k ", %8, %" #k "\n\t"
#define I_ADD_SGPR(k) "v_add_f32_e32 %" #k ", s20, %" #k "\n\t"
#define I_ADD_LIT(k) "v_add_f32_e32 %" #k ", 0x3f99999a, %" #k "\n\t"
#define I_MUL_LEGACY(k) "v_mul_legacy_f32_e64 %" #k ", %" #k ", %8\n\t"
#define I_XOR3(k) "v_xor3_b32 %" #k ", %" #k ", %8, %9\n\t"
#define I_OR3(k) "v_or3_b32 %" #k ", %" #k ", %8, %9\n\t"
#define I_LSHL_OR(k) "v_lshl_or_b32 %" #k ", %" #k ", 3, %8\n\t"
#define I_MUL_HI(k) "v_mul_hi_u32 %" #k ", %" #k ", %8\n\t"
#define I_RSQ(k) "v_rsq_f32_e32 %" #k ", %" #k "\n\t"
#define I_READLANE(k) "v_readlane_b32 s22, %" #k ", 3\n\t"
#define I_PK_MUL(k) ""
#define KERNEL(NAME, MAC)                                                                          \
    __global__ void __launch_bounds__(256) NAME(float* out, float seed) {                           \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
        float b = seed * 0.5f + 1.0f, c = seed * 0.25f + 2.0f;                                     \
        unsigned long long sm = 0x5555aaaa3333ccccull;                                              \
        for (int i = 0; i < kIters; i++) asm volatile(R8(MAC) R8(MAC) OPS : "vcc", "s20", "s21", "s22"); \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;         \
    }
KERNEL(k_cndmask_vcc, I_CNDMASK_VCC) KERNEL(k_cndmask_sgpr, I_CNDMASK_SGPR) KERNEL(k_cmp_f32_vcc, I_CMP_F32_VCC)
KERNEL(k_cmp_f32_sgpr, I_CMP_F32_SGPR) KERNEL(k_cmp_u32, I_CMP_U32_VCC) KERNEL(k_cmp_eq_u32, I_CMP_EQ_U32)
KERNEL(k_max_f32, I_MAX_F32) KERNEL(k_min_u32, I_MIN_U32) KERNEL(k_med3, I_MED3) KERNEL(k_max3, I_MAX3) KERNEL(k_bfi, I_BFI)
KERNEL(k_and_or, I_AND_OR) KERNEL(k_lshl_add, I_LSHL_ADD) KERNEL(k_add3, I_ADD3) KERNEL(k_mad_u24, I_MAD_U24)
KERNEL(k_mul_u24, I_MUL_U24) KERNEL(k_bfe, I_BFE) KERNEL(k_cvt_u32_f32, I_CVT_U32_F32) KERNEL(k_floor, I_FLOOR)
KERNEL(k_fract, I_FRACT) KERNEL(k_mbcnt, I_MBCNT) KERNEL(k_fma_same, I_FMA_SAME) KERNEL(k_fma_neg, I_FMA_NEG)
KERNEL(k_mul_abs, I_MUL_ABS) KERNEL(k_add_neg, I_ADD_NEG) KERNEL(k_subrev, I_SUBREV) KERNEL(k_add_sgpr, I_ADD_SGPR)
KERNEL(k_add_lit, I_ADD_LIT) KERNEL(k_mul_legacy, I_MUL_LEGACY) KERNEL(k_or3, I_OR3)
KERNEL(k_lshl_or, I_LSHL_OR) KERNEL(k_mul_hi, I_MUL_HI) KERNEL(k_rsq, I_RSQ) KERNEL(k_readlane, I_READLANE)
struct Entry { const char* name; void (*fn)(float*, float); };
int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* out; CHECK(hipMalloc(&out, sizeof(float) * 256 * cus * 8));
    std::vector<Entry> es = {
        {"v_cndmask_b32 (vcc)", k_cndmask_vcc}, {"v_cndmask_b32 (sgpr pair)", k_cndmask_sgpr}, {"v_cmp_lt_f32 -> vcc", k_cmp_f32_vcc},
        {"v_cmp_lt_f32 -> sgpr pair", k_cmp_f32_sgpr}, {"v_cmp_lt_u32 -> vcc", k_cmp_u32}, {"v_cmp_eq_u32 -> vcc", k_cmp_eq_u32},
        {"v_max_f32", k_max_f32}, {"v_min_u32", k_min_u32}, {"v_med3_f32", k_med3}, {"v_max3_f32", k_max3}, {"v_bfi_b32", k_bfi},
        {"v_and_or_b32", k_and_or}, {"v_lshl_add_u32", k_lshl_add}, {"v_add3_u32", k_add3}, {"v_mad_u32_u24", k_mad_u24},
        {"v_mul_u32_u24", k_mul_u24}, {"v_bfe_u32", k_bfe}, {"v_cvt_u32_f32", k_cvt_u32_f32}, {"v_floor_f32", k_floor},
        {"v_fract_f32", k_fract}, {"v_mbcnt_lo", k_mbcnt}, {"v_fma_f32 d=d*b+d", k_fma_same}, {"v_fma_f32 d=d*d-b", k_fma_neg},
        {"v_mul_f32_e64 |a|*b", k_mul_abs}, {"v_add_f32_e64 -a+b", k_add_neg}, {"v_subrev_f32", k_subrev}, {"v_add_f32 sgpr operand", k_add_sgpr},
        {"v_add_f32 literal", k_add_lit}, {"v_mul_legacy_f32", k_mul_legacy}, {"v_or3_b32", k_or3},
        {"v_lshl_or_b32", k_lshl_or}, {"v_mul_hi_u32", k_mul_hi}, {"v_rsq_f32", k_rsq}, {"v_readlane_b32", k_readlane},
    };
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-28s %10s %14s\n", "instruction", "ms", "cyc/inst/SIMD @2.4GHz (4 waves/SIMD)");
    for (auto& e : es) {
        const int blocks = cus * 4;
        hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0f); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        double winst = (double)blocks * 4 * kIters * 16;
        printf("%-28s %10.4f %14.3f\n", e.name, ms, 1.0 / (winst / (ms * 1e-3) / (cus * 4.0) / 2.4e9));
    }
    return 0;
}
