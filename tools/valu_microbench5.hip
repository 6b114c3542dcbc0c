// Fifth VALU survey for gfx950: do VGPR operand banks matter?  Explicit registers: sources from the same bank (index mod 4)
// against sources from different banks, for two- and three-source instructions.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/valu_microbench5 tools/valu_microbench5.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kIters = 2048;
#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41"
#define X4(a) a a a a
#define KERN(NAME, BODY)                                                                   \
    __global__ void __launch_bounds__(512) NAME(float* out, float v) {                      \
        asm volatile("v_mov_b32 v10, %0\n\tv_mov_b32 v11, %0\n\tv_mov_b32 v12, %0\n\tv_mov_b32 v13, %0\n\tv_mov_b32 v14, %0\n\tv_mov_b32 v15, %0\n\t" \
                     "v_mov_b32 v16, %0\n\tv_mov_b32 v17, %0\n\tv_mov_b32 v18, %0\n\tv_mov_b32 v19, %0\n\tv_mov_b32 v20, %0\n\tv_mov_b32 v21, %0\n\t" \
                     "v_mov_b32 v22, %0\n\tv_mov_b32 v23, %0\n\tv_mov_b32 v24, %0\n\tv_mov_b32 v25, %0\n\tv_mov_b32 v26, %0\n\tv_mov_b32 v27, %0\n\t" \
                     "v_mov_b32 v28, %0\n\tv_mov_b32 v29, %0\n\tv_mov_b32 v30, %0\n\tv_mov_b32 v31, %0\n\tv_mov_b32 v32, %0\n\tv_mov_b32 v33, %0\n\t" :: "v"(v) : CLOB); \
        for (int i = 0; i < kIters; i++) asm volatile(BODY ::: CLOB);                        \
        float r; asm volatile("v_add_f32 %0, v10, v11\n\tv_add_f32 %0, %0, v12\n\tv_add_f32 %0, %0, v13" : "=v"(r) :: CLOB); \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                      \
    }
// 16 instructions per trip, 4 independent destinations; sources chosen by bank
KERN(k_add_diff, X4("v_add_f32 v10, v20, v21\n\tv_add_f32 v11, v22, v23\n\tv_add_f32 v12, v24, v25\n\tv_add_f32 v13, v26, v27\n\t"))
KERN(k_add_same, X4("v_add_f32 v10, v20, v24\n\tv_add_f32 v11, v21, v25\n\tv_add_f32 v12, v22, v26\n\tv_add_f32 v13, v23, v27\n\t"))
KERN(k_add_samedst, X4("v_add_f32 v10, v14, v18\n\tv_add_f32 v11, v15, v19\n\tv_add_f32 v12, v16, v20\n\tv_add_f32 v13, v17, v21\n\t"))
KERN(k_fma_diff, X4("v_fma_f32 v10, v20, v21, v22\n\tv_fma_f32 v11, v23, v24, v25\n\tv_fma_f32 v12, v26, v27, v28\n\tv_fma_f32 v13, v29, v30, v31\n\t"))
KERN(k_fma_same, X4("v_fma_f32 v10, v20, v24, v28\n\tv_fma_f32 v11, v21, v25, v29\n\tv_fma_f32 v12, v22, v26, v30\n\tv_fma_f32 v13, v23, v27, v31\n\t"))
KERN(k_fma_two, X4("v_fma_f32 v10, v20, v24, v21\n\tv_fma_f32 v11, v21, v25, v22\n\tv_fma_f32 v12, v22, v26, v23\n\tv_fma_f32 v13, v23, v27, v20\n\t"))
KERN(k_fmac_diff, X4("v_fmac_f32 v10, v20, v21\n\tv_fmac_f32 v11, v22, v23\n\tv_fmac_f32 v12, v24, v25\n\tv_fmac_f32 v13, v26, v27\n\t"))
KERN(k_fmac_same, X4("v_fmac_f32 v10, v14, v18\n\tv_fmac_f32 v11, v15, v19\n\tv_fmac_f32 v12, v16, v20\n\tv_fmac_f32 v13, v17, v21\n\t"))
KERN(k_min3_diff, X4("v_min3_u32 v10, v20, v21, v22\n\tv_min3_u32 v11, v23, v24, v25\n\tv_min3_u32 v12, v26, v27, v28\n\tv_min3_u32 v13, v29, v30, v31\n\t"))
KERN(k_min3_same, X4("v_min3_u32 v10, v20, v24, v28\n\tv_min3_u32 v11, v21, v25, v29\n\tv_min3_u32 v12, v22, v26, v30\n\tv_min3_u32 v13, v23, v27, v31\n\t"))
struct Entry { const char* name; void (*fn)(float*, float); };
int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* out; CHECK(hipMalloc(&out, sizeof(float) * 512 * cus * 8));
    std::vector<Entry> es = {{"v_add_f32 sources in different banks", k_add_diff}, {"v_add_f32 sources in one bank", k_add_same},
        {"v_add_f32 sources + dest in one bank", k_add_samedst}, {"v_fma_f32 3 sources, 3 banks", k_fma_diff}, {"v_fma_f32 3 sources, one bank", k_fma_same},
        {"v_fma_f32 3 sources, two in one bank", k_fma_two}, {"v_fmac_f32 different banks", k_fmac_diff}, {"v_fmac_f32 all one bank", k_fmac_same},
        {"v_min3_u32 3 banks", k_min3_diff}, {"v_min3_u32 one bank", k_min3_same}};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps : {2, 6}) {
        printf("---- %d waves per SIMD\n%-42s %10s %s\n", wps, "stream", "ms", "cycles per wave-instruction per SIMD @2.4 GHz");
        for (auto& e : es) {
            const int blocks = cus * (wps / 2);
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(512), 0, 0, out, 1.0f); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 5; r++) hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(512), 0, 0, out, 1.0f);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
            const double winst = (double)blocks * 8 * kIters * 16;
            printf("%-42s %10.4f %8.3f\n", e.name, ms, (ms * 1e-3) * 2.4e9 * (cus * 4.0) / winst);
        }
    }
    return 0;
}
