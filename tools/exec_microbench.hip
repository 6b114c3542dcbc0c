// Does gfx950 skip the second 32-lane pass of a wave64 VALU instruction when one half of EXEC is zero?
// (RDNA wave64 does; GCN did not.)  Decides whether lane regrouping in the path tracer should target whole waves or
// 32-lane halves.  Same 8-chain loop as valu_microbench.hip, executed under different EXEC masks.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/exec_microbench tools/exec_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kIters = 4096;
#define BODY1(INS)                                                                                        \
    asm volatile(INS " %0, %0, %8\n\t" INS " %1, %1, %8\n\t" INS " %2, %2, %8\n\t" INS " %3, %3, %8\n\t"  \
                 INS " %4, %4, %8\n\t" INS " %5, %5, %8\n\t" INS " %6, %6, %8\n\t" INS " %7, %7, %8\n\t"  \
                 INS " %0, %0, %8\n\t" INS " %1, %1, %8\n\t" INS " %2, %2, %8\n\t" INS " %3, %3, %8\n\t"  \
                 INS " %4, %4, %8\n\t" INS " %5, %5, %8\n\t" INS " %6, %6, %8\n\t" INS " %7, %7, %8\n\t"  \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b))
#define BODY_UN(INS)                                                                                      \
    asm volatile(INS " %0, %0\n\t" INS " %1, %1\n\t" INS " %2, %2\n\t" INS " %3, %3\n\t"                  \
                 INS " %4, %4\n\t" INS " %5, %5\n\t" INS " %6, %6\n\t" INS " %7, %7\n\t"                  \
                 INS " %0, %0\n\t" INS " %1, %1\n\t" INS " %2, %2\n\t" INS " %3, %3\n\t"                  \
                 INS " %4, %4\n\t" INS " %5, %5\n\t" INS " %6, %6\n\t" INS " %7, %7\n\t"                  \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))
// mode: 0 all lanes, 1 lanes 0-31, 2 lanes 32-63, 3 even lanes, 4 lane 0 only, 5 lanes 0-15
template <int OP>
__global__ void __launch_bounds__(256) k(float* out, float seed, int mode) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    float b = seed * 0.5f + 1.0f;
    const unsigned lane = threadIdx.x & 63u;
    bool on = mode == 0 || (mode == 1 && lane < 32) || (mode == 2 && lane >= 32) || (mode == 3 && !(lane & 1)) ||
              (mode == 4 && lane == 0) || (mode == 5 && lane < 16);
    if (on) {
        for (int i = 0; i < kIters; i++) {
            if (OP == 0) BODY1("v_add_f32"); else if (OP == 1) BODY_UN("v_rcp_f32"); else BODY1("v_max_f32");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* out; CHECK(hipMalloc(&out, sizeof(float) * 256 * cus * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[] = {"all 64", "lanes 0-31", "lanes 32-63", "even lanes", "lane 0", "lanes 0-15"};
    const char* ops[] = {"v_add_f32", "v_rcp_f32", "v_max_f32"};
    for (int op = 0; op < 3; op++)
        for (int mode = 0; mode < 6; mode++) {
            const int blocks = cus * 4;   // 4 waves per SIMD
            auto launch = [&]() {
                if (op == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, mode);
                else if (op == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, mode);
                else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, mode);
            };
            launch(); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 5; r++) launch();
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
            double winst = (double)blocks * 4 * kIters * 16;
            printf("%-10s %-12s %8.4f ms  %6.3f cyc/inst/SIMD @2.4GHz\n", ops[op], names[mode], ms, 1.0 / (winst / (ms * 1e-3) / (cus * 4.0) / 2.4e9));
        }
    return 0;
}
