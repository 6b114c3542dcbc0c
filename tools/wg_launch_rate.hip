// How fast does gfx950 start workgroups?  K1 (fp32 Mandelbrot, 3200 x 2400) is 120 000 one-wave workgroups, most of which leave after
// ~90 iterations (~3000 cycles): if the dispatcher needs a comparable time per wave, the launch shape — not the arithmetic — is the bound.
// Kernels of N waves in total, as blocks of 64 / 256 / 1024 threads, each wave spinning for a given number of dependent adds.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/wg_launch_rate tools/wg_launch_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void spin(float* out, int iters, float seed) {
    float a = seed + threadIdx.x;
    for (int i = 0; i < iters; i++) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(a) : "v"(seed));
    if (a == 123.456f) out[blockIdx.x] = a;   // never true: no store traffic
}
int main() {
    CHECK(hipSetDevice(0));
    float* out; CHECK(hipMalloc(&out, 1 << 22));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int waves = 120000;
    printf("%d waves in total; time per launch (ms) and ns per wave\n%8s %8s %10s %10s\n", waves, "threads", "adds", "ms", "ns/wave");
    for (int iters : {0, 256, 1024, 4096}) {
        for (int threads : {64, 256, 1024}) {
            const int blocks = waves / (threads / 64);
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 20; r++) hipLaunchKernelGGL(spin, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
            printf("%8d %8d %10.4f %10.2f\n", threads, iters, ms, ms * 1e6 / waves);
        }
    }
    return 0;
}
