// Where does a cold process spend its first 100 ms, and what of it overlaps?  (the apps' cold K2 `kernel` reads
// 30 ms against 14 steady; K4 allocates its pinned buffer for 87 ms before a 69 ms render that needs it only for the final copy.)
// One fresh process per mode; every line is "t_start..t_end  what" in ms since main():
//   seq   the app's order, HIP calls one by one: runtime start-up, context, pinned buffer (K2 and K4 size), a 16 x 8 x 16 path trace
//         (the first use of the path tracer's code object), K2 cold, K2 warm
//   alloc the K4-sized pinned buffer on a helper thread started first; the main thread creates the context and renders K2
//   warm  a helper thread with a context of its own renders the 16 x 8 x 16 image while the main thread creates its context and
//         allocates; joined before the main thread's K2
//   both  alloc + warm together
//   hostmem  what the storage buffer's host side costs, 629 MB / 157 MB / 8.64 MB, after the runtime is up: hipHostMalloc; mmap + first
//         touch on T threads (+ MADV_HUGEPAGE) + hipHostRegister; plain mmap left untouched — each followed by two device -> host copies
//   build: hipcc -O2 -std=c++17 -I include tools/cold_timeline.cpp -o tools/bin/cold_timeline -L vulkan-compute-tests_amd/lib -lmc_compute
//          -Wl,-rpath,$PWD/vulkan-compute-tests_amd/lib -lpthread
//   GPU box: for m in seq alloc warm both seq; do tools/bin/cold_timeline $m; done > gpurun_out/r06_cold_timeline.txt
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "mc_compute.h"

static std::chrono::steady_clock::time_point g_t0;
static std::mutex g_mu;
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g_t0).count(); }

template <class F>
static void span(const char* who, const char* what, F f) {
    const double a = now_ms();
    f();
    const double b = now_ms();
    std::lock_guard<std::mutex> lk(g_mu);
    printf("  %-6s %8.2f .. %8.2f  (%7.2f)  %s\n", who, a, b, b - a, what);
}

static const float *g_pl, *g_sp;
static uint32_t g_np, g_ns;

static void tiny_pathtrace(mc_context* c, const char* who) {
    mc_pathtrace_params p;
    mc_pathtrace_default_params(16, 8, 16, &p);
    p.math_mode = MC_PT_MATH_FAST;
    std::vector<float> out(16 * 8 * 4);
    span(who, "path trace 16 x 8 x 16 (first use of the code object)", [&] { mc_pathtrace_render(c, &p, g_pl, g_np, g_sp, g_ns, out.data()); });
}

static void k2(mc_context* c, float* buf, const char* label) {
    mc_pathtrace_params p;
    mc_pathtrace_default_params(900, 600, 500, &p);
    p.math_mode = MC_PT_MATH_FAST;
    double k = 0, cp = 0;
    span("main", label, [&] { mc_pathtrace_render(c, &p, g_pl, g_np, g_sp, g_ns, buf); });
    mc_context_last_timing(c, &k, &cp);
    printf("         (device: kernel %.2f ms, copy %.2f ms)\n", k, cp);
}

static double timed(const std::function<void()>& f) { const double a = now_ms(); f(); return now_ms() - a; }

static void touch_parallel(char* p, size_t bytes, int threads) {
    std::vector<std::thread> th;
    const size_t chunk = ((bytes / threads + 4095) / 4096) * 4096;
    for (int t = 0; t < threads; t++)
        th.emplace_back([=] { for (size_t i = (size_t)t * chunk; i < std::min(bytes, (size_t)(t + 1) * chunk); i += 4096) p[i] = 0; });
    for (auto& x : th) x.join();
}

static int hostmem() {
    (void)hipSetDevice(0);
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const size_t sizes[3] = {7680ull * 5120 * 16, 7680ull * 5120 * 4, 900ull * 600 * 16};
    void* d = nullptr;
    (void)hipMalloc(&d, sizes[0]);
    (void)hipMemset(d, 1, sizes[0]);
    (void)hipDeviceSynchronize();
    auto copy2 = [&](void* h, size_t n, double out[2]) {
        for (int k = 0; k < 2; k++) out[k] = timed([&] { (void)hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); });
    };
    printf("# %-44s %9s %9s %9s %9s %9s   (ms)\n", "host side of the storage buffer", "alloc", "touch", "register", "copy 1", "copy 2");
    for (size_t n : sizes) {
        printf("%.2f MB\n", n / 1e6);
        double c[2];
        { void* h = nullptr; const double a = timed([&] { (void)hipHostMalloc(&h, n, hipHostMallocPortable); }); copy2(h, n, c);
          printf("  %-44s %9.2f %9s %9s %9.2f %9.2f\n", "hipHostMalloc", a, "-", "-", c[0], c[1]); (void)hipHostFree(h); }
        for (int huge = 0; huge < 2; huge++)
            for (int threads : {1, 4, 8, 16}) {
                char* h = nullptr;
                const double a = timed([&] { h = (char*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
                                             if (huge) madvise(h, n, MADV_HUGEPAGE); });
                const double t = timed([&] { touch_parallel(h, n, threads); });
                hipError_t e = hipSuccess;
                const double r = timed([&] { e = hipHostRegister(h, n, hipHostRegisterPortable); });
                copy2(h, n, c);
                char label[96];
                snprintf(label, sizeof label, "mmap%s + touch x%d + hipHostRegister%s", huge ? " + MADV_HUGEPAGE" : "", threads, e == hipSuccess ? "" : " (FAILED)");
                printf("  %-44s %9.2f %9.2f %9.2f %9.2f %9.2f\n", label, a, t, r, c[0], c[1]);
                if (e == hipSuccess) (void)hipHostUnregister(h);
                munmap(h, n);
            }
        for (int threads : {0, 16}) {
            char* h = (char*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            const double t = threads ? timed([&] { touch_parallel(h, n, threads); }) : 0.0;
            copy2(h, n, c);
            printf("  %-44s %9.2f %9.2f %9s %9.2f %9.2f\n", threads ? "mmap + touch x16, pageable (no register)" : "mmap untouched, pageable", 0.0, t, "-", c[0], c[1]);
            munmap(h, n);
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    g_t0 = std::chrono::steady_clock::now();
    const std::string mode = argc > 1 ? argv[1] : "seq";
    printf("mode %s\n", mode.c_str());
    if (mode == "hostmem") return hostmem();
    mc_pathtrace_default_scene(&g_pl, &g_np, &g_sp, &g_ns);
    const size_t k2_bytes = 900ull * 600 * 16, k4_bytes = 7680ull * 5120 * 16;
    void *h2 = nullptr, *h4 = nullptr;
    mc_context* ctx = nullptr;
    if (mode == "seq") {
        int n = 0;
        span("main", "hipGetDeviceCount (runtime start-up)", [&] { (void)hipGetDeviceCount(&n); });
        span("main", "hipSetDevice(0)", [&] { (void)hipSetDevice(0); });
        hipDeviceProp_t pr;
        span("main", "hipGetDeviceProperties", [&] { (void)hipGetDeviceProperties(&pr, 0); });
        hipStream_t s;
        span("main", "hipStreamCreateWithFlags", [&] { (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); });
        span("main", "mc_context_create (second stream)", [&] { mc_context_create(0, &ctx); });
        void* d = nullptr;
        span("main", "hipMalloc 8.64 MB (first device allocation)", [&] { (void)hipMalloc(&d, k2_bytes); });
        span("main", "mc_host_alloc 8.64 MB", [&] { mc_host_alloc(k2_bytes, &h2); });
        span("main", "mc_host_alloc 629 MB", [&] { mc_host_alloc(k4_bytes, &h4); });
        tiny_pathtrace(ctx, "main");
        k2(ctx, (float*)h2, "K2 render, first full-size launch");
        k2(ctx, (float*)h2, "K2 render, second");
    } else {
        const bool do_alloc = mode == "alloc" || mode == "both", do_warm = mode == "warm" || mode == "both";
        std::thread ta, tw;
        if (do_alloc) ta = std::thread([&] { span("alloc", "mc_host_alloc 629 MB (helper thread)", [&] { mc_host_alloc(k4_bytes, &h4); }); });
        if (do_warm)
            tw = std::thread([&] {
                mc_context* w = nullptr;
                span("warm", "mc_context_create (helper's own)", [&] { mc_context_create(0, &w); });
                tiny_pathtrace(w, "warm");
                span("warm", "mc_context_destroy", [&] { mc_context_destroy(w); });
            });
        span("main", "mc_context_create", [&] { mc_context_create(0, &ctx); });
        span("main", "mc_host_alloc 8.64 MB", [&] { mc_host_alloc(k2_bytes, &h2); });
        if (tw.joinable()) span("main", "join warm-up", [&] { tw.join(); });
        k2(ctx, (float*)h2, "K2 render, first full-size launch");
        if (ta.joinable()) span("main", "join alloc", [&] { ta.join(); });
        k2(ctx, (float*)h2, "K2 render, second");
    }
    printf("  total %.2f ms\n", now_ms());
    mc_host_free(h2);
    mc_host_free(h4);
    mc_context_destroy(ctx);
    return 0;
}
