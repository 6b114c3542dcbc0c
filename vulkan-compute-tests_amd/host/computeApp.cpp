#include "computeApp.h"

#include "pngWriter.h"

#include <algorithm>
#include <chrono>
#include <thread>

#ifdef MC_HAVE_REFERENCE_PNG
#include "lodepng.h"   // the reference checkout's own header (make REFERENCE=<checkout>: -I<checkout>/src/external/lodepng)
#endif

namespace {
double msSince(std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
}
}  // namespace

bool ComputeApp::referencePngAvailable() {
#ifdef MC_HAVE_REFERENCE_PNG
    return true;
#else
    return false;
#endif
}

void ComputeApp::setReferencePng(bool r) {
    if (r && !referencePngAvailable())
        throw std::runtime_error("--reference-png: this binary was built without the reference's PNG codec; rebuild with "
                                 "`make REFERENCE=<checkout of pjhusky/vulkan-compute-tests>` (its lodepng is compiled where it lies)");
    referencePng = r;
}

ComputeApp::~ComputeApp() {
    progressive.abandon();   // a streamed save whose render failed: its workers stop before the buffers they read are released below
    if (warmThread.joinable()) warmThread.join();
    // cleanupVulkanResources (vulkanComputeApp.cpp:673-695)
    if (multi) mc_multi_destroy(multi);
    if (ctx) mc_context_destroy(ctx);
}

void ComputeApp::check(int status, const char* what) {
    if (status != MC_OK) {
        std::string msg = std::string(what) + ": " + mc_error_string(status);
        const char* d = mc_last_error_detail();
        if (d && d[0]) msg += std::string(" (") + d + ")";
        throw std::runtime_error(msg);
    }
}

void ComputeApp::init() {
    int n = 0;
    int rc = mc_device_count(&n);
    if (rc != MC_OK || n == 0) throw std::runtime_error("could not find a device with HIP support");   // cf. vulkanComputeApp.cpp:78
    if (numGpus > 1) {
        check(mc_multi_create(numGpus, &multi), "mc_multi_create");
    } else {
        check(mc_context_create(deviceIndex, &ctx), "mc_context_create");
        if (!quiet) {
            char name[256]; int cus = 0, khz = 0;
            mc_context_device_info(ctx, name, sizeof(name), &cus, &khz);
            printf("using device %d: %s (%d CUs)\n", deviceIndex, name, cus);
        }
        if (overlapStart)   // the code object, tables and device scratch of the request run() will make, while the caller goes on
            warmThread = std::thread([this] {
                auto t0 = std::chrono::steady_clock::now();
                warmStatus = warmup();
                if (warmStatus != MC_OK) { const char* d = mc_last_error_detail(); warmError = d ? d : ""; }
                times.warmupMs = msSince(t0);
            });
    }
}

void ComputeApp::waitWarmup() {
    if (!warmThread.joinable()) return;
    auto t0 = std::chrono::steady_clock::now();
    warmThread.join();
    times.warmupWaitMs = msSince(t0);
    if (warmStatus != MC_OK)   // a warm-up that fails would fail the render the same way: report it as the render's error
        throw std::runtime_error(std::string("warm-up: ") + mc_error_string(warmStatus) + (warmError.empty() ? "" : " (" + warmError + ")"));
}

void HostStorage::allocate(uint64_t bytes) {
    release();
    void* p = nullptr;
    const int rc = mc_host_alloc((size_t)bytes, &p);
    if (rc != MC_OK) {
        std::string msg = std::string("mc_host_alloc: ") + mc_error_string(rc);
        const char* d = mc_last_error_detail();
        if (d && d[0]) msg += std::string(" (") + d + ")";
        throw std::runtime_error(msg);
    }
    ptr_ = p;
    bytes_ = (size_t)bytes;
}

void ComputeApp::createBuffer(uint64_t bufferSizeBytes) {
    // In preRun(), where the reference allocates (vulkanComputeApp.cpp:489-533) — and BEFORE the render is launched: registering 629 MB
    // with the runtime while a kernel runs (tried in round 6: allocate between the launch and the copy of a two-phase render call, since withdrawn) stalls the device —
    // K4's kernel read 82 ms instead of 60 and the copy that followed 38 GB/s instead of 57 (profiles/r06_end_to_end_rejected_b.txt).
    // mc_host_alloc makes the buffer in 4 ms; there is nothing left to hide.
    auto t0 = std::chrono::steady_clock::now();
    if (gpuPostprocess) rgba8.allocate(bufferSizeBytes / 4);   // 16 B/pixel of fp32 -> 4 B/pixel of RGBA8
    else buffer.allocate(bufferSizeBytes);
    times.allocMs = msSince(t0);
}

void ComputeApp::convertStorage(std::vector<uint8_t>& image, uint32_t resx, uint32_t resy, float scale, bool rotate180) const {
    image.resize((size_t)resx * resy * 4);
    pngwriter::convertStorage(buffer.data(), image.data(), resx, resy, scale, rotate180, pngThreads);
}

void ComputeApp::run() {
    if (!quiet) { printf("in run()\n"); fflush(stdout); }
    createCommandBuffer();
    auto t0 = std::chrono::steady_clock::now();
    waitWarmup();
    runCommandBuffer();
    auto t1 = std::chrono::steady_clock::now();
    lastRunMs = std::chrono::duration<double, std::milli>(t1 - t0).count();
    times.runMs = lastRunMs;
    times.kernelMs = times.copyMs = 0.0;
    if (ctx) (void)mc_context_last_timing(ctx, &times.kernelMs, &times.copyMs);
    if (!quiet) { printf("run() finished in %.3f ms\n", lastRunMs); fflush(stdout); }
}

std::string ComputeApp::writePngFromStorage(const char* filename, uint32_t w, uint32_t h, float scale, bool rotate180) const {
    return pngwriter::encodeStorageFile(filename, buffer.data(), w, h, scale, rotate180, pngThreads);
}

std::string ComputeApp::writePng(const char* filename, const uint8_t* rgba8, uint32_t w, uint32_t h) const {
#ifdef MC_HAVE_REFERENCE_PNG
    if (referencePng) {   // mandelbrotApp.h:181-183 / pathtracerApp.h:245-247: the reference's call and its error text
        const unsigned error = lodepng::encode(filename, rgba8, w, h);
        return error ? std::to_string(error) + ": " + lodepng_error_text(error) : std::string();
    }
#endif
    return pngwriter::encodeFile(filename, rgba8, w, h, pngThreads);
}
