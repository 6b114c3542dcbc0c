#include "computeApp.h"

#include "pngWriter.h"

#include <algorithm>
#include <chrono>
#include <thread>

ComputeApp::~ComputeApp() {
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
    }
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
    auto t0 = std::chrono::steady_clock::now();
    if (gpuPostprocess) rgba8.allocate(bufferSizeBytes / 4);   // 16 B/pixel of fp32 -> 4 B/pixel of RGBA8
    else buffer.allocate(bufferSizeBytes);
    times.allocMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

void ComputeApp::convertStorage(std::vector<uint8_t>& image, uint32_t resx, uint32_t resy, float scale, bool rotate180) const {
    struct Pixel { float r, g, b, a; };
    const Pixel* p = reinterpret_cast<const Pixel*>(buffer.data());
    image.resize((size_t)resx * resy * 4);
    uint8_t* out = image.data();
    // Destination pixel of source pixel (x, y) under the reference's swap loop: every pixel with x < resx / 2 changes places with its
    // point reflection; for an odd width the middle column (x = resx / 2) is left where it is (pathtracerApp.h:238: `x < resx / 2`).
    auto rows = [&](uint32_t y0, uint32_t y1) {
        for (uint32_t y = y0; y < y1; y++)
            for (uint32_t x = 0; x < resx; x++) {
                const Pixel& s = p[(size_t)y * resx + x];
                size_t to = (size_t)y * resx + x;
                if (rotate180 && !((resx & 1u) && x == resx / 2)) to = (size_t)(resy - 1 - y) * resx + (resx - 1 - x);
                uint8_t* o = out + 4 * to;
                o[0] = x86FloatToU8(scale * s.r); o[1] = x86FloatToU8(scale * s.g); o[2] = x86FloatToU8(scale * s.b); o[3] = 255u;
            }
    };
    unsigned n = pngThreads > 0 ? (unsigned)pngThreads : std::thread::hardware_concurrency();
    n = std::max(1u, std::min(n, std::max(1u, resy / 16u)));
    if (n == 1) { rows(0, resy); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < n; t++) th.emplace_back(rows, (uint32_t)((uint64_t)resy * t / n), (uint32_t)((uint64_t)resy * (t + 1) / n));
    for (auto& t : th) t.join();
}

void ComputeApp::run() {
    if (!quiet) { printf("in run()\n"); fflush(stdout); }
    createCommandBuffer();
    auto t0 = std::chrono::steady_clock::now();
    runCommandBuffer();
    auto t1 = std::chrono::steady_clock::now();
    lastRunMs = std::chrono::duration<double, std::milli>(t1 - t0).count();
    times.runMs = lastRunMs;
    times.kernelMs = times.copyMs = 0.0;
    if (ctx) (void)mc_context_last_timing(ctx, &times.kernelMs, &times.copyMs);
    if (!quiet) { printf("run() finished in %.3f ms\n", lastRunMs); fflush(stdout); }
}

std::string ComputeApp::writePng(const char* filename, const uint8_t* rgba8, uint32_t w, uint32_t h) const {
    return pngwriter::encodeFile(filename, rgba8, w, h, pngThreads);
}
