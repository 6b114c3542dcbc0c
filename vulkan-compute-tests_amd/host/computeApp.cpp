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
    image.resize((size_t)resx * resy * 4);
    pngwriter::convertStorage(buffer.data(), image.data(), resx, resy, scale, rotate180, pngThreads);
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
