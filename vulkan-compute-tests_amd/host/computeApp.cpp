#include "computeApp.h"

#include "pngWriter.h"

#include <chrono>

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

void ComputeApp::createBuffer(uint64_t bufferSizeBytes) { buffer.assign(bufferSizeBytes / sizeof(float), 0.0f); }

void ComputeApp::run() {
    if (!quiet) { printf("in run()\n"); fflush(stdout); }
    createCommandBuffer();
    auto t0 = std::chrono::steady_clock::now();
    runCommandBuffer();
    auto t1 = std::chrono::steady_clock::now();
    lastRunMs = std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (!quiet) { printf("run() finished in %.3f ms\n", lastRunMs); fflush(stdout); }
}

std::string ComputeApp::writePng(const char* filename, const uint8_t* rgba8, uint32_t w, uint32_t h) const {
    return pngwriter::encodeFile(filename, rgba8, w, h, pngThreads);
}
