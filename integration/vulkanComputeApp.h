// Replacement for the reference's src/vulkanComputeApp.h + src/vulkanComputeApp.cpp (825 lines of Vulkan plumbing):
// the same virtual surface `main` drives (src/main.cpp:28-33, src/vulkanComputeApp.h:30-67), implemented on the C ABI of
// libmc_compute.so (include/mc_compute.h).  This is INTEGRATION.md route B: a maintainer of the reference drops this file
// in place of the original header, deletes vulkanComputeApp.cpp, and replaces the Vulkan-specific member functions of
// MandelbrotApp / PathtracerApp with the two `runCommandBuffer` bodies shown in INTEGRATION.md.  It is compiled and run by
// tests/test_integration_stub.py (integration/stub_check.cpp plays the part of the two apps).
#ifndef MC_INTEGRATION_VULKAN_COMPUTE_APP_H_
#define MC_INTEGRATION_VULKAN_COMPUTE_APP_H_

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "mc_compute.h"   // <repo>/include

struct VulkanComputeApp {   // name kept so that main.cpp and the two apps compile unchanged
    virtual ~VulkanComputeApp() {
        if (ctx) mc_context_destroy(ctx);   // cleanupVulkanResources (vulkanComputeApp.cpp:673-695)
        if (buffer.ptr) mc_host_free(buffer.ptr);   // vkFreeMemory(bufferMemory) / vkDestroyBuffer(buffer) (:684-685)
    }
    // createInstance / findPhysicalDevice / createDevice (vulkanComputeApp.cpp:443-449); a missing device throws the
    // runtime_error main() turns into EXIT_FAILURE (src/main.cpp:35-38), as findPhysicalDevice does (:78)
    void init() { check(mc_context_create(0, &ctx), "mc_context_create"); }
    virtual void preRun() {}
    virtual void run() {   // vulkanComputeApp.cpp:451-466
        createCommandBuffer();
        runCommandBuffer();
    }
    virtual void createCommandBuffer() {}
    virtual void runCommandBuffer() = 0;
    virtual void saveRenderedImage(const char* png_filename) = 0;

protected:
    // createBuffer (vulkanComputeApp.cpp:489-533): the host-visible storage buffer, one vec4 fp32 per pixel — page-locked host memory
    // here (mc_host_alloc), into which mc_*_render copies the rendered buffer from HBM once
    void createBuffer(uint32_t bytes) {
        if (buffer.ptr) mc_host_free(buffer.ptr);
        void* p = nullptr;
        check(mc_host_alloc(bytes, &p), "mc_host_alloc");
        buffer.ptr = static_cast<float*>(p);
        buffer.floats = bytes / sizeof(float);
    }
    static void check(int rc, const char* what) {
        if (rc != MC_OK)
            throw std::runtime_error(std::string(what) + ": " + mc_error_string(rc) + " " + mc_last_error_detail());
    }
    mc_context* ctx = nullptr;
    struct Mapped {              // what vkMapMemory(bufferMemory) exposes (mandelbrotApp.h:153, pathtracerApp.h:206)
        float* ptr = nullptr;
        size_t floats = 0;
        float* data() { return ptr; }
        const float* data() const { return ptr; }
        size_t size() const { return floats; }
    } buffer;
};

#endif
