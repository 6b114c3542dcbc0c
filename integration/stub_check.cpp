// Plays the part of the reference's two apps on top of integration/vulkanComputeApp.h: the `runCommandBuffer` bodies are
// the ones INTEGRATION.md gives a maintainer, driven through init() / preRun() / run() exactly as src/main.cpp:28-33 does.
// Prints an FNV-1a hash of each storage buffer; tests/test_integration_stub.py compares them with the oracle's buffers.
//   stub_check mandelbrot W H M   |   stub_check pathtracer W H spp
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "vulkanComputeApp.h"

namespace {

unsigned long long fnv1a(const void* data, size_t n) {
    const unsigned char* p = static_cast<const unsigned char*>(data);
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

class MandelbrotStub : public VulkanComputeApp {
public:
    MandelbrotStub(uint32_t x, uint32_t y, uint32_t m) : resx(x), resy(y), max_iter(m) {}
    void preRun() override { createBuffer(resx * resy * 4u * (uint32_t)sizeof(float)); }   // mandelbrotApp.h:187-189
    void runCommandBuffer() override {   // replaces mandelbrotApp.h:27-147
        mc_mandelbrot_params p;
        mc_mandelbrot_default_params(resx, resy, &p);   // M = 128, centre (-0.445, 0), scale 2.34, kColor {0.1, 0.7, 0.6, 0}
        p.max_iter = max_iter;
        check(mc_mandelbrot_render(ctx, &p, buffer.data(), nullptr), "mc_mandelbrot_render");
    }
    void saveRenderedImage(const char*) override {
        std::printf("mandelbrot %llu\n", fnv1a(buffer.data(), buffer.size() * sizeof(float)));
    }
private:
    uint32_t resx, resy, max_iter;
};

class PathtracerStub : public VulkanComputeApp {
public:
    PathtracerStub(uint32_t x, uint32_t y, uint32_t s) : resx(x), resy(y), spp(s) {}
    void preRun() override { createBuffer(resx * resy * 4u * (uint32_t)sizeof(float)); }
    void runCommandBuffer() override {   // replaces pathtracerApp.h:129-198,250-378
        mc_pathtrace_params p;
        mc_pathtrace_default_params(resx, resy, spp, &p);   // samps = [0, spp), maxDepth 12, strict math
        const float *planes, *spheres;                      // a real PathtracerApp passes its file-static tables (:14-39)
        uint32_t np, ns;
        check(mc_pathtrace_default_scene(&planes, &np, &spheres, &ns), "mc_pathtrace_default_scene");
        check(mc_pathtrace_render(ctx, &p, planes, np, spheres, ns, buffer.data()), "mc_pathtrace_render");
    }
    void saveRenderedImage(const char*) override {
        std::printf("pathtracer %llu\n", fnv1a(buffer.data(), buffer.size() * sizeof(float)));
    }
private:
    uint32_t resx, resy, spp;
};

}  // namespace

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: stub_check mandelbrot|pathtracer W H M|spp\n"); return 2; }
    const uint32_t w = (uint32_t)std::atoi(argv[2]), h = (uint32_t)std::atoi(argv[3]), k = (uint32_t)std::atoi(argv[4]);
    try {   // src/main.cpp:26-39
        VulkanComputeApp* app = !std::strcmp(argv[1], "mandelbrot") ? static_cast<VulkanComputeApp*>(new MandelbrotStub(w, h, k))
                                                                     : static_cast<VulkanComputeApp*>(new PathtracerStub(w, h, k));
        app->init();
        app->preRun();
        app->run();
        app->saveRenderedImage("unused.png");
        delete app;
    } catch (const std::runtime_error& e) {
        std::printf("%s\n", e.what());
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
