// PathtracerApp — mirrors src/pathtracerApp.h:42-397 of the reference on top of ComputeApp.
#ifndef PATHTRACERAPP_H_
#define PATHTRACERAPP_H_

#include <chrono>
#include <cstring>

#include "computeApp.h"
#include "pngWriter.h"

struct PathtracerApp : public ComputeApp {
    struct pushConst_t {          // pathtracerApp.h:44-47
        uint32_t imgdim[2];       // { WIDTH, HEIGHT }
        uint32_t samps[2];        // { 0, spp }
    } pushConst;

    // Same signature as the reference (pathtracerApp.h:49); `quiet` (main.cpp --quiet) only silences the constructor's
    // "in PathtracerApp ctor" line, which the reference always prints.
    PathtracerApp(const uint32_t resx, const uint32_t resy, const int32_t spp, const uint32_t workgroupSize = 16,
                  const bool quiet = false) {
        setQuiet(quiet);
        this->resx = resx;
        this->resy = resy;
        this->spp = spp;
        this->workgroupSize = workgroupSize;
        bufferSize = (uint64_t)sizeof(Pixel) * resx * resy;
        if (!quiet) printf("in PathtracerApp ctor\n");
        pushConst.imgdim[0] = resx;
        pushConst.imgdim[1] = resy;
        pushConst.samps[0] = 0;
        pushConst.samps[1] = (uint32_t)spp;
        mc_pathtrace_default_params(resx, resy, (uint32_t)spp, &params);
        const float *pl, *sp; uint32_t np, ns;
        mc_pathtrace_default_scene(&pl, &np, &sp, &ns);   // the tables of pathtracerApp.h:14-39
        planes.assign(pl, pl + 12 * np);
        spheres.assign(sp, sp + 12 * ns);
    }
    virtual ~PathtracerApp() { joinWarmupQuietly(); }   // (the helper reads planes / spheres)

    // -- additions --
    void setMathMode(uint32_t mode) { params.math_mode = mode; }   // MC_PT_MATH_STRICT / MC_PT_MATH_FAST / MC_PT_MATH_FAST_CAREFUL
    // The reference's precision experiment (pathtracerApp.h:11, emulateDouble.h.glsl:13-26), a run-time switch here:
    // which sphere-test branch of pathTracer.comp:132-256 is active, and the sphere-walled scene of :28-35.
    void setSpherePrecision(uint32_t prec) { params.flags = (params.flags & ~MC_PT_PRECISION(0xf)) | MC_PT_PRECISION(prec); }
    void useLargeSphereWalls() {
        static const float dummyPlane[12] = {1, 0, 0, 1000, 0, 0, 0, 0, 1, 1, 1, 1};   // "must have at least one plane" (:22-23)
        static const float walls[6 * 12] = {
            (float)(1e5 - 2.6), 0, 0, (float)1e5, 0, 0, 0, 0, (float).85, (float).25, (float).25, 1,    // Left
            (float)(1e5 + 2.6), 0, 0, (float)1e5, 0, 0, 0, 0, (float).25, (float).35, (float).85, 1,    // Right
            0, (float)(1e5 + 2), 0, (float)1e5, 0, 0, 0, 0, (float).75, (float).75, (float).75, 1,      // Top
            0, (float)(-1e5 - 2), 0, (float)1e5, 0, 0, 0, 0, (float).75, (float).75, (float).75, 1,     // Bottom
            0, 0, (float)(-1e5 - 2.8), (float)1e5, 0, 0, 0, 0, (float).85, (float).85, (float).25, 1,   // Back
            0, 0, (float)(1e5 + 7.9), (float)1e5, 0, 0, 0, 0, (float)0.1, (float)0.7, (float)0.7, 1,    // Front
        };
        std::vector<float> sp(walls, walls + 72);
        sp.insert(sp.end(), spheres.begin(), spheres.end());   // then mirror, glass, light (:36-38)
        planes.assign(dummyPlane, dummyPlane + 12);
        spheres.swap(sp);
    }
    void setScene(const float* pl, uint32_t np, const float* sp, uint32_t ns) {
        planes.assign(pl, pl + 12 * np);
        spheres.assign(sp, sp + 12 * ns);
    }

    virtual void preRun() override {
        if (!quiet) { printf(" * before createBuffer()\n"); fflush(stdout); }
        createBuffer(bufferSize);   // output buffer; the two scene SSBOs of the reference (pathtracerApp.h:129-198)
                                    // become kernel arguments, there is nothing to upload here
    }

    // The reference records spp dispatches, one push-constant update each, with no barrier in between
    // (pathtracerApp.h:361-376).  Here the whole samps.x range [0,spp) is ONE launch with the sample
    // loop fused in registers, accumulating in the barrier-serialised order s = 0..spp-1.
    virtual void createCommandBuffer() override {
        if (!quiet) { printf("\n   ### recording fused spp loop: samples [0,%d) ###\n\n", spp); fflush(stdout); }
        params.width = pushConst.imgdim[0]; params.height = pushConst.imgdim[1];
        params.spp = pushConst.samps[1];
        params.sample_begin = 0; params.sample_end = pushConst.samps[1];
        params.row_begin = 0; params.row_end = resy;
    }

    virtual int warmup() override {   // helper thread of init(): what run() is going to ask for (the setters were called before init())
        mc_pathtrace_params q = params;
        q.width = resx; q.height = resy; q.spp = (uint32_t)spp; q.sample_begin = 0; q.sample_end = q.spp; q.row_begin = 0; q.row_end = resy;
        return mc_context_warmup_pathtrace(ctx, &q, planes.data(), (uint32_t)planes.size() / 12, spheres.data(),
                                           (uint32_t)spheres.size() / 12, gpuPostprocess ? 1 : 0);
    }

    virtual void runCommandBuffer() override {
        const uint32_t np = (uint32_t)planes.size() / 12, ns = (uint32_t)spheres.size() / 12;
        if (gpuPostprocess) {   // render + float->u8 + 180-degree rotation on the device (pathtracerApp.h:202-243), 4 B/pixel copied
            if (multi) check(mc_multi_pathtrace_render_rgba8(multi, &params, planes.data(), np, spheres.data(), ns, rgba8.bytes()),
                             "mc_multi_pathtrace_render_rgba8");
            else check(mc_pathtrace_render_rgba8(ctx, &params, planes.data(), np, spheres.data(), ns, rgba8.bytes()),
                       "mc_pathtrace_render_rgba8");
            return;
        }
        if (multi) check(mc_multi_pathtrace_render(multi, &params, planes.data(), np, spheres.data(), ns, buffer.data()),
                         "mc_multi_pathtrace_render");
        else check(mc_pathtrace_render(ctx, &params, planes.data(), np, spheres.data(), ns, buffer.data()), "mc_pathtrace_render");
    }

    // pathtracerApp.h:202-223
    void getRenderedImage(std::vector<uint8_t>& image, const uint32_t resx, const uint32_t resy, float floatScaleFactor) {
        convertStorage(image, resx, resy, floatScaleFactor, false);   // the loop of :212-219, row stripes in parallel
    }

    virtual void saveRenderedImage(const char* png_filename = "pathtracer.png") override {
        std::vector<uint8_t> image;
        constexpr float scaleFactor = 1.0f;   // pathtracerApp.h:227
        printf("writing %s\n", png_filename);
        auto t0 = std::chrono::steady_clock::now();
        if (fusedSave()) {   // getRenderedImage + the rotation of :235-243 + the encoder in one pass over the storage buffer
            std::string err = writePngFromStorage(png_filename, resx, resy, scaleFactor, true);
            if (!err.empty()) printf("encoder error: %s", err.c_str());
            times.convertMs = 0.0;
            times.pngMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            return;
        }
        if (!gpuPostprocess) {   // (else: converted and rotated on the device)
            // getRenderedImage, then: due to the pinhole camera the image is upside-down and mirrored — undo that (pathtracerApp.h:235-243:
            // every pixel of the left half swapped with its point reflection).  One pass here: each converted pixel is written to the
            // place the swap loop would leave it in (the same bytes, incl. an odd width's untouched middle column).
            convertStorage(image, resx, resy, scaleFactor, true);
        }
        auto t1 = std::chrono::steady_clock::now();
        std::string err = writePng(png_filename, gpuPostprocess ? rgba8.bytes() : image.data(), resx, resy);
        if (!err.empty()) printf("encoder error: %s", err.c_str());
        times.convertMs = std::chrono::duration<double, std::milli>(t1 - t0).count();
        times.pngMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    }

    const HostStorage& storageBuffer() const { return buffer; }

private:
    struct Pixel { float r, g, b, a; };
    uint64_t bufferSize;
    uint32_t resx, resy;
    int32_t spp;
    uint32_t workgroupSize;
    mc_pathtrace_params params;
    std::vector<float> planes, spheres;
};

#endif  // PATHTRACERAPP_H_
