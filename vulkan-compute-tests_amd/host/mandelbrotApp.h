// MandelbrotApp — mirrors src/mandelbrotApp.h:8-194 of the reference on top of ComputeApp.
#ifndef MANDELBROTAPP_H_
#define MANDELBROTAPP_H_

#include <algorithm>
#include <chrono>

#include "computeApp.h"
#include "pngWriter.h"

struct MandelbrotApp : public ComputeApp {
    // Same signature as the reference (mandelbrotApp.h:10).  workgroupSize is accepted for source
    // compatibility; the HIP kernels choose their own tiling (8x8 pixels per wave64).
    MandelbrotApp(const uint32_t resx, const uint32_t resy, const uint32_t workgroupSize = 32) {
        this->resx = resx;
        this->resy = resy;
        this->workgroupSize = workgroupSize;
        bufferSize = (uint64_t)sizeof(Pixel) * resx * resy;   // mandelbrotApp.h:16 (uint32_t there)
        mc_mandelbrot_default_params(resx, resy, &params);    // M=128, centre (-0.445,0), scale 2.34, kColor {0.1,0.7,0.6,0}
    }
    virtual ~MandelbrotApp() { joinWarmupQuietly(); }

    // -- additions: the reference hard-codes these in the shader (mandelbrot.comp:5-6,38,40) --
    void setMaxIter(uint32_t m) { params.max_iter = m; }
    void setView(double cx, double cy, double sx, double sy) {
        split(cx, params.centre_x_hi, params.centre_x_lo);
        split(cy, params.centre_y_hi, params.centre_y_lo);
        split(sx, params.scale_x_hi, params.scale_x_lo);
        split(sy, params.scale_y_hi, params.scale_y_lo);
    }
    void setPrecision(uint32_t precision) { params.precision = precision; }   // MC_PRECISION_F32 / MC_PRECISION_DS

    virtual void preRun() override {
        if (!quiet) { printf(" * before createBuffer()\n"); fflush(stdout); }
        createBuffer(bufferSize);   // output buffer
    }

    virtual void createCommandBuffer() override {
        // push constant kColor (mandelbrotApp.h:139-141) and ONE dispatch over the whole image (:146)
        params.k_color[0] = 0.1f; params.k_color[1] = 0.7f; params.k_color[2] = 0.6f; params.k_color[3] = 0.0f;
        params.row_begin = 0; params.row_end = resy;
    }

    virtual int warmup() override {   // helper thread of init(): tables + code object of the request run() will make
        mc_mandelbrot_params q = params;
        q.k_color[0] = 0.1f; q.k_color[1] = 0.7f; q.k_color[2] = 0.6f; q.k_color[3] = 0.0f;   // as createCommandBuffer sets it
        q.row_begin = 0; q.row_end = resy;
        return mc_context_warmup_mandelbrot(ctx, &q, (gpuPostprocess ? 1 : 0) | (streaming() ? 2 : 0));   // (bit 1: the banded render's second stream)
    }

    virtual void runCommandBuffer() override {
        if (streaming()) { runStreamed(); return; }
        if (gpuPostprocess) {   // render + float->u8 on the device: 4 B/pixel cross PCIe instead of 16
            if (multi) check(mc_multi_mandelbrot_render_rgba8(multi, &params, rgba8.bytes()), "mc_multi_mandelbrot_render_rgba8");
            else check(mc_mandelbrot_render_rgba8(ctx, &params, rgba8.bytes()), "mc_mandelbrot_render_rgba8");
            return;
        }
        if (multi) check(mc_multi_mandelbrot_render(multi, &params, buffer.data(), nullptr), "mc_multi_mandelbrot_render");
        else check(mc_mandelbrot_render(ctx, &params, buffer.data(), nullptr), "mc_mandelbrot_render");
    }

    // mandelbrotApp.h:149-170: u8 = static_cast<uint8_t>(scale * c), alpha 255.  The cast is UB out of
    // range; the reference binary on x86-64 truncates to int32 and keeps the low byte — stated explicitly.
    void getRenderedImage(std::vector<uint8_t>& image, const uint32_t resx, const uint32_t resy, float floatScaleFactor) {
        convertStorage(image, resx, resy, floatScaleFactor, false);   // the loop of :159-166, row stripes in parallel
    }

    virtual void saveRenderedImage(const char* png_filename = "mandelbrot.png") override {
        std::vector<uint8_t> image;
        constexpr float scaleFactor = 255.0f;   // mandelbrotApp.h:174
        auto t0 = std::chrono::steady_clock::now();
        if (progressive.active()) {   // run() streamed the bands to the PNG workers: wait for the last stripes, write the file
            printf("writing %s\n", png_filename);
            std::vector<uint8_t> png;
            std::string err = progressive.finish(png);
            times.pngJoinMs = progressive.lastJoinMs(); times.pngAssembleMs = progressive.lastAssembleMs();
            auto tw = std::chrono::steady_clock::now();
            if (err.empty()) err = pngwriter::writeFile(png_filename, png);
            times.pngWriteMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw).count();
            if (!err.empty()) printf("encoder error: %s", err.c_str());
            times.convertMs = 0.0;
            times.pngMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            return;
        }
        if (fusedSave()) {   // getRenderedImage's cast inside the PNG writer's stripe workers: one pass over the storage buffer
            printf("writing %s\n", png_filename);
            std::string err = writePngFromStorage(png_filename, resx, resy, scaleFactor, false);
            if (!err.empty()) printf("encoder error: %s", err.c_str());
            times.convertMs = 0.0;
            times.pngMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            return;
        }
        if (!gpuPostprocess) getRenderedImage(image, resx, resy, scaleFactor);   // (else: converted on the device, same cast semantics)
        auto t1 = std::chrono::steady_clock::now();
        printf("writing %s\n", png_filename);
        std::string err = writePng(png_filename, gpuPostprocess ? rgba8.bytes() : image.data(), resx, resy);
        if (!err.empty()) printf("encoder error: %s", err.c_str());   // printed, not thrown (mandelbrotApp.h:183)
        times.convertMs = std::chrono::duration<double, std::milli>(t1 - t0).count();
        times.pngMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    }

    const HostStorage& storageBuffer() const { return buffer; }

private:
    // The image in row bands of kBandRows through mc_mandelbrot_render_banded: band k + 1 is launched on a second stream before band k has
    // finished, and every band is handed to the PNG workers as it arrives.  At K4 (5120 rows: 8 bands) the 50 ms of filter + deflate run
    // beside the 60 ms of rendering instead of after them.  (A blocking render per band was measured first: every band's last tiles then
    // drain on an otherwise empty device — K4's kernels 72 ms instead of 60, profiles/r06_streamed_save_probe_blocking_bands.txt.)
    static constexpr uint32_t kBandRows = 640;
    // Where streaming pays: K4 (7680 x 5120, M = 50 000: 60 ms of rendering beside 50 ms of PNG work) gains 29 ms of 204; K1 (3200 x 2400,
    // M = 1000: a 0.2 ms kernel, 2 ms of copy) LOSES 10 to the second stream's first launch (profiles/r06_streamed_save_probe.txt).  The
    // work bound W x H x M tells the two apart before anything has run; the line sits a decade above K1 and a decade below K4.
    virtual bool worthStreaming() const override { return (double)resx * (double)resy * (double)params.max_iter >= 1e11; }
    static void bandArrived(uint32_t rowsDone, void* self) { static_cast<MandelbrotApp*>(self)->progressive.rowsReady(rowsDone); }
    void runStreamed() {
        constexpr float scaleFactor = 255.0f;   // mandelbrotApp.h:174
        if (gpuPostprocess) progressive.beginOpaqueRgba8(rgba8.bytes(), resx, resy, pngThreads);   // (the device conversion writes alpha 255)
        else progressive.beginStorage(buffer.data(), resx, resy, scaleFactor, pngThreads);
        check(mc_mandelbrot_render_banded(ctx, &params, gpuPostprocess ? nullptr : buffer.data(), gpuPostprocess ? rgba8.bytes() : nullptr,
                                          kBandRows, &MandelbrotApp::bandArrived, this), "mc_mandelbrot_render_banded");
        times.streamedBands = (int)((resy + kBandRows - 1) / kBandRows);
    }
    struct Pixel { float r, g, b, a; };   // mandelbrotApp.h:187-189
    static void split(double d, float& hi, float& lo) { hi = (float)d; lo = (float)(d - (double)hi); }
    uint64_t bufferSize;
    uint32_t resx, resy;
    uint32_t workgroupSize;
    mc_mandelbrot_params params;
};

#endif  // MANDELBROTAPP_H_
