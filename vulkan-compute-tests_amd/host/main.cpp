// Entry point — same shape as the reference's src/main.cpp:15-41: mode selected by macro, the path
// tracer takes argv[1] = spp (default 500) and argv[2] = resy (default 600) with resx = resy*3/2, the
// Mandelbrot app renders 2000x2000; lifecycle init() -> preRun() -> run() -> saveRenderedImage();
// std::runtime_error -> message + EXIT_FAILURE.  Options (never reinterpreting the two positional
// arguments) expose what the reference hard-codes: --gpus N, --out FILE, --quiet, and per mode
// --width/--height/--max-iter/--centre X Y/--scale SX SY/--precision f32|ds  or  --math strict|fast|careful,
// --large-sphere-walls, --sphere-precision f32|fp64|ds|df64 (the reference's compile-time precision experiment);
// --reference-png writes the file through the reference's own lodepng (a build with `make REFERENCE=<checkout>`): its bytes.
// A value none of these lists name is an error (EXIT_FAILURE) — never a silent default.
#include <chrono>
#include <cstdlib>
#include <initializer_list>
#include <utility>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

// make sure that one token is defined
#if !defined(MANDELBROT_MODE) && !defined(PATHTRACER_MODE)
#define PATHTRACER_MODE
#endif

#if defined(MANDELBROT_MODE)
#include "mandelbrotApp.h"
#elif defined(PATHTRACER_MODE)
#include "pathtracerApp.h"
#endif

int main(int argc, char* argv[]) {
    printf("starting main!\n");

    // split options from positional arguments
    std::vector<const char*> pos;
    int gpus = 1;
    bool quiet = false, gpuPost = false, timingJson = false, referencePng = false, overlapStart = true, fullTeardown = false;
    int streamedSave = ComputeApp::kStreamAuto;
    int pngThreads = 0;
    const char* outFile = nullptr;
    uint32_t width = 2000, height = 2000, maxIter = 128, precision = MC_PRECISION_F32, mathMode = MC_PT_MATH_STRICT;
    double cx = -0.445, cy = 0.0, sx = 2.34, sy = 2.34;
    bool viewSet = false, largeSpheres = false;
    uint32_t spherePrec = MC_PT_PREC_F32;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto need = [&](int n) { if (i + n >= argc) { printf("missing value for %s\n", a.c_str()); exit(EXIT_FAILURE); } };
        // one of a closed list of words -> its code; anything else ends the run (a typo must not render something else)
        auto choice = [&](const char* v, std::initializer_list<std::pair<const char*, uint32_t>> words) -> uint32_t {
            std::string all;
            for (const auto& w : words) { if (std::strcmp(v, w.first) == 0) return w.second; all += (all.empty() ? "" : " | ") + std::string(w.first); }
            printf("%s %s: not one of %s\n", a.c_str(), v, all.c_str());
            exit(EXIT_FAILURE);
        };
        if (a == "--gpus") { need(1); gpus = atoi(argv[++i]); }
        else if (a == "--out") { need(1); outFile = argv[++i]; }
        else if (a == "--quiet") quiet = true;
        else if (a == "--gpu-postprocess") gpuPost = true;     // float->u8 (+rotation) on the device, RGBA8-only download
        else if (a == "--png-threads") { need(1); pngThreads = atoi(argv[++i]); }   // 0 = all cores (default), 1 = serial
        else if (a == "--timing-json") timingJson = true;      // one JSON line: where the wall time of this run went (bench.py end_to_end)
        else if (a == "--fast-png")                             // (round-3 command lines: the stripe-parallel writer is the only one now)
            printf("note: --fast-png has no effect — the apps always write their own standard PNG (identical pixels; the reference "
                   "codec's exact file bytes come from a reference tree that calls this library, INTEGRATION.md route B)\n");
        else if (a == "--width") { need(1); width = (uint32_t)atoi(argv[++i]); }
        else if (a == "--height") { need(1); height = (uint32_t)atoi(argv[++i]); }
        else if (a == "--max-iter") { need(1); maxIter = (uint32_t)atoi(argv[++i]); }
        else if (a == "--centre") { need(2); cx = atof(argv[++i]); cy = atof(argv[++i]); viewSet = true; }
        else if (a == "--scale") { need(2); sx = atof(argv[++i]); sy = atof(argv[++i]); viewSet = true; }
        else if (a == "--precision") { need(1); precision = choice(argv[++i], {{"f32", MC_PRECISION_F32}, {"ds", MC_PRECISION_DS}}); }
        else if (a == "--math") {   // strict (the default: bit-identical to the oracle) | fast | careful (mc_compute.h MC_PT_MATH_*)
            need(1);
            mathMode = choice(argv[++i], {{"strict", MC_PT_MATH_STRICT}, {"fast", MC_PT_MATH_FAST}, {"careful", MC_PT_MATH_FAST_CAREFUL}});
        }
        else if (a == "--reference-png") referencePng = true;       // the reference's lodepng::encode (make REFERENCE=<checkout>)
        else if (a == "--no-streamed-save") streamedSave = ComputeApp::kStreamOff;   // render everything, then encode (see setStreamedSave)
        else if (a == "--streamed-save") streamedSave = ComputeApp::kStreamOn;       // ... stream whatever the size (default: where it pays)
        else if (a == "--full-teardown") fullTeardown = true;        // run the destructors and the HIP runtime's exit handlers (see the end of main)
        else if (a == "--serial-start") overlapStart = false;        // measurements: the round-5 start-up order (no warm-up thread)
        else if (a == "--large-sphere-walls") largeSpheres = true;   // TEST_PRECISION_WITH_LARGE_SPHERE_WALLS (pathtracerApp.h:11)
        else if (a == "--sphere-precision") {                        // which #if branch of pathTracer.comp:132-256 is active
            need(1);
            spherePrec = choice(argv[++i], {{"f32", MC_PT_PREC_F32}, {"fp64", MC_PT_PREC_FP64}, {"ds", MC_PT_PREC_DS}, {"df64", MC_PT_PREC_DF64}});
        }
        else if (a.size() > 2 && a[0] == '-' && a[1] == '-') { printf("unknown option %s\n", a.c_str()); exit(EXIT_FAILURE); }
        else pos.push_back(argv[i]);
    }
    (void)width; (void)height; (void)maxIter; (void)precision; (void)mathMode; (void)cx; (void)cy; (void)sx; (void)sy; (void)viewSet; (void)largeSpheres; (void)spherePrec;

#if defined(MANDELBROT_MODE)
    MandelbrotApp app = MandelbrotApp(width, height);   // reference: 2000 x 2000 (main.cpp:20)
    app.setMaxIter(maxIter);
    if (viewSet) app.setView(cx, cy, sx, sy);
    app.setPrecision(precision);
#elif defined(PATHTRACER_MODE)
    const int32_t spp = pos.size() > 0 ? atoi(pos[0]) : 500;                           // samples per pixel
    const uint32_t resy = pos.size() > 1 ? static_cast<uint32_t>(atoi(pos[1])) : 600;  // vertical pixel resolution
    const uint32_t resx = resy * 3 / 2;                                                // horizontal pixel resolution
    PathtracerApp app = PathtracerApp(resx, resy, spp, 16, quiet);
    app.setMathMode(mathMode);
    if (largeSpheres) app.useLargeSphereWalls();
    app.setSpherePrecision(spherePrec);
#endif
    app.setNumGpus(gpus);
    app.setQuiet(quiet);
    app.setGpuPostprocess(gpuPost);
    app.setPngThreads(pngThreads);
    app.setOverlapStart(overlapStart);
    app.setStreamedSave(streamedSave);   // this program always saves what it renders (the Mandelbrot app streams; the path tracer's first
                                         // output rows are its last storage rows, and its PNG is 3 ms beside a 14 ms kernel)
    if (referencePng && !ComputeApp::referencePngAvailable()) {   // said before anything is rendered
        printf("--reference-png: this binary was built without the reference's PNG codec; rebuild with `make REFERENCE=<checkout of "
               "pjhusky/vulkan-compute-tests>` (its src/external/lodepng is compiled where it lies)\n");
        return EXIT_FAILURE;
    }
    app.setReferencePng(referencePng);

    const auto tStart = std::chrono::steady_clock::now();
    auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    try {
        // the reference calls init()/preRun() outside its try block (main.cpp:28-29); a missing device
        // then terminates via an uncaught exception.  Kept inside here so the failure is reported.
        app.init();
        const double initMs = since(tStart);
        app.preRun();
        printf("now running app!\n");
        app.run();
        auto t0 = std::chrono::steady_clock::now();
        if (outFile) app.saveRenderedImage(outFile);
        else app.saveRenderedImage();
        double saveMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (!quiet) printf("saveRenderedImage() finished in %.3f ms\n", saveMs);
        if (timingJson) {   // init = context creation (HIP start-up); alloc = the pinned storage buffer; run = the blocking render call, of
                            // which kernel + copy are device time; convert = float -> u8 (+ rotation) on the host (0: done on the device);
                            // png = encode + write; total = process wall time up to here
            const ComputeApp::Timing& t = app.timing();
            // (warmup = the warm-up call on its helper thread, warmup_wait = what run() still waited for it: computeApp.h)
            printf("{\"timing_ms\": {\"init\": %.3f, \"alloc\": %.3f, \"run\": %.3f, \"kernel\": %.3f, \"copy\": %.3f, \"convert\": %.3f, "
                   "\"png\": %.3f, \"total\": %.3f, \"warmup\": %.3f, \"warmup_wait\": %.3f, \"streamed_bands\": %d, "
                   "\"png_join\": %.3f, \"png_assemble\": %.3f, \"png_write\": %.3f}, "
                   "\"gpu_postprocess\": %s, \"gpus\": %d, \"overlap_start\": %s, \"reference_png\": %s, "
                   "\"main_at_ms\": %.3f, \"end_at_ms\": %.3f}\n",
                   initMs, t.allocMs, t.runMs, t.kernelMs, t.copyMs, t.convertMs, t.pngMs, since(tStart), t.warmupMs,
                   t.warmupWaitMs, t.streamedBands, t.pngJoinMs, t.pngAssembleMs, t.pngWriteMs, gpuPost ? "true" : "false", gpus, overlapStart ? "true" : "false", referencePng ? "true" : "false",
                   // CLOCK_MONOTONIC at main()'s first timed statement and now: a parent that reads the same clock around the process
                   // gets what `total` cannot contain — loading + static initialisers before main(), teardown after it
                   std::chrono::duration<double, std::milli>(tStart.time_since_epoch()).count(),
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count());
        }
    } catch (const std::runtime_error& e) {
        printf("%s\n", e.what());
        return EXIT_FAILURE;
    }

    // The picture is on disk and everything is printed.  Destroying the context, unregistering the storage buffer and the HIP runtime's
    // own exit handlers take another 45 - 50 ms (profiles/r06_init_spread_probe.txt: a third of a K2 process) to give back what the
    // operating system reclaims at exit anyway: leave at once unless asked (--full-teardown: leak checkers, sanitizers, the tests).
    if (!fullTeardown) {
        fflush(stdout);
        fflush(stderr);
        std::_Exit(EXIT_SUCCESS);
    }
    return EXIT_SUCCESS;
}
