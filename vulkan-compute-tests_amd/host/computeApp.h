// ComputeApp — the HIP/MI355X stand-in for the reference's `VulkanComputeApp`
// (src/vulkanComputeApp.h:30-127): same public lifecycle as driven by src/main.cpp:28-33
//     init() -> preRun() -> run() -> saveRenderedImage(filename)
// but underneath it owns an mc_context (include/mc_compute.h) instead of a VkInstance/VkDevice/
// VkPipeline/VkCommandBuffer.  The Vulkan plumbing (descriptor sets, pipeline layout, SPIR-V loading,
// memory-type search, fences) has no counterpart: a HIP launch needs none of it.
//
// Error convention: the reference prints+asserts on VkResult (vulkanComputeApp.h:16-24), exit(-1)s
// (:11-12) or throws std::runtime_error (vulkanComputeApp.cpp:78,218,...).  Here every non-zero
// C-ABI status becomes a std::runtime_error, so main's catch block (main.cpp:35-38) keeps its
// behaviour: message on stdout, EXIT_FAILURE.
#ifndef COMPUTEAPP_H_
#define COMPUTEAPP_H_

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "mc_compute.h"
#include "pngWriter.h"

// The storage buffer's host side: page-locked memory from mc_host_alloc — what stands where the reference allocates its output
// buffer HOST_VISIBLE | HOST_COHERENT and maps it (vulkanComputeApp.cpp:489-533, mandelbrotApp.h:153, pathtracerApp.h:206).
class HostStorage {
public:
    HostStorage() = default;
    HostStorage(const HostStorage&) = delete;
    HostStorage& operator=(const HostStorage&) = delete;
    ~HostStorage() { release(); }
    void allocate(uint64_t bytes);             // throws std::runtime_error; the contents are unspecified until run() has filled them
    void release() { if (ptr_) mc_host_free(ptr_); ptr_ = nullptr; bytes_ = 0; }
    float* data() { return static_cast<float*>(ptr_); }
    const float* data() const { return static_cast<const float*>(ptr_); }
    uint8_t* bytes() { return static_cast<uint8_t*>(ptr_); }
    const uint8_t* bytes() const { return static_cast<const uint8_t*>(ptr_); }
    size_t size() const { return bytes_ / sizeof(float); }   // in floats
    size_t sizeBytes() const { return bytes_; }
    bool empty() const { return bytes_ == 0; }
    const float* begin() const { return data(); }
    const float* end() const { return data() + size(); }
    float operator[](size_t i) const { return data()[i]; }
private:
    void* ptr_ = nullptr;
    size_t bytes_ = 0;
};

struct ComputeApp {
    virtual ~ComputeApp();

    void init();                 // vulkanComputeApp.cpp:443-449 (createInstance/findPhysicalDevice/createDevice)
    virtual void preRun() {}     // apps allocate the output storage buffer here (mandelbrotApp.h:22-25)
    virtual void run();          // vulkanComputeApp.cpp:451-466: record the dispatch(es), submit, wait

    // Records what run() will execute (the reference records vkCmdDispatch calls here).
    virtual void createCommandBuffer() {}
    // Submits and blocks until the device is done (vulkanComputeApp.cpp:645-671).
    virtual void runCommandBuffer() = 0;

    virtual void saveRenderedImage(const char* png_filename) = 0;   // vulkanComputeApp.h:67

    // -- additions (reference defaults when untouched) --
    void setNumGpus(int n) { numGpus = n < 1 ? 1 : n; }   // row-tiled multi-GPU render + RCCL gather
    void setDevice(int d) { deviceIndex = d; }
    void setQuiet(bool q) { quiet = q; }
    // float->u8 conversion (+ rotation) on the device so only RGBA8 crosses PCIe; the fp32 storage buffer is then
    // not copied to the host (storageBuffer() stays empty).  Same cast semantics, same bytes.
    void setGpuPostprocess(bool g) { gpuPostprocess = g; }
    void setPngThreads(int t) { pngThreads = t; }   // 0 = all cores, 1 = serial deflate; the host float -> u8 loop uses the same count
    // saveRenderedImage through the REFERENCE's own codec — lodepng::encode(filename, image, w, h), mandelbrotApp.h:181 /
    // pathtracerApp.h:245 — compiled from a reference checkout by `make REFERENCE=<checkout>` (nothing of it lives in this tree):
    // the file is then the reference's byte for byte.  In a build without it the setter throws.
    void setReferencePng(bool r);
    static bool referencePngAvailable();
    // Cold-start hiding, on by default, switchable for measurements:
    //   overlapStart   init() hands mc_context_warmup_* to a helper thread as soon as the context exists (joined before run() touches
    //                  the context): the kernel family's code object, tables and device scratch are ready when run() launches, and
    //                  the helper works while the caller allocates its storage buffer in preRun();
    //   false          the round-5 order: the first launch, with everything it drags in, inside run().
    void setOverlapStart(bool o) { overlapStart = o; }
    // Streamed save (round 6, the Mandelbrot app; the caller says beforehand that the image WILL be saved — main.cpp does): run() renders
    // the image in pipelined row bands (mc_mandelbrot_render_banded) and hands every band that has arrived to the PNG writer's stripe
    // workers, which filter and deflate it while the device renders the next ones; saveRenderedImage() then waits for the last stripes
    // and writes the file.  The storage buffer (or the RGBA8
    // image) is complete after run() as ever, the file is byte for byte the one the unstreamed save writes.  Off by default: run() alone
    // does what the reference's does.  Not with --reference-png (lodepng::encode takes the finished image) or more than one GPU.
    // kStreamAuto: only where it pays (worthStreaming(): the second stream of the pipelined render costs 9 ms to make and 9 ms at its
    // first launch in a cold process — more than a render of a few milliseconds could hide).
    enum { kStreamOff = 0, kStreamOn = 1, kStreamAuto = 2 };
    void setStreamedSave(int mode) { streamedSave = mode; }
    // saveRenderedImage's file: a standard PNG of exactly the RGBA8 pixels the reference converts its buffer to (mandelbrotApp.h:159-174,
    // pathtracerApp.h:202-243), deflated stripe-parallel by pngWriter.h.  The reference encodes the same pixels with its vendored
    // third-party codec (lodepng::encode, mandelbrotApp.h:181 / pathtracerApp.h:245): a reference tree that calls this library
    // (INTEGRATION.md route B) keeps that call and therefore its exact bytes; the standalone apps do not re-implement that codec.
    std::string writePng(const char* filename, const uint8_t* rgba8, uint32_t w, uint32_t h) const;
    // The storage-buffer route's saveRenderedImage in one pass (round 6): the stripe workers of the PNG writer convert the rows they
    // filter, so the RGBA8 intermediate image is never materialised; --reference-png needs that image (lodepng::encode takes it) and
    // goes the two-step way.  Same file bytes as convertStorage + writePng.
    bool fusedSave() const { return !gpuPostprocess && !referencePng; }
    std::string writePngFromStorage(const char* filename, uint32_t w, uint32_t h, float scale, bool rotate180) const;
    double lastRunMilliseconds() const { return lastRunMs; }
    // Where the time of the last run() / saveRenderedImage() went (milliseconds; SURVEY §8d "end-to-end ... reported separately"):
    // device time of the kernels and of the device -> host copy (mc_context_last_timing; 0 for multi-GPU runs), the host
    // float -> u8 (+ rotation) loop, the PNG encoder + file write.
    // allocMs = the storage buffer's allocation (preRun()); warmupMs = the warm-up call on its helper thread, warmupWaitMs = what
    // run() still waited for it.
    // streamedBands = row bands run() rendered with the PNG workers running beside it (0: the save was not streamed; then pngMs is all
    // of the PNG work, otherwise what was left of it after run())
    struct Timing { double allocMs = 0, runMs = 0, kernelMs = 0, copyMs = 0, convertMs = 0, pngMs = 0, warmupMs = 0, warmupWaitMs = 0; int streamedBands = 0;
                    double pngJoinMs = 0, pngAssembleMs = 0, pngWriteMs = 0; };   // a streamed save's tail: last stripes, file image, write
    const Timing& timing() const { return times; }

protected:
    void createBuffer(uint64_t bufferSizeBytes);   // vulkanComputeApp.cpp:489-533: the output storage buffer (gpuPostprocess: a quarter
                                                   // of it, for the RGBA8 image)
    void waitWarmup();                             // before the first call on ctx after init(): joins the warm-up helper
    // First statement of every derived destructor: the helper calls the derived class's warmup(), which reads the derived object's scene
    // tables and parameters — an object torn down before run() (preRun() threw) must not release them under the helper's feet.
    void joinWarmupQuietly() noexcept { if (warmThread.joinable()) warmThread.join(); }
    virtual int warmup() { return MC_OK; }                // apps: mc_context_warmup_* for the request run() will make (helper thread)
    static void check(int status, const char* what);

    mc_context* ctx = nullptr;
    mc_multi* multi = nullptr;
    int numGpus = 1;
    int deviceIndex = 0;   // the reference always takes devices[0] (vulkanComputeApp.cpp:163)
    bool quiet = false;
    bool gpuPostprocess = false;
    int pngThreads = 0;
    bool referencePng = false;
    bool overlapStart = true;
    int streamedSave = kStreamOff;
    virtual bool worthStreaming() const { return false; }
    bool streaming() const { return !multi && !referencePng && (streamedSave == kStreamOn || (streamedSave == kStreamAuto && worthStreaming())); }
    pngwriter::Progressive progressive;      // the save in progress while run() renders (streaming())
    std::thread warmThread;
    int warmStatus = MC_OK;
    std::string warmError;
    HostStorage rgba8;            // gpuPostprocess: the RGBA8 image run() fills (page-locked too: 4 B/pixel cross PCIe), allocated by
                                  // preRun INSTEAD of the 16-B/pixel storage buffer, which such a run never copies to the host
    double lastRunMs = 0.0;
    Timing times;
    // The storage buffer, host side: vec4 fp32 per pixel, row-major (what vkMapMemory exposes to
    // getRenderedImage in the reference).
    HostStorage buffer;
    // getRenderedImage's loop (mandelbrotApp.h:159-166, pathtracerApp.h:212-219) over row stripes on `pngThreads` host threads
    // (0 = all): u8 = static_cast<uint8_t>(scale * c) with the reference binary's x86-64 semantics, alpha 255; rotate180 = the path
    // tracer's swap loop (pathtracerApp.h:236-243, incl. its odd-width middle column) applied while writing.  Same bytes as the
    // serial loops, which tests/test_gpu_output.py holds them to.
    void convertStorage(std::vector<uint8_t>& image, uint32_t resx, uint32_t resy, float scale, bool rotate180) const;
};

// Drop-in alias: code written against the reference's base-class name keeps compiling.
using VulkanComputeApp = ComputeApp;

#endif  // COMPUTEAPP_H_
