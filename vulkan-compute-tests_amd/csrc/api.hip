// C ABI of libmc_compute.so (include/mc_compute.h): context lifecycle, host-buffer entry points,
// defaults.  (The device self-test hooks of the parity suite live in test_hooks.hip / libmc_compute_test.so.)
#include <algorithm>
#include <cstring>

#include "ds_arith.h"
#include "mc_internal.h"
#include "mc_math.h"

namespace mc {

static thread_local std::string g_detail;
void set_error_detail(const std::string& s) { g_detail = s; }

int DeviceBuffer::reserve(size_t need) {
    if (need <= bytes && ptr) return MC_OK;
    if (ptr) {
        MC_HIP_TRY(hipFree(ptr));
        ptr = nullptr; bytes = 0;
    }
    MC_HIP_TRY(hipMalloc(&ptr, need));
    bytes = need;
    return MC_OK;
}
void DeviceBuffer::release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr; bytes = 0;
}

}  // namespace mc

int mc_context::note_launch(hipStream_t s) {
    for (auto& e : launch_events)
        if (e.first == s) { MC_HIP_TRY(hipEventRecord(e.second, s)); return MC_OK; }
    hipEvent_t ev = nullptr;
    MC_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    launch_events.emplace_back(s, ev);
    MC_HIP_TRY(hipEventRecord(ev, s));
    return MC_OK;
}
int mc_context::ensure_status() {
    if (status.ptr) return MC_OK;
    int rc = status.reserve(256);
    if (rc) return rc;
    MC_HIP_TRY(hipMemset(status.ptr, 0, 256));
    return MC_OK;
}
int mc_context::check_status() {
    if (!status.ptr) return MC_OK;
    uint32_t word = 0;
    MC_HIP_TRY(hipMemcpy(&word, status.ptr, sizeof(word), hipMemcpyDeviceToHost));
    if (!word) return MC_OK;
    MC_HIP_TRY(hipMemset(status.ptr, 0, sizeof(word)));
    mc::set_error_detail("path tracer scheduler tripped a loop bound (status " + std::to_string(word) + "): the image is incomplete");
    return MC_ERR_HIP;
}
int mc_context::drain_launch_streams() {
    for (auto& e : launch_events) MC_HIP_TRY(hipEventSynchronize(e.second));
    return MC_OK;
}

namespace mc {

// The reference's scene tables — DATA from src/pathtracerApp.h:14-39 (double literals rounded to fp32
// exactly as `static float planes[] = { .85, ... }` does).
static const float kDefaultPlanes[6 * 12] = {
    -1.0f, +0.0f, +0.0f, +2.6f, 0, 0, 0, 0, (float).85, (float).25, (float).25, 1,   // Left
    +1.0f, +0.0f, +0.0f, +2.6f, 0, 0, 0, 0, (float).25, (float).35, (float).85, 1,   // Right
    +0.0f, +1.0f, +0.0f, +2.0f, 0, 0, 0, 0, (float).75, (float).75, (float).75, 1,   // Top
    +0.0f, -1.0f, +0.0f, +2.0f, 0, 0, 0, 0, (float).75, (float).75, (float).75, 1,   // Bottom
    +0.0f, +0.0f, -1.0f, +2.8f, 0, 0, 0, 0, (float).85, (float).85, (float).25, 1,   // Back
    +0.0f, +0.0f, +1.0f, +7.9f, 0, 0, 0, 0, (float)0.1, (float)0.7, (float)0.7, 1,   // Front
};
static const float kDefaultSpheres[3 * 12] = {
    (float)-1.3, (float)-1.2, (float)-1.3, (float)0.8, 0, 0, 0, 0, (float).999, (float).999, (float).999, 2,   // mirror
    (float)1.3,  (float)-1.2, (float)-0.2, (float)0.8, 0, 0, 0, 0, (float).999, (float).999, (float).999, 3,   // glass
    0, (float)(2 * 0.8), 0, (float)0.2, 100, 100, 100, 0, 0, 0, 0, 1,                                            // light
};

// ---- shader clock under load (mc_context_measure_clock) ------------------------------------------------
// Every wave runs a dependent fp32 chain for ~`trips` x 64 instructions and stamps s_memtime (shader clock) and
// s_memrealtime (constant 100 MHz) around it; the ratio is the clock the chip actually holds with every SIMD busy
// (MI355X_MICROARCH.md, DVFS give-back item 6).  Stamps go to a buffer of their own; nothing else reads them.
__global__ void __launch_bounds__(256) clock_probe_kernel(unsigned long long* __restrict__ stamps, float* __restrict__ sink, int trips) {
    float a0 = (float)threadIdx.x, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f;
    const float b = 1.0000001f, c = 1e-7f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < trips; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) { a0 = a0 * b + c; a1 = a1 * b + c; a2 = a2 * b + c; a3 = a3 * b + c; }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63u) == 0u) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x / 64u) + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
    if (a0 + a1 + a2 + a3 == 12345.678f) sink[0] = a0;   // keeps the chain alive
}

}  // namespace mc

using namespace mc;

extern "C" {

int mc_abi_version(void) { return MC_ABI_VERSION; }

const char* mc_error_string(int status) {
    switch (status) {
        case MC_OK: return "ok";
        case MC_ERR_INVALID_ARGUMENT: return "invalid argument";
        case MC_ERR_NO_DEVICE: return "could not find a HIP device";   // cf. vulkanComputeApp.cpp:78
        case MC_ERR_HIP: return "HIP runtime error";
        case MC_ERR_RCCL: return "RCCL error";
        case MC_ERR_UNSUPPORTED: return "unsupported configuration";
        case MC_ERR_OUT_OF_MEMORY: return "out of device memory";
        default: return "unknown error";
    }
}

const char* mc_last_error_detail(void) { return g_detail.c_str(); }

int mc_device_count(int* count) {
    if (!count) return MC_ERR_INVALID_ARGUMENT;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        set_error_detail(std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
        return MC_ERR_NO_DEVICE;
    }
    *count = n;
    return MC_OK;
}

int mc_context_create(int device, mc_context** out_ctx) {
    if (!out_ctx) return MC_ERR_INVALID_ARGUMENT;
    *out_ctx = nullptr;
    int n = 0;
    int rc = mc_device_count(&n);
    if (rc) return rc;
    if (device < 0 || device >= n) {
        set_error_detail("device index out of range");
        return MC_ERR_NO_DEVICE;
    }
    MC_HIP_TRY(hipSetDevice(device));
    mc_context* ctx = new mc_context();
    ctx->device = device;
    hipError_t e = hipGetDeviceProperties(&ctx->props, device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error_detail(std::string("context creation: ") + hipGetErrorString(e));
        delete ctx;
        return MC_ERR_HIP;
    }
    *out_ctx = ctx;
    return MC_OK;
}

int mc_context_destroy(mc_context* ctx) {
    if (!ctx) return MC_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->stream);
    }
    for (auto& e : ctx->launch_events) { (void)hipEventSynchronize(e.second); (void)hipEventDestroy(e.second); }
    ctx->launch_events.clear();
    ctx->lut.release();
    ctx->ctab.release();
    ctx->scratch_rgba.release();
    ctx->scratch_iters.release();
    ctx->scratch_u8.release();
    ctx->scene_buf.release();
    ctx->status.release();
    delete ctx;
    return MC_OK;
}

int mc_context_device_info(mc_context* ctx, char* name, size_t name_len, int* compute_units, int* clock_khz) {
    if (!ctx) return MC_ERR_INVALID_ARGUMENT;
    if (name && name_len) {
        std::strncpy(name, ctx->props.name, name_len - 1);
        name[name_len - 1] = 0;
    }
    if (compute_units) *compute_units = ctx->props.multiProcessorCount;
    if (clock_khz) *clock_khz = ctx->props.clockRate;
    return MC_OK;
}

int mc_context_synchronize(mc_context* ctx) {
    if (!ctx) return MC_ERR_INVALID_ARGUMENT;
    MC_HIP_TRY(hipSetDevice(ctx->device));
    MC_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ctx->check_status();
}

int mc_context_measure_clock(mc_context* ctx, double* sclk_mhz) {
    if (!ctx || !sclk_mhz) return MC_ERR_INVALID_ARGUMENT;
    MC_HIP_TRY(hipSetDevice(ctx->device));
    const int blocks = ctx->props.multiProcessorCount * 8, waves = blocks * 4;   // 8 waves per SIMD: every CU full
    DeviceBuffer stamps, sink;
    struct Release { DeviceBuffer &a, &b; ~Release() { a.release(); b.release(); } } release{stamps, sink};
    int rc;
    if ((rc = stamps.reserve(sizeof(unsigned long long) * 2 * waves))) return rc;
    if ((rc = sink.reserve(256))) return rc;
    std::vector<unsigned long long> host(2 * (size_t)waves);
    for (int pass = 0; pass < 2; pass++) {   // pass 0 warms the clock governor up; pass 1 (~2 ms of full-chip VALU work) is read
        hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (unsigned long long*)stamps.ptr,
                           (float*)sink.ptr, pass == 0 ? 2000 : 4000);
        MC_HIP_TRY(hipGetLastError());
    }
    MC_HIP_TRY(hipMemcpyAsync(host.data(), stamps.ptr, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    MC_HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::vector<double> mhz;
    for (int w = 0; w < waves; w++)
        if (host[2 * w + 1]) mhz.push_back(100.0 * (double)host[2 * w] / (double)host[2 * w + 1]);
    if (mhz.empty()) return MC_ERR_HIP;
    std::nth_element(mhz.begin(), mhz.begin() + mhz.size() / 2, mhz.end());
    *sclk_mhz = mhz[mhz.size() / 2];
    return MC_OK;
}

int mc_pathtrace_scene_class(const float* planes, uint32_t n_planes, const float* spheres, uint32_t n_spheres,
                             uint32_t* out_class) {
    if (!out_class || (!planes && n_planes) || (!spheres && n_spheres)) return MC_ERR_INVALID_ARGUMENT;
    *out_class = pathtrace_scene_class(planes, n_planes, spheres, n_spheres);
    return MC_OK;
}

int mc_pathtrace_select_kernel(const mc_pathtrace_params* p, const float* planes, uint32_t n_planes, const float* spheres,
                               uint32_t n_spheres, mc_pathtrace_kernel_info* out) {
    return pathtrace_select_kernel(p, planes, n_planes, spheres, n_spheres, out);
}

uint32_t mc_row_block(void) { return kRowBlock; }

uint32_t mc_tile_rows(uint32_t row_begin, uint32_t row_end, uint32_t row_block, uint32_t row_stride) {
    return tile_rows(row_begin, row_end, row_stride ? row_block : 0u, row_stride);
}

int mc_deinterleave_rows_device_async(mc_context* ctx, const void* d_tiles, uint32_t width, uint32_t height,
                                      uint32_t n_tiles, uint32_t row_block, uint32_t tile_rows_padded,
                                      uint32_t bytes_per_pixel, void* d_out, void* stream) {
    if (!ctx) return MC_ERR_INVALID_ARGUMENT;
    MC_HIP_TRY(hipSetDevice(ctx->device));
    return deinterleave_rows_launch(ctx, d_tiles, width, height, n_tiles, row_block, tile_rows_padded, bytes_per_pixel,
                                    d_out, pick_stream(ctx, stream));
}

int mc_mandelbrot_assemble_device_async(mc_context* ctx, const mc_mandelbrot_params* p, const void* d_tiles, uint32_t iters_bytes,
                                        uint32_t n_tiles, uint32_t row_block, uint32_t tile_rows_padded, void* d_rgba_f32,
                                        void* d_iters, void* stream) {
    if (!ctx) return MC_ERR_INVALID_ARGUMENT;
    MC_HIP_TRY(hipSetDevice(ctx->device));
    return mandelbrot_assemble_launch(ctx, p, d_tiles, iters_bytes, n_tiles, row_block, tile_rows_padded, d_rgba_f32, d_iters,
                                      pick_stream(ctx, stream));
}

// Experiment only (tools/mandel_order_probe.py; not declared in the header): dispatch the Mandelbrot tiles in the order given
// (device array of n tile indices, tile = tile_y * tiles_x + tile_x), or in natural order again with d_order = NULL.
int mc_debug_mandelbrot_tile_order(mc_context* ctx, const void* d_order, size_t n) {
    if (!ctx) return MC_ERR_INVALID_ARGUMENT;
    ctx->debug_tile_order = d_order;
    ctx->debug_tile_order_n = d_order ? n : 0;
    return MC_OK;
}

// ---- Mandelbrot -------------------------------------------------------------------------------------
int mc_mandelbrot_default_params(uint32_t width, uint32_t height, mc_mandelbrot_params* p) {
    if (!p) return MC_ERR_INVALID_ARGUMENT;
    std::memset(p, 0, sizeof(*p));
    p->width = width; p->height = height;
    p->max_iter = 128;                               // mandelbrot.comp:40
    p->precision = MC_PRECISION_F32;
    p->centre_x_hi = -0.445f; p->centre_y_hi = 0.0f; // mandelbrot.comp:38
    p->scale_x_hi = 2.34f; p->scale_y_hi = 2.34f;    // 2.0+1.7*0.2 folds to 2.34f
    p->k_color[0] = 0.1f; p->k_color[1] = 0.7f; p->k_color[2] = 0.6f; p->k_color[3] = 0.0f;   // mandelbrotApp.h:139
    p->row_begin = 0; p->row_end = height;
    return MC_OK;
}

int mc_mandelbrot_colour_lut(uint32_t max_iter, const float k_color[4], float* lut_f32) {
    if (!max_iter || !k_color || !lut_f32) return MC_ERR_INVALID_ARGUMENT;
    mandelbrot_build_lut(max_iter, k_color, lut_f32);
    return MC_OK;
}

int mc_mandelbrot_render_device_async(mc_context* ctx, const mc_mandelbrot_params* p, void* d_rgba_f32, void* d_iters,
                                      void* stream) {
    if (!ctx) return MC_ERR_INVALID_ARGUMENT;
    MC_HIP_TRY(hipSetDevice(ctx->device));
    return mandelbrot_launch(ctx, p, d_rgba_f32, d_iters, pick_stream(ctx, stream));
}

int mc_mandelbrot_render(mc_context* ctx, const mc_mandelbrot_params* p, float* out_rgba_f32, uint32_t* out_iters) {
    if (!ctx || !p || (!out_rgba_f32 && !out_iters)) return MC_ERR_INVALID_ARGUMENT;
    if (p->row_end > p->height || p->row_begin >= p->row_end || !p->width) return MC_ERR_INVALID_ARGUMENT;
    // the host form's iteration plane is uint32_t*: the uint16 exchange format exists for the device form only
    if (p->flags & MC_MANDEL_ITERS_U16) {
        set_error_detail("MC_MANDEL_ITERS_U16 applies to mc_mandelbrot_render_device_async only (out_iters is uint32_t*)");
        return MC_ERR_INVALID_ARGUMENT;
    }
    MC_HIP_TRY(hipSetDevice(ctx->device));
    const size_t npix = (size_t)mc_tile_rows(p->row_begin, p->row_end, p->row_block, p->row_stride) * p->width;
    int rc;
    if (out_rgba_f32 && (rc = ctx->scratch_rgba.reserve(npix * 16))) return rc;
    if (out_iters && (rc = ctx->scratch_iters.reserve(npix * 4))) return rc;
    void* d_rgba = out_rgba_f32 ? ctx->scratch_rgba.ptr : nullptr;
    void* d_it = out_iters ? ctx->scratch_iters.ptr : nullptr;
    rc = mandelbrot_launch(ctx, p, d_rgba, d_it, ctx->stream);
    if (rc) return rc;
    if (out_rgba_f32) MC_HIP_TRY(hipMemcpyAsync(out_rgba_f32, d_rgba, npix * 16, hipMemcpyDeviceToHost, ctx->stream));
    if (out_iters) MC_HIP_TRY(hipMemcpyAsync(out_iters, d_it, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    MC_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MC_OK;
}

// ---- Path tracer ------------------------------------------------------------------------------------
int mc_pathtrace_default_params(uint32_t width, uint32_t height, uint32_t spp, mc_pathtrace_params* p) {
    if (!p) return MC_ERR_INVALID_ARGUMENT;
    std::memset(p, 0, sizeof(*p));
    p->width = width; p->height = height; p->spp = spp;
    p->sample_begin = 0; p->sample_end = spp;
    p->max_depth = 12;                               // pathTracer.comp:367
    p->row_begin = 0; p->row_end = height;
    p->math_mode = MC_PT_MATH_STRICT;
    return MC_OK;
}

int mc_pathtrace_default_scene(const float** planes, uint32_t* n_planes, const float** spheres, uint32_t* n_spheres) {
    if (!planes || !n_planes || !spheres || !n_spheres) return MC_ERR_INVALID_ARGUMENT;
    *planes = kDefaultPlanes; *n_planes = 6;
    *spheres = kDefaultSpheres; *n_spheres = 3;
    return MC_OK;
}

int mc_pathtrace_render_device_async(mc_context* ctx, const mc_pathtrace_params* p, const float* planes, uint32_t n_planes,
                                     const float* spheres, uint32_t n_spheres, void* d_rgba_f32, void* stream) {
    if (!ctx) return MC_ERR_INVALID_ARGUMENT;
    MC_HIP_TRY(hipSetDevice(ctx->device));
    return pathtrace_launch(ctx, p, planes, n_planes, spheres, n_spheres, d_rgba_f32, pick_stream(ctx, stream));
}

int mc_pathtrace_render(mc_context* ctx, const mc_pathtrace_params* p, const float* planes, uint32_t n_planes,
                        const float* spheres, uint32_t n_spheres, float* out_rgba_f32) {
    if (!ctx || !p || !out_rgba_f32) return MC_ERR_INVALID_ARGUMENT;
    if (p->row_end > p->height || p->row_begin >= p->row_end || !p->width) return MC_ERR_INVALID_ARGUMENT;
    MC_HIP_TRY(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)mc_tile_rows(p->row_begin, p->row_end, p->row_block, p->row_stride) * p->width * 16;
    int rc = ctx->scratch_rgba.reserve(bytes);
    if (rc) return rc;
    if (p->sample_begin > 0)   // progressive continuation: the caller's buffer holds the accumulator
        MC_HIP_TRY(hipMemcpyAsync(ctx->scratch_rgba.ptr, out_rgba_f32, bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = pathtrace_launch(ctx, p, planes, n_planes, spheres, n_spheres, ctx->scratch_rgba.ptr, ctx->stream);
    if (rc) return rc;
    MC_HIP_TRY(hipMemcpyAsync(out_rgba_f32, ctx->scratch_rgba.ptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MC_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ctx->check_status();
}

// ---- post-process -----------------------------------------------------------------------------------
int mc_convert_rgba8_device_async(mc_context* ctx, const void* d_rgba_f32, uint32_t width, uint32_t height, float scale,
                                  int rotate180, void* d_rgba8, void* stream) {
    if (!ctx) return MC_ERR_INVALID_ARGUMENT;
    MC_HIP_TRY(hipSetDevice(ctx->device));
    return convert_rgba8_launch(ctx, d_rgba_f32, width, height, scale, rotate180, d_rgba8, pick_stream(ctx, stream));
}

int mc_convert_rgba8(mc_context* ctx, const float* rgba_f32, uint32_t width, uint32_t height, float scale, int rotate180,
                     uint8_t* rgba8) {
    if (!ctx || !rgba_f32 || !rgba8 || !width || !height) return MC_ERR_INVALID_ARGUMENT;
    MC_HIP_TRY(hipSetDevice(ctx->device));
    const size_t npix = (size_t)width * height;
    int rc;
    if ((rc = ctx->scratch_rgba.reserve(npix * 16))) return rc;
    if ((rc = ctx->scratch_u8.reserve(npix * 4))) return rc;
    MC_HIP_TRY(hipMemcpyAsync(ctx->scratch_rgba.ptr, rgba_f32, npix * 16, hipMemcpyHostToDevice, ctx->stream));
    rc = convert_rgba8_launch(ctx, ctx->scratch_rgba.ptr, width, height, scale, rotate180, ctx->scratch_u8.ptr, ctx->stream);
    if (rc) return rc;
    MC_HIP_TRY(hipMemcpyAsync(rgba8, ctx->scratch_u8.ptr, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    MC_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MC_OK;
}

// ---- render + post-process fused on the device ------------------------------------------------------------
static int download_rgba8(mc_context* ctx, uint32_t W, uint32_t H, float scale, int rotate180, uint8_t* out_rgba8) {
    const size_t npix = (size_t)W * H;
    int rc = ctx->scratch_u8.reserve(npix * 4);
    if (rc) return rc;
    rc = convert_rgba8_launch(ctx, ctx->scratch_rgba.ptr, W, H, scale, rotate180, ctx->scratch_u8.ptr, ctx->stream);
    if (rc) return rc;
    MC_HIP_TRY(hipMemcpyAsync(out_rgba8, ctx->scratch_u8.ptr, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    MC_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ctx->check_status();
}

int mc_mandelbrot_render_rgba8(mc_context* ctx, const mc_mandelbrot_params* p, uint8_t* out_rgba8) {
    if (!ctx || !p || !out_rgba8) return MC_ERR_INVALID_ARGUMENT;
    if (p->row_begin != 0 || p->row_end != p->height || p->row_stride || !p->width || !p->height) return MC_ERR_INVALID_ARGUMENT;
    if (p->flags & MC_MANDEL_ITERS_U16) return MC_ERR_INVALID_ARGUMENT;   // (no iteration plane leaves this entry point)
    MC_HIP_TRY(hipSetDevice(ctx->device));
    int rc = ctx->scratch_rgba.reserve((size_t)p->width * p->height * 16);
    if (rc) return rc;
    rc = mandelbrot_launch(ctx, p, ctx->scratch_rgba.ptr, nullptr, ctx->stream);
    if (rc) return rc;
    return download_rgba8(ctx, p->width, p->height, 255.0f, 0, out_rgba8);      // mandelbrotApp.h:159-174
}

int mc_pathtrace_render_rgba8(mc_context* ctx, const mc_pathtrace_params* p, const float* planes, uint32_t n_planes,
                              const float* spheres, uint32_t n_spheres, uint8_t* out_rgba8) {
    if (!ctx || !p || !out_rgba8) return MC_ERR_INVALID_ARGUMENT;
    if (p->row_begin != 0 || p->row_end != p->height || p->row_stride || p->sample_begin != 0 || p->sample_end != p->spp ||
        !p->width || !p->height)
        return MC_ERR_INVALID_ARGUMENT;
    MC_HIP_TRY(hipSetDevice(ctx->device));
    int rc = ctx->scratch_rgba.reserve((size_t)p->width * p->height * 16);
    if (rc) return rc;
    rc = pathtrace_launch(ctx, p, planes, n_planes, spheres, n_spheres, ctx->scratch_rgba.ptr, ctx->stream);
    if (rc) return rc;
    return download_rgba8(ctx, p->width, p->height, 1.0f, 1, out_rgba8);        // pathtracerApp.h:202-243
}

}  // extern "C"
