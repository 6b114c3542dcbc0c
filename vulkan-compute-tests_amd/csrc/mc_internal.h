// Internal declarations shared by the translation units of libmc_compute.so (not installed).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/mc_compute.h"
#include "../../include/mc_compute_test.h"   // the measurement switches the library honours for tools/ and tests/ (flag bits only)

namespace mc {

void set_error_detail(const std::string& s);

#define MC_HIP_TRY(expr)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            ::mc::set_error_detail(std::string(#expr) + ": " + hipGetErrorString(e_));                \
            return e_ == hipErrorOutOfMemory ? MC_ERR_OUT_OF_MEMORY : MC_ERR_HIP;                     \
        }                                                                                             \
    } while (0)

// Device-resident scratch that grows on demand and is reused between calls.
struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
    int reserve(size_t need);
    void release();
};

}  // namespace mc

struct mc_context {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // mc_mandelbrot_render_banded: odd bands (created on first use, or by the warm-up when told)
    int ensure_stream2();
    hipDeviceProp_t props{};
    // colour LUT cache (Mandelbrot): rebuilt when (max_iter, k_color) changes
    mc::DeviceBuffer lut;
    uint32_t lut_max_iter = 0xffffffffu;
    float lut_kcolor[4] = {0, 0, 0, 0};
    // per-column / per-row c tables (Mandelbrot), keyed by (W, H, precision, view)
    mc::DeviceBuffer ctab;
    std::vector<float> ctab_key;
    // scratch for the host-buffer entry points
    mc::DeviceBuffer scratch_rgba, scratch_iters, scratch_u8;
    // generic path-tracer scenes: device copy of [records | emissive indices] and the host copy it mirrors
    mc::DeviceBuffer scene_buf;
    std::vector<float> scene_host;
    // One event per stream this context has launched on, re-recorded after every launch that reads a cached table
    // (LUT, c table, scene records).  Replacing a table waits for these events only — never hipDeviceSynchronize():
    // other contexts and unrelated streams keep running.  The stream handle is only a key (an event outlives its stream).
    std::vector<std::pair<hipStream_t, hipEvent_t>> launch_events;
    int note_launch(hipStream_t s);
    int drain_launch_streams();
    // Device status word: a kernel whose scheduler trips one of its loop bounds ORs a bit in instead of spinning
    // (pathtrace_pool.h: bit 1).  Checked — and cleared — by the blocking entry points and mc_context_synchronize().
    mc::DeviceBuffer status;
    int ensure_status();
    int check_status();   // call after the stream is idle: MC_OK, or MC_ERR_HIP with a detail message
    // The blocking host-buffer entry points bracket their kernel launches and their device -> host copy with events on the context's
    // stream (mc_context_last_timing): t[0] before the first launch, t[1] after the last kernel, t[2] after the copy.
    hipEvent_t t_ev[3] = {nullptr, nullptr, nullptr};
    bool timing_valid = false;
    bool banded_timing = false;          // the last blocking call was mc_mandelbrot_render_banded: its own two figures below
    double banded_kernel_ms = 0.0, banded_copy_ms = 0.0;
    int mark(int k);      // records t_ev[k] on the context's stream (creating the events on first use)
};

namespace mc {

// mandelbrot.hip
int mandelbrot_launch(mc_context* ctx, const mc_mandelbrot_params* p, void* d_rgba, void* d_iters, hipStream_t s);
int mandelbrot_warmup(mc_context* ctx, const mc_mandelbrot_params* p, void* d_iters_scratch, hipStream_t s);
void mandelbrot_build_lut(uint32_t max_iter, const float k_color[4], float* lut);
int mandelbrot_lut_device(mc_context* ctx, const mc_mandelbrot_params* p, hipStream_t s, const void** d_lut);
// pathtrace.hip
uint32_t pathtrace_scene_class(const float* planes, uint32_t n_planes, const float* spheres, uint32_t n_spheres);
int pathtrace_select_kernel(const mc_pathtrace_params* p, const float* planes, uint32_t n_planes, const float* spheres,
                            uint32_t n_spheres, mc_pathtrace_kernel_info* out);
int pathtrace_launch(mc_context* ctx, const mc_pathtrace_params* p, const float* planes, uint32_t n_planes,
                     const float* spheres, uint32_t n_spheres, void* d_rgba, hipStream_t s);
// postprocess.hip
int convert_rgba8_launch(mc_context* ctx, const void* d_rgba_f32, uint32_t W, uint32_t H, float scale, int rotate180,
                         void* d_rgba8, hipStream_t s);
int deinterleave_rows_launch(mc_context* ctx, const void* d_tiles, uint32_t W, uint32_t H, uint32_t n_tiles, uint32_t B,
                             uint32_t tile_rows_padded, uint32_t bytes_per_pixel, void* d_out, hipStream_t s);
int assemble_rgba8_launch(mc_context* ctx, const void* d_tiles_u8, uint32_t W, uint32_t H, uint32_t n_tiles, uint32_t B,
                          uint32_t tile_rows_padded, int rotate180, void* d_rgba8, hipStream_t s);
int mandelbrot_assemble_launch(mc_context* ctx, const mc_mandelbrot_params* p, const void* d_tiles, uint32_t iters_bytes,
                               uint32_t n_tiles, uint32_t B, uint32_t tile_rows_padded, void* d_rgba, void* d_iters, hipStream_t s);

// Interleaved row-block tiling (include/mc_compute.h: row_block,row_stride).  Tile-local row ty maps to
// storage row row_begin + (ty / B) * stride + ty % B; B == 0 means contiguous.
__host__ __device__ inline uint32_t tile_row_to_storage(uint32_t ty, uint32_t row_begin, uint32_t B, uint32_t stride) {
    if (B == 0u) return row_begin + ty;
    uint32_t k = ty / B;
    return row_begin + k * stride + (ty - k * B);
}
inline uint32_t tile_rows(uint32_t row_begin, uint32_t row_end, uint32_t B, uint32_t stride) {
    if (row_begin >= row_end) return 0;
    if (B == 0u || stride == 0u) return row_end - row_begin;
    uint32_t span = row_end - row_begin;
    uint32_t full = span / stride, rem = span - full * stride;
    return full * B + (rem < B ? rem : B);
}

// Rows per interleave block of the multi-GPU row tiling — ONE value for mc_multi_* (multi.hip), sharding.py (through
// mc_row_block()) and the docs.  8: every BASELINE configuration splits into whole blocks per rank at N = 1, 2, 4, 8
// (K2 weak-scaled 600*N rows -> 75 blocks per rank, K3 2560 -> 320 blocks, K4 5120 -> 640), a wave's 8x8 / 4x4 / 2x2
// pixel tile stays inside one block, and Mandelbrot's interior rows are spread evenly.  Any value is legal:
// tile-local rows map to storage rows one by one (tile_row_to_storage).
constexpr uint32_t kRowBlock = 8;

inline hipStream_t pick_stream(mc_context* ctx, void* stream) { return stream ? (hipStream_t)stream : ctx->stream; }

}  // namespace mc
