// Storage-buffer -> RGBA8 conversion on the GPU.
//
// Replaces the scalar host loops of getRenderedImage (src/mandelbrotApp.h:149-170 with scale 255,
// src/pathtracerApp.h:202-223 with scale 1) and the in-place 180-degree rotation of
// saveRenderedImage (src/pathtracerApp.h:236-243), so only 4 B/pixel instead of 16 B/pixel cross
// PCIe.  HBM-bound: 16 B read + 4 B written per pixel, one pixel per lane (float4 load, u32 store).
//
// u8 = static_cast<uint8_t>(scale * c) has undefined behaviour out of range in C++; the reference
// binary on x86-64 executes cvttss2si (truncate to int32; "indefinite" 0x80000000 when out of range
// or NaN) and keeps the low byte.  That behaviour is reproduced explicitly.
#include "mc_internal.h"

namespace mc {

namespace {

__device__ __forceinline__ uint32_t x86_float_to_u8(float v) {
    // in-range: truncate toward zero; out of int32 range or NaN: 0x80000000 -> low byte 0
    bool in_range = (v > -2147483648.0f) && (v < 2147483648.0f);
    int32_t i = in_range ? (int32_t)v : (int32_t)0x80000000;
    return (uint32_t)i & 0xffu;
}

__global__ void __launch_bounds__(256) convert_rgba8_kernel(const float4* __restrict__ src, uint32_t* __restrict__ dst,
                                                            uint32_t W, uint32_t H, float scale, int rotate180) {
    const size_t npix = (size_t)W * H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        size_t j = i;
        if (rotate180) {
            // pathtracerApp.h:237-243 swaps (x,y) <-> (W-1-x, H-1-y) for x < W/2 only: with odd W the
            // middle column stays in place (latent reference behaviour, kept).
            uint32_t y = (uint32_t)(i / W), x = (uint32_t)(i - (size_t)y * W);
            bool middle = (W & 1u) && x == W / 2u;
            if (!middle) j = (size_t)(H - 1u - y) * W + (W - 1u - x);
        }
        float4 c = src[j];
        uint32_t r = x86_float_to_u8(scale * c.x), g = x86_float_to_u8(scale * c.y), b = x86_float_to_u8(scale * c.z);
        dst[i] = r | (g << 8) | (b << 16) | 0xff000000u;
    }
}

// Rebuilds the storage buffer from interleaved tiles (the layout an RCCL gather leaves on the root):
// storage row r = (k*n_tiles + t)*B + j  <-  tile t, tile row k*B + j.  HBM-bound; 16-B granules (uint4) whenever a row
// is a multiple of 16 B, 4-B granules otherwise (the uint32 iteration plane of an image whose width is not a multiple of 4).
template <class G>
__global__ void __launch_bounds__(256) deinterleave_rows_kernel(const G* __restrict__ src, G* __restrict__ dst,
                                                                uint32_t row_granules, uint32_t H, uint32_t n_tiles,
                                                                uint32_t B, uint32_t tile_rows_padded) {
    const size_t total = (size_t)row_granules * H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t r = (uint32_t)(i / row_granules), g = (uint32_t)(i - (size_t)r * row_granules);
        uint32_t blk = r / B, j = r - blk * B;
        uint32_t t = blk % n_tiles, k = blk / n_tiles;
        size_t srow = (size_t)t * tile_rows_padded + (size_t)k * B + j;
        dst[i] = src[srow * row_granules + g];
    }
}

// Mandelbrot exchange on the root (multi-GPU): the ranks send their ITERATION COUNTS (2 or 4 B/pixel instead of the 16-B vec4),
// and the root rebuilds the storage buffer in one pass — de-interleave the tiles as above and write lut[n], the very vec4 the
// render kernel writes for that count (mandelbrot.comp:50-59: the colour is a function of n alone).  2-4 B read + 16 B
// (+ 4 B for the optional count plane) written per pixel: HBM-bound.
template <class T>
__global__ void __launch_bounds__(256) mandelbrot_assemble_kernel(const T* __restrict__ tiles, const float4* __restrict__ lut,
                                                                  float4* __restrict__ rgba, uint32_t* __restrict__ iters, uint32_t W,
                                                                  uint32_t H, uint32_t max_iter, uint32_t n_tiles, uint32_t B,
                                                                  uint32_t tile_rows_padded) {
    const size_t total = (size_t)W * H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(i / W), x = (uint32_t)(i - (size_t)r * W);
        const uint32_t blk = r / B, j = r - blk * B;
        const uint32_t t = blk % n_tiles, k = blk / n_tiles;
        uint32_t n = (uint32_t)tiles[((size_t)t * tile_rows_padded + (size_t)k * B + j) * W + x];
        if (n > max_iter) n = max_iter;   // (never: the counts come from the render kernel; keeps the table read in range)
        if (rgba) rgba[i] = lut[n];
        if (iters) iters[i] = n;
    }
}

// Path-tracer RGBA8 exchange on the root (multi-GPU, SURVEY §8(f)1: "only 4 B/px cross xGMI"): every rank converts its own tile
// (convert_rgba8_kernel, no rotation) and sends 4 B/pixel; the root de-interleaves the gathered byte tiles and applies the point
// reflection of pathtracerApp.h:236-243 in the same pass — output pixel (x, y) is storage pixel (W-1-x, H-1-y), except an odd
// width's middle column, which the reference's swap loop (x < W/2 only) leaves where it was.  4 B read + 4 B written per pixel.
__global__ void __launch_bounds__(256) assemble_rgba8_kernel(const uint32_t* __restrict__ tiles, uint32_t* __restrict__ dst, uint32_t W,
                                                             uint32_t H, uint32_t n_tiles, uint32_t B, uint32_t tile_rows_padded,
                                                             int rotate180) {
    const size_t npix = (size_t)W * H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t r = (uint32_t)(i / W), x = (uint32_t)(i - (size_t)r * W);
        if (rotate180 && !((W & 1u) && x == W / 2u)) { r = H - 1u - r; x = W - 1u - x; }
        const uint32_t blk = r / B, j = r - blk * B;
        const uint32_t t = blk % n_tiles, k = blk / n_tiles;
        dst[i] = tiles[((size_t)t * tile_rows_padded + (size_t)k * B + j) * W + x];
    }
}

}  // namespace

int mandelbrot_assemble_launch(mc_context* ctx, const mc_mandelbrot_params* p, const void* d_tiles, uint32_t iters_bytes,
                               uint32_t n_tiles, uint32_t B, uint32_t tile_rows_padded, void* d_rgba, void* d_iters, hipStream_t s) {
    if (!ctx || !p || !d_tiles || (!d_rgba && !d_iters) || !p->width || !p->height || !p->max_iter || !n_tiles || !B)
        return MC_ERR_INVALID_ARGUMENT;
    if (iters_bytes != 2u && iters_bytes != 4u) return MC_ERR_INVALID_ARGUMENT;
    if (iters_bytes == 2u && p->max_iter > 65535u) return MC_ERR_INVALID_ARGUMENT;
    const void* d_lut = nullptr;
    if (d_rgba) {
        int rc = mandelbrot_lut_device(ctx, p, s, &d_lut);
        if (rc) return rc;
    }
    const size_t total = (size_t)p->width * p->height;
    uint32_t blocks = (uint32_t)((total + 255) / 256);
    const uint32_t cap = (uint32_t)ctx->props.multiProcessorCount * 8u;
    if (blocks > cap) blocks = cap;
    if (iters_bytes == 2u)
        hipLaunchKernelGGL(mandelbrot_assemble_kernel<uint16_t>, dim3(blocks), dim3(256), 0, s, (const uint16_t*)d_tiles,
                           (const float4*)d_lut, (float4*)d_rgba, (uint32_t*)d_iters, p->width, p->height, p->max_iter, n_tiles, B,
                           tile_rows_padded);
    else
        hipLaunchKernelGGL(mandelbrot_assemble_kernel<uint32_t>, dim3(blocks), dim3(256), 0, s, (const uint32_t*)d_tiles,
                           (const float4*)d_lut, (float4*)d_rgba, (uint32_t*)d_iters, p->width, p->height, p->max_iter, n_tiles, B,
                           tile_rows_padded);
    MC_HIP_TRY(hipGetLastError());
    return d_rgba ? ctx->note_launch(s) : MC_OK;   // (reads the cached colour table)
}

int deinterleave_rows_launch(mc_context* ctx, const void* d_tiles, uint32_t W, uint32_t H, uint32_t n_tiles, uint32_t B,
                             uint32_t tile_rows_padded, uint32_t bytes_per_pixel, void* d_out, hipStream_t s) {
    if (!ctx || !d_tiles || !d_out || !W || !H || !n_tiles || !B) return MC_ERR_INVALID_ARGUMENT;
    size_t row_bytes = (size_t)W * bytes_per_pixel;
    if (bytes_per_pixel != 16 && bytes_per_pixel != 4) return MC_ERR_INVALID_ARGUMENT;
    // 16-B granules need 16-B row starts: a row length that is a multiple of 16 B AND 16-B aligned bases (a caller may pass a
    // sliced or offset device pointer); anything else takes the 4-B granule kernel
    const bool wide = (row_bytes % 16) == 0 && ((reinterpret_cast<uintptr_t>(d_tiles) | reinterpret_cast<uintptr_t>(d_out)) % 16) == 0;
    uint32_t row_granules = (uint32_t)(row_bytes / (wide ? 16 : 4));
    size_t total = (size_t)row_granules * H;
    uint32_t blocks = (uint32_t)((total + 255) / 256);
    uint32_t cap = (uint32_t)ctx->props.multiProcessorCount * 8u;
    if (blocks > cap) blocks = cap;
    if (wide)
        hipLaunchKernelGGL(deinterleave_rows_kernel<uint4>, dim3(blocks), dim3(256), 0, s, (const uint4*)d_tiles, (uint4*)d_out,
                           row_granules, H, n_tiles, B, tile_rows_padded);
    else
        hipLaunchKernelGGL(deinterleave_rows_kernel<uint32_t>, dim3(blocks), dim3(256), 0, s, (const uint32_t*)d_tiles,
                           (uint32_t*)d_out, row_granules, H, n_tiles, B, tile_rows_padded);
    MC_HIP_TRY(hipGetLastError());
    return MC_OK;
}

int assemble_rgba8_launch(mc_context* ctx, const void* d_tiles_u8, uint32_t W, uint32_t H, uint32_t n_tiles, uint32_t B,
                          uint32_t tile_rows_padded, int rotate180, void* d_rgba8, hipStream_t s) {
    if (!ctx || !d_tiles_u8 || !d_rgba8 || !W || !H || !n_tiles || !B) return MC_ERR_INVALID_ARGUMENT;
    // every storage row must exist in its tile: tile t holds the blocks t, t + n, ... — rank 0's tile is the longest
    if (tile_rows_padded < tile_rows(0, H, B, B * n_tiles)) return MC_ERR_INVALID_ARGUMENT;
    const size_t npix = (size_t)W * H;
    uint32_t blocks = (uint32_t)((npix + 255) / 256);
    const uint32_t cap = (uint32_t)ctx->props.multiProcessorCount * 8u;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(assemble_rgba8_kernel, dim3(blocks), dim3(256), 0, s, (const uint32_t*)d_tiles_u8, (uint32_t*)d_rgba8, W, H,
                       n_tiles, B, tile_rows_padded, rotate180);
    MC_HIP_TRY(hipGetLastError());
    return MC_OK;
}

int convert_rgba8_launch(mc_context* ctx, const void* d_rgba_f32, uint32_t W, uint32_t H, float scale, int rotate180,
                         void* d_rgba8, hipStream_t s) {
    if (!ctx || !d_rgba_f32 || !d_rgba8 || !W || !H) return MC_ERR_INVALID_ARGUMENT;
    size_t npix = (size_t)W * H;
    uint32_t blocks = (uint32_t)((npix + 255) / 256);
    uint32_t cap = (uint32_t)ctx->props.multiProcessorCount * 8u;   // grid-stride beyond 8 blocks/CU
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(convert_rgba8_kernel, dim3(blocks), dim3(256), 0, s, (const float4*)d_rgba_f32, (uint32_t*)d_rgba8, W,
                       H, scale, rotate180);
    MC_HIP_TRY(hipGetLastError());
    return MC_OK;
}

}  // namespace mc
