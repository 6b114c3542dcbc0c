// Single-process multi-GPU render: interleaved row blocks per GPU, one RCCL gather of the tiles to
// device 0 over xGMI, de-interleave on device 0, one D2H copy (north_star).
//
// The reference is single-device (vulkanComputeApp.cpp:163 takes devices[0]); this is the scale-out
// the north_star adds.  Every pixel is independent and keyed by its absolute coordinates
// (pathTracer.comp:357,393), so a rank rendering rows with the GLOBAL (gx, gy, W, H) produces the
// same bits as the single-GPU run: N-GPU output must memcmp-equal the 1-GPU output.  Samples are
// never split across GPUs (the fp32 accumulation order is part of the contract).
//
// A gather to one root on the fully connected 8-GPU xGMI mesh uses the root's 7 inbound links concurrently.  The path tracer
// exchanges its fp32 tiles (K3: 19.7 MB per rank next to 0.3 s of render) — or, in the RGBA8 form, the tiles each device has
// converted itself, 4 B/pixel (K3: 4.9 MB per rank); the Mandelbrot exchanges iteration counts only —
// 2 B/pixel — and device 0 rebuilds the vec4 buffer from them (K4: 9.8 MB per rank next to 7 ms of render).
//
// RCCL is loaded ON DEMAND (dlopen in mc_multi_create, only when more than one device takes part): librccl.so is a 573 MB library
// whose device code every process that links it registers with the HIP runtime at start-up; a single-GPU render — the reference's
// own case, vulkanComputeApp.cpp:163 — never needs it and must not pay for it (profiles/r06_cold_timeline.txt).  The types come
// from <rccl/rccl.h>; the six entry points used are resolved by name.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "mc_internal.h"

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Gather)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

// Loads librccl once per process; returns nullptr (the reason is rccl_error()) when the library or a symbol is missing.
RcclApi g_rccl;
std::once_flag g_rccl_once;

RcclApi* rccl() {
    std::call_once(g_rccl_once, [] {
        RcclApi& api = g_rccl;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if ((api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!api.handle) {
            const char* why = dlerror();
            api.error = std::string("dlopen(librccl.so.1): ") + (why ? why : "not found");
            return;
        }
        auto sym = [&](const char* name) -> void* {
            void* p = dlsym(api.handle, name);
            if (!p && api.error.empty()) api.error = std::string("librccl.so.1 does not export ") + name;
            return p;
        };
        api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.Gather = reinterpret_cast<decltype(api.Gather)>(sym("ncclGather"));   // an RCCL extension (rccl.h)
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return g_rccl.error.empty() ? &g_rccl : nullptr;
}

std::string rccl_error() { (void)rccl(); return g_rccl.error; }

}  // namespace

struct mc_multi {
    int n = 0;
    std::vector<mc_context*> ctx;
    std::vector<ncclComm_t> comms;
    std::vector<mc::DeviceBuffer> tile_rgba, tile_iters, tile_u8;
    mc::DeviceBuffer gather_u8;   // on device 0
    mc::DeviceBuffer gather_rgba, gather_iters, full_rgba, full_iters, full_u8;   // on device 0
    bool use_rccl = false;
};

namespace mc {

#define MC_NCCL_TRY(expr)                                                                       \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess) {                                                                \
            ::mc::set_error_detail(std::string(#expr) + ": " + rccl()->GetErrorString(r_));         \
            return MC_ERR_RCCL;                                                                 \
        }                                                                                       \
    } while (0)

// Gathers equal-sized tiles (tile_bytes per rank) to device 0: `gathered` holds them back to back in rank order.
static int gather_tiles(mc_multi* m, std::vector<DeviceBuffer>& tiles, DeviceBuffer& gathered, size_t tile_bytes) {
    mc_context* c0 = m->ctx[0];
    MC_HIP_TRY(hipSetDevice(c0->device));
    int rc;
    if ((rc = gathered.reserve(tile_bytes * m->n))) return rc;
    if (m->use_rccl) {
        MC_NCCL_TRY(rccl()->GroupStart());
        for (int i = 0; i < m->n; i++) {
            // one thread drives all devices: make rank i's device current for its call inside the group
            if (hipSetDevice(m->ctx[i]->device) != hipSuccess) {
                (void)rccl()->GroupEnd();
                set_error_detail("hipSetDevice failed inside the gather group");
                return MC_ERR_HIP;
            }
            // ncclGather is an RCCL extension (rccl.h); bytes are moved as ncclUint8 so vec4 and u32 tiles share the path
            ncclResult_t r = rccl()->Gather(tiles[i].ptr, i == 0 ? gathered.ptr : nullptr, tile_bytes, ncclUint8, 0, m->comms[i],
                                        m->ctx[i]->stream);
            if (r != ncclSuccess) {
                (void)rccl()->GroupEnd();
                set_error_detail(std::string("ncclGather: ") + rccl()->GetErrorString(r));
                return MC_ERR_RCCL;
            }
        }
        MC_NCCL_TRY(rccl()->GroupEnd());
    } else {
        // n == 1 without a communicator: the "gather" is the tile itself
        MC_HIP_TRY(hipMemcpyAsync(gathered.ptr, tiles[0].ptr, tile_bytes, hipMemcpyDeviceToDevice, c0->stream));
    }
    MC_HIP_TRY(hipSetDevice(c0->device));
    return MC_OK;
}

// Gathers the tiles of a bpp-byte-per-pixel plane to device 0 and de-interleaves them into `full`.
static int gather_and_assemble(mc_multi* m, std::vector<DeviceBuffer>& tiles, DeviceBuffer& gathered, DeviceBuffer& full,
                               uint32_t W, uint32_t H, uint32_t tile_rows_padded, uint32_t bpp) {
    int rc;
    if ((rc = gather_tiles(m, tiles, gathered, (size_t)tile_rows_padded * W * bpp))) return rc;
    mc_context* c0 = m->ctx[0];
    if ((rc = full.reserve((size_t)W * H * bpp))) return rc;
    if (m->n == 1) {   // one tile = the image in storage order already: no row shuffle, any width
        MC_HIP_TRY(hipMemcpyAsync(full.ptr, gathered.ptr, (size_t)W * H * bpp, hipMemcpyDeviceToDevice, c0->stream));
        return MC_OK;
    }
    return deinterleave_rows_launch(c0, gathered.ptr, W, H, (uint32_t)m->n, kRowBlock, tile_rows_padded, bpp, full.ptr,
                                    c0->stream);
}

static uint32_t padded_tile_rows(uint32_t H, int n) {
    // rank 0 always owns the most rows
    return tile_rows(0, H, kRowBlock, kRowBlock * (uint32_t)n);
}

}  // namespace mc

using namespace mc;

extern "C" {

int mc_multi_create(int n_devices, mc_multi** out) {
    if (!out || n_devices < 1) return MC_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int have = 0;
    int rc = mc_device_count(&have);
    if (rc) return rc;
    if (n_devices > have) {
        set_error_detail("fewer HIP devices than requested");
        return MC_ERR_NO_DEVICE;
    }
    mc_multi* m = new mc_multi();
    m->n = n_devices;
    m->ctx.resize(n_devices, nullptr);
    m->tile_rgba.resize(n_devices);
    m->tile_iters.resize(n_devices);
    m->tile_u8.resize(n_devices);
    for (int i = 0; i < n_devices; i++) {
        rc = mc_context_create(i, &m->ctx[i]);
        if (rc) { mc_multi_destroy(m); return rc; }
    }
    const char* force = std::getenv("MC_MULTI_FORCE_RCCL");
    m->use_rccl = n_devices > 1 || (force && force[0] == '1');
    if (m->use_rccl) {
        std::vector<int> devs(n_devices);
        for (int i = 0; i < n_devices; i++) devs[i] = i;
        if (!rccl()) {   // loaded here, on first need (see the head of this file)
            m->use_rccl = false;
            set_error_detail("RCCL is needed for a render on more than one device: " + rccl_error());
            mc_multi_destroy(m);
            return MC_ERR_RCCL;
        }
        m->comms.resize(n_devices);
        ncclResult_t r = rccl()->CommInitAll(m->comms.data(), n_devices, devs.data());
        if (r != ncclSuccess) {
            m->comms.clear();
            set_error_detail(std::string("ncclCommInitAll: ") + rccl()->GetErrorString(r));
            mc_multi_destroy(m);
            return MC_ERR_RCCL;
        }
    }
    *out = m;
    return MC_OK;
}

int mc_multi_destroy(mc_multi* m) {
    if (!m) return MC_OK;
    for (size_t i = 0; i < m->comms.size(); i++)
        if (m->comms[i]) (void)rccl()->CommDestroy(m->comms[i]);
    for (int i = 0; i < m->n; i++) {
        if (!m->ctx[i]) continue;
        (void)hipSetDevice(m->ctx[i]->device);
        m->tile_rgba[i].release();
        m->tile_iters[i].release();
        m->tile_u8[i].release();
    }
    if (m->n && m->ctx[0]) {
        (void)hipSetDevice(m->ctx[0]->device);
        m->gather_rgba.release(); m->gather_iters.release(); m->full_rgba.release(); m->full_iters.release(); m->full_u8.release(); m->gather_u8.release();
    }
    for (int i = 0; i < m->n; i++) mc_context_destroy(m->ctx[i]);
    delete m;
    return MC_OK;
}

static int multi_mandelbrot(mc_multi* m, const mc_mandelbrot_params* p, float* out_rgba_f32, uint32_t* out_iters,
                            uint8_t* out_rgba8) {
    if (!m || !p || (!out_rgba_f32 && !out_iters && !out_rgba8)) return MC_ERR_INVALID_ARGUMENT;
    const bool want_rgba = out_rgba_f32 || out_rgba8;
    if (p->row_begin != 0 || p->row_end != p->height || p->row_stride) return MC_ERR_INVALID_ARGUMENT;   // whole image only
    const uint32_t W = p->width, H = p->height;
    const uint32_t padded = padded_tile_rows(H, m->n);
    // The exchange carries ITERATION COUNTS — 2 B/pixel (max_iter <= 65535) or 4 — never the 16-B vec4: the colour is a function
    // of the count alone (mandelbrot.comp:50-59), so device 0 rebuilds the storage buffer from the gathered counts through the
    // same table the render kernel reads.  K4 on 8 GPUs: 9.8 MB per rank instead of 78.6 MB.
    const bool narrow = p->max_iter <= 65535u;
    const uint32_t ib = narrow ? 2u : 4u;
    int rc;
    for (int i = 0; i < m->n; i++) {
        mc_context* c = m->ctx[i];
        MC_HIP_TRY(hipSetDevice(c->device));
        if ((rc = m->tile_iters[i].reserve((size_t)padded * W * ib))) return rc;
        mc_mandelbrot_params q = *p;
        q.row_begin = (uint32_t)i * kRowBlock; q.row_end = H;
        q.row_block = kRowBlock; q.row_stride = kRowBlock * (uint32_t)m->n;
        q.flags = narrow ? (p->flags | MC_MANDEL_ITERS_U16) : (p->flags & ~(uint32_t)MC_MANDEL_ITERS_U16);
        if (q.row_begin >= H) continue;   // more GPUs than row blocks
        rc = mandelbrot_launch(c, &q, nullptr, m->tile_iters[i].ptr, c->stream);
        if (rc) return rc;
    }
    mc_context* c0 = m->ctx[0];
    if ((rc = gather_tiles(m, m->tile_iters, m->gather_iters, (size_t)padded * W * ib))) return rc;
    if (want_rgba && (rc = m->full_rgba.reserve((size_t)W * H * 16))) return rc;
    if (out_iters && (rc = m->full_iters.reserve((size_t)W * H * 4))) return rc;
    if ((rc = mandelbrot_assemble_launch(c0, p, m->gather_iters.ptr, ib, (uint32_t)m->n, kRowBlock, padded,
                                         want_rgba ? m->full_rgba.ptr : nullptr, out_iters ? m->full_iters.ptr : nullptr, c0->stream)))
        return rc;
    if (out_rgba_f32)
        MC_HIP_TRY(hipMemcpyAsync(out_rgba_f32, m->full_rgba.ptr, (size_t)W * H * 16, hipMemcpyDeviceToHost, c0->stream));
    if (out_rgba8) {   // mandelbrotApp.h:159-174 on device 0: only 4 B/pixel leave the GPU
        if ((rc = m->full_u8.reserve((size_t)W * H * 4))) return rc;
        if ((rc = convert_rgba8_launch(c0, m->full_rgba.ptr, W, H, 255.0f, 0, m->full_u8.ptr, c0->stream))) return rc;
        MC_HIP_TRY(hipMemcpyAsync(out_rgba8, m->full_u8.ptr, (size_t)W * H * 4, hipMemcpyDeviceToHost, c0->stream));
    }
    if (out_iters)
        MC_HIP_TRY(hipMemcpyAsync(out_iters, m->full_iters.ptr, (size_t)W * H * 4, hipMemcpyDeviceToHost, c0->stream));
    for (int i = 0; i < m->n; i++) {
        MC_HIP_TRY(hipSetDevice(m->ctx[i]->device));
        MC_HIP_TRY(hipStreamSynchronize(m->ctx[i]->stream));
    }
    return MC_OK;
}

int mc_multi_mandelbrot_render(mc_multi* m, const mc_mandelbrot_params* p, float* out_rgba_f32, uint32_t* out_iters) {
    if (!out_rgba_f32 && !out_iters) return MC_ERR_INVALID_ARGUMENT;
    return multi_mandelbrot(m, p, out_rgba_f32, out_iters, nullptr);
}
int mc_multi_mandelbrot_render_rgba8(mc_multi* m, const mc_mandelbrot_params* p, uint8_t* out_rgba8) {
    if (!out_rgba8) return MC_ERR_INVALID_ARGUMENT;
    return multi_mandelbrot(m, p, nullptr, nullptr, out_rgba8);
}

static int multi_pathtrace(mc_multi* m, const mc_pathtrace_params* p, const float* planes, uint32_t n_planes,
                           const float* spheres, uint32_t n_spheres, float* out_rgba_f32, uint8_t* out_rgba8) {
    if (!m || !p || (!out_rgba_f32 && !out_rgba8)) return MC_ERR_INVALID_ARGUMENT;
    if (p->row_begin != 0 || p->row_end != p->height || p->row_stride) return MC_ERR_INVALID_ARGUMENT;
    if (p->sample_begin != 0) return MC_ERR_UNSUPPORTED;   // progressive continuation: single-GPU entry points
    // the RGBA8 form converts a FINISHED render (:453 applied): refused before anything is launched on any device
    if (out_rgba8 && p->sample_end != p->spp) return MC_ERR_INVALID_ARGUMENT;
    const uint32_t W = p->width, H = p->height;
    const uint32_t padded = padded_tile_rows(H, m->n);
    int rc;
    for (int i = 0; i < m->n; i++) {
        mc_context* c = m->ctx[i];
        MC_HIP_TRY(hipSetDevice(c->device));
        if ((rc = m->tile_rgba[i].reserve((size_t)padded * W * 16))) return rc;
        if (out_rgba8 && (rc = m->tile_u8[i].reserve((size_t)padded * W * 4))) return rc;
        mc_pathtrace_params q = *p;
        q.row_begin = (uint32_t)i * kRowBlock; q.row_end = H;
        q.row_block = kRowBlock; q.row_stride = kRowBlock * (uint32_t)m->n;
        if (q.row_begin >= H) continue;
        rc = pathtrace_launch(c, &q, planes, n_planes, spheres, n_spheres, m->tile_rgba[i].ptr, c->stream);
        if (rc) return rc;
        // RGBA8 form: every device converts ITS tile (pathtracerApp.h:212-219, scale 1; tile rows in tile order, no rotation yet), so
        // 4 B/pixel cross xGMI instead of 16
        if (out_rgba8 && (rc = convert_rgba8_launch(c, m->tile_rgba[i].ptr, W, tile_rows(q.row_begin, H, kRowBlock, q.row_stride), 1.0f, 0,
                                                    m->tile_u8[i].ptr, c->stream)))
            return rc;
    }
    mc_context* c0 = m->ctx[0];
    if (out_rgba_f32) {
        if ((rc = gather_and_assemble(m, m->tile_rgba, m->gather_rgba, m->full_rgba, W, H, padded, 16))) return rc;
        MC_HIP_TRY(hipMemcpyAsync(out_rgba_f32, m->full_rgba.ptr, (size_t)W * H * 16, hipMemcpyDeviceToHost, c0->stream));
    }
    if (out_rgba8) {   // the byte tiles gathered to device 0, de-interleaved and point-reflected there (pathtracerApp.h:236-243) in one pass
        if ((rc = gather_tiles(m, m->tile_u8, m->gather_u8, (size_t)padded * W * 4))) return rc;
        if ((rc = m->full_u8.reserve((size_t)W * H * 4))) return rc;
        if ((rc = assemble_rgba8_launch(c0, m->gather_u8.ptr, W, H, (uint32_t)m->n, kRowBlock, padded, 1, m->full_u8.ptr, c0->stream))) return rc;
        MC_HIP_TRY(hipMemcpyAsync(out_rgba8, m->full_u8.ptr, (size_t)W * H * 4, hipMemcpyDeviceToHost, c0->stream));
    }
    for (int i = 0; i < m->n; i++) {
        MC_HIP_TRY(hipSetDevice(m->ctx[i]->device));
        MC_HIP_TRY(hipStreamSynchronize(m->ctx[i]->stream));
        if ((rc = m->ctx[i]->check_status())) return rc;   // a pool kernel's tripped scheduling bound: never a silent partial image
    }
    return MC_OK;
}

int mc_multi_pathtrace_render(mc_multi* m, const mc_pathtrace_params* p, const float* planes, uint32_t n_planes,
                              const float* spheres, uint32_t n_spheres, float* out_rgba_f32) {
    if (!out_rgba_f32) return MC_ERR_INVALID_ARGUMENT;
    return multi_pathtrace(m, p, planes, n_planes, spheres, n_spheres, out_rgba_f32, nullptr);
}
int mc_multi_pathtrace_render_rgba8(mc_multi* m, const mc_pathtrace_params* p, const float* planes, uint32_t n_planes,
                                    const float* spheres, uint32_t n_spheres, uint8_t* out_rgba8) {
    if (!out_rgba8) return MC_ERR_INVALID_ARGUMENT;
    return multi_pathtrace(m, p, planes, n_planes, spheres, n_spheres, nullptr, out_rgba8);
}

}  // extern "C"
