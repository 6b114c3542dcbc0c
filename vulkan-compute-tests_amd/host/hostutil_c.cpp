// C shim over the host-side helpers (PNG writer, x86 float->u8 cast) so the Python test-suite can call them
// through ctypes: lib/libmc_hostutil.so.  No GPU code here.
#include <cstdlib>
#include <cstring>

#include <sys/mman.h>
#include <unistd.h>

#include "pngWriter.h"

extern "C" {

// Encodes RGBA8 into a malloc'ed PNG with `threads` deflate workers (0 = all cores); caller frees with mcu_free.
int mcu_png_encode_mt(const uint8_t* rgba8, uint32_t w, uint32_t h, int threads, uint8_t** out, size_t* out_len) {
    std::vector<uint8_t> png;
    if (!pngwriter::encode(png, rgba8, w, h, threads).empty()) return 1;
    *out = (uint8_t*)std::malloc(png.size());
    if (!*out) return 2;
    std::memcpy(*out, png.data(), png.size());
    *out_len = png.size();
    return 0;
}
int mcu_png_encode(const uint8_t* rgba8, uint32_t w, uint32_t h, uint8_t** out, size_t* out_len) {
    return mcu_png_encode_mt(rgba8, w, h, 0, out, out_len);
}
// The storage buffer straight to a PNG (conversion fused into the stripe workers): same bytes as mcu_convert_storage + mcu_png_encode_mt.
int mcu_png_encode_storage(const float* vec4, uint32_t w, uint32_t h, float scale, int rotate180, int threads, uint8_t** out, size_t* out_len) {
    std::vector<uint8_t> png;
    if (!pngwriter::encodeStorage(png, vec4, w, h, scale, rotate180 != 0, threads).empty()) return 1;
    *out = (uint8_t*)std::malloc(png.size());
    if (!*out) return 2;
    std::memcpy(*out, png.data(), png.size());
    *out_len = png.size();
    return 0;
}
// The progressive encoder (pngwriter::Progressive) fed in `bands` steps from a source that is FILLED band by band: the destination the
// workers read starts as garbage and each band is copied in just before it is declared ready — a worker that read a row too early
// would encode the garbage.  rgba8_route = 0: the fp32 storage buffer; 1: an RGBA8 image with alpha 255.  Must give the one-shot file.
int mcu_png_encode_progressive(const void* image, uint32_t w, uint32_t h, float scale, int rgba8_route, int threads, int bands, uint8_t** out,
                               size_t* out_len) {
    const size_t row = (size_t)w * (rgba8_route ? 4 : 16);
    std::vector<uint8_t> live(row * h, 0xA5);
    pngwriter::Progressive enc;
    if (rgba8_route) enc.beginOpaqueRgba8(live.data(), w, h, threads);
    else enc.beginStorage(reinterpret_cast<const float*>(live.data()), w, h, scale, threads);
    if (bands < 1) bands = 1;
    for (int b = 0; b < bands; b++) {
        const uint32_t y0 = (uint32_t)((uint64_t)h * b / bands), y1 = (uint32_t)((uint64_t)h * (b + 1) / bands);
        std::memcpy(live.data() + row * y0, static_cast<const uint8_t*>(image) + row * y0, row * (y1 - y0));
        enc.rowsReady(y1);
    }
    std::vector<uint8_t> png;
    if (!enc.finish(png).empty()) return 1;
    *out = (uint8_t*)std::malloc(png.size());
    if (!*out) return 2;
    std::memcpy(*out, png.data(), png.size());
    *out_len = png.size();
    return 0;
}
// An image that is never finished (the render that was to fill it failed): the encoder is begun on a mapping whose rows from
// `ready_rows` on are INACCESSIBLE (PROT_NONE from the next page boundary), hears of the first `ready_rows` rows and is then destroyed.
// A worker that read a row it was never promised — the old destructor declared every row ready and let the workers run out — faults.
// Returns 0 when the object was torn down without such a read, and the number of stripes' worth of time is bounded (no worker waits).
int mcu_png_progressive_abandon(uint32_t w, uint32_t h, int rgba8_route, int threads, uint32_t ready_rows) {
    const size_t row = (size_t)w * (rgba8_route ? 4 : 16), page = (size_t)sysconf(_SC_PAGESIZE);
    const size_t bytes = ((row * h + page - 1) / page + 1) * page;
    void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) return 2;
    std::memset(m, 0x3c, bytes);
    if (ready_rows > h) ready_rows = h;
    const size_t first_dead = ((row * ready_rows + page - 1) / page) * page;
    if (first_dead < bytes && mprotect(static_cast<char*>(m) + first_dead, bytes - first_dead, PROT_NONE) != 0) { munmap(m, bytes); return 3; }
    {
        pngwriter::Progressive enc;
        if (rgba8_route) enc.beginOpaqueRgba8(static_cast<const uint8_t*>(m), w, h, threads);
        else enc.beginStorage(static_cast<const float*>(m), w, h, 1.0f, threads);
        enc.rowsReady(ready_rows);
    }   // ~Progressive: abandon()
    {
        pngwriter::Progressive enc;   // ... and abandon() by name, after which the object is idle again
        enc.beginOpaqueRgba8(static_cast<const uint8_t*>(m), w, h, threads);
        enc.abandon();
        if (enc.active()) { munmap(m, bytes); return 4; }
        // ... and a begin on top of an image in progress (run() twice, no save between) gives the first one up the same way
        enc.beginOpaqueRgba8(static_cast<const uint8_t*>(m), w, h, threads);
        enc.rowsReady(ready_rows);
        enc.beginOpaqueRgba8(static_cast<const uint8_t*>(m), w, h, threads);
        if (!enc.active()) { munmap(m, bytes); return 5; }
        enc.abandon();
    }
    munmap(m, bytes);
    return 0;
}
void mcu_free(void* p) { std::free(p); }

void mcu_float_to_u8(const float* in, uint8_t* out, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = x86FloatToU8(in[i]);
}

// The apps' storage-buffer -> RGBA8 conversion (+ the path tracer's 180-degree rotation), row stripes on `threads` host threads.
void mcu_convert_storage(const float* vec4, uint8_t* rgba8, uint32_t w, uint32_t h, float scale, int rotate180, int threads) {
    pngwriter::convertStorage(vec4, rgba8, w, h, scale, rotate180 != 0, threads);
}

}  // extern "C"
