// C shim over the host-side helpers (PNG writer, x86 float->u8 cast) so the Python test-suite can call them
// through ctypes: lib/libmc_hostutil.so.  No GPU code here.
#include <cstdlib>
#include <cstring>

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
void mcu_free(void* p) { std::free(p); }

void mcu_float_to_u8(const float* in, uint8_t* out, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = x86FloatToU8(in[i]);
}

// The apps' storage-buffer -> RGBA8 conversion (+ the path tracer's 180-degree rotation), row stripes on `threads` host threads.
void mcu_convert_storage(const float* vec4, uint8_t* rgba8, uint32_t w, uint32_t h, float scale, int rotate180, int threads) {
    pngwriter::convertStorage(vec4, rgba8, w, h, scale, rotate180 != 0, threads);
}

}  // extern "C"
