// Minimal PNG encoder for the apps' RGBA8 output (zlib deflate).  Stands where the reference calls its
// vendored third-party codec, lodepng::encode(filename, image, w, h) (mandelbrotApp.h:181,
// pathtracerApp.h:245).  The files decode to exactly the RGBA8 pixels handed in; byte-identity with
// lodepng's own deflate stream is NOT a goal (a drop-in build inside the reference tree keeps using
// the reference's lodepng).  Like lodepng's auto_convert, all-opaque images are
// stored as 8-bit RGB.
#ifndef PNGWRITER_H_
#define PNGWRITER_H_

#include <cstdint>
#include <string>
#include <vector>

// static_cast<uint8_t>(v) as the reference binary executes it on x86-64: cvttss2si (truncate to int32,
// 0x80000000 when out of range / NaN), low byte kept.
inline uint8_t x86FloatToU8(float v) {
    if (!(v > -2147483648.0f && v < 2147483648.0f)) return 0;
    return (uint8_t)((int32_t)v & 0xff);
}

namespace pngwriter {
// Returns an empty string on success, else an error description.  The image is cut into row stripes that are
// filtered and deflated by `threads` workers in parallel (0 = all hardware threads, 1 = serial) and concatenated
// into one valid zlib stream (sync-flushed raw-deflate pieces + combined Adler-32): at 7680x5120 the
// single-threaded deflate of the reference's lodepng path is the dominant end-to-end cost.
std::string encode(std::vector<uint8_t>& out, const uint8_t* rgba8, uint32_t w, uint32_t h, int threads = 0);
std::string encodeFile(const char* filename, const uint8_t* rgba8, uint32_t w, uint32_t h, int threads = 0);
}  // namespace pngwriter

#endif  // PNGWRITER_H_
