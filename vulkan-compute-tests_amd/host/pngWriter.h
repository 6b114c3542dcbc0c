// Minimal PNG encoder for the apps' RGBA8 output (zlib deflate).  Stands where the reference calls its
// vendored third-party codec, lodepng::encode(filename, image, w, h) (mandelbrotApp.h:181,
// pathtracerApp.h:245).  The files decode to exactly the RGBA8 pixels handed in; byte-identity with
// lodepng's own deflate stream is NOT a goal (a drop-in build inside the reference tree keeps using
// the reference's lodepng).  Like lodepng's auto_convert, all-opaque images are
// stored as 8-bit RGB.
#ifndef PNGWRITER_H_
#define PNGWRITER_H_

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

// static_cast<uint8_t>(v) as the reference binary executes it on x86-64: cvttss2si (truncate to int32,
// 0x80000000 when out of range / NaN), low byte kept.
inline uint8_t x86FloatToU8(float v) {
    if (!(v > -2147483648.0f && v < 2147483648.0f)) return 0;
    return (uint8_t)((int32_t)v & 0xff);
}

namespace pngwriter {
// Worker threads `threads = 0` stands for: min(hardware, affinity mask, cgroup CPU quota) — not the 256 a 16-CPU container reports.
int usableThreads();
// getRenderedImage's loop (mandelbrotApp.h:159-166, pathtracerApp.h:212-219) over row stripes on `threads` host threads (0 = all):
// out[4 i + c] = static_cast<uint8_t>(scale * vec4[i][c]) with the reference binary's x86-64 semantics (x86FloatToU8), alpha 255.
// rotate180 = the path tracer's swap loop (pathtracerApp.h:236-243) applied while writing: every pixel changes places with its point
// reflection, except — for an odd width — the middle column, which that loop (`x < resx / 2`) never touches.  Same bytes as the
// reference's serial loops.  `rgba8` receives w * h * 4 bytes.
void convertStorage(const float* vec4, uint8_t* rgba8, uint32_t w, uint32_t h, float scale, bool rotate180, int threads = 0);

// Returns an empty string on success, else an error description.  The image is cut into row stripes that are
// filtered and deflated by `threads` workers in parallel (0 = all hardware threads, 1 = serial) and concatenated
// into one valid zlib stream (sync-flushed raw-deflate pieces + combined Adler-32): at 7680x5120 the
// single-threaded deflate of the reference's lodepng path is the dominant end-to-end cost.
std::string encode(std::vector<uint8_t>& out, const uint8_t* rgba8, uint32_t w, uint32_t h, int threads = 0);
std::string encodeFile(const char* filename, const uint8_t* rgba8, uint32_t w, uint32_t h, int threads = 0);
// The fp32 vec4 storage buffer straight to a PNG (round 6): every stripe worker converts the rows it filters — convertStorage's cast and
// point reflection, row by row — so the RGBA8 intermediate (K4: 157 MB written and read back) does not exist.  The file is byte for byte
// the one convertStorage + encode produce with the same thread count.
std::string encodeStorage(std::vector<uint8_t>& out, const float* vec4, uint32_t w, uint32_t h, float scale, bool rotate180, int threads = 0);
std::string encodeStorageFile(const char* filename, const float* vec4, uint32_t w, uint32_t h, float scale, bool rotate180, int threads = 0);

// The same encoder fed while the image is still being PRODUCED, top rows first (round 6: the Mandelbrot app renders its image in row
// bands and the stripe workers filter and deflate band k while the device renders band k + 1 — at K4 the 50 ms of PNG work were a quarter
// of the process).  begin*() starts the workers, which wait; rowsReady(y) declares the source's rows [0, y) final; finish() waits for the
// last stripe and returns the file.  Same stripes, same bytes as encodeStorage / encode with the same thread count.  The source buffer
// must stay alive until finish() or abandon() has returned (an object destroyed with an image in progress abandons it: rows that were never
// declared ready are never read).  No point reflection here (the path tracer's first output rows are its storage buffer's LAST ones).
class Progressive {
public:
    Progressive();
    ~Progressive();
    Progressive(const Progressive&) = delete;
    Progressive& operator=(const Progressive&) = delete;
    void beginStorage(const float* vec4, uint32_t w, uint32_t h, float scale, int threads = 0);
    void beginOpaqueRgba8(const uint8_t* rgba8, uint32_t w, uint32_t h, int threads = 0);   // alpha is 255 by construction: stored as RGB
    void rowsReady(uint32_t upTo);
    bool active() const;   // an image is in progress (begun, neither finished nor abandoned)
    std::string finish(std::vector<uint8_t>& png);
    double lastJoinMs() const { return joinMs; }          // finish(): waiting for the last stripes ...
    double lastAssembleMs() const { return assembleMs; }  // ... and putting the file image together
    void abandon();   // the image will never be complete: stops the workers without another read of the source (the destructor does this)
private:
    struct Impl;
    std::unique_ptr<Impl> impl;
    double joinMs = 0.0, assembleMs = 0.0;
};
std::string writeFile(const char* filename, const std::vector<uint8_t>& png);
}  // namespace pngwriter

#endif  // PNGWRITER_H_
