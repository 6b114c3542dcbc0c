// PNG encoder whose OUTPUT BYTES equal those of the reference's codec for the images the apps write.
//
// The reference saves its images with its vendored third-party codec at default settings —
// lodepng::encode(filename, image, w, h), src/mandelbrotApp.h:172-184, src/pathtracerApp.h:245 (lodepng 20161127) — and
// north_star asks for a bit-identical PNG.  Identical RGBA8 pixels determine that file only through the codec's choices, so
// this encoder makes the same choices, written from the codec's documented behaviour (none of its source is in this repository;
// the build container compares the two byte for byte through oracle/_ref, tests/test_host_png.py):
//   * colour reduction of an opaque 8-bit RGBA image: greyscale at 1 / 2 / 4 / 8 bits when every pixel has r = g = b, a palette
//     at 1 / 2 / 4 / 8 bits (colours in order of first appearance) when there are at most 256 colours and the image has at
//     least twice as many pixels, else 8-bit RGB;
//   * filtering: none for palette / sub-byte images, else per row the filter (0..4, first wins) with the smallest sum of
//     |residual| — residual bytes above 127 counted as 255 - b, filter 0 by its plain byte sum;
//   * deflate: 2048-byte window, hash chains of at most 256 links over a 3-byte shift-xor hash with a second chain over runs of
//     zeros, lazy matching up to length 64, matches of length 3 refused beyond distance 4096, "nice" length 128; blocks of
//     max(65536, min(262144, n / 8 + 8)) bytes, each with dynamic Huffman codes from a boundary package-merge (limits 15 / 15 / 7,
//     stable order of equal weights), code lengths run-length coded the codec's way; zlib header 78 01; one IDAT chunk.
// Images with any alpha below 255 are outside this contract (the apps never produce them): encode() reports it and the caller
// falls back to pngwriter::encode.  Single-threaded by construction (the LZ77 state runs through the whole stream); the parallel
// writer of pngWriter.h remains available (apps: --fast-png).
#ifndef PNGREFERENCE_H_
#define PNGREFERENCE_H_

#include <cstdint>
#include <string>
#include <vector>

namespace pngref {
// Returns an empty string on success, else an error description ("alpha": the image is not opaque).
std::string encode(std::vector<uint8_t>& out, const uint8_t* rgba8, uint32_t w, uint32_t h);
std::string encodeFile(const char* filename, const uint8_t* rgba8, uint32_t w, uint32_t h);
}  // namespace pngref

#endif  // PNGREFERENCE_H_
