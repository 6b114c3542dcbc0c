#include "pngWriter.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace pngwriter {

namespace {

void put32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}

void chunk(std::vector<uint8_t>& out, const char type[4], const uint8_t* data, size_t len) {
    put32(out, (uint32_t)len);
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    if (len) out.insert(out.end(), data, data + len);
    uint32_t c = (uint32_t)crc32(0L, out.data() + start, (uInt)(len + 4));
    put32(out, c);
}

inline int paeth(int a, int b, int c) {
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

}  // namespace

std::string encode(std::vector<uint8_t>& out, const uint8_t* rgba8, uint32_t w, uint32_t h) {
    if (!rgba8 || !w || !h) return "empty image";
    bool opaque = true;
    for (size_t i = 0; i < (size_t)w * h && opaque; i++) opaque = rgba8[4 * i + 3] == 255;
    const int bpp = opaque ? 3 : 4;
    const size_t stride = (size_t)w * bpp;
    // filtered scanlines: per row pick the filter with the smallest sum of absolute residuals
    std::vector<uint8_t> raw((stride + 1) * h);
    std::vector<uint8_t> cur(stride), prev(stride, 0), cand(stride), best(stride);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* src = rgba8 + (size_t)y * w * 4;
        for (uint32_t x = 0; x < w; x++) std::memcpy(&cur[(size_t)x * bpp], src + 4 * (size_t)x, bpp);
        uint64_t best_sum = ~0ull;
        int best_f = 0;
        for (int f = 0; f < 5; f++) {
            uint64_t sum = 0;
            for (size_t i = 0; i < stride; i++) {
                int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
                int pred = f == 0 ? 0 : f == 1 ? a : f == 2 ? b : f == 3 ? ((a + b) >> 1) : paeth(a, b, c);
                uint8_t r = (uint8_t)(cur[i] - pred);
                cand[i] = r;
                sum += r < 128 ? r : 256 - r;
            }
            if (sum < best_sum) { best_sum = sum; best_f = f; best.swap(cand); }
        }
        uint8_t* dst = &raw[(stride + 1) * y];
        dst[0] = (uint8_t)best_f;
        std::memcpy(dst + 1, best.data(), stride);
        prev = cur;
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return "zlib compress2 failed";
    out.clear();
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    out.insert(out.end(), sig, sig + 8);
    std::vector<uint8_t> ihdr;
    put32(ihdr, w); put32(ihdr, h);
    ihdr.push_back(8); ihdr.push_back(opaque ? 2 : 6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr.data(), ihdr.size());
    chunk(out, "IDAT", z.data(), zlen);
    chunk(out, "IEND", nullptr, 0);
    return "";
}

std::string encodeFile(const char* filename, const uint8_t* rgba8, uint32_t w, uint32_t h) {
    std::vector<uint8_t> png;
    std::string err = encode(png, rgba8, w, h);
    if (!err.empty()) return err;
    FILE* f = std::fopen(filename, "wb");
    if (!f) return std::string("cannot open ") + filename;
    size_t n = std::fwrite(png.data(), 1, png.size(), f);
    std::fclose(f);
    return n == png.size() ? "" : "short write";
}

}  // namespace pngwriter
