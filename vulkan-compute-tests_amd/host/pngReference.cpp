// See pngReference.h.  Every routine states the codec behaviour it reproduces; the byte-for-byte check against the reference's
// own codec (built where it lies, oracle/_ref) is tests/test_host_png.py::test_reference_compatible_encoder_matches_lodepng.
#include "pngReference.h"

#include <zlib.h>   // crc32 / adler32 only: the deflate stream below is produced here, not by zlib

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>

namespace pngref {
namespace {

// ---------------------------------------------------------------------------------------------------------------- bit output
struct BitSink {
    std::vector<uint8_t> bytes;
    size_t nbits = 0;
    void bit(unsigned b) {
        if ((nbits & 7u) == 0) bytes.push_back(0);
        bytes.back() |= (uint8_t)((b & 1u) << (nbits & 7u));
        nbits++;
    }
    void lsb_first(unsigned value, unsigned n) { for (unsigned i = 0; i < n; i++) bit(value >> i); }          // plain fields
    void msb_first(unsigned code, unsigned n) { for (unsigned i = 0; i < n; i++) bit(code >> (n - 1 - i)); }   // Huffman codes
};

// ---------------------------------------------------------------------------------- length-limited prefix code (package-merge)
// Boundary package-merge with two look-ahead chains per list.  What makes the result reproducible among equally good codes:
// leaves sorted by weight with equal weights in symbol order; a package wins a tie against the next leaf (a leaf is taken only
// when the package is strictly heavier).
struct Chain { int weight; unsigned count; int tail; };   // count = number of leaves to the left of the boundary
struct Leaf { int weight; unsigned symbol; };

class PackageMerge {
public:
    PackageMerge(const std::vector<Leaf>& leaves, unsigned lists) : lv(leaves), n((unsigned)leaves.size()), c0(lists), c1(lists) {
        const int a = make(lv[0].weight, 1, -1), b = make(lv[1].weight, 2, -1);
        for (unsigned i = 0; i < lists; i++) { c0[i] = a; c1[i] = b; }
        for (unsigned run = 2; run != 2 * n - 2; run++) step((int)lists - 1, (int)run);
        last = c1[lists - 1];
    }
    // every chain element of the last list covers `count` leaves: each adds one bit to those leaves' code lengths
    void lengths(std::vector<unsigned>& out) const {
        for (int k = last; k >= 0; k = pool[k].tail)
            for (unsigned i = 0; i < pool[k].count; i++) out[lv[i].symbol]++;
    }

private:
    const std::vector<Leaf>& lv;
    unsigned n;
    std::vector<Chain> pool;
    std::vector<int> c0, c1;
    int last = -1;
    int make(int w, unsigned count, int tail) { pool.push_back(Chain{w, count, tail}); return (int)pool.size() - 1; }
    void step(int c, int run) {
        const unsigned taken = pool[c1[c]].count;
        if (c == 0) {
            if (taken >= n) return;
            c0[0] = c1[0];
            c1[0] = make(lv[taken].weight, taken + 1, -1);
            return;
        }
        const int package = pool[c0[c - 1]].weight + pool[c1[c - 1]].weight;
        c0[c] = c1[c];
        if (taken < n && package > lv[taken].weight) {
            c1[c] = make(lv[taken].weight, taken + 1, pool[c1[c]].tail);
            return;
        }
        c1[c] = make(package, taken, c1[c - 1]);
        if (run + 1 < (int)(2 * n - 2)) { step(c - 1, run); step(c - 1, run); }   // the two chains just consumed are replaced
    }
};

struct PrefixCode {
    std::vector<unsigned> len, code;
    unsigned size() const { return (unsigned)len.size(); }
};

// Code over symbols [0, count) where count = the alphabet cut after its last used symbol, but not below `keep`.
PrefixCode make_code(const std::vector<unsigned>& freq, unsigned keep, unsigned maxbits) {
    unsigned count = (unsigned)freq.size();
    while (count > keep && freq[count - 1] == 0) count--;
    PrefixCode pc;
    pc.len.assign(count, 0);
    std::vector<Leaf> leaves;
    for (unsigned s = 0; s < count; s++)
        if (freq[s]) leaves.push_back(Leaf{(int)freq[s], s});
    if (leaves.empty()) {
        pc.len[0] = pc.len[1] = 1;                       // decoders want two codes even for an unused alphabet
    } else if (leaves.size() == 1) {
        pc.len[leaves[0].symbol] = 1;
        pc.len[leaves[0].symbol == 0 ? 1 : 0] = 1;
    } else {
        std::stable_sort(leaves.begin(), leaves.end(), [](const Leaf& a, const Leaf& b) { return a.weight < b.weight; });
        PackageMerge(leaves, maxbits).lengths(pc.len);
    }
    // canonical codes (RFC 1951 §3.2.2)
    std::vector<unsigned> per_len(maxbits + 1, 0), next(maxbits + 1, 0);
    for (unsigned l : pc.len) per_len[l]++;
    per_len[0] = 0;
    for (unsigned b = 1; b <= maxbits; b++) next[b] = (next[b - 1] + per_len[b - 1]) << 1;
    pc.code.assign(count, 0);
    for (unsigned s = 0; s < count; s++)
        if (pc.len[s]) pc.code[s] = next[pc.len[s]]++;
    return pc;
}

// ------------------------------------------------------------------------------------------------------------------- LZ77
constexpr unsigned kWindow = 2048, kMaxMatch = 258, kChainLimit = kWindow / 8, kLazyLimit = 64, kNice = 128;
const unsigned kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const unsigned kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const unsigned kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const unsigned kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
unsigned base_index(const unsigned* table, unsigned n, unsigned v) {   // last entry <= v
    unsigned k = 0;
    while (k + 1 < n && table[k + 1] <= v) k++;
    return k;
}

// The matcher's memory: it lives across the blocks of one stream.  Positions are kept modulo the window.
struct Matcher {
    std::vector<int> head = std::vector<int>(65536, -1);          // hash -> newest window position
    std::vector<int> hash_at = std::vector<int>(kWindow, -1);      // window position -> hash it was entered under
    std::vector<uint16_t> prev = std::vector<uint16_t>(kWindow);   // window position -> older position of the same hash
    std::vector<int> zhead = std::vector<int>(kMaxMatch + 1, -1);  // zero-run length -> newest window position
    std::vector<uint16_t> zprev = std::vector<uint16_t>(kWindow), zrun = std::vector<uint16_t>(kWindow, 0);
    Matcher() { for (unsigned i = 0; i < kWindow; i++) prev[i] = zprev[i] = (uint16_t)i; }   // self link = no older entry
    void enter(unsigned w, unsigned h, unsigned zeros) {
        hash_at[w] = (int)h;
        if (head[h] != -1) prev[w] = (uint16_t)head[h];
        head[h] = (int)w;
        zrun[w] = (uint16_t)zeros;
        if (zhead[zeros] != -1) zprev[w] = (uint16_t)zhead[zeros];
        zhead[zeros] = (int)w;
    }
};

unsigned hash3(const uint8_t* d, size_t end, size_t pos) {
    unsigned h = 0;
    if (pos + 2 < end) h = (unsigned)d[pos] ^ ((unsigned)d[pos + 1] << 4) ^ ((unsigned)d[pos + 2] << 8);
    else for (size_t i = 0; pos + i < end; i++) h ^= (unsigned)d[pos + i] << (8 * i);
    return h & 65535u;
}
unsigned zero_run(const uint8_t* d, size_t end, size_t pos) {
    size_t stop = std::min(end, pos + kMaxMatch), p = pos;
    while (p != stop && d[p] == 0) p++;
    return (unsigned)(p - pos);
}

// Symbols of one block: a literal, or {257 + length code, length extra, distance code, distance extra}.
void lz77_block(std::vector<unsigned>& out, Matcher& m, const uint8_t* d, size_t begin, size_t end) {
    unsigned zeros = 0, pending_len = 0, pending_dist = 0;
    bool pending = false;
    auto index = [&](size_t pos) {   // enters position `pos` into both chains; returns its hash
        const unsigned h = hash3(d, end, pos);
        if (h == 0) {
            if (zeros == 0) zeros = zero_run(d, end, pos);
            else if (pos + zeros > end || d[pos + zeros - 1] != 0) zeros--;
        } else {
            zeros = 0;
        }
        m.enter((unsigned)(pos & (kWindow - 1)), h, zeros);
        return h;
    };
    for (size_t pos = begin; pos < end; pos++) {
        const unsigned w = (unsigned)(pos & (kWindow - 1));
        const unsigned h = index(pos);
        unsigned best_len = 0, best_dist = 0, links = 0, last_dist = 0;
        unsigned cand = m.prev[w];
        const size_t limit = std::min(end, pos + kMaxMatch);
        for (;;) {
            if (links++ >= kChainLimit) break;
            const unsigned dist = cand <= w ? w - cand : w - cand + kWindow;
            if (dist < last_dist) break;                 // once around the window
            last_dist = dist;
            if (dist > 0) {
                size_t f = pos, b = pos - dist;
                if (zeros >= 3) { const unsigned skip = std::min<unsigned>(m.zrun[cand], zeros); f += skip; b += skip; }
                while (f != limit && d[b] == d[f]) { f++; b++; }
                const unsigned len = (unsigned)(f - pos);
                if (len > best_len) {
                    best_len = len; best_dist = dist;
                    if (len >= kNice) break;
                }
            }
            if (cand == m.prev[cand]) break;             // no older entry
            if (zeros >= 3 && best_len > zeros) {
                cand = m.zprev[cand];
                if (m.zrun[cand] != zeros) break;
            } else {
                cand = m.prev[cand];
                if (m.hash_at[cand] != (int)h) break;    // the slot was re-used by another hash since
            }
        }
        // lazy evaluation: hold a short match back for one byte and keep it unless the next position does better by two
        if (!pending && best_len >= 3 && best_len <= kLazyLimit && best_len < kMaxMatch) {
            pending = true; pending_len = best_len; pending_dist = best_dist;
            continue;
        }
        if (pending) {
            pending = false;
            if (best_len > pending_len + 1) {
                out.push_back(d[pos - 1]);
            } else {
                best_len = pending_len; best_dist = pending_dist;
                m.head[h] = -1; m.zhead[zeros] = -1;     // this position is entered again below: forget the first entry
                pos--;
            }
        }
        if (best_len < 3 || (best_len == 3 && best_dist > 4096)) {
            out.push_back(d[pos]);
        } else {
            const unsigned lc = base_index(kLenBase, 29, best_len), dc = base_index(kDistBase, 30, best_dist);
            out.push_back(257 + lc); out.push_back(best_len - kLenBase[lc]);
            out.push_back(dc); out.push_back(best_dist - kDistBase[dc]);
            for (unsigned i = 1; i < best_len; i++) index(++pos);
        }
    }
}

// --------------------------------------------------------------------------------------------------- one dynamic-code block
const unsigned kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

void write_block(BitSink& o, Matcher& m, const uint8_t* d, size_t begin, size_t end, bool final_block) {
    std::vector<unsigned> sym;
    lz77_block(sym, m, d, begin, end);
    std::vector<unsigned> f_ll(286, 0), f_d(30, 0);
    for (size_t i = 0; i < sym.size(); i++) {
        f_ll[sym[i]]++;
        if (sym[i] > 256) { f_d[sym[i + 2]]++; i += 3; }
    }
    f_ll[256] = 1;
    const PrefixCode ll = make_code(f_ll, 257, 15), dd = make_code(f_d, 2, 15);
    // both length tables in a row, then run-length coded: 16 = repeat previous 3..6, 17 = 3..10 zeros, 18 = 11..138 zeros
    std::vector<unsigned> lens(ll.len);
    lens.insert(lens.end(), dd.len.begin(), dd.len.end());
    std::vector<unsigned> rle;
    for (size_t i = 0; i < lens.size(); i++) {
        unsigned same = 0;                               // further entries equal to entry i
        while (i + same + 1 < lens.size() && lens[i + same + 1] == lens[i]) same++;
        if (lens[i] == 0 && same >= 2) {
            unsigned run = same + 1;
            if (run <= 10) { rle.push_back(17); rle.push_back(run - 3); }
            else { run = std::min(run, 138u); rle.push_back(18); rle.push_back(run - 11); }
            i += run - 1;
        } else if (same >= 3) {
            const unsigned sixes = same / 6, rest = same % 6;
            rle.push_back(lens[i]);
            for (unsigned k = 0; k < sixes; k++) { rle.push_back(16); rle.push_back(3); }
            if (rest >= 3) { rle.push_back(16); rle.push_back(rest - 3); }
            else same -= rest;                           // the last one or two are written out on their own
            i += same;
        } else {
            rle.push_back(lens[i]);
        }
    }
    std::vector<unsigned> f_cl(19, 0);
    for (size_t i = 0; i < rle.size(); i++) { f_cl[rle[i]]++; if (rle[i] >= 16) i++; }
    const PrefixCode cl = make_code(f_cl, 19, 7);
    std::vector<unsigned> cl_lens(19);
    for (unsigned i = 0; i < 19; i++) cl_lens[i] = cl.len[kClOrder[i]];
    unsigned n_cl = 19;
    while (n_cl > 4 && cl_lens[n_cl - 1] == 0) n_cl--;

    o.bit(final_block ? 1u : 0u); o.bit(0); o.bit(1);    // BTYPE = 2
    o.lsb_first(ll.size() - 257, 5); o.lsb_first(dd.size() - 1, 5); o.lsb_first(n_cl - 4, 4);
    for (unsigned i = 0; i < n_cl; i++) o.lsb_first(cl_lens[i], 3);
    for (size_t i = 0; i < rle.size(); i++) {
        const unsigned s = rle[i];
        o.msb_first(cl.code[s], cl.len[s]);
        if (s == 16) o.lsb_first(rle[++i], 2);
        else if (s == 17) o.lsb_first(rle[++i], 3);
        else if (s == 18) o.lsb_first(rle[++i], 7);
    }
    for (size_t i = 0; i < sym.size(); i++) {
        const unsigned s = sym[i];
        o.msb_first(ll.code[s], ll.len[s]);
        if (s > 256) {
            o.lsb_first(sym[i + 1], kLenExtra[s - 257]);
            const unsigned dc = sym[i + 2];
            o.msb_first(dd.code[dc], dd.len[dc]);
            o.lsb_first(sym[i + 3], kDistExtra[dc]);
            i += 3;
        }
    }
    o.msb_first(ll.code[256], ll.len[256]);
}

std::vector<uint8_t> zlib_stream(const std::vector<uint8_t>& raw) {
    BitSink o;
    o.bytes.push_back(0x78); o.bytes.push_back(0x01); o.nbits = 16;   // CM 8, CINFO 7, FLEVEL 0, FCHECK
    size_t block = raw.size() / 8 + 8;
    block = std::max<size_t>(65536, std::min<size_t>(262144, block));
    size_t blocks = (raw.size() + block - 1) / block;
    if (blocks == 0) blocks = 1;
    Matcher m;
    for (size_t b = 0; b < blocks; b++) {
        const size_t begin = b * block, end = std::min(raw.size(), begin + block);
        write_block(o, m, raw.data(), begin, end, b + 1 == blocks);
    }
    uLong ad = adler32(0L, Z_NULL, 0);
    for (size_t pos = 0; pos < raw.size();) {
        const uInt n = (uInt)std::min<size_t>(raw.size() - pos, 1u << 30);
        ad = adler32(ad, raw.data() + pos, n);
        pos += n;
    }
    for (int s = 24; s >= 0; s -= 8) o.bytes.push_back((uint8_t)(ad >> s));
    return std::move(o.bytes);
}

// ------------------------------------------------------------------------------------------------- colour model and filtering
unsigned grey_bits(uint8_t v) {   // depth at which an 8-bit grey value is representable (scaling by 255, 85, 17)
    if (v == 0 || v == 255) return 1;
    if (v % 17 == 0) return v % 85 == 0 ? 2 : 4;
    return 8;
}

void put32(std::vector<uint8_t>& v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s)); }
void chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data) {
    put32(out, (uint32_t)data.size());
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    uLong c = crc32(0L, Z_NULL, 0);
    for (size_t pos = start; pos < out.size();) {
        const uInt n = (uInt)std::min<size_t>(out.size() - pos, 1u << 30);
        c = crc32(c, out.data() + pos, n);
        pos += n;
    }
    put32(out, (uint32_t)c);
}

int paeth(int a, int b, int c) {
    const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - c - c);
    if (pc < pa && pc < pb) return c;
    return pb < pa ? b : a;
}

}  // namespace

std::string encode(std::vector<uint8_t>& out, const uint8_t* rgba8, uint32_t w, uint32_t h) {
    if (!rgba8 || !w || !h) return "empty image";
    const size_t npix = (size_t)w * h;
    // ---- what the image needs: colours (up to 257, in order of first appearance), colour vs grey, grey depth
    bool coloured = false;
    unsigned gbits = 1;
    std::map<uint32_t, unsigned> index_of;
    std::vector<uint32_t> palette;
    bool counting = true;
    for (size_t i = 0; i < npix; i++) {
        const uint8_t r = rgba8[4 * i], g = rgba8[4 * i + 1], b = rgba8[4 * i + 2], a = rgba8[4 * i + 3];
        if (a != 255) return "alpha";
        if (!coloured) {
            if (r != g || r != b) { coloured = true; gbits = 8; }
            else gbits = std::max(gbits, grey_bits(r));
        }
        if (counting) {
            const uint32_t key = (uint32_t)r | ((uint32_t)g << 8) | ((uint32_t)b << 16);
            if (index_of.emplace(key, (unsigned)palette.size()).second) {
                palette.push_back(key);
                if (palette.size() >= 257) counting = false;
            }
        }
        if (coloured && !counting) break;               // (alpha is still checked below)
    }
    for (size_t i = 0; i < npix; i++) if (rgba8[4 * i + 3] != 255) return "alpha";
    const unsigned n = (unsigned)palette.size();
    const unsigned pbits = n <= 2 ? 1 : (n <= 4 ? 2 : (n <= 16 ? 4 : 8));
    bool use_palette = n <= 256 && npix >= 2 * (size_t)n;
    if (!coloured && gbits <= pbits) use_palette = false;           // grey costs no PLTE chunk
    const unsigned colour_type = use_palette ? 3u : (coloured ? 2u : 0u);
    const unsigned depth = use_palette ? pbits : (coloured ? 8u : gbits);
    const unsigned bpp = colour_type == 2 ? 24u : depth;             // bits per pixel
    const size_t line = ((size_t)w * bpp + 7) / 8, step = (bpp + 7) / 8;
    // ---- pixels in the file's colour model (samples below 8 bits packed from the most significant bit)
    std::vector<uint8_t> img(line * h, 0);
    for (uint32_t y = 0; y < h; y++) {
        uint8_t* row = img.data() + (size_t)y * line;
        for (uint32_t x = 0; x < w; x++) {
            const uint8_t* p = rgba8 + 4 * ((size_t)y * w + x);
            if (colour_type == 2) { row[3 * x] = p[0]; row[3 * x + 1] = p[1]; row[3 * x + 2] = p[2]; continue; }
            unsigned v;
            if (colour_type == 3) v = index_of[(uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16)];
            else v = depth == 8 ? p[0] : ((unsigned)p[0] >> (8 - depth)) & ((1u << depth) - 1u);
            if (depth == 8) row[x] = (uint8_t)v;
            else {
                const size_t bitpos = (size_t)x * depth;
                row[bitpos >> 3] |= (uint8_t)(v << (8 - depth - (bitpos & 7)));
            }
        }
    }
    // ---- filter
    std::vector<uint8_t> raw((line + 1) * h);
    const bool adaptive = colour_type != 3 && depth >= 8;
    std::vector<uint8_t> trial[5];
    for (auto& t : trial) t.resize(line);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* cur = img.data() + (size_t)y * line;
        const uint8_t* up = y ? cur - line : nullptr;
        uint8_t* dst = raw.data() + (size_t)y * (line + 1);
        if (!adaptive) { dst[0] = 0; std::memcpy(dst + 1, cur, line); continue; }
        size_t best_sum = 0;
        unsigned best = 0;
        for (unsigned f = 0; f < 5; f++) {
            uint8_t* t = trial[f].data();
            for (size_t i = 0; i < line; i++) {
                const int a = i >= step ? cur[i - step] : 0, b = up ? up[i] : 0, c = (up && i >= step) ? up[i - step] : 0;
                int pred = 0;
                if (f == 1) pred = a;
                else if (f == 2) pred = b;
                else if (f == 3) pred = (a + b) >> 1;
                else if (f == 4) pred = paeth(a, b, c);
                t[i] = (uint8_t)(cur[i] - pred);
            }
            size_t sum = 0;
            if (f == 0) for (size_t i = 0; i < line; i++) sum += t[i];
            else for (size_t i = 0; i < line; i++) sum += t[i] < 128 ? t[i] : 255u - t[i];
            if (f == 0 || sum < best_sum) { best_sum = sum; best = f; }
        }
        dst[0] = (uint8_t)best;
        std::memcpy(dst + 1, trial[best].data(), line);
    }
    // ---- the file
    const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    out.assign(sig, sig + 8);
    std::vector<uint8_t> ihdr;
    put32(ihdr, w); put32(ihdr, h);
    ihdr.push_back((uint8_t)depth); ihdr.push_back((uint8_t)colour_type); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    if (colour_type == 3) {
        std::vector<uint8_t> plte;
        for (uint32_t c : palette) { plte.push_back((uint8_t)c); plte.push_back((uint8_t)(c >> 8)); plte.push_back((uint8_t)(c >> 16)); }
        chunk(out, "PLTE", plte);
    }
    chunk(out, "IDAT", zlib_stream(raw));
    chunk(out, "IEND", {});
    return "";
}

std::string encodeFile(const char* filename, const uint8_t* rgba8, uint32_t w, uint32_t h) {
    std::vector<uint8_t> png;
    std::string err = encode(png, rgba8, w, h);
    if (!err.empty()) return err;
    FILE* f = std::fopen(filename, "wb");
    if (!f) return std::string("cannot open ") + filename;
    const bool ok = std::fwrite(png.data(), 1, png.size(), f) == png.size();
    std::fclose(f);
    return ok ? "" : "short write";
}

}  // namespace pngref
