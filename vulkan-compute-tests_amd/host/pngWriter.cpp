#include "pngWriter.h"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sched.h>

#include <condition_variable>
#include <mutex>
#include <thread>

namespace pngwriter {

namespace {

void put32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}

void chunk_header_and_crc(std::vector<uint8_t>& out, const char type[4], const uint8_t* data, size_t len) {
    put32(out, (uint32_t)len);
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    if (len) out.insert(out.end(), data, data + len);
    uLong c = crc32(0L, out.data() + start, 4);
    // crc32 takes a 32-bit length: feed large IDAT payloads in pieces
    size_t pos = start + 4, left = len;
    while (left) {
        uInt n = (uInt)std::min<size_t>(left, 1u << 30);
        c = crc32(c, out.data() + pos, n);
        pos += n; left -= n;
    }
    put32(out, (uint32_t)c);
}

// Filters rows [y0,y1) of the RGBA8 image into `raw` ((stride+1) bytes per row): per row the PNG filter with the
// smallest sum of absolute residuals (the heuristic of the PNG specification, 12.8).  Row y needs row y-1 of the SOURCE image only,
// so stripes are independent.  One tight loop per filter type over byte arrays (the compiler vectorises None / Sub / Up / Average;
// Paeth's three-way choice is written branch-free); the sums are taken first, then ONLY the winning filter's residuals are stored:
// K1's 7.68 Mpx image took 0.46 s of CPU with the one-loop-for-all form of rounds 2-5 (a `switch` per byte, five stored candidates),
// three quarters of it here and not in deflate.
namespace {
inline uint32_t absres(uint8_t r) { return r < 128 ? r : 256u - r; }

// a = left, b = up, c = up-left; `cur` / `prev` point at the row's first byte, bytes left of the row count as 0.
template <int F> inline uint8_t predict(const uint8_t* cur, const uint8_t* prev, size_t i, int bpp) {
    const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
    if (F == 0) return 0;
    if (F == 1) return (uint8_t)a;
    if (F == 2) return (uint8_t)b;
    if (F == 3) return (uint8_t)((a + b) >> 1);
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (uint8_t)((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
}
template <int F> uint64_t filter_sum(const uint8_t* cur, const uint8_t* prev, size_t stride, int bpp) {
    uint64_t sum = 0;
    for (size_t i = 0; i < (size_t)bpp && i < stride; i++) sum += absres((uint8_t)(cur[i] - predict<F>(cur, prev, i, bpp)));
    uint32_t acc = 0;   // (a row of 7680 RGB pixels sums to < 2^32 / 128 ... keep 64-bit blocks of 2^16 bytes to be safe)
    for (size_t i0 = (size_t)bpp; i0 < stride; i0 += 65536) {
        const size_t i1 = std::min(stride, i0 + 65536);
        acc = 0;
        for (size_t i = i0; i < i1; i++) {
            const int a = cur[i - bpp], b = prev[i], c = prev[i - bpp];
            int pred;
            if (F == 0) pred = 0;
            else if (F == 1) pred = a;
            else if (F == 2) pred = b;
            else if (F == 3) pred = (a + b) >> 1;
            else {
                const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            }
            acc += absres((uint8_t)(cur[i] - pred));
        }
        sum += acc;
    }
    return sum;
}
template <int F> void filter_apply(const uint8_t* cur, const uint8_t* prev, size_t stride, int bpp, uint8_t* dst) {
    for (size_t i = 0; i < stride; i++) dst[i] = (uint8_t)(cur[i] - predict<F>(cur, prev, i, bpp));
}
}  // namespace

// Where a stripe's rows come from: an RGBA8 image (bpp 3 or 4 bytes kept per pixel), or — the apps' storage-buffer route since round 6 —
// the fp32 vec4 storage buffer itself, converted row by row INSIDE the stripe worker (getRenderedImage's cast, mandelbrotApp.h:159-166 /
// pathtracerApp.h:212-219, + the point reflection of :236-243): the 4 B/pixel intermediate image is never written or read back.
struct RowSource {
    const uint8_t* rgba8 = nullptr;   // either this ...
    const float* vec4 = nullptr;      // ... or the storage buffer with
    float scale = 1.0f;
    bool rotate180 = false;
    uint32_t w = 0, h = 0;
    int bpp = 3;
    void load(uint32_t y, uint8_t* dst) const {
        if (rgba8) {
            const uint8_t* src = rgba8 + (size_t)y * w * 4;
            if (bpp == 4) std::memcpy(dst, src, (size_t)w * 4);
            else for (uint32_t x = 0; x < w; x++) std::memcpy(dst + (size_t)x * 3, src + 4 * (size_t)x, 3);
            return;
        }
        // output pixel (x, y) is storage pixel (w-1-x, h-1-y) — except an odd width's middle column, which the reference's swap loop
        // (x < resx / 2) leaves where it was: convertStorage's rule, read from the output's side
        const uint32_t mid = (w & 1u) ? w / 2 : w;   // (w: no such column)
        const float* same = vec4 + (size_t)y * w * 4;
        const float* refl = vec4 + (size_t)(h - 1 - y) * w * 4;
        for (uint32_t x = 0; x < w; x++) {
            const float* s = (rotate180 && x != mid) ? refl + 4 * (size_t)(w - 1 - x) : same + 4 * (size_t)x;
            dst[3 * (size_t)x] = x86FloatToU8(scale * s[0]);
            dst[3 * (size_t)x + 1] = x86FloatToU8(scale * s[1]);
            dst[3 * (size_t)x + 2] = x86FloatToU8(scale * s[2]);
        }
    }
};

void filter_rows(const RowSource& src, uint32_t y0, uint32_t y1, uint8_t* raw) {
    const uint32_t w = src.w;
    const int bpp = src.bpp;
    const size_t stride = (size_t)w * bpp;
    std::vector<uint8_t> cur(stride), prev(stride, 0);
    auto load = [&](uint32_t y, std::vector<uint8_t>& dst) { src.load(y, dst.data()); };
    if (y0 > 0) load(y0 - 1, prev);
    for (uint32_t y = y0; y < y1; y++) {
        load(y, cur);
        const uint8_t *c = cur.data(), *p = prev.data();
        const uint64_t sums[5] = {filter_sum<0>(c, p, stride, bpp), filter_sum<1>(c, p, stride, bpp), filter_sum<2>(c, p, stride, bpp),
                                  filter_sum<3>(c, p, stride, bpp), filter_sum<4>(c, p, stride, bpp)};
        int best_f = 0;
        for (int f = 1; f < 5; f++)
            if (sums[f] < sums[best_f]) best_f = f;   // (ties keep the lower filter number, as the one-loop form did)
        uint8_t* dst = raw + (stride + 1) * (size_t)(y - y0);
        dst[0] = (uint8_t)best_f;
        switch (best_f) {
            case 0: filter_apply<0>(c, p, stride, bpp, dst + 1); break;
            case 1: filter_apply<1>(c, p, stride, bpp, dst + 1); break;
            case 2: filter_apply<2>(c, p, stride, bpp, dst + 1); break;
            case 3: filter_apply<3>(c, p, stride, bpp, dst + 1); break;
            default: filter_apply<4>(c, p, stride, bpp, dst + 1); break;
        }
        prev.swap(cur);
    }
}

// Bytes that are NOT zero-filled on allocation: a std::vector sized to deflateBound() memsets — and thereby faults in — as many bytes as
// the raw image has, of which deflate then writes a twentieth; the filtered rows are overwritten in full anyway.
struct Bytes {
    std::unique_ptr<uint8_t[]> p;
    size_t n = 0;
    void allocate(size_t bytes) { p.reset(new uint8_t[bytes]); n = bytes; }
    uint8_t* data() { return p.get(); }
    const uint8_t* data() const { return p.get(); }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    const uint8_t* begin() const { return p.get(); }
    const uint8_t* end() const { return p.get() + n; }
};

struct Stripe {
    uint32_t y0 = 0, y1 = 0;
    Bytes z;   // raw deflate bytes of this stripe (sync-flushed, or finished for the last one); z.n = the bytes deflate wrote
    uLong adler = 1;
    uLong crc = 0;            // CRC-32 of z: the IDAT chunk's CRC is combined from these (crc32_combine), not recomputed over the stream
    size_t raw_len = 0;
    bool ok = false;
};

// One stripe: filter + raw deflate.  Non-final stripes end on a byte boundary with Z_SYNC_FLUSH (an empty stored
// block), so the concatenation of all stripes is ONE valid deflate stream (the pigz construction).
// `scratch` is the calling worker's buffer for the filtered rows, reused from stripe to stripe: a fresh one per stripe faulted in as many
// pages as the raw image has — in a cold process (the apps are one) every one of them for the first time.
void compress_stripe(const RowSource& src, bool last, Stripe& s, Bytes& scratch) {
    const size_t stride = (size_t)src.w * src.bpp;
    const size_t raw_bytes = (stride + 1) * (size_t)(s.y1 - s.y0);
    if (scratch.size() < raw_bytes) scratch.allocate(raw_bytes);   // (every byte used is written by filter_rows)
    struct View { uint8_t* p; size_t n; uint8_t* data() const { return p; } size_t size() const { return n; } } raw{scratch.data(), raw_bytes};
    filter_rows(src, s.y0, s.y1, raw.data());
    s.raw_len = raw.size();
    uLong ad = adler32(0L, Z_NULL, 0);
    for (size_t pos = 0; pos < raw.size();) {
        uInt n = (uInt)std::min<size_t>(raw.size() - pos, 1u << 30);
        ad = adler32(ad, raw.data() + pos, n);
        pos += n;
    }
    s.adler = ad;
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    // Z_RLE: matches at distance 1 only.  On FILTERED image rows that is where the redundancy is (runs of equal residuals): measured on
    // K1's 3200 x 2400 image and on a 900 x 600 path trace against the default strategy at level 6 — 159 against 228 ms and 23 against
    // 66 ms of CPU, and the files are SMALLER (909 871 / 915 268 bytes, 1 233 340 / 1 251 549): the hash-chain search finds nothing better.
    if (deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_RLE) != Z_OK) return;
    s.z.allocate(deflateBound(&zs, (uLong)raw.size()) + 16);
    zs.next_in = raw.data(); zs.avail_in = (uInt)raw.size();
    zs.next_out = s.z.data(); zs.avail_out = (uInt)s.z.size();
    int rc = deflate(&zs, last ? Z_FINISH : Z_SYNC_FLUSH);
    s.ok = last ? rc == Z_STREAM_END : (rc == Z_OK && zs.avail_in == 0);
    s.z.n = s.z.size() - zs.avail_out;   // (the allocation stays as large as it was; only these bytes were touched)
    deflateEnd(&zs);
    uLong c = crc32(0L, Z_NULL, 0);
    for (size_t pos = 0; pos < s.z.size();) {
        uInt n = (uInt)std::min<size_t>(s.z.size() - pos, 1u << 30);
        c = crc32(c, s.z.data() + pos, n);
        pos += n;
    }
    s.crc = c;
}

}  // namespace

// CPUs this process may actually run on: the smallest of the hardware count, the affinity mask and the cgroup CPU quota.  A one-GPU box
// of the pool shows 256 CPUs and grants 16: 64 deflate workers or 256 conversion threads there are throttled by the quota, and K4's PNG
// took 55 or 105 ms from one run to the next (profiles/r05_end_to_end.txt, r06_end_to_end.txt).
int usableThreads() {
    unsigned n = std::thread::hardware_concurrency();
    if (!n) n = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min<unsigned>(n, (unsigned)std::max(1, CPU_COUNT(&set)));
    auto quota = [](const char* path, const char* period_path) -> double {   // cgroup v2: "max 100000" | "1600000 100000"; v1: two files
        FILE* f = std::fopen(path, "r");
        if (!f) return 0.0;
        char a[64] = {0};
        double q = 0.0, per = 0.0;
        if (period_path) {
            if (std::fscanf(f, "%lf", &q) != 1) q = 0.0;
            std::fclose(f);
            f = std::fopen(period_path, "r");
            if (!f) return 0.0;
            if (std::fscanf(f, "%lf", &per) != 1) per = 0.0;
        } else if (std::fscanf(f, "%63s %lf", a, &per) == 2) {
            q = std::strcmp(a, "max") == 0 ? 0.0 : std::atof(a);
        }
        std::fclose(f);
        return (q > 0.0 && per > 0.0) ? q / per : 0.0;
    };
    double q = quota("/sys/fs/cgroup/cpu.max", nullptr);
    if (q <= 0.0) q = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
    if (q > 0.0) n = std::min<unsigned>(n, (unsigned)std::max(1.0, q + 0.5));
    return (int)std::max(1u, n);
}

namespace {
// The stripe plan, the workers and the assembly of one image.  The rows of the source may still be ARRIVING, in order (Progressive
// below): a worker takes the stripes in ascending order and, before it filters one, waits until the source's rows [0, y1) are final —
// rows_ready(h) before the start makes this the plain encoder.  The stripes, and therefore the file, do not depend on how the rows came.
// Stripes per worker thread.  Eight: when the rows arrive in bands (Progressive) the LAST band's stripes are all that is left to do once the
// device is done, and with four per thread a band of an eighth of the image was 8 stripes for 16 workers — half of them idle for the 12 ms
// that mattered.  With Z_RLE (no match window to lose) a stripe boundary costs an empty stored block: K4's file grows by 0.02 %.
#ifndef MC_PNG_STRIPES_PER_THREAD
#define MC_PNG_STRIPES_PER_THREAD 8
#endif
struct EncodeJob {
    RowSource src;
    int threads = 1;
    uint32_t n_stripes = 1;
    std::vector<Stripe> stripes;
    std::atomic<uint32_t> next{0};
    std::mutex mu;
    std::condition_variable cv;
    uint32_t ready = 0;            // rows [0, ready) of the source are final (guarded by mu)
    bool cancelled = false;        // the image will never be finished: workers take no further stripe (guarded by mu)
    std::vector<std::thread> workers;

    EncodeJob(const RowSource& source, int nthreads) : src(source) {
        const uint32_t h = src.h;
        // stripes of >= 64 KiB of raw data, at most MC_PNG_STRIPES_PER_THREAD per worker thread
        threads = nthreads <= 0 ? usableThreads() : nthreads;
        threads = std::max(1, std::min(threads, 64));
        const size_t row_bytes = (size_t)src.w * src.bpp + 1;
        const uint32_t min_rows = (uint32_t)std::max<size_t>(1, (64 * 1024 + row_bytes - 1) / row_bytes);
        n_stripes = std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)threads * (uint32_t)MC_PNG_STRIPES_PER_THREAD, h / min_rows ? h / min_rows : 1));
        if (threads == 1) n_stripes = 1;
        stripes.resize(n_stripes);
        for (uint32_t i = 0; i < n_stripes; i++) {
            stripes[i].y0 = (uint32_t)((uint64_t)h * i / n_stripes);
            stripes[i].y1 = (uint32_t)((uint64_t)h * (i + 1) / n_stripes);
        }
    }
    void rows_ready(uint32_t up_to) {
        { std::lock_guard<std::mutex> lk(mu); ready = std::max(ready, std::min(up_to, src.h)); }
        cv.notify_all();
    }
    // An abandoned image (the render that was to fill the source failed): no worker reads a row it has not been promised, none takes
    // another stripe, and join() returns once the stripes already in hand — whose rows WERE final — are done.
    void cancel() {
        { std::lock_guard<std::mutex> lk(mu); cancelled = true; }
        cv.notify_all();
    }
    void work() {
        Bytes scratch;   // this worker's filtered rows (compress_stripe)
        for (;;) {
            const uint32_t i = next.fetch_add(1);
            if (i >= n_stripes) break;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return cancelled || ready >= stripes[i].y1; });
                if (cancelled) break;
            }
            compress_stripe(src, i + 1 == n_stripes, stripes[i], scratch);
        }
    }
    void start() {   // background workers (Progressive); run() below works on the calling thread as well
        const int nt = (int)std::min<uint32_t>((uint32_t)threads, n_stripes);
        for (int t = 0; t < nt; t++) workers.emplace_back([this] { work(); });
    }
    void join() {
        for (auto& t : workers) t.join();
        workers.clear();
    }
    std::string assemble(std::vector<uint8_t>& out) {
        const uint32_t w = src.w, h = src.h;
        const bool opaque = src.bpp == 3;
        size_t zlen = 2 + 4;
        uLong adler = adler32(0L, Z_NULL, 0);
        for (auto& s : stripes) {
            if (!s.ok) return "zlib deflate failed";
            zlen += s.z.size();
            adler = adler32_combine(adler, s.adler, (z_off_t)s.raw_len);
        }
        // One copy of every stripe, straight into the file image, and NO pass over the assembled stream: the chunk's CRC is joined from
        // the stripes' own (computed by the workers) with crc32_combine.  The serial tail of a save was two copies and a CRC pass over
        // the whole stream — at K4 (7.9 MB) most of what was left to do after the last band had arrived.
        const uint8_t zhdr[2] = {0x78, 0x9c};   // zlib header: deflate, 32 KiB window, default compression
        const uint8_t ztail[4] = {(uint8_t)(adler >> 24), (uint8_t)(adler >> 16), (uint8_t)(adler >> 8), (uint8_t)adler};
        uLong crc = crc32(0L, reinterpret_cast<const Bytef*>("IDAT"), 4);
        crc = crc32(crc, zhdr, 2);
        for (auto& s : stripes)
            if (!s.z.empty()) crc = crc32_combine(crc, s.crc, (z_off_t)s.z.size());
        crc = crc32(crc, ztail, 4);

        out.clear();
        out.reserve(zlen + 8 + 25 + 12 + 12);
        static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
        out.insert(out.end(), sig, sig + 8);
        std::vector<uint8_t> ihdr;
        put32(ihdr, w); put32(ihdr, h);
        ihdr.push_back(8); ihdr.push_back(opaque ? 2 : 6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
        chunk_header_and_crc(out, "IHDR", ihdr.data(), ihdr.size());
        put32(out, (uint32_t)zlen);
        static const uint8_t idat[4] = {'I', 'D', 'A', 'T'};
        out.insert(out.end(), idat, idat + 4);
        out.insert(out.end(), zhdr, zhdr + 2);
        for (auto& s : stripes) out.insert(out.end(), s.z.begin(), s.z.end());
        out.insert(out.end(), ztail, ztail + 4);
        put32(out, (uint32_t)crc);
        chunk_header_and_crc(out, "IEND", nullptr, 0);
        return "";
    }
};

std::string encode_rows(std::vector<uint8_t>& out, const RowSource& src, int threads) {
    EncodeJob job(src, threads);
    job.rows_ready(src.h);
    if (std::min<uint32_t>((uint32_t)job.threads, job.n_stripes) <= 1) job.work();
    else { job.start(); job.join(); }
    return job.assemble(out);
}

std::string write_file(const char* filename, const std::vector<uint8_t>& png) {
    FILE* f = std::fopen(filename, "wb");
    if (!f) return std::string("cannot open ") + filename;
    size_t n = std::fwrite(png.data(), 1, png.size(), f);
    std::fclose(f);
    return n == png.size() ? "" : "short write";
}
}  // namespace

std::string encode(std::vector<uint8_t>& out, const uint8_t* rgba8, uint32_t w, uint32_t h, int threads) {
    if (!rgba8 || !w || !h) return "empty image";
    bool opaque = true;
    for (size_t i = 0; i < (size_t)w * h && opaque; i++) opaque = rgba8[4 * i + 3] == 255;
    RowSource src;
    src.rgba8 = rgba8; src.w = w; src.h = h; src.bpp = opaque ? 3 : 4;
    return encode_rows(out, src, threads);
}

// The storage buffer straight to a PNG: alpha is 255 by construction (mandelbrotApp.h:165, pathtracerApp.h:218), so the image is 8-bit RGB.
// Same pixels as convertStorage + encode — the SAME FILE, in fact: the rows a stripe filters are the rows convertStorage would have written.
std::string encodeStorage(std::vector<uint8_t>& out, const float* vec4, uint32_t w, uint32_t h, float scale, bool rotate180, int threads) {
    if (!vec4 || !w || !h) return "empty image";
    RowSource src;
    src.vec4 = vec4; src.scale = scale; src.rotate180 = rotate180; src.w = w; src.h = h; src.bpp = 3;
    return encode_rows(out, src, threads);
}

struct Progressive::Impl { EncodeJob job; bool finished = false; Impl(const RowSource& s, int t) : job(s, t) {} };
Progressive::Progressive() = default;
Progressive::~Progressive() { abandon(); }
// The image will not be finished (run() threw): the workers stop WITHOUT reading rows that never arrived — by the time an application
// object is torn down its storage buffer may be gone already (members are destroyed in reverse order of declaration).
void Progressive::abandon() {
    if (!impl) return;
    if (!impl->finished) { impl->job.cancel(); impl->job.join(); }
    impl.reset();
}
bool Progressive::active() const { return impl && !impl->finished; }
void Progressive::beginStorage(const float* vec4, uint32_t w, uint32_t h, float scale, int threads) {
    RowSource src;
    src.vec4 = vec4; src.scale = scale; src.rotate180 = false; src.w = w; src.h = h; src.bpp = 3;
    abandon();   // (an image still in progress — run() called again before a save — is given up, its workers stopped, not destroyed under them)
    impl.reset(new Impl(src, threads));
    impl->job.start();
}
void Progressive::beginOpaqueRgba8(const uint8_t* rgba8, uint32_t w, uint32_t h, int threads) {
    RowSource src;
    src.rgba8 = rgba8; src.w = w; src.h = h; src.bpp = 3;
    abandon();   // (an image still in progress — run() called again before a save — is given up, its workers stopped, not destroyed under them)
    impl.reset(new Impl(src, threads));
    impl->job.start();
}
void Progressive::rowsReady(uint32_t upTo) { if (impl) impl->job.rows_ready(upTo); }
std::string Progressive::finish(std::vector<uint8_t>& png) {
    if (!impl) return "no image in progress";
    if (impl->finished) return "no image in progress";
    auto ms = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    impl->job.rows_ready(impl->job.src.h);
    auto t0 = std::chrono::steady_clock::now();
    impl->job.join();
    joinMs = ms(t0);
    t0 = std::chrono::steady_clock::now();
    std::string err = impl->job.assemble(png);
    assembleMs = ms(t0);
    // The stripes' buffers are released by the next begin*() or the destructor, not here: the caller still has the file to write, and an
    // application that leaves with _Exit once it is written never pays for giving the memory back (5 ms at K4).
    impl->finished = true;
    return err;
}
std::string writeFile(const char* filename, const std::vector<uint8_t>& png) { return write_file(filename, png); }

std::string encodeFile(const char* filename, const uint8_t* rgba8, uint32_t w, uint32_t h, int threads) {
    std::vector<uint8_t> png;
    std::string err = encode(png, rgba8, w, h, threads);
    return err.empty() ? write_file(filename, png) : err;
}

std::string encodeStorageFile(const char* filename, const float* vec4, uint32_t w, uint32_t h, float scale, bool rotate180, int threads) {
    std::vector<uint8_t> png;
    std::string err = encodeStorage(png, vec4, w, h, scale, rotate180, threads);
    return err.empty() ? write_file(filename, png) : err;
}


void convertStorage(const float* vec4, uint8_t* rgba8, uint32_t w, uint32_t h, float scale, bool rotate180, int threads) {
    struct Pixel { float r, g, b, a; };
    const Pixel* p = reinterpret_cast<const Pixel*>(vec4);
    auto rows = [=](uint32_t y0, uint32_t y1) {
        for (uint32_t y = y0; y < y1; y++)
            for (uint32_t x = 0; x < w; x++) {
                const Pixel& s = p[(size_t)y * w + x];
                size_t to = (size_t)y * w + x;
                if (rotate180 && !((w & 1u) && x == w / 2)) to = (size_t)(h - 1 - y) * w + (w - 1 - x);
                uint8_t* o = rgba8 + 4 * to;
                o[0] = x86FloatToU8(scale * s.r); o[1] = x86FloatToU8(scale * s.g); o[2] = x86FloatToU8(scale * s.b); o[3] = 255u;
            }
    };
    unsigned n = threads > 0 ? (unsigned)threads : (unsigned)usableThreads();
    n = std::max(1u, std::min(n, std::max(1u, h / 16u)));
    if (n == 1) { rows(0, h); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < n; t++) th.emplace_back(rows, (uint32_t)((uint64_t)h * t / n), (uint32_t)((uint64_t)h * (t + 1) / n));
    for (auto& t : th) t.join();
}

}  // namespace pngwriter
