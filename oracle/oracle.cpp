// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_core.h header).  C entry points for ctypes.
// Multithreaded by rows (std::thread) so it doubles as the "CPU restatement baseline" of
// BASELINE.md §3 plan 2.  Build: oracle/Makefile (g++ -O2 -ffp-contract=off -fno-fast-math -mfma).
#include "oracle_core.h"

#include <atomic>
#include <thread>
#include <vector>

using namespace oracle;

namespace {

template <class F>
void parallel_rows(uint32_t r0, uint32_t r1, int nthreads, F&& body) {
    if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
    if (nthreads < 1) nthreads = 1;
    uint32_t rows = r1 > r0 ? r1 - r0 : 0;
    if ((uint32_t)nthreads > rows) nthreads = rows ? (int)rows : 1;
    std::atomic<uint32_t> next{r0};
    auto worker = [&](int tid) {
        for (;;) {
            uint32_t r = next.fetch_add(1);
            if (r >= r1) break;
            body(r, tid);
        }
    };
    if (nthreads == 1) { worker(0); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back(worker, t);
    for (auto& t : th) t.join();
}

}  // namespace

extern "C" {

// view[8] = {cx_hi, cx_lo, cy_hi, cy_lo, sx_hi, sx_lo, sy_hi, sy_lo}; precision 0 = fp32, 1 = ds.
// iters: tile-local, (row_end-row_begin)*W entries, row-major (row = gl_GlobalInvocationID.y).
int oracle_mandelbrot_iters(uint32_t W, uint32_t H, uint32_t maxIter, const float* view, int precision,
                            uint32_t row_begin, uint32_t row_end, uint32_t* iters, int nthreads) {
    if (!W || !H || row_end > H || row_begin > row_end) return 1;
    MandelView v{view[0], view[1], view[2], view[3], view[4], view[5], view[6], view[7]};
    parallel_rows(row_begin, row_end, nthreads, [&](uint32_t gy, int) {
        uint32_t* dst = iters + (size_t)(gy - row_begin) * W;
        for (uint32_t gx = 0; gx < W; gx++)
            dst[gx] = precision ? mandel_ds_pixel(gx, gy, W, H, maxIter, v) : mandel_f32_pixel(gx, gy, W, H, maxIter, v);
    });
    return 0;
}

// (maxIter+1)-entry colour LUT in the storage-buffer format (vec4 fp32) and after the host cast (RGBA8).
void oracle_mandel_lut(uint32_t maxIter, const float* kColor, float* lut_f32, uint8_t* lut_u8) {
    for (uint32_t n = 0; n <= maxIter; n++) {
        float c[4];
        mandel_colour(n, maxIter, kColor, c);
        for (int k = 0; k < 4; k++) lut_f32[4 * n + k] = c[k];
        lut_u8[4 * n + 0] = x86_float_to_u8(255.0f * c[0]);   // mandelbrotApp.h:162-165, scale :174
        lut_u8[4 * n + 1] = x86_float_to_u8(255.0f * c[1]);
        lut_u8[4 * n + 2] = x86_float_to_u8(255.0f * c[2]);
        lut_u8[4 * n + 3] = 255u;
    }
}

// Sum over pixels of executed loop bodies (n+1 for escaping pixels, maxIter for interior): the
// "pixel-iters" unit of BASELINE.json's metric.
uint64_t oracle_mandel_pixel_iters(const uint32_t* iters, uint64_t count, uint32_t maxIter) {
    uint64_t s = 0;
    for (uint64_t i = 0; i < count; i++) s += iters[i] < maxIter ? iters[i] + 1 : maxIter;
    return s;
}

// static_cast<uint8_t>(scale * v) with x86 semantics, alpha forced to 255 (mandelbrotApp.h:159-166).
void oracle_float_to_rgba8(uint64_t npix, float scale, const float* rgba_f32, uint8_t* rgba8) {
    for (uint64_t i = 0; i < npix; i++) {
        rgba8[4 * i + 0] = x86_float_to_u8(scale * rgba_f32[4 * i + 0]);
        rgba8[4 * i + 1] = x86_float_to_u8(scale * rgba_f32[4 * i + 1]);
        rgba8[4 * i + 2] = x86_float_to_u8(scale * rgba_f32[4 * i + 2]);
        rgba8[4 * i + 3] = 255u;
    }
}

// 180-degree rotation exactly as pathtracerApp.h:236-243 (note x < resx/2: odd widths keep the middle column).
void oracle_rotate180_rgba8(uint32_t W, uint32_t H, uint8_t* rgba8) {
    uint32_t* p = reinterpret_cast<uint32_t*>(rgba8);
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W / 2; x++) {
            uint32_t from = x + y * W;
            uint32_t to = (W - 1) - x + ((H - 1) - y) * W;
            uint32_t t = p[from]; p[from] = p[to]; p[to] = t;
        }
}

// Path tracer.  out: tile-local storage-buffer rows [row_begin,row_end) (buffer row r holds pix.y = H-1-r,
// pathTracer.comp:349), vec4 fp32 per pixel.  For sample_begin > 0 `out` must hold the accumulator
// left by the previous range.  counts (optional, 12 x uint64): per-op totals, see OpCounts.
int oracle_pathtrace(uint32_t W, uint32_t H, uint32_t spp, uint32_t sample_begin, uint32_t sample_end,
                     uint32_t maxDepth, const float* planes, uint32_t nPlanes, const float* spheres,
                     uint32_t nSpheres, int mathMode, uint32_t row_begin, uint32_t row_end, float* out,
                     int nthreads, uint64_t* counts) {
    const int precMode = (mathMode >> 8) & 0xff;   // bits 8..15: sphere-test precision branch (0 fp32, 1 fp64, 2 DS, 3 DF64)
    mathMode &= 0xff;
    if (!W || !H || row_end > H || row_begin > row_end || sample_end > spp || sample_begin > sample_end) return 1;
    if (counts) {
        int nt = nthreads <= 0 ? (int)std::thread::hardware_concurrency() : nthreads;
        if (nt < 1) nt = 1;
        std::vector<OpCounts> per(nt);
        PT<CountPolicy> pt{planes, nPlanes, spheres, nSpheres, mathMode, precMode};
        parallel_rows(row_begin, row_end, nt, [&](uint32_t r, int tid) {
            tls_counts() = &per[tid];
            uint32_t gy = H - 1 - r;
            for (uint32_t gx = 0; gx < W; gx++)
                pt.pixel(gx, gy, W, H, spp, sample_begin, sample_end, maxDepth, out + ((size_t)(r - row_begin) * W + gx) * 4);
        });
        OpCounts tot;
        for (auto& c : per) tot += c;
        uint64_t v[12] = {tot.add, tot.mul, tot.div, tot.sqrt, tot.trig, tot.pow_, tot.cmp, tot.iop, tot.cvt,
                          tot.intersect_calls, tot.bounces, tot.samples};
        for (int i = 0; i < 12; i++) counts[i] = v[i];
    } else {
        PT<PlainPolicy> pt{planes, nPlanes, spheres, nSpheres, mathMode, precMode};
        parallel_rows(row_begin, row_end, nthreads, [&](uint32_t r, int) {
            uint32_t gy = H - 1 - r;
            for (uint32_t gx = 0; gx < W; gx++)
                pt.pixel(gx, gy, W, H, spp, sample_begin, sample_end, maxDepth, out + ((size_t)(r - row_begin) * W + gx) * 4);
        });
    }
    return 0;
}

// Per-sample radiance (before the /spp accumulation) for one pixel — debugging aid for parity tests.
void oracle_pathtrace_sample(uint32_t gx, uint32_t gy, uint32_t W, uint32_t H, uint32_t samp, uint32_t maxDepth,
                             const float* planes, uint32_t nPlanes, const float* spheres, uint32_t nSpheres,
                             int mathMode, float* rgb) {
    PT<PlainPolicy> pt{planes, nPlanes, spheres, nSpheres, mathMode & 0xff, (mathMode >> 8) & 0xff};
    v3 r = pt.sample(gx, gy, W, H, samp, maxDepth);
    rgb[0] = r.x; rgb[1] = r.y; rgb[2] = r.z;
}

// ---- unit-level known-answer helpers ------------------------------------------------------------
void oracle_rand01(uint64_t n, const uint32_t* xyz, float* out) {
    for (uint64_t i = 0; i < n; i++) {
        v3 r = rand01<PlainPolicy>(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
        out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
    }
}
// op: 0 ds_add, 1 ds_sub, 2 ds_mul, 3 ds_compare (out[2i] = -1/0/1, out[2i+1] = 0), 4 ds_sqrt(a),
//     5 df64_add, 6 df64_mult, 7 df64_sqrt(a), 8 ds_twoProd(a.hi, b.hi)
void oracle_ds_op(int op, uint64_t n, const float* a, const float* b, float* out) {
    for (uint64_t i = 0; i < n; i++) {
        ds2 x{a[2 * i], a[2 * i + 1]}, y{b[2 * i], b[2 * i + 1]}, r{0, 0};
        switch (op) {
            case 0: r = ds_add(x, y); break;
            case 1: r = ds_sub(x, y); break;
            case 2: r = ds_mul(x, y); break;
            case 4: r = ds_sqrt(x); break;
            case 5: r = df64_add(x, y); break;
            case 6: r = df64_mult(x, y); break;
            case 7: r = df64_sqrt(x); break;
            case 8: r = ds_twoProd(x.x, y.x); break;
            case 9: r = ds_div(x, y); break;
            case 10: r = twoDiff(x.x, y.x); break;
            case 11: r = ds2{df64_eq(x, y) ? 1.0f : 0.0f, df64_neq(x, y) ? 1.0f : 0.0f}; break;
            default: r = ds2{ds_compare(x, y), 0.0f}; break;
        }
        out[2 * i] = r.x; out[2 * i + 1] = r.y;
    }
}
// fn: 0 mc_sin, 1 mc_cos, 2 mc_log2, 3 mc_exp2, 4 mc_pow(x, 0.45)
void oracle_mc_math(int fn, uint64_t n, const float* in, float* out) {
    for (uint64_t i = 0; i < n; i++) {
        float x = in[i];
        switch (fn) {
            case 0: out[i] = mcmath::mc_sin(x); break;
            case 1: out[i] = mcmath::mc_cos(x); break;
            case 2: out[i] = mcmath::mc_log2(x); break;
            case 3: out[i] = mcmath::mc_exp2(x); break;
            default: out[i] = mcmath::mc_pow(x, 0.45f); break;
        }
    }
}

int oracle_hardware_threads(void) { return (int)std::thread::hardware_concurrency(); }

}  // extern "C"
