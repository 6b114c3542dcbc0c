// Mandelbrot escape-time kernels for gfx950 (MI355X).
//
// Replaces shaders/mandelbrot.comp:21-60 (fp32) and adds the two-float deep-zoom variant composed
// from the reference's ds_* primitives (shaders/emulateDouble.h.glsl:59-139).
//
// Design (MI355X-first, not a translation of the 32x32 Vulkan workgroup):
//  * one work-item per pixel; a wave64 owns an 8x8 pixel tile (coherent trip counts, and every
//    128-B output segment of a row is written whole); one wave per workgroup.
//  * the escape test is a wave ballot (`v_cmp_* sgpr-pair`), not an exec-mask update: the loop body
//    is straight-line VALU for U iterations and all bookkeeping (OR of the U ballots, "every lane
//    done" early-out, iteration counter) runs on the scalar unit.  Per-lane iteration counts are
//    reconstructed from the saved ballots only in the (rare) blocks where some lane escapes.
//  * v_cmp_*_f32 costs 4 issue cycles on gfx950 but v_or_b32 only 2 (tools/valu_microbench.hip): the fp32
//    fast path ORs the bit patterns of the block's |z|^2 values (bit 30 set <=> value >= 2) and compares
//    once per block; the exact ballots are recomputed from the block's saved state only when an unfinished
//    lane may have escaped.  Interior tiles issue 8 fp32 ops + 1 v_or per pixel-iteration (18 cycles).
//  * unfused IEEE fp32 in GLSL source order: this TU is built with -ffp-contract=off.
//    zx*zx and zy*zy are computed once per iteration and reused by the magnitude test and the next
//    update — identical values, so identical results to the literal `dot(z,z)` (mandelbrot.comp:43-44).
//  * the colour is a host-built (max_iter+1)-entry vec4 LUT: only max_iter+1 distinct
//    colours exist and GLSL cos() precision is implementation-defined.
#include <cmath>
#include <cstring>

#include "ds_arith.h"
#include "mc_internal.h"

namespace mc {

namespace {

struct MandelArgs {
    uint32_t W, H, max_iter;
    uint32_t row_begin, row_end, row_block, row_stride;
    float cx_hi, cx_lo, cy_hi, cy_lo;
    float sx_hi, sx_lo, sy_hi, sy_lo;
    float4* __restrict__ out_rgba;      // tile-local, may be null
    uint32_t* __restrict__ out_iters;   // tile-local, may be null
    uint16_t* __restrict__ out_iters16; // the same plane as 16-bit counts (MC_MANDEL_ITERS_U16, max_iter <= 65535), may be null
    const float4* __restrict__ lut;     // max_iter+1 entries (null when out_rgba is null)
    // c = centre + (uv - 0.5) * scale per column / per row, evaluated once on the host with the shader's fp32
    // operation sequence (mandelbrot.comp:30-31,38): fp32 [cx[W] | cy[H]], two-float [cx(hi,lo)[W] | cy(hi,lo)[H]].
    // Replaces two IEEE divisions (~50 instructions) per pixel; exterior tiles only run a handful of iterations.
    const float* __restrict__ c_tab;
};

// ---- per-pixel state machines: step() advances one iteration and reports "escaped now" ----------
template <bool FMA>
struct StateF32 {
    float cx, cy, zx, zy, sx, sy;   // sx = zx*zx, sy = zy*zy of the current z
    __device__ __forceinline__ void init(uint32_t gx, uint32_t gy, const MandelArgs& a) {
        cx = a.c_tab[gx];                          // mandelbrot.comp:30,38 (host-evaluated, see build_c_table)
        cy = a.c_tab[a.W + gy];                    // :31,38
        zx = zy = sx = sy = 0.0f;
    }
    // One iteration (:43), returns |z|^2 = dot(z,z) of the new z (:44).
    __device__ __forceinline__ float advance() {
        float nzx, nzy;
        if (FMA) {   // NON-PARITY diagnostic variant (MC_MANDEL_FMA)
            nzx = __builtin_fmaf(zx, zx, -sy) + cx;
            nzy = __builtin_fmaf(2.0f * zx, zy, cy);
        } else {
            nzx = (sx - sy) + cx;
            nzy = ((2.0f * zx) * zy) + cy;
        }
        zx = nzx; zy = nzy;
        sx = zx * zx; sy = zy * zy;
        return sx + sy;
    }
    __device__ __forceinline__ bool step() { return advance() > 2.0f; }   // :44
    // z is the orbit's complete state (sx, sy are functions of it): equal z => equal future (see escape_time)
    __device__ __forceinline__ bool same_z(const StateF32& o) const { return zx == o.zx && zy == o.zy; }
    static constexpr uint32_t kCycleCheckBlocks = 1;   // compare with the reference state after every block (8 iterations)
    // Conservative escape filter on the bit pattern of |z|^2 (>= 0, or NaN after an overflow): every value
    // > 2.0f has bit 30 set or is 0x40000001..., every value < 2.0f has bit 30 clear, so the bitwise OR of a
    // block's magnitudes exceeds 0x40000000 whenever any of them exceeded 2.0f (no false negatives; the only
    // false positives involve a magnitude of exactly 2.0f).  v_or_b32 issues in 2 cycles, v_cmp_*_f32 in 4.
    static constexpr bool kHasFastBlock = true;
    using Acc = uint32_t;
    __device__ __forceinline__ Acc acc_init() const { return 0u; }
    __device__ __forceinline__ void advance_fast(Acc& acc) { acc |= __float_as_uint(advance()); }
    static __device__ __forceinline__ bool needs_exact(Acc or_of_bits) { return or_of_bits > 0x40000000u; }
};

struct StateDS {
    ds2 cx, cy, zx, zy, sx, sy;
    __device__ __forceinline__ void init(uint32_t gx, uint32_t gy, const MandelArgs& a) {
        const float2* tab = reinterpret_cast<const float2*>(a.c_tab);
        float2 tx = tab[gx], ty = tab[a.W + gy];
        cx = ds2{tx.x, tx.y};
        cy = ds2{ty.x, ty.y};
        zx = zy = sx = sy = ds_set(0.0f);
    }
    // Fast block: the same iteration with
    //  (1) the Dekker error term of each product replaced by ONE fma — bit-identical whenever the error term is
    //      representable (tools/dekker_vs_fma.c; error-free transformation), which holds for |operand| >= 2^-50;
    //      `mn` tracks the smallest square seen, a block containing a smaller operand (or an exact zero) is redone
    //      with the literal sequence;
    //  (2) the 11-flop ds_add + 3-way compare of the escape test replaced by t = sx.hi + sy.hi and an integer
    //      max: ds_add(sx, sy).hi differs from t by < 4 ulp (|t2| <= 1/2 ulp(t) + |sx.lo| + |sy.lo|), so
    //      t < 2 - 16 ulp proves "not escaped"; a block that gets closer is redone exactly.
    // Both substitutions leave the state words of every unfinished lane bit-identical to step()'s.
    static constexpr bool kHasFastBlock = true;
    struct Acc {
        uint32_t mx;   // max over the block of bits(t), unsigned: a negative or NaN t reads as "large"
        int32_t mn;    // min over the block of bits(square.hi), signed: a negative square reads as "small"
    };
    __device__ __forceinline__ Acc acc_init() const {
        int32_t a = __float_as_int(sx.hi), b = __float_as_int(sy.hi);
        return Acc{0u, a < b ? a : b};   // the incoming z is an operand of this block's first zx*zy
    }
    __device__ __forceinline__ void advance_fast(Acc& acc) {
        ds2 zxy = ds_mul_fma(zx, zy);
        ds2 twoxy = ds2{2.0f * zxy.hi, 2.0f * zxy.lo};
        ds2 nzx = ds_add(ds_sub(sx, sy), cx);
        ds2 nzy = ds_add(twoxy, cy);
        zx = nzx; zy = nzy;
        sx = ds_mul_fma(zx, zx); sy = ds_mul_fma(zy, zy);
        int32_t bx = __float_as_int(sx.hi), by = __float_as_int(sy.hi);
        uint32_t bt = __float_as_uint(sx.hi + sy.hi);
        acc.mx = acc.mx > bt ? acc.mx : bt;
        int32_t m = bx < by ? bx : by;
        acc.mn = acc.mn < m ? acc.mn : m;
    }
    static __device__ __forceinline__ bool needs_exact(Acc acc) {
        // 0x3ffffff0 = 2 - 16 ulp; 0x0e800000 = 2^-98 > (2^-50)^2 (sx.hi is within an ulp of zx.hi^2)
        return acc.mx >= 0x3ffffff0u || acc.mn < 0x0e800000;
    }
    __device__ __forceinline__ bool same_z(const StateDS& o) const {
        return zx.hi == o.zx.hi && zx.lo == o.zx.lo && zy.hi == o.zy.hi && zy.lo == o.zy.lo;
    }
    static constexpr uint32_t kCycleCheckBlocks = 4;   // every 4 blocks (16 iterations): deep-zoom views have few cycling pixels
    __device__ __forceinline__ bool step() {
        ds2 zxy = ds_mul(zx, zy);
        ds2 twoxy = ds2{2.0f * zxy.hi, 2.0f * zxy.lo};   // exact
        ds2 nzx = ds_add(ds_sub(sx, sy), cx);
        ds2 nzy = ds_add(twoxy, cy);
        zx = nzx; zy = nzy;
        sx = ds_sqr(zx); sy = ds_sqr(zy);
        return ds_greater(ds_add(sx, sy), ds_set(2.0f));
    }
};

// Runs the escape-time loop for the 64 pixels of a wave.  Returns n in [0,max_iter] per lane:
// the number of iterations that did not escape (mandelbrot.comp:40-46).
//
// CONVERGED TILES (north_star: "wavefront ballot/any for early-out on converged Mandelbrot tiles").  The iteration is a
// deterministic map of the state z (c is fixed per lane; sx, sy are functions of z), so an orbit that returns to a value it
// has held before repeats that stretch for ever.  If lane L has not escaped up to iteration i and z_i == z_j for an earlier
// j (compared as VALUES: +0 and -0 are interchangeable operands of +, -, x and of the comparison, and a NaN never compares
// equal), no iteration of the cycle j..i escaped, so none ever will: the shader's loop would run to max_iter and leave
// n = max_iter — exactly what this lane returns.  Brent's scheme at block granularity: a reference state is kept per lane,
// compared with the state at the end of a block of U iterations (fp32: 2 compares per 8 iterations) and replaced when the
// number of comparisons made reaches 1, 2, 4, 8, ...  A wave leaves as soon as every lane has escaped or is known to cycle:
// at K1 that is 91 % of the interior pixels (median: iteration 88 of 1000), 2.44x fewer issued instructions and
// 0.38 -> 0.21 ms; the iteration plane is bit-identical (tests, fuzz).  fp32 orbits inside the set collapse
// onto a short exact cycle near their attractor; the two-float orbits of a deep zoom rarely do (checked every 16 iterations).
template <class State, int U>
__device__ __forceinline__ uint32_t escape_time(State& st, uint32_t max_iter, bool valid) {
    const uint32_t lane = __lane_id();
    const uint64_t lanebit = 1ull << lane;
    uint64_t done = ~__ballot(valid);   // lanes outside the image never hold the wave
    uint32_t n = max_iter;
    uint32_t i = 0;
    State ref = st;                     // Brent reference state (z_0 = 0: a cycle through the origin is caught too)
    uint32_t checks = 0;                // comparisons made so far (wave-uniform)
    for (; i + U <= max_iter; i += U) {
        if (i != 0 && (i / (uint32_t)U) % State::kCycleCheckBlocks == 0u) {
            // cycling lanes are finished with n = max_iter (their state stays on the cycle: harmless to keep iterating)
            done |= __ballot(st.same_z(ref));
            if (done == ~0ull) return n;
            checks++;
            if ((checks & (checks - 1u)) == 0u) ref = st;   // wave-uniform: at 1, 2, 4, 8, ... comparisons
        }
        if (State::kHasFastBlock && i != 0) {   // the first block is evaluated exactly: most tiles escape right there
            // fast path: U iterations without per-iteration compares/ballots, ONE test per block; the exact
            // per-iteration ballots below are evaluated (from the saved state) only if some unfinished lane may
            // have escaped (or, two-float state, may have left the fast arithmetic's precondition)
            State probe = st;
            typename State::Acc acc = st.acc_init();
#pragma unroll
            for (int k = 0; k < U; k++) probe.advance_fast(acc);
            if ((__ballot(State::needs_exact(acc)) & ~done) == 0ull) {
                st = probe;
                continue;
            }
        }
        uint64_t b[U];
        uint64_t any = 0;
#pragma unroll
        for (int k = 0; k < U; k++) {
            b[k] = __ballot(st.step());
            any |= b[k];
        }
        uint64_t newly = any & ~done;
        if (newly) {   // wave-uniform: some lane escaped for the first time in this block
#pragma unroll
            for (int k = U - 1; k >= 0; k--)
                if (b[k] & ~done & lanebit) n = i + k;
            done |= any;
            if (done == ~0ull) return n;
        }
    }
    for (; i < max_iter; i++) {   // tail: max_iter % U iterations
        uint64_t b = __ballot(st.step());
        uint64_t newly = b & ~done;
        if (newly) {
            if (newly & lanebit) n = i;
            done |= b;
            if (done == ~0ull) return n;
        }
    }
    return n;
}

template <class State, int U>
__global__ void __launch_bounds__(64) mandelbrot_kernel(MandelArgs a) {
    // workgroup = one wave = one 8x8 pixel tile, lane = (lx, ly) inside the tile.  Tiles finish anywhere between 1 and
    // max_iter iterations apart, so the unit the hardware schedules is the tile itself: a 4-wave block would keep its
    // place on the CU until its slowest tile is through (K1: 0.208 -> 0.200 ms).
    const uint32_t lane = threadIdx.x;
    const uint32_t tile_x = blockIdx.x, tile_y = blockIdx.y;
    const uint32_t gx = tile_x * 8u + (lane & 7u);
    const uint32_t ty = tile_y * 8u + (lane >> 3);   // tile-local row
    const uint32_t gy = tile_row_to_storage(ty, a.row_begin, a.row_block, a.row_stride);
    const bool valid = gx < a.W && gy < a.row_end;                            // mandelbrot.comp:27-28
    State st;
    st.init(valid ? gx : 0u, valid ? gy : 0u, a);
    uint32_t n = escape_time<State, U>(st, a.max_iter, valid);
    if (valid) {
        size_t idx = (size_t)ty * a.W + gx;                                    // :59 (row-major, tile-local)
        if (a.out_iters) a.out_iters[idx] = n;
        if (a.out_iters16) a.out_iters16[idx] = (uint16_t)n;
        if (a.out_rgba) a.out_rgba[idx] = a.lut[n];
    }
}

}  // namespace

// colour(n) = d + e*cos(6.28318*(f*t+g)), t = n/M — mandelbrot.comp:50-56, evaluated in fp32 in source
// order on the host (d = kColor.rgb, mandelbrotApp.h:139-141).  alpha = 1.0.
void mandelbrot_build_lut(uint32_t max_iter, const float k_color[4], float* lut) {
    const float e[3] = {-0.2f, -0.3f, -0.5f};
    const float f[3] = {2.1f, 2.0f, 3.0f};
    const float g[3] = {0.0f, 0.1f, 0.0f};
    for (uint32_t n = 0; n <= max_iter; n++) {
        float t = (float)n / (float)max_iter;
        for (int c = 0; c < 3; c++) {
            float arg = 6.28318f * (f[c] * t + g[c]);
            lut[4 * (size_t)n + c] = k_color[c] + e[c] * cosf(arg);
        }
        lut[4 * (size_t)n + 3] = 1.0f;
    }
}

static int ensure_lut(mc_context* ctx, const mc_mandelbrot_params* p, hipStream_t s) {
    if (ctx->lut_max_iter == p->max_iter && std::memcmp(ctx->lut_kcolor, p->k_color, sizeof(float) * 4) == 0 && ctx->lut.ptr)
        return MC_OK;
    size_t bytes = ((size_t)p->max_iter + 1) * 4 * sizeof(float);
    std::vector<float> host(((size_t)p->max_iter + 1) * 4);
    mandelbrot_build_lut(p->max_iter, p->k_color, host.data());
    // a previous launch of this context may still be reading the old table (possibly on another stream): wait for
    // the streams this context has launched on — not the whole device — before replacing it
    int rc = ctx->drain_launch_streams();
    if (rc) return rc;
    if ((rc = ctx->lut.reserve(bytes))) return rc;
    MC_HIP_TRY(hipMemcpyAsync(ctx->lut.ptr, host.data(), bytes, hipMemcpyHostToDevice, s));
    MC_HIP_TRY(hipStreamSynchronize(s));   // host vector goes out of scope
    ctx->lut_max_iter = p->max_iter;
    std::memcpy(ctx->lut_kcolor, p->k_color, sizeof(float) * 4);
    return MC_OK;
}

// Per-column / per-row c tables.  x = float(gx)/float(W) (mandelbrot.comp:30), c.x = centre.x + (x - 0.5)*scale.x (:38) in
// fp32 source order; the two-float variant composes ds_add(centre, ds_mul(ds_set(x - 0.5), scale)).
// Host and device execute the same IEEE operations (no contraction), so the tables hold exactly the values the
// kernel used to compute per pixel.  Cached in the context, keyed by (W, H, precision, view).
static int ensure_c_table(mc_context* ctx, const mc_mandelbrot_params* p, hipStream_t s) {
    std::vector<float> key = {(float)p->width, (float)p->height, (float)p->precision, p->centre_x_hi, p->centre_x_lo,
                              p->centre_y_hi, p->centre_y_lo, p->scale_x_hi, p->scale_x_lo, p->scale_y_hi, p->scale_y_lo};
    if (ctx->ctab.ptr && ctx->ctab_key.size() == key.size() &&
        std::memcmp(ctx->ctab_key.data(), key.data(), key.size() * sizeof(float)) == 0)
        return MC_OK;
    const uint32_t W = p->width, H = p->height;
    const bool ds = p->precision == MC_PRECISION_DS;
    std::vector<float> tab(((size_t)W + H) * (ds ? 2 : 1));
    for (uint32_t g = 0; g < W + H; g++) {
        const bool is_x = g < W;
        const float u = is_x ? (float)g / (float)W : (float)(g - W) / (float)H;
        const float c_hi = is_x ? p->centre_x_hi : p->centre_y_hi, c_lo = is_x ? p->centre_x_lo : p->centre_y_lo;
        const float s_hi = is_x ? p->scale_x_hi : p->scale_y_hi, s_lo = is_x ? p->scale_x_lo : p->scale_y_lo;
        if (ds) {
            ds2 c = ds_add(ds2{c_hi, c_lo}, ds_mul(ds_set(u - 0.5f), ds2{s_hi, s_lo}));
            tab[2 * (size_t)g] = c.hi;
            tab[2 * (size_t)g + 1] = c.lo;
        } else {
            tab[g] = c_hi + (u - 0.5f) * s_hi;
        }
    }
    int rc = ctx->drain_launch_streams();   // an earlier launch of this context may still read the old table
    if (rc) return rc;
    if ((rc = ctx->ctab.reserve(tab.size() * sizeof(float)))) return rc;
    MC_HIP_TRY(hipMemcpyAsync(ctx->ctab.ptr, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice, s));
    MC_HIP_TRY(hipStreamSynchronize(s));
    ctx->ctab_key = key;
    return MC_OK;
}

// The device colour table of (max_iter, k_color), uploaded on first use (mandelbrot_assemble_launch, postprocess.hip).
int mandelbrot_lut_device(mc_context* ctx, const mc_mandelbrot_params* p, hipStream_t s, const void** d_lut) {
    int rc = ensure_lut(ctx, p, s);
    if (rc) return rc;
    *d_lut = ctx->lut.ptr;
    return MC_OK;
}

// warm = the cold-start warm-up (mc_context_warmup_mandelbrot): the tables of the REAL request are built and uploaded, then ONE 8 x 8
// tile is run for at most 32 iterations into d_iters — enough for the runtime to load this code object and create the kernel.
static int launch_impl(mc_context* ctx, const mc_mandelbrot_params* p, void* d_rgba, void* d_iters, hipStream_t s, bool warm) {
    if (!ctx || !p || (!d_rgba && !d_iters)) return MC_ERR_INVALID_ARGUMENT;
    if (!p->width || !p->height || !p->max_iter || p->row_end > p->height || p->row_begin >= p->row_end)
        return MC_ERR_INVALID_ARGUMENT;
    if (p->precision != MC_PRECISION_F32 && p->precision != MC_PRECISION_DS) return MC_ERR_INVALID_ARGUMENT;
    if (p->row_stride && (!p->row_block || p->row_block > p->row_stride)) return MC_ERR_INVALID_ARGUMENT;
    if (d_rgba || warm) {
        int rc = ensure_lut(ctx, p, s);
        if (rc) return rc;
    }
    {
        int rc = ensure_c_table(ctx, p, s);
        if (rc) return rc;
    }
    MandelArgs a;
    a.c_tab = (const float*)ctx->ctab.ptr;
    a.W = p->width; a.H = p->height; a.max_iter = p->max_iter;
    a.row_begin = p->row_begin; a.row_end = p->row_end;
    a.row_block = p->row_stride ? p->row_block : 0u; a.row_stride = p->row_stride;
    a.cx_hi = p->centre_x_hi; a.cx_lo = p->centre_x_lo; a.cy_hi = p->centre_y_hi; a.cy_lo = p->centre_y_lo;
    a.sx_hi = p->scale_x_hi; a.sx_lo = p->scale_x_lo; a.sy_hi = p->scale_y_hi; a.sy_lo = p->scale_y_lo;
    a.out_rgba = warm ? nullptr : (float4*)d_rgba;
    const bool narrow = (p->flags & MC_MANDEL_ITERS_U16) != 0u;
    if (narrow && p->max_iter > 65535u) return MC_ERR_INVALID_ARGUMENT;
    a.out_iters = narrow ? nullptr : (uint32_t*)d_iters;
    a.out_iters16 = narrow ? (uint16_t*)d_iters : nullptr;
    a.lut = a.out_rgba ? (const float4*)ctx->lut.ptr : nullptr;
    const uint32_t rows = tile_rows(p->row_begin, p->row_end, a.row_block, a.row_stride);
    dim3 grid((p->width + 7u) / 8u, (rows + 7u) / 8u), block(64);
    if (warm) {   // one tile, a handful of iterations; d_iters holds at least rows x W counts (the caller's scratch)
        grid = dim3(1, 1);
        a.max_iter = p->max_iter < 32u ? p->max_iter : 32u;
    }
    if (p->precision == MC_PRECISION_DS) {
        hipLaunchKernelGGL((mandelbrot_kernel<StateDS, 4>), grid, block, 0, s, a);
    } else if (p->flags & MC_MANDEL_FMA) {
        hipLaunchKernelGGL((mandelbrot_kernel<StateF32<true>, 8>), grid, block, 0, s, a);
    } else {
        hipLaunchKernelGGL((mandelbrot_kernel<StateF32<false>, 8>), grid, block, 0, s, a);
    }
    MC_HIP_TRY(hipGetLastError());
    return ctx->note_launch(s);
}

int mandelbrot_launch(mc_context* ctx, const mc_mandelbrot_params* p, void* d_rgba, void* d_iters, hipStream_t s) {
    return launch_impl(ctx, p, d_rgba, d_iters, s, false);
}

int mandelbrot_warmup(mc_context* ctx, const mc_mandelbrot_params* p, void* d_iters_scratch, hipStream_t s) {
    return launch_impl(ctx, p, nullptr, d_iters_scratch, s, true);
}

}  // namespace mc
