// Path tracer, "two path slots per lane" variant for gfx950 — same arithmetic as pathtrace_kernel.h, different
// scheduling of the lanes.  Measured divergence of the round-synchronous kernel (tools/pt_region_stats.py): the
// glass / mirror blocks run ~19x per sample round with 3-5 active lanes, and Russian roulette leaves the second
// half of every round half empty.  Here a wave owns P = 8 pixels and ALL their samples; every lane carries two
// path states, an active one (A) and a parked one (B):
//   * BOUNCE phase: lanes whose A path is runnable do intersect + prologue + diffuse shading (the common case);
//   * a path that hits a specular surface or dies is parked and the lane swaps in its other path (14 v_swap_b32);
//   * SPEC phase: runs only when >= kSpecThreshold lanes hold a specular-pending path (or nothing else can run) and
//     processes all of them at once; REGEN phase likewise batches "deposit result, fetch next (pixel, sample),
//     generate camera ray" for finished paths.
// Samples therefore finish out of order.  The fp32 accumulation order is part of the parity contract
// (pathTracer.comp:451-452), so each pixel has a reorder ring in LDS: a finished sample deposits
// accrad/spp tagged with its index, and the pixel's owner lane folds the ring into the accumulator strictly in
// sample order.  Every per-sample value is computed by the same expressions as in pathtrace_kernel.h, so the
// output is bit-identical to it (and, in strict mode, to the oracle) — tests/test_gpu_parity.py.
#pragma once
#include "pathtrace_kernel.h"

namespace mc {
namespace pt {

constexpr uint32_t kPqPixels = 8;         // pixels per wave: 4 x 2
constexpr uint32_t kPqTileW = 4, kPqTileH = 2;
constexpr uint32_t kPqRing = 64;          // reorder window per pixel (samples): long paths time-share a lane, so
                                          // a sample can stay in flight while ~30 later ones of its pixel finish
// PTArgs::pq_regen_threshold / pq_spec_threshold: lanes with a finished / specular-pending path before a REGEN /
// SPEC phase is worth running (defaults set on the host, tunable through MC_PT_PQ_THRESHOLDS for experiments).

enum : uint32_t { PQ_RUN = 0, PQ_SPEC = 1, PQ_DEAD = 2, PQ_FRESH = 3, PQ_EMPTY = 4 };

// One path.  st packs: bits 0-2 state, bit 3 emissive flag, bits 4-7 depth, bits 8-15 hit id (specular-pending).
// key packs: bits 0-2 pixel of the wave tile, bits 3-31 sample index (samps.x).
struct PqPath {
    v3 ro, rd, accmat, accrad;
    uint32_t key, st;
};

__device__ __forceinline__ uint32_t pq_state(const PqPath& p) { return p.st & 7u; }
__device__ __forceinline__ uint32_t pq_depth(const PqPath& p) { return (p.st >> 4) & 15u; }
__device__ __forceinline__ void pq_set(PqPath& p, uint32_t state, uint32_t emissive, uint32_t depth, uint32_t id) {
    p.st = state | (emissive << 3) | (depth << 4) | (id << 8);
}

// Exchanges the two path states of the lanes active in the enclosing branch (v_swap_b32 obeys EXEC).
__device__ __forceinline__ void pq_swap(PqPath& a, PqPath& b) {
#define MC_PQ_SWAPF(x, y) asm volatile("v_swap_b32 %0, %1" : "+v"(x), "+v"(y))
    MC_PQ_SWAPF(a.ro.x, b.ro.x); MC_PQ_SWAPF(a.ro.y, b.ro.y); MC_PQ_SWAPF(a.ro.z, b.ro.z);
    MC_PQ_SWAPF(a.rd.x, b.rd.x); MC_PQ_SWAPF(a.rd.y, b.rd.y); MC_PQ_SWAPF(a.rd.z, b.rd.z);
    MC_PQ_SWAPF(a.accmat.x, b.accmat.x); MC_PQ_SWAPF(a.accmat.y, b.accmat.y); MC_PQ_SWAPF(a.accmat.z, b.accmat.z);
    MC_PQ_SWAPF(a.accrad.x, b.accrad.x); MC_PQ_SWAPF(a.accrad.y, b.accrad.y); MC_PQ_SWAPF(a.accrad.z, b.accrad.z);
    MC_PQ_SWAPF(a.key, b.key); MC_PQ_SWAPF(a.st, b.st);
#undef MC_PQ_SWAPF
}

template <bool Fast>
__global__ void __launch_bounds__(256) pathtrace_pq_kernel(PTArgs a) {
    constexpr int NP = 6, NS = 3;
    // dynamic LDS: [scene records | per-wave reorder rings]
    extern __shared__ float lds_dyn[];
    float* lds_obj = lds_dyn;
    constexpr uint32_t kObjFloats = (NP + NS) * 12u;
    for (uint32_t i = threadIdx.x; i < kObjFloats; i += blockDim.x) lds_obj[i] = a.scene.obj[i];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float4* ring = reinterpret_cast<float4*>(lds_dyn + kObjFloats + 4u) + (size_t)wave * kPqPixels * kPqRing;   // 16-B aligned
    for (uint32_t i = lane; i < kPqPixels * kPqRing; i += 64u) ring[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // tag 0 = empty
    __syncthreads();
    const SceneArgs& sc = a.scene;
    const float* __restrict__ uobj = sc.obj;

    // ---- the wave's pixel tile (4 x 2 pixels); all of it is wave-uniform ----
    const uint32_t tile_x0 = (blockIdx.x * 2u + (wave & 1u)) * kPqTileW;
    const uint32_t tile_y0 = (blockIdx.y * 2u + (wave >> 1)) * kPqTileH;      // tile-local storage row of pixel row 0
    uint32_t row_r[kPqTileH];
    bool row_ok[kPqTileH];
#pragma unroll
    for (uint32_t j = 0; j < kPqTileH; j++) {
        row_r[j] = tile_row_to_storage(tile_y0 + j, a.row_begin, a.row_block, a.row_stride);
        row_ok[j] = row_r[j] < a.row_end;
    }
    const uint32_t n_samples = a.sample_end - a.sample_begin;
    const uint32_t total_items = kPqPixels * n_samples;      // item q -> pixel q % 8, sample sample_begin + q / 8
    uint32_t next_item = 0;                                   // wave-uniform
    const float fspp = (float)a.spp;

    // ---- commit state of the pixel this lane owns (lanes 0..7) ----
    const uint32_t own_p = lane & 7u;
    const uint32_t own_gx = tile_x0 + (own_p & 3u);
    const bool own_valid = lane < kPqPixels && own_gx < a.W && row_ok[own_p >> 2];
    const size_t own_idx = (size_t)(tile_y0 + (own_p >> 2)) * a.W + own_gx;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (own_valid && a.sample_begin > 0) acc = a.out[own_idx];   // progressive continuation (samps.x protocol)
    uint32_t next_commit = a.sample_begin;                        // next sample index this pixel folds in

    PqPath A, B;
    A.ro = A.rd = A.accmat = A.accrad = v3{0.0f, 0.0f, 0.0f}; A.key = 0u; pq_set(A, PQ_FRESH, 0u, 0u, 0u);
    B = A;

    // Exit condition every wave reaches: the scheduler below makes progress in every iteration (a path advances a
    // bounce, is deposited, regenerated or finished), and a hard bound far above the worst case (every item needs at
    // most max_depth bounce phases plus as many specular/regeneration phases, two slots per lane) backs that up so a
    // logic error can never leave a wave spinning on the device.
    const uint32_t guard_limit = (total_items / 64u + 2u) * (a.max_depth + 4u) * 8u + 4096u;
    for (uint32_t guard = 0; guard < guard_limit; guard++) {
        MC_REGION(9);    // scheduler iteration
        // ------------------------------------------------------------------ votes (scalar)
        const uint32_t sa = pq_state(A), sb = pq_state(B);
        const bool items_left = next_item < total_items;
        const bool want_regen = sa == PQ_DEAD || sb == PQ_DEAD || (items_left && (sa == PQ_FRESH || sb == PQ_FRESH));
        const bool want_spec = sa == PQ_SPEC || sb == PQ_SPEC;
        const unsigned long long m_run = __ballot(sa == PQ_RUN || sb == PQ_RUN);
        const unsigned long long m_regen = __ballot(want_regen);
        const unsigned long long m_spec = __ballot(want_spec);
        if ((m_run | m_regen | m_spec) == 0ull) break;           // every path finished, every item issued

        // ------------------------------------------------------------------ REGEN phase (batched)
        if (m_regen && ((uint32_t)__popcll(m_regen) >= a.pq_regen_threshold || m_run == 0ull)) {
            if (want_regen) { MC_REGION(10); }   // REGEN phase: lanes taking part
            if (want_regen) {
                // bring the slot to work on into A: prefer a DEAD one (its result must be deposited)
                const bool a_is_it = sa == PQ_DEAD || (sb != PQ_DEAD && sa == PQ_FRESH);
                if (!a_is_it) pq_swap(A, B);
                if (pq_state(A) == PQ_DEAD) {
                    // deposit accrad / spp (pathTracer.comp:452) tagged with sample+1 into the pixel's reorder ring
                    const uint32_t p = A.key & 7u, s = A.key >> 3;
                    v3 q = divs<Fast>(A.accrad, fspp);
                    ring[p * kPqRing + (s % kPqRing)] = make_float4(q.x, q.y, q.z, __uint_as_float(s + 1u));
                    pq_set(A, PQ_FRESH, 0u, 0u, 0u);
                }
            }
            // fold the rings into the accumulators, strictly in sample order (lanes 0..7, one pixel each)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < kPqPixels) {
                for (uint32_t k = 0; k < kPqRing; k++) {
                    const float4 e = ring[own_p * kPqRing + (next_commit % kPqRing)];
                    if (__float_as_uint(e.w) != next_commit + 1u) break;
                    if (next_commit == 0u) acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);     // :451
                    acc.x += e.x; acc.y += e.y; acc.z += e.z; acc.w += 0.0f;              // :452
                    next_commit++;
                }
            }
            // hand out new (pixel, sample) items in order, limited by the reorder window of the slowest pixel
            uint32_t min_commit = __shfl(next_commit, 0);
#pragma unroll
            for (int p = 1; p < (int)kPqPixels; p++) min_commit = min(min_commit, (uint32_t)__shfl(next_commit, p));
            const uint32_t window_end_item = (min_commit - a.sample_begin + kPqRing) * kPqPixels;   // first item NOT allowed
            const uint32_t limit = min(total_items, window_end_item);
            const bool pull = want_regen && pq_state(A) == PQ_FRESH;
            const unsigned long long m_pull = __ballot(pull);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_pull >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_pull, 0u));
            const uint32_t item = next_item + rank;
            const bool got = pull && item < limit;
            const uint32_t n_got = min((uint32_t)__popcll(m_pull), limit > next_item ? limit - next_item : 0u);
            next_item += n_got;
            if (pull && !got && next_item >= total_items) pq_set(A, PQ_EMPTY, 0u, 0u, 0u);   // nothing left to do for this slot
            if (got) {
                MC_REGION(0);    // ray generation
                const uint32_t p = item & 7u, samp = a.sample_begin + (item >> 3);
                const uint32_t gx = tile_x0 + (p & 3u);
                const bool pvalid = gx < a.W && ((p >> 2) ? row_ok[1] : row_ok[0]);
                const uint32_t gy = a.H - 1u - ((p >> 2) ? row_r[1] : row_r[0]);               // :349
                A.key = p | (samp << 3);
                A.accrad = v3{0.0f, 0.0f, 0.0f};
                if (!pvalid) {
                    pq_set(A, PQ_DEAD, 0u, 0u, 0u);      // pixel outside the image: contributes a zero that nobody reads
                } else {
                    // -- sample sensor (pathTracer.comp:357-362), identical to trace_sample()
                    v3 r0 = rand01(gx, gy, samp);
                    float rnd2x = 2.0f * r0.x, rnd2y = 2.0f * r0.y;
                    float tentx = rnd2x < 1.0f ? dm::fsqrt<Fast>(rnd2x) - 1.0f : 1.0f - dm::fsqrt<Fast>(2.0f - rnd2x);
                    float tenty = rnd2y < 1.0f ? dm::fsqrt<Fast>(rnd2y) - 1.0f : 1.0f - dm::fsqrt<Fast>(2.0f - rnd2y);
                    float stratx = (float)((samp / 2u) % 2u), straty = (float)(samp % 2u);
                    float sx = (dm::fdiv<Fast>((float)gx + 0.5f * ((0.5f + stratx) + tentx), (float)a.W) - 0.5f) * 0.036f;
                    float sy = (dm::fdiv<Fast>((float)gy + 0.5f * ((0.5f + straty) + tenty), (float)a.H) - 0.5f) * 0.024f;
                    v3 spos = (a.cam_o + a.cx * sx) + a.cy * sy;
                    A.accmat = v3{1.0f, 1.0f, 1.0f};
                    A.ro = a.lc;
                    A.rd = normalize<Fast>(a.lc - spos);
                    pq_set(A, a.max_depth ? PQ_RUN : PQ_DEAD, 1u, 0u, 0u);
                }
            }
        }

        // ------------------------------------------------------------------ SPEC phase (batched)
        {
            const uint32_t sa2 = pq_state(A), sb2 = pq_state(B);
            const bool has_spec = sa2 == PQ_SPEC || sb2 == PQ_SPEC;
            const unsigned long long ms = __ballot(has_spec);
            const unsigned long long mr = __ballot(sa2 == PQ_RUN || sb2 == PQ_RUN);
            if (ms && ((uint32_t)__popcll(ms) >= a.pq_spec_threshold || mr == 0ull)) {
                if (has_spec) {
                    MC_REGION(11);   // SPEC phase
                    if (sa2 != PQ_SPEC) pq_swap(A, B);
                    const uint32_t p = A.key & 7u, samp = A.key >> 3, depth = pq_depth(A);
                    const int id = (int)((A.st >> 8) & 0xffu);
                    const uint32_t gx = tile_x0 + (p & 3u);
                    const uint32_t gy = a.H - 1u - ((p >> 2) ? row_r[1] : row_r[0]);
                    const v3 x = A.ro, rd = A.rd;                         // ro already holds the hit point
                    const float* obj = lds_obj + 12 * id;
                    v3 geo{obj[0], obj[1], obj[2]};
                    const int mat = (int)__builtin_floorf(obj[11] + 0.5f);
                    v3 n = id >= NP ? normalize<Fast>(x - geo) : geo;     // same expression as in the bounce prologue
                    v3 nl = dot(n, rd) < 0.0f ? n : -n;
                    v3 rnd = rand01(gx, gy, samp * a.max_depth + depth);  // :393 (recomputed: cheaper than parking it)
                    v3 nrd = rd;
                    if (mat == 2) {                                       // :432 mirror
                        nrd = reflect(rd, n);
                    } else {                                              // :437 glass
                        bool into = (n.x == nl.x) && (n.y == nl.y) && (n.z == nl.z);
                        const float nc = 1.0f, nt = 1.5f;
                        float nnt = into ? dm::fdiv<Fast>(nc, nt) : dm::fdiv<Fast>(nt, nc);
                        float ddn = dot(rd, nl);
                        float cos2t = 1.0f - (nnt * nnt) * (1.0f - ddn * ddn);
                        v3 refl = reflect(rd, n);
                        if (cos2t >= 0.0f) {
                            float k = (into ? 1.0f : -1.0f) * (ddn * nnt + dm::fsqrt<Fast>(cos2t));
                            v3 tdir = normalize<Fast>(rd * nnt - n * k);
                            float aa = nt - nc, bb = nt + nc;
                            float R0 = dm::fdiv<Fast>(aa * aa, bb * bb);
                            float c = 1.0f - (into ? -ddn : dot(tdir, n));
                            float Re = R0 + (((((1.0f - R0) * c) * c) * c) * c) * c;
                            float Tr = 1.0f - Re;
                            float Pr = 0.25f + 0.5f * Re;
                            float RP = dm::fdiv<Fast>(Re, Pr), TP = dm::fdiv<Fast>(Tr, 1.0f - Pr);
                            bool pick_refl = rnd.x < Pr;
                            nrd = select(pick_refl, refl, tdir);
                            A.accmat = A.accmat * (pick_refl ? RP : TP);
                        } else {
                            nrd = refl;
                        }
                    }
                    A.rd = nrd;
                    const uint32_t nd = depth + 1u;
                    pq_set(A, nd >= a.max_depth ? PQ_DEAD : PQ_RUN, 1u, nd, 0u);          // emissive = 1 (:434,:447)
                }
            }
        }

        // ------------------------------------------------------------------ make A the runnable slot
        // ... and when both are runnable, the OLDER sample: a freshly generated path must not pre-empt the one that
        // was running, or old samples starve in the parked slot, their pixel's commit pointer stalls and the reorder
        // window closes (no new items can be issued).
        {
            const bool b_run = pq_state(B) == PQ_RUN;
            const bool a_run = pq_state(A) == PQ_RUN;
            if (b_run && (!a_run || (B.key >> 3) < (A.key >> 3))) { MC_REGION(14); pq_swap(A, B); }
        }

#ifdef MC_PT_REGION_STATS
        if (pq_state(A) != PQ_RUN) {   // idle lane in this BOUNCE phase: what is it waiting for?
            const uint32_t x = pq_state(A), y = pq_state(B);
            if (x == PQ_SPEC || y == PQ_SPEC) { MC_REGION(12); }
            if (x == PQ_DEAD || y == PQ_DEAD) { MC_REGION(13); }
            if ((x == PQ_FRESH || x == PQ_EMPTY) && (y == PQ_FRESH || y == PQ_EMPTY)) { MC_REGION(15); }
        }
#endif
        // ------------------------------------------------------------------ BOUNCE phase
        if (pq_state(A) == PQ_RUN) {
            MC_REGION(1);    // BOUNCE phase: intersect + prologue
            const uint32_t p = A.key & 7u, samp = A.key >> 3, depth = pq_depth(A);
            const uint32_t gx = tile_x0 + (p & 3u);
            const uint32_t gy = a.H - 1u - ((p >> 2) ? row_r[1] : row_r[0]);
            float emissive = (float)((A.st >> 3) & 1u);
            v3 ro = A.ro, rd = A.rd;
            float t;
            int id = intersect<Fast, NP, NS, true, 0>(sc, uobj, ro, rd, t);
            if (id < 0) {
                pq_set(A, PQ_DEAD, 0u, 0u, 0u);           // :369 a miss repeats at every later depth: the path is over
            } else {
                v3 x = ro + rd * t;                                               // :374
                const float* obj = lds_obj + 12 * id;
                const bool is_sphere = id >= NP;
                v3 geo{obj[0], obj[1], obj[2]};
                v3 emi{obj[4], obj[5], obj[6]};
                v3 col{obj[8], obj[9], obj[10]};
                int mat = (int)__builtin_floorf(obj[11] + 0.5f);                  // :378/:384
                v3 n = is_sphere ? normalize<Fast>(x - geo) : geo;                // :381/:387
                v3 nl = dot(n, rd) < 0.0f ? n : -n;                               // :390
                A.accrad = A.accrad + (A.accmat * emi) * emissive;                // :391
                A.accmat = A.accmat * col;                                        // :392
                v3 rnd = rand01(gx, gy, samp * a.max_depth + depth);              // :393
                float pr = dm::gmax(dm::gmax(col.x, col.y), col.z);              // :394
                bool dead = false;
                if (depth > 5u) {                                                 // :395
                    if (rnd.z >= pr) dead = true;                                 // :396
                    else A.accmat = divs<Fast>(A.accmat, pr);                     // :397
                }
                if (dead) {
                    pq_set(A, PQ_DEAD, 0u, 0u, 0u);
                } else if (mat == 1) {                                            // :400 diffuse
                    MC_REGION(3);
#pragma unroll
                    for (int i = 0; i < NS; i++) {                                // :403
                        if (!((sc.emissive_mask >> i) & 1u)) continue;            // :407 (uniform)
                        const float* ls = uobj + 12 * (NP + i);
                        v3 le{ls[4], ls[5], ls[6]};
                        v3 xc = v3{ls[0], ls[1], ls[2]} - x;                      // :408
                        v3 sw = normalize<Fast>(xc);                              // :409
                        v3 su = normalize<Fast>(cross((__builtin_fabsf(sw.x) > 0.1f ? v3{0, 1, 0} : v3{1, 0, 0}), sw));
                        v3 sv = cross(sw, su);
                        float cos_a_max = dm::fsqrt<Fast>(1.0f - dm::fdiv<Fast>(sc.r2[i], dot(xc, xc)));   // :410
                        float cos_a = (1.0f - rnd.x) + rnd.x * cos_a_max;         // :411
                        float sin_a = dm::fsqrt<Fast>(1.0f - cos_a * cos_a);
                        float phi = (2.0f * kPi) * rnd.y;                         // :412
                        float sphi, cphi;
                        dm::sincos_angle<Fast>(phi, rnd.y, sphi, cphi);
                        v3 l = normalize<Fast>(((su * cphi) * sin_a + (sv * sphi) * sin_a) + sw * cos_a);   // :413
                        float tne;
                        int idne = intersect<Fast, NP, NS, true, 0>(sc, uobj, x, l, tne, sc.nee_skip_planes != 0u);   // :420 shadow ray
                        if (idne == NP + i) {
                            float omega = (2.0f * kPi) * (1.0f - cos_a_max);      // :421
                            A.accrad = A.accrad + ((divs<Fast>(A.accmat, kPi) * dm::gmax(dot(l, nl), 0.0f)) * le) * omega;   // :422
                        }
                    }
                    float r1 = (2.0f * kPi) * rnd.x, r2 = rnd.y, r2s = dm::fsqrt<Fast>(r2);   // :426
                    v3 w = nl;
                    v3 u = normalize<Fast>(cross((__builtin_fabsf(w.x) > 0.1f ? v3{0, 1, 0} : v3{1, 0, 0}), w));   // :427
                    v3 v = cross(w, u);
                    float s1, c1;
                    dm::sincos_angle<Fast>(r1, rnd.x, s1, c1);
                    A.rd = normalize<Fast>(((u * c1) * r2s + (v * s1) * r2s) + w * dm::fsqrt<Fast>(1.0f - r2));   // :428
                    A.ro = x;
                    const uint32_t nd = depth + 1u;
                    pq_set(A, nd >= a.max_depth ? PQ_DEAD : PQ_RUN, 0u, nd, 0u);  // emissive = 0 (:429)
                } else if (mat == 2 || mat == 3) {
                    A.ro = x;                                                     // park: the SPEC phase finishes this bounce
                    pq_set(A, PQ_SPEC, (A.st >> 3) & 1u, depth, (uint32_t)id);
                } else {
                    // unknown material: the shader changes neither the ray nor `emissive` (:400-448) and goes on
                    const uint32_t nd = depth + 1u;
                    pq_set(A, nd >= a.max_depth ? PQ_DEAD : PQ_RUN, (A.st >> 3) & 1u, nd, 0u);
                }
            }
        }
    }

    // every sample has been deposited and folded in (the last REGEN phase ran with nothing else runnable)
    if (own_valid) {
        if (a.sample_end == a.spp) {                                                // :453 after sample spp-1
            acc.x = dm::fpow<Fast>(dm::gmin(dm::gmax(acc.x, 0.0f), 1.0f), 0.45f) * 255.0f + 0.5f;
            acc.y = dm::fpow<Fast>(dm::gmin(dm::gmax(acc.y, 0.0f), 1.0f), 0.45f) * 255.0f + 0.5f;
            acc.z = dm::fpow<Fast>(dm::gmin(dm::gmax(acc.z, 0.0f), 1.0f), 0.45f) * 255.0f + 0.5f;
        }
        a.out[own_idx] = acc;
    }
}

inline size_t pq_lds_bytes() { return ((size_t)(6 + 3) * 12u + 4u) * sizeof(float) + (size_t)4 * kPqPixels * kPqRing * sizeof(float4); }

template <bool Fast>
inline void launch_pq(const PTArgs& a, uint32_t tile_rows, hipStream_t s) {
    dim3 grid((a.W + 2u * kPqTileW - 1u) / (2u * kPqTileW), (tile_rows + 2u * kPqTileH - 1u) / (2u * kPqTileH));
    hipLaunchKernelGGL((pathtrace_pq_kernel<Fast>), grid, dim3(256), pq_lds_bytes(), s, a);
}

}  // namespace pt
}  // namespace mc
