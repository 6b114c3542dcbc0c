// Path tracer, lane-regrouping scheduler for gfx950 — same per-path arithmetic as pathtrace_kernel.h, different
// assignment of paths to lanes.  Included by pathtrace_fast.hip / pathtrace_strict.hip.
//
// Why (profiles/r01d_pt_fast_pmc_summary.json, tools/sched_sim.cpp): the round-synchronous kernel is VALU-issue bound
// and only 37 of 64 lanes are active per issued instruction — Russian roulette thins every round after depth 5, the
// 320-instruction diffuse block runs with the ~80 % of the lanes that hit a diffuse surface, and the glass / mirror blocks
// run in almost every iteration for 3-5 lanes.  gfx950 does not skip a half-empty wave64 pass
// (profiles/r02_exec_microbench.txt), so the unit to fill is the whole wave.
//
// How: a workgroup of NW = 4 waves owns a 4x4 pixel tile and ALL its samples (items: sample-major, so the pixels advance
// together).  Paths move between lanes through two small workgroup-shared FIFOs in LDS:
//   * SQ: paths whose next step is a specular bounce (mirror / glass), parked right after intersect + prologue;
//   * DQ: paths whose next step is the diffuse block ("D-ready"), spilled by a wave that is about to run a batch.
// Every scheduler iteration of a wave is  [swap point] -> heads (diffuse | mirror | glass | camera) -> intersect + prologue:
//   * at the swap point the wave parks its specular lanes in SQ and refills its empty lanes with D-ready paths from DQ, so
//     the diffuse block and intersect + prologue run with (nearly) all 64 lanes;
//   * when DQ cannot fill the vacancies and SQ + fresh camera samples amount to most of a wave, the wave spills its own
//     D-ready lanes to DQ (the other waves' vacancies drain them) and runs a BATCH: 64 lanes of parked specular paths and
//     new camera rays, whose heads are cheap and whose intersect + prologue is again full width;
//   * whenever a queue is full or too empty for that, the lanes simply keep what they have and the iteration runs mixed,
//     exactly like the round-synchronous kernel — correctness never depends on the policy, only utilisation does.
// Samples therefore finish out of order, in any wave.  The fp32 accumulation order is part of the parity contract
// (pathTracer.comp:451-452): a finished sample deposits accrad/spp in the workgroup's reorder ring (LDS,
// indexed by item), and whichever wave holds the commit lock folds the ring into the per-pixel accumulators strictly in
// sample order; new camera samples are only handed out inside the ring's window.  Every per-sample value is computed by
// the same expressions as in pathtrace_kernel.h, so strict mode stays bit-identical to the oracle.
//
// Synchronisation: LDS only, no barriers after start-up.  The FIFOs are bounded multi-producer / multi-consumer rings with
// a sequence word per slot (ticket t is written when seq == t, read when seq == t+1, freed with seq = t+CAP); tickets are
// reserved by one lane with a compare-and-swap, so a reservation never exceeds what exists / fits.  Every wait is on a
// wave that is between its reservation and its data access (never itself waiting), every loop is bounded, and a tripped
// bound raises the context's status word (the host returns MC_ERR_HIP) instead of leaving a wave spinning.
#pragma once
#include "pathtrace_kernel.h"

namespace mc {
namespace pt {

enum : uint32_t { RG_EMPTY = 0, RG_D = 1, RG_M = 2, RG_G = 3, RG_CAM = 4 };

template <int NW> struct RgGeom;
template <> struct RgGeom<4> {
    static constexpr uint32_t PB = 16, PW = 4, PH = 4;        // pixels per workgroup
    static constexpr uint32_t DQ = 64, SQ = 64, WIN = 512;    // FIFO capacities (records), reorder window (items)
};
template <> struct RgGeom<1> {
    static constexpr uint32_t PB = 4, PW = 2, PH = 2;
    static constexpr uint32_t DQ = 64, SQ = 64, WIN = 256;
};
constexpr uint32_t kRgBatchMin = 48;          // parked specular + available camera items that justify a batch
constexpr uint32_t kRgSpinLimit = 1u << 22;   // bound of every wait loop (a healthy wait is a few hundred cycles)
constexpr uint32_t kRgMaxDepth = 63;          // meta packs the depth into 6 bits
constexpr uint32_t kRgMaxItems = 1u << 24;    // ... and the item index into 24

// meta: bits 0-23 item (sample-major: item = (samp - sample_begin) * PB + pixel), bits 24-29 depth, bit 30-31 unused
__device__ __forceinline__ uint32_t rg_item(uint32_t meta) { return meta & 0xffffffu; }
__device__ __forceinline__ uint32_t rg_depth(uint32_t meta) { return (meta >> 24) & 63u; }

template <int NW>
struct RgShared {
    using G = RgGeom<NW>;
    float obj[(6 + 3) * 12];                   // scene records for the per-lane material fetch
    uint32_t dq_head, dq_tail, sq_head, sq_tail;
    uint32_t next_item, committed, commit_lock, waves_done;
    uint32_t pixkey[G::PB];                    // gx | gy << 16 of the workgroup's pixels, 0xffffffff outside the image
    float4 acc[G::PB];                         // xyz accumulator, w = bits of the number of samples folded in so far
    uint32_t dq_seq[G::DQ], sq_seq[G::SQ];
    float4 dq_rec[4][G::DQ];                   // (x, rnd.x) (nl, rnd.y) (accmat, meta) (accrad, pixkey)
    float4 sq_rec[5][G::SQ];                   // (x, rnd.x) (rd, kind) (accmat, meta) (accrad, pixkey) (n, -)
    float4 ring[G::WIN];                       // xyz = accrad / spp of a finished sample awaiting its turn; w = bits of
                                               // item + 1, written (as its own dword) after xyz is in place
};

__device__ __forceinline__ uint32_t rg_ld(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void rg_st(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void rg_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
__device__ __forceinline__ void rg_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
__device__ __forceinline__ uint32_t rg_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t rg_rank(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Reserves n tickets for a push (lane 0 decides; wave-uniform result): the base ticket, or 0xffffffff when the FIFO cannot
// take n more records.  `head` only grows, so a stale value is conservative.
__device__ __forceinline__ uint32_t rg_reserve_push(uint32_t* tail, const uint32_t* head, uint32_t n, uint32_t cap, uint32_t lane) {
    uint32_t base = 0xffffffffu;
    if (lane == 0) {
        uint32_t t = rg_ld(tail);
        for (int k = 0; k < 64; k++) {
            const uint32_t h = rg_ld(head);
            if (t - h + n > cap) break;
            const uint32_t o = atomicCAS(tail, t, t + n);
            if (o == t) { base = t; break; }
            t = o;
        }
    }
    return rg_uniform(base);
}
// Reserves up to n tickets for a pop: returns the base ticket and the number granted (0 when the FIFO is empty).
__device__ __forceinline__ uint32_t rg_reserve_pop(uint32_t* head, const uint32_t* tail, uint32_t n, uint32_t lane, uint32_t& granted) {
    uint32_t base = 0, got = 0;
    if (lane == 0) {
        uint32_t h = rg_ld(head);
        for (int k = 0; k < 64; k++) {
            const uint32_t t = rg_ld(tail);
            const uint32_t avail = t - h;                    // tail is read after head: never behind it
            const uint32_t g = avail < n ? avail : n;
            if (g == 0u || avail > 0x7fffffffu) break;
            const uint32_t o = atomicCAS(head, h, h + g);
            if (o == h) { base = h; got = g; break; }
            h = o;
        }
    }
    granted = rg_uniform(got);
    return rg_uniform(base);
}
// Waits (bounded) until the sequence word of this lane's slot shows `expect`; inactive lanes pass `on = false`.
__device__ __forceinline__ bool rg_wait_seq(const uint32_t* seq, uint32_t expect, bool on) {
    for (uint32_t spins = 0; spins < kRgSpinLimit; spins++) {
        const bool ready = !on || rg_ld(seq) == expect;
        if (__ballot(!ready) == 0ull) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// State of the path a lane holds between two scheduler iterations (= one FIFO record).
struct RgPath {
    v3 X;          // RG_D / RG_M / RG_G: the hit point x (the next ray origin); RG_CAM: unused
    v3 V;          // RG_D: nl; RG_M / RG_G: the incoming ray direction
    v3 N;          // RG_M / RG_G: the surface normal n
    v3 accmat, accrad;
    float rx, ry;  // rnd.x, rnd.y of the bounce (pathTracer.comp:393)
    uint32_t meta, pixkey;
};

template <bool Fast, int NW>
__global__ void __launch_bounds__(64 * NW, 6) pathtrace_regroup_kernel(PTArgs a) {
    using G = RgGeom<NW>;
    constexpr int NP = 6, NS = 3;
    __shared__ RgShared<NW> sh;
    const uint32_t lane = threadIdx.x & 63u;
    const SceneArgs& sc = a.scene;
    const float* __restrict__ uobj = sc.obj;
    const float* lds_obj = sh.obj;

    // ---- start-up (the only barrier) ----
    for (uint32_t i = threadIdx.x; i < (NP + NS) * 12u; i += 64u * NW) sh.obj[i] = sc.obj[i];
    for (uint32_t i = threadIdx.x; i < G::DQ; i += 64u * NW) sh.dq_seq[i] = i;
    for (uint32_t i = threadIdx.x; i < G::SQ; i += 64u * NW) sh.sq_seq[i] = i;
    for (uint32_t i = threadIdx.x; i < G::WIN; i += 64u * NW) sh.ring[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // tag 0: empty
    if (threadIdx.x == 0) {
        sh.dq_head = sh.dq_tail = sh.sq_head = sh.sq_tail = 0u;
        sh.next_item = sh.committed = sh.commit_lock = sh.waves_done = 0u;
    }
    const uint32_t tile_x0 = blockIdx.x * G::PW, tile_y0 = blockIdx.y * G::PH;   // tile-local storage rows
    if (threadIdx.x < G::PB) {
        const uint32_t p = threadIdx.x;
        const uint32_t gx = tile_x0 + p % G::PW, ty = tile_y0 + p / G::PW;
        const uint32_t r = tile_row_to_storage(ty, a.row_begin, a.row_block, a.row_stride);
        const bool valid = gx < a.W && r < a.row_end;                               // pathTracer.comp:348
        sh.pixkey[p] = valid ? (gx | ((a.H - 1u - r) << 16)) : 0xffffffffu;          // :349 gid = (H-1-y)*W + x
        float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (valid && a.sample_begin > 0) acc = a.out[(size_t)ty * a.W + gx];         // progressive continuation; s == 0 resets (:451)
        acc.w = __uint_as_float(0u);
        sh.acc[p] = acc;
    }
    __syncthreads();

    const uint32_t n_samples = a.sample_end - a.sample_begin;
    const uint32_t total_items = G::PB * n_samples;
    const float fspp = (float)a.spp;
    bool failed = false;

    // Folds finished samples into the accumulators in sample order, if no other wave is doing so (wave-uniform call).
    auto try_commit = [&]() {
        uint32_t got = 1u;
        if (lane == 0) got = atomicCAS(&sh.commit_lock, 0u, 1u);
        if (rg_uniform(got) != 0u) return;
        rg_acquire();
        uint32_t c = 0xffffffffu;
        if (lane < G::PB) {
            float4 acc = sh.acc[lane];
            c = __float_as_uint(acc.w);
            for (uint32_t k = 0; k < G::WIN / G::PB + 1u && c < n_samples; k++) {
                const uint32_t item = c * G::PB + lane, slot = item % G::WIN;
                if (rg_ld(reinterpret_cast<const uint32_t*>(&sh.ring[slot].w)) != item + 1u) break;
                rg_acquire();
                const float4 e = sh.ring[slot];
                acc.x += e.x; acc.y += e.y; acc.z += e.z;                                 // pathTracer.comp:452 (w += 0: dropped)
                c++;
            }
            acc.w = __uint_as_float(c);
            sh.acc[lane] = acc;
        }
        // every item below min(c) * PB is folded in: their ring slots may be reused
#pragma unroll
        for (int off = 1; off < (int)G::PB; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)c, off);
            c = o < c ? o : c;
        }
        rg_release();
        if (lane == 0) {
            rg_st(&sh.committed, c * G::PB);
            rg_release();
            rg_st(&sh.commit_lock, 0u);
        }
    };

    RgPath P;
    P.X = P.V = P.N = P.accmat = P.accrad = v3{0.0f, 0.0f, 0.0f};
    P.rx = P.ry = 0.0f; P.meta = 0u; P.pixkey = 0u;
    uint32_t kind = RG_EMPTY;
    uint32_t idle = 0;
    // Hard bound far above the worst case (every item needs at most max_depth + 1 iterations of ONE lane): a logic error can
    // never leave a wave spinning on the device.
    const uint32_t guard_limit = total_items * (a.max_depth + 2u) + 1024u;
    for (uint32_t guard = 0; guard < guard_limit; guard++) {
        MC_REGION(9);    // scheduler iteration
        // ================================================================== swap point
        {   // park the specular lanes
            const bool is_spec = kind == RG_M || kind == RG_G;
            const unsigned long long m = __ballot(is_spec);
            if (m) {
                const uint32_t base = rg_reserve_push(&sh.sq_tail, &sh.sq_head, (uint32_t)__popcll(m), G::SQ, lane);
                if (base != 0xffffffffu) {
                    const uint32_t ticket = base + rg_rank(m), slot = ticket % G::SQ;
                    if (!rg_wait_seq(&sh.sq_seq[slot], ticket, is_spec)) { failed = true; break; }
                    if (is_spec) {
                        sh.sq_rec[0][slot] = make_float4(P.X.x, P.X.y, P.X.z, P.rx);
                        sh.sq_rec[1][slot] = make_float4(P.V.x, P.V.y, P.V.z, __uint_as_float(kind));
                        sh.sq_rec[2][slot] = make_float4(P.accmat.x, P.accmat.y, P.accmat.z, __uint_as_float(P.meta));
                        sh.sq_rec[3][slot] = make_float4(P.accrad.x, P.accrad.y, P.accrad.z, __uint_as_float(P.pixkey));
                        sh.sq_rec[4][slot] = make_float4(P.N.x, P.N.y, P.N.z, 0.0f);
                        rg_release();
                        rg_st(&sh.sq_seq[slot], ticket + 1u);
                        kind = RG_EMPTY;
                    }
                }
            }
        }
        unsigned long long m_empty = __ballot(kind == RG_EMPTY);
        if (m_empty) {
            const uint32_t v = (uint32_t)__popcll(m_empty);
            // one snapshot of the control words (uniform addresses: broadcast reads)
            const uint32_t dqh = rg_uniform(rg_ld(&sh.dq_head)), dqt = rg_uniform(rg_ld(&sh.dq_tail));
            const uint32_t sqh = rg_uniform(rg_ld(&sh.sq_head)), sqt = rg_uniform(rg_ld(&sh.sq_tail));
            const uint32_t ni = rg_uniform(rg_ld(&sh.next_item)), cm = rg_uniform(rg_ld(&sh.committed));
            const uint32_t d_avail = dqt - dqh, s_avail = sqt - sqh;
            const uint32_t limit = total_items < cm + G::WIN ? total_items : cm + G::WIN;
            const uint32_t cam_avail = limit > ni ? limit - ni : 0u;
            bool batch = false;
            if (d_avail < v && s_avail + cam_avail >= kRgBatchMin) {
                // BATCH: spill the D-ready lanes (the other waves' vacancies drain them), take specular + camera work
                MC_REGION(11);   // batch
                const bool is_d = kind == RG_D;
                const unsigned long long m = __ballot(is_d);
                batch = true;
                if (m) {
                    const uint32_t base = rg_reserve_push(&sh.dq_tail, &sh.dq_head, (uint32_t)__popcll(m), G::DQ, lane);
                    if (base == 0xffffffffu) {
                        batch = false;
                    } else {
                        const uint32_t ticket = base + rg_rank(m), slot = ticket % G::DQ;
                        if (!rg_wait_seq(&sh.dq_seq[slot], ticket, is_d)) { failed = true; break; }
                        if (is_d) {
                            sh.dq_rec[0][slot] = make_float4(P.X.x, P.X.y, P.X.z, P.rx);
                            sh.dq_rec[1][slot] = make_float4(P.V.x, P.V.y, P.V.z, P.ry);
                            sh.dq_rec[2][slot] = make_float4(P.accmat.x, P.accmat.y, P.accmat.z, __uint_as_float(P.meta));
                            sh.dq_rec[3][slot] = make_float4(P.accrad.x, P.accrad.y, P.accrad.z, __uint_as_float(P.pixkey));
                            rg_release();
                            rg_st(&sh.dq_seq[slot], ticket + 1u);
                            kind = RG_EMPTY;
                        }
                    }
                }
            }
            bool more = batch || d_avail < v;   // also take specular / camera work (mixed iteration unless it is a batch)
            if (!batch && d_avail) {
                // refill the vacated lanes with D-ready paths
                const bool want = kind == RG_EMPTY;
                const unsigned long long m = __ballot(want);
                uint32_t granted;
                const uint32_t base = rg_reserve_pop(&sh.dq_head, &sh.dq_tail, (uint32_t)__popcll(m), lane, granted);
                const uint32_t rank = rg_rank(m);
                const bool mine = want && rank < granted;
                const uint32_t ticket = base + rank, slot = ticket % G::DQ;
                if (granted) {
                    if (!rg_wait_seq(&sh.dq_seq[slot], ticket + 1u, mine)) { failed = true; break; }
                    rg_acquire();
                    if (mine) {
                        const float4 r0 = sh.dq_rec[0][slot], r1 = sh.dq_rec[1][slot], r2 = sh.dq_rec[2][slot], r3 = sh.dq_rec[3][slot];
                        P.X = v3{r0.x, r0.y, r0.z}; P.rx = r0.w;
                        P.V = v3{r1.x, r1.y, r1.z}; P.ry = r1.w;
                        P.accmat = v3{r2.x, r2.y, r2.z}; P.meta = __float_as_uint(r2.w);
                        P.accrad = v3{r3.x, r3.y, r3.z}; P.pixkey = __float_as_uint(r3.w);
                        kind = RG_D;
                        rg_release();                                   // the reads above are complete before the slot is freed
                        rg_st(&sh.dq_seq[slot], ticket + G::DQ);
                    }
                }
            }
            if (more) {
                {   // parked specular paths
                    const bool want = kind == RG_EMPTY;
                    const unsigned long long m = __ballot(want);
                    if (m && s_avail) {
                        uint32_t granted;
                        const uint32_t base = rg_reserve_pop(&sh.sq_head, &sh.sq_tail, (uint32_t)__popcll(m), lane, granted);
                        const uint32_t rank = rg_rank(m);
                        const bool mine = want && rank < granted;
                        const uint32_t ticket = base + rank, slot = ticket % G::SQ;
                        if (granted) {
                            if (!rg_wait_seq(&sh.sq_seq[slot], ticket + 1u, mine)) { failed = true; break; }
                            rg_acquire();
                            if (mine) {
                                const float4 r0 = sh.sq_rec[0][slot], r1 = sh.sq_rec[1][slot], r2 = sh.sq_rec[2][slot],
                                             r3 = sh.sq_rec[3][slot], r4 = sh.sq_rec[4][slot];
                                P.X = v3{r0.x, r0.y, r0.z}; P.rx = r0.w;
                                P.V = v3{r1.x, r1.y, r1.z}; kind = __float_as_uint(r1.w);
                                P.accmat = v3{r2.x, r2.y, r2.z}; P.meta = __float_as_uint(r2.w);
                                P.accrad = v3{r3.x, r3.y, r3.z}; P.pixkey = __float_as_uint(r3.w);
                                P.N = v3{r4.x, r4.y, r4.z};
                                rg_release();
                                rg_st(&sh.sq_seq[slot], ticket + G::SQ);
                            }
                        }
                    }
                }
                {   // fresh camera samples, inside the reorder window
                    const bool want = kind == RG_EMPTY;
                    const unsigned long long m = __ballot(want);
                    if (m && cam_avail) {
                        const uint32_t n = (uint32_t)__popcll(m);
                        uint32_t base = 0, got = 0;
                        if (lane == 0) {
                            uint32_t i = rg_ld(&sh.next_item);
                            for (int k = 0; k < 64; k++) {
                                const uint32_t c2 = rg_ld(&sh.committed);
                                const uint32_t lim = total_items < c2 + G::WIN ? total_items : c2 + G::WIN;
                                const uint32_t avail = lim > i ? lim - i : 0u;
                                const uint32_t g = avail < n ? avail : n;
                                if (g == 0u) break;
                                const uint32_t o = atomicCAS(&sh.next_item, i, i + g);
                                if (o == i) { base = i; got = g; break; }
                                i = o;
                            }
                        }
                        base = rg_uniform(base); got = rg_uniform(got);
                        const uint32_t rank = rg_rank(m);
                        if (want && rank < got) { kind = RG_CAM; P.meta = base + rank; }
                    }
                }
            }
            if (batch || cam_avail < v) try_commit();   // a batch is a natural cadence; otherwise only under window pressure
        }

        // ================================================================== nothing to run in this wave?
        if (__ballot(kind != RG_EMPTY) == 0ull) {
            const uint32_t dqh = rg_uniform(rg_ld(&sh.dq_head)), dqt = rg_uniform(rg_ld(&sh.dq_tail));
            const uint32_t sqh = rg_uniform(rg_ld(&sh.sq_head)), sqt = rg_uniform(rg_ld(&sh.sq_tail));
            const uint32_t ni = rg_uniform(rg_ld(&sh.next_item));
            // everything handed out and nothing parked: whatever is still in flight sits in the lanes of other waves,
            // which finish it themselves (a wave pops what it pushed if nobody else does)
            if (ni >= total_items && dqt == dqh && sqt == sqh) break;
            MC_REGION(12);   // idle spin
            try_commit();                        // the window may be what holds the camera samples back
            __builtin_amdgcn_s_sleep(8);
            if (++idle > kRgSpinLimit) { failed = true; break; }
            continue;
        }
        idle = 0;

        // ================================================================== heads: bring every lane to "ray ready"
        // After this block X = ray origin, V = ray direction, and `depth` is the depth of the intersect that follows.
        uint32_t depth = rg_depth(P.meta);
        float emissive = 1.0f;                                                // pathTracer.comp:365,434,447
        bool fin = false;
        if (kind == RG_CAM) {
            MC_REGION(0);   // ray generation
            const uint32_t item = P.meta;                                      // meta = item, depth 0
            P.pixkey = sh.pixkey[item % G::PB];
            P.accrad = v3{0.0f, 0.0f, 0.0f};
            P.accmat = v3{1.0f, 1.0f, 1.0f};                                   // :361
            if (P.pixkey == 0xffffffffu) {
                fin = true;       // pixel outside the image: deposits a zero that nobody stores
            } else {
                const uint32_t cgx = P.pixkey & 0xffffu, cgy = P.pixkey >> 16;
                const uint32_t csamp = a.sample_begin + item / G::PB;
                // -- sample sensor (pathTracer.comp:357-362), identical to trace_sample()
                v3 r0 = rand01(cgx, cgy, csamp);
                float rnd2x = 2.0f * r0.x, rnd2y = 2.0f * r0.y;
                float tentx = rnd2x < 1.0f ? dm::fsqrt<Fast>(rnd2x) - 1.0f : 1.0f - dm::fsqrt<Fast>(2.0f - rnd2x);
                float tenty = rnd2y < 1.0f ? dm::fsqrt<Fast>(rnd2y) - 1.0f : 1.0f - dm::fsqrt<Fast>(2.0f - rnd2y);
                float stratx = (float)((csamp / 2u) % 2u), straty = (float)(csamp % 2u);
                float sx = (dm::fdiv<Fast>((float)cgx + 0.5f * ((0.5f + stratx) + tentx), (float)a.W) - 0.5f) * 0.036f;
                float sy = (dm::fdiv<Fast>((float)cgy + 0.5f * ((0.5f + straty) + tenty), (float)a.H) - 0.5f) * 0.024f;
                v3 spos = (a.cam_o + a.cx * sx) + a.cy * sy;                   // :360
                P.X = a.lc;
                P.V = normalize<Fast>(a.lc - spos);                            // :362
                fin = a.max_depth == 0u;
            }
        } else if (kind == RG_D) {
            MC_REGION(3);   // diffuse: NEE set-up + shadow ray + bounce direction
            const v3 x = P.X, nl = P.V;
            for (int i = 0; i < NS; i++) {                                    // :403
                if (!((sc.emissive_mask >> i) & 1u)) continue;                // :407 (uniform)
                const float* ls = uobj + 12 * (NP + i);
                const float lr2 = sc.r2[i];
                v3 le{ls[4], ls[5], ls[6]};
                v3 xc = v3{ls[0], ls[1], ls[2]} - x;                          // :408
                v3 sw = normalize<Fast>(xc);                                  // :409
                v3 su = normalize<Fast>(cross((__builtin_fabsf(sw.x) > 0.1f ? v3{0, 1, 0} : v3{1, 0, 0}), sw));
                v3 sv = cross(sw, su);
                float cos_a_max = dm::fsqrt<Fast>(1.0f - dm::fdiv<Fast>(lr2, dot(xc, xc)));   // :410
                float cos_a = (1.0f - P.rx) + P.rx * cos_a_max;               // :411
                float sin_a = dm::fsqrt<Fast>(1.0f - cos_a * cos_a);
                float phi = (2.0f * kPi) * P.ry;                              // :412
                float sphi, cphi;
                dm::sincos_angle<Fast>(phi, P.ry, sphi, cphi);
                v3 l = normalize<Fast>(((su * cphi) * sin_a + (sv * sphi) * sin_a) + sw * cos_a);   // :413
                float tne;
                int idne = intersect<Fast, NP, NS, true, 0>(sc, uobj, x, l, tne, sc.nee_skip_planes != 0u);   // :420 shadow ray
                if (idne == NP + i) {
                    float omega = (2.0f * kPi) * (1.0f - cos_a_max);          // :421
                    P.accrad = P.accrad + ((divs<Fast>(P.accmat, kPi) * dm::gmax(dot(l, nl), 0.0f)) * le) * omega;   // :422
                }
            }
            float r1 = (2.0f * kPi) * P.rx, r2 = P.ry, r2s = dm::fsqrt<Fast>(r2);   // :426
            v3 w = nl;
            v3 u = normalize<Fast>(cross((__builtin_fabsf(w.x) > 0.1f ? v3{0, 1, 0} : v3{1, 0, 0}), w));   // :427
            v3 vv = cross(w, u);
            float s1, c1;
            dm::sincos_angle<Fast>(r1, P.rx, s1, c1);
            P.V = normalize<Fast>(((u * c1) * r2s + (vv * s1) * r2s) + w * dm::fsqrt<Fast>(1.0f - r2));   // :428
            emissive = 0.0f;                                                  // :429
            depth++;
            fin = depth >= a.max_depth;
        } else if (kind == RG_M) {                                            // :432 mirror
            MC_REGION(5);
            P.V = reflect(P.V, P.N);
            depth++;
            fin = depth >= a.max_depth;
        } else if (kind == RG_G) {                                            // :437 glass
            MC_PT_DECISION_FP
            MC_REGION(6);
            const v3 rd0 = P.V, n = P.N;
            const v3 nl = dot(n, rd0) < 0.0f ? n : -n;                        // :390 (same expression, same values)
            bool into = (n.x == nl.x) && (n.y == nl.y) && (n.z == nl.z);      // :438
            const float nc = 1.0f, nt = 1.5f;
            float nnt = into ? dm::fdiv<Fast>(nc, nt) : dm::fdiv<Fast>(nt, nc);   // :439
            float ddn = dot(rd0, nl);
            float cos2t = 1.0f - (nnt * nnt) * (1.0f - ddn * ddn);            // :440
            v3 refl = reflect(rd0, n);
            if (cos2t >= 0.0f) {
                float k = (into ? 1.0f : -1.0f) * (ddn * nnt + dm::fsqrt<Fast>(cos2t));
                v3 tdir = normalize<Fast>(rd0 * nnt - n * k);                 // :441
                float aa = nt - nc, bb = nt + nc;
                float R0 = dm::fdiv<Fast>(aa * aa, bb * bb);                  // :442
                float c = 1.0f - (into ? -ddn : dot(tdir, n));
                float Re = R0 + (((((1.0f - R0) * c) * c) * c) * c) * c;      // :443
                float Tr = 1.0f - Re;
                float Pp = 0.25f + 0.5f * Re;
                float RP = dm::fdiv<Fast>(Re, Pp), TP = dm::fdiv<Fast>(Tr, 1.0f - Pp);
                bool pick_refl = P.rx < Pp;
                P.V = select(pick_refl, refl, tdir);                          // :444
                P.accmat = P.accmat * (pick_refl ? RP : TP);                  // :445
            } else {
                P.V = refl;                                                   // :446
            }
            depth++;
            fin = depth >= a.max_depth;
        }

        // ================================================================== intersect + bounce prologue
        if (kind != RG_EMPTY && !fin) {
            MC_REGION(1);
            const uint32_t gx = P.pixkey & 0xffffu, gy = P.pixkey >> 16;
            const uint32_t samp = a.sample_begin + rg_item(P.meta) / G::PB;
            const v3 ro = P.X, rd = P.V;
            float t;
            const int id = intersect<Fast, NP, NS, true, 0>(sc, uobj, ro, rd, t);
            if (id < 0) {
                fin = true;   // :369 `continue` with an unchanged ray misses again at every later depth: the path is over
            } else {
                MC_REGION(2);
                v3 x = ro + rd * t;                                           // :374
                const float* obj = lds_obj + 12 * id;                         // per-lane fetch from LDS
                const bool is_sphere = id >= NP;
                v3 geo{obj[0], obj[1], obj[2]};
                v3 emi{obj[4], obj[5], obj[6]};
                v3 col{obj[8], obj[9], obj[10]};
                const int mat = (int)__builtin_floorf(obj[11] + 0.5f);        // :378/:384
                v3 n = is_sphere ? normalize<Fast>(x - geo) : geo;            // :381/:387
                v3 nl = dot(n, rd) < 0.0f ? n : -n;                           // :390
                P.accrad = P.accrad + (P.accmat * emi) * emissive;            // :391
                P.accmat = P.accmat * col;                                    // :392
                v3 rnd = rand01(gx, gy, samp * a.max_depth + depth);       // :393
                float p = dm::gmax(dm::gmax(col.x, col.y), col.z);           // :394
                if (depth > 5u) {                                             // :395
                    if (rnd.z >= p) fin = true;                               // :396
                    else P.accmat = divs<Fast>(P.accmat, p);                  // :397
                }
                P.X = x;
                P.N = n;
                P.V = mat == 1 ? nl : rd;
                P.rx = rnd.x; P.ry = rnd.y;
                P.meta = rg_item(P.meta) | (depth << 24);
                kind = (uint32_t)mat;                                         // 1 diffuse, 2 mirror, 3 glass (host-checked)
            }
        }

        // ================================================================== finished samples -> reorder ring
        if (kind != RG_EMPTY && fin) {
            MC_REGION(10);
            const uint32_t item = rg_item(P.meta), slot = item % G::WIN;
            const v3 q = divs<Fast>(P.accrad, fspp);                          // :452 accrad / samps.y
            sh.ring[slot] = make_float4(q.x, q.y, q.z, 0.0f);
            rg_release();
            rg_st(reinterpret_cast<uint32_t*>(&sh.ring[slot].w), item + 1u);
            kind = RG_EMPTY;
        }
    }

    // ---- the last wave to leave folds what remains and writes the tile ----
    if (failed && lane == 0 && a.status) atomicOr(a.status, 1u);
    rg_release();
    uint32_t prev = 0;
    if (lane == 0) prev = atomicAdd(&sh.waves_done, 1u);
    if (rg_uniform(prev) != (uint32_t)NW - 1u) return;
    rg_acquire();
    for (uint32_t k = 0; k < n_samples / (G::WIN / G::PB) + 2u; k++) {   // the lock is free: every other wave has left
        try_commit();
        if (rg_uniform(rg_ld(&sh.committed)) >= total_items) break;
    }
    if (lane < G::PB) {
        float4 acc = sh.acc[lane];
        const uint32_t key = sh.pixkey[lane];
        if (key != 0xffffffffu) {
            if (__float_as_uint(acc.w) != n_samples && a.status) atomicOr(a.status, 2u);   // a sample never arrived
            acc.w = 0.0f;                                                      // :452 w accumulates +0
            if (a.sample_begin > 0) acc.w = a.out[(size_t)(tile_y0 + lane / G::PW) * a.W + tile_x0 + lane % G::PW].w + 0.0f;
            if (a.sample_end == a.spp) {                                       // :453 after sample spp-1
                acc.x = dm::fpow<Fast>(dm::gmin(dm::gmax(acc.x, 0.0f), 1.0f), 0.45f) * 255.0f + 0.5f;
                acc.y = dm::fpow<Fast>(dm::gmin(dm::gmax(acc.y, 0.0f), 1.0f), 0.45f) * 255.0f + 0.5f;
                acc.z = dm::fpow<Fast>(dm::gmin(dm::gmax(acc.z, 0.0f), 1.0f), 0.45f) * 255.0f + 0.5f;
            }
            a.out[(size_t)(tile_y0 + lane / G::PW) * a.W + tile_x0 + lane % G::PW] = acc;
        }
    }
}

template <bool Fast, int NW>
inline int launch_regroup(const PTArgs& a, uint32_t tile_rows, hipStream_t s) {
    using G = RgGeom<NW>;
    dim3 grid((a.W + G::PW - 1u) / G::PW, (tile_rows + G::PH - 1u) / G::PH);
    hipLaunchKernelGGL((pathtrace_regroup_kernel<Fast, NW>), grid, dim3(64 * NW), 0, s, a);
    return MC_OK;
}

}  // namespace pt
}  // namespace mc
