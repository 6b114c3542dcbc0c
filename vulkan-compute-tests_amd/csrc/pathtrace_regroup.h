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
    static constexpr uint32_t PB = 16, PW = 4, PH = 4;   // pixels per workgroup
    static constexpr uint32_t QCAP = 64;                 // capacity of each FIFO (records)
    static constexpr uint32_t WIN = 1024;                // reorder window (items): ~4 rounds of the 256 lanes; multi-bounce
                                                         // specular paths wait for a batch per bounce and hold the window back
};
constexpr uint32_t kRgBatchMin = 48;          // parked specular + available camera items that justify a batch
constexpr uint32_t kRgSpinLimit = 1u << 22;   // bound of every wait loop (a healthy wait is a few hundred cycles)
constexpr uint32_t kRgMaxDepth = 63;          // meta packs the depth into 6 bits
constexpr uint32_t kRgMaxItems = 1u << 24;    // ... and the item index into 24
constexpr int kRgFields = 19;                 // dwords per record

// meta: bits 0-23 item (sample-major: item = (samp - sample_begin) * PB + pixel), bits 24-29 depth
__device__ __forceinline__ uint32_t rg_item(uint32_t meta) { return meta & 0xffffffu; }
__device__ __forceinline__ uint32_t rg_depth(uint32_t meta) { return (meta >> 24) & 63u; }

template <int NW>
struct RgShared {
    using G = RgGeom<NW>;
    float obj[(6 + 3) * 12];                   // scene records for the per-lane material fetch
    // The four ticket counters of the two FIFOs in ONE word, so that a wave makes all its reservations of a swap point with a
    // single compare-and-swap: bits 0-15 dq_head, 16-31 dq_tail, 32-47 sq_head, 48-63 sq_tail (tickets wrap at 2^16).
    unsigned long long qctl;
    uint32_t next_item, committed, commit_lock, waves_done;
    uint32_t pixkey[G::PB];                    // gx | gy << 16 of the workgroup's pixels, 0xffffffff outside the image
    float4 acc[G::PB];                         // xyz accumulator, w = bits of the number of samples folded in so far
    uint32_t seq[2 * G::QCAP];                 // per slot: DQ = slots 0..QCAP-1, SQ = slots QCAP..2*QCAP-1
    float rec[kRgFields][2 * G::QCAP];         // records, one dword array per field (conflict-free ds_read/write_b32, no
                                               // register-tuple constraints): X.xyz V.xyz N.xyz accmat.xyz accrad.xyz rx ry meta
                                               // meta | kind << 30, pixkey
    float ring[3][G::WIN];                     // accrad of finished samples awaiting their turn (divided by spp at the fold)
    uint8_t ring_tag[G::WIN];                  // low byte of (lap + 1), lap = item / WIN, once the three floats are in place
                                               // (a slot's previous occupant left `lap`, so a stale entry never matches)
};

__device__ __forceinline__ uint32_t rg_ld(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void rg_st(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void rg_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
__device__ __forceinline__ void rg_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
// Order between this wave's own DS operations: they execute in issue order, so only the compiler has to keep them in place.
__device__ __forceinline__ void rg_order() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
__device__ __forceinline__ uint32_t rg_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ unsigned long long rg_uniform64(unsigned long long v) {
    return (unsigned long long)rg_uniform((uint32_t)v) | ((unsigned long long)rg_uniform((uint32_t)(v >> 32)) << 32);
}
__device__ __forceinline__ uint32_t rg_rank(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// State of the path a lane holds between two scheduler iterations (= one FIFO record).
struct RgPath {
    v3 X;          // RG_D / RG_M / RG_G: the hit point x (the next ray origin)
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
    constexpr uint32_t Q = G::QCAP;
    __shared__ RgShared<NW> sh;
    const uint32_t lane = threadIdx.x & 63u;
    const SceneArgs& sc = a.scene;
    const float* __restrict__ uobj = sc.obj;
    const float* lds_obj = sh.obj;

    // ---- start-up (the only barrier) ----
    stage_records(sh.obj, sc.obj, NP + NS, true);
    for (uint32_t i = threadIdx.x; i < 2u * Q; i += 64u * NW) sh.seq[i] = i & (Q - 1u);   // slot s is first written by ticket s
    for (uint32_t i = threadIdx.x; i < G::WIN; i += 64u * NW) sh.ring_tag[i] = 0u;         // lap 0 expects 1
    if (threadIdx.x == 0) {
        sh.qctl = 0ull;
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
                if (__hip_atomic_load(&sh.ring_tag[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != (uint8_t)(item / G::WIN + 1u)) break;
                rg_acquire();
                const v3 q = divs<Fast>(v3{sh.ring[0][slot], sh.ring[1][slot], sh.ring[2][slot]}, fspp);   // :452 accrad / samps.y
                acc.x += q.x; acc.y += q.y; acc.z += q.z;                                 // (acc.w += 0: dropped)
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

    HotSlab hot;
    hot.load<false>(sc);   // this kernel has no register headroom: scene constants stay scalar operands
    RgPath P;
    P.X = P.V = P.N = P.accmat = P.accrad = v3{0.0f, 0.0f, 0.0f};
    P.rx = P.ry = 0.0f; P.meta = 0u; P.pixkey = 0u;
    uint32_t kind = RG_EMPTY;
    uint32_t idle = 0;
    // Hard bound far above the worst case (every item needs at most max_depth + 1 iterations of ONE lane): a logic error can
    // never leave a wave spinning on the device.
    const uint32_t guard_limit = total_items * (a.max_depth + 2u) + 1024u;
    MC_TIME_INIT;
    for (uint32_t guard = 0; guard < guard_limit; guard++) {
        MC_REGION(9);    // scheduler iteration
        MC_TIME_DECL;
        // ================================================================== swap point
        // Latency matters as much as instruction count here: the LDS round trips of one swap point are (1) the control
        // words, (2) the compare-and-swap TOGETHER WITH the sequence words of the slots it would grant, (3) the record
        // writes and reads.  DS operations of a wave execute in issue order, so "data, then sequence word" needs no wait in
        // between — only the compiler must keep the order (wavefront-scope fences).
        const bool is_spec = (kind - RG_M) < 2u;
        const unsigned long long m_spec = __ballot(is_spec);
        const unsigned long long m_empty0 = __ballot(kind == RG_EMPTY);
        if (m_spec | m_empty0) {
            const bool is_d = kind == RG_D;
            const unsigned long long m_d = __ballot(is_d);
            const uint32_t n_s = (uint32_t)__popcll(m_spec), v0 = (uint32_t)__popcll(m_empty0), nd = (uint32_t)__popcll(m_d);
            unsigned long long wv = __hip_atomic_load(&sh.qctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t ni = rg_uniform(rg_ld(&sh.next_item)), cm = rg_uniform(rg_ld(&sh.committed));
            const uint32_t limit = total_items < cm + G::WIN ? total_items : cm + G::WIN;
            const uint32_t cam_avail = rg_uniform(limit > ni ? limit - ni : 0u);
            uint32_t push_s = 0, push_d = 0, pop_d = 0, pop_s = 0, s_avail_dbg = 0;
            bool batch = false, more = false, reserved = false, bad = false;
            bool push_on = false, pop_on = false;
            uint32_t push_ticket = 0, push_slot = 0, pop_ticket = 0, pop_slot = 0, rank_e = 0;
            unsigned long long m_e = m_empty0;
            for (int tries = 0; tries < 16; tries++) {
                // ---- all reservations of this swap point from one snapshot (scalar unit) ----
                const uint32_t wlo = rg_uniform((uint32_t)wv), whi = rg_uniform((uint32_t)(wv >> 32));
                const uint32_t dqh = wlo & 0xffffu, dqt = wlo >> 16, sqh = whi & 0xffffu, sqt = whi >> 16;
                const uint32_t d_avail = (dqt - dqh) & 0xffffu, s_avail = (sqt - sqh) & 0xffffu;
                s_avail_dbg = s_avail;
                push_s = (s_avail + n_s <= Q) ? n_s : 0u;             // park the specular lanes (all or none)
                const uint32_t v = v0 + push_s;                       // lanes to refill
                push_d = 0u; pop_s = 0u; batch = false;
                if (d_avail >= v) {
                    pop_d = v; more = false;                          // the common case: D-ready paths for every vacancy
                } else {
                    more = true;
                    batch = s_avail + cam_avail >= kRgBatchMin && d_avail + nd <= Q;
                    if (batch) {   // spill the D-ready lanes (other waves' vacancies drain them), take specular + camera work
                        push_d = nd; pop_d = 0u;
                        pop_s = v + nd < s_avail ? v + nd : s_avail;
                    } else {       // mixed iteration: whatever there is
                        pop_d = d_avail;
                        pop_s = v - pop_d < s_avail ? v - pop_d : s_avail;
                    }
                }
                m_e = m_empty0 | (push_s ? m_spec : 0ull) | (push_d ? m_d : 0ull);   // lanes empty once the pushes are out
                rank_e = rg_rank(m_e);
                if ((push_s | push_d | pop_d | pop_s) == 0u) { reserved = true; break; }
                // ---- per-lane roles under this (tentative) reservation ----
                push_on = (is_spec && push_s != 0u) || (is_d && push_d != 0u);
                push_ticket = sqt + rg_rank(m_spec);
                if (push_d) push_ticket = is_spec ? push_ticket : dqt + rg_rank(m_d);
                push_slot = (push_ticket & (Q - 1u)) | (is_spec ? Q : 0u);
                const bool from_d = rank_e < pop_d;
                pop_on = ((m_e >> lane) & 1ull) != 0ull && rank_e < pop_d + pop_s;
                pop_ticket = from_d ? dqh + rank_e : sqh + (rank_e - pop_d);
                pop_slot = (pop_ticket & (Q - 1u)) | (from_d ? 0u : Q);
                // ---- one round trip: the compare-and-swap and the sequence words of the slots it would grant ----
                const uint32_t nlo = ((dqh + pop_d) & 0xffffu) | ((dqt + push_d) << 16);
                const uint32_t nhi = ((sqh + pop_s) & 0xffffu) | ((sqt + push_s) << 16);
                unsigned long long ov = wv;
                if (lane == 0) ov = atomicCAS(&sh.qctl, wv, (unsigned long long)nlo | ((unsigned long long)nhi << 32));
                uint32_t sq_push = push_on ? rg_ld(&sh.seq[push_slot]) : 0u;
                uint32_t sq_pop = pop_on ? rg_ld(&sh.seq[pop_slot]) : 0u;
                const uint32_t olo = rg_uniform((uint32_t)ov), ohi = rg_uniform((uint32_t)(ov >> 32));
                if (olo != wlo || ohi != whi) {                       // another wave got in between: decide again
                    wv = (unsigned long long)olo | ((unsigned long long)ohi << 32);
                    continue;
                }
                reserved = true;
                // the slot's previous reader (push) / the record's writer (pop) may still be at work: rarely, briefly
                for (uint32_t spins = 0;; spins++) {
                    const bool ok = (!push_on || sq_push == (push_ticket & 0xffffu)) && (!pop_on || sq_pop == ((pop_ticket + 1u) & 0xffffu));
                    if (__ballot(!ok) == 0ull) break;
                    if (spins >= kRgSpinLimit) { bad = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                    if (push_on) sq_push = rg_ld(&sh.seq[push_slot]);
                    if (pop_on) sq_pop = rg_ld(&sh.seq[pop_slot]);
                }
                break;
            }
            MC_TIME_MARK(4);   // swap: control words + reservation
            if (bad) { failed = true; break; }
            if (!reserved) { push_s = push_d = pop_d = pop_s = 0u; batch = false; more = false; push_on = pop_on = false; m_e = m_empty0; rank_e = rg_rank(m_e); }
            if (batch) { MC_REGION(11); }
            if (!reserved) { MC_REGION(13); }
            if (more && !batch) { MC_REGION(14); }
            if (n_s && !push_s) { MC_REGION(15); }
            MC_ACCUM(7, cam_avail);
            MC_ACCUM(8, s_avail_dbg);
            MC_ACCUM(2, ni - cm);
            (void)s_avail_dbg;

            // ---- pushes: specular lanes -> SQ, (batch) D-ready lanes -> DQ; one write sequence, then the sequence word ----
            if (push_s | push_d) {
                if (push_on) {
                    const uint32_t slot = push_slot;
                    sh.rec[0][slot] = P.X.x; sh.rec[1][slot] = P.X.y; sh.rec[2][slot] = P.X.z;
                    sh.rec[3][slot] = P.V.x; sh.rec[4][slot] = P.V.y; sh.rec[5][slot] = P.V.z;
                    sh.rec[6][slot] = P.N.x; sh.rec[7][slot] = P.N.y; sh.rec[8][slot] = P.N.z;
                    sh.rec[9][slot] = P.accmat.x; sh.rec[10][slot] = P.accmat.y; sh.rec[11][slot] = P.accmat.z;
                    sh.rec[12][slot] = P.accrad.x; sh.rec[13][slot] = P.accrad.y; sh.rec[14][slot] = P.accrad.z;
                    sh.rec[15][slot] = P.rx; sh.rec[16][slot] = P.ry;
                    sh.rec[17][slot] = __uint_as_float(P.meta | (kind << 30));   // item, depth | kind (1..3)
                    sh.rec[18][slot] = __uint_as_float(P.pixkey);
                    rg_order();
                    rg_st(&sh.seq[slot], (push_ticket + 1u) & 0xffffu);
                    kind = RG_EMPTY;
                }
            }
            // ---- pops: D-ready paths, then parked specular ones, into the empty lanes; one read sequence ----
            const bool is_empty = ((m_e >> lane) & 1ull) != 0ull;
            if (pop_d | pop_s) {
                rg_order();
                if (pop_on) {
                    const uint32_t slot = pop_slot;
                    P.X = v3{sh.rec[0][slot], sh.rec[1][slot], sh.rec[2][slot]};
                    P.V = v3{sh.rec[3][slot], sh.rec[4][slot], sh.rec[5][slot]};
                    P.N = v3{sh.rec[6][slot], sh.rec[7][slot], sh.rec[8][slot]};
                    P.accmat = v3{sh.rec[9][slot], sh.rec[10][slot], sh.rec[11][slot]};
                    P.accrad = v3{sh.rec[12][slot], sh.rec[13][slot], sh.rec[14][slot]};
                    P.rx = sh.rec[15][slot]; P.ry = sh.rec[16][slot];
                    const uint32_t km = __float_as_uint(sh.rec[17][slot]);
                    P.pixkey = __float_as_uint(sh.rec[18][slot]);
                    kind = km >> 30; P.meta = km & 0x3fffffffu;
                    rg_order();                                     // issued after the reads: the slot is freed behind them
                    rg_st(&sh.seq[slot], (pop_ticket + Q) & 0xffffu);
                }
            }
            MC_TIME_MARK(5);   // swap: record writes + reads
            // ---- fresh camera samples for the lanes still empty, inside the reorder window ----
            const uint32_t want_cam = more ? (uint32_t)__popcll(m_e) - (pop_d + pop_s) : 0u;
            if (want_cam && cam_avail) {
                uint32_t base = 0, got = 0, iv = ni;
                for (int k = 0; k < 16; k++) {
                    const uint32_t i = rg_uniform(iv);
                    const uint32_t c2 = rg_uniform(rg_ld(&sh.committed));
                    const uint32_t lim = total_items < c2 + G::WIN ? total_items : c2 + G::WIN;
                    const uint32_t avail = lim > i ? lim - i : 0u;
                    const uint32_t g = avail < want_cam ? avail : want_cam;
                    if (g == 0u) break;
                    uint32_t o = i;
                    if (lane == 0) o = atomicCAS(&sh.next_item, i, i + g);
                    o = rg_uniform(o);
                    if (o == i) { base = i; got = g; break; }
                    iv = o;
                }
                const uint32_t r = rank_e - (pop_d + pop_s);          // (wraps for the lanes that popped: r >= got)
                if (is_empty && rank_e >= pop_d + pop_s && r < got) { kind = RG_CAM; P.meta = base + r; }
            }
            MC_TIME_MARK(6);   // swap: camera items
            // a batch is the natural cadence; otherwise fold as soon as the window (not the work) limits the camera samples
            if (batch || cam_avail < 64u + v0 + n_s) try_commit();
            MC_TIME_MARK(7);   // swap: commit
        }

        MC_TIME_MARK(0);   // swap point
        // ================================================================== nothing to run in this wave?
        if (__ballot(kind != RG_EMPTY) == 0ull) {
            const unsigned long long w = rg_uniform64(__hip_atomic_load(&sh.qctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            const uint32_t ni = rg_uniform(rg_ld(&sh.next_item));
            // everything handed out and nothing parked: whatever is still in flight sits in the lanes of other waves,
            // which finish it themselves (a wave pops what it pushed if nobody else does)
            if (ni >= total_items && (uint32_t)(w & 0xffffu) == (uint32_t)((w >> 16) & 0xffffu) &&
                (uint32_t)((w >> 32) & 0xffffu) == (uint32_t)(w >> 48))
                break;
            MC_REGION(12);   // idle spin
            try_commit();                        // the window may be what holds the camera samples back
            __builtin_amdgcn_s_sleep(8);
            if (++idle > kRgSpinLimit) { failed = true; break; }
            continue;
        }
        idle = 0;

        // ================================================================== heads: bring every lane to "ray ready"
        // After this block X = ray origin, V = ray direction, and `depth` is the depth of the intersect that follows.
        // (depth / emissive / fin are functions of the kind: computed outside the branches so that the heads only touch the
        //  path fields they change — fewer values to merge behind the branches)
        const uint32_t depth = rg_depth(P.meta) + (kind != RG_CAM ? 1u : 0u);   // depth of the intersect that follows
        const float emissive = kind == RG_D ? 0.0f : 1.0f;                    // pathTracer.comp:365,429,434,447
        bool fin = depth >= a.max_depth;                                      // (camera: depth 0)
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
            }
        }
        if (kind == RG_D) {
            MC_REGION(3);   // diffuse: NEE set-up + shadow ray + bounce direction
            const v3 x = P.X, nl = P.V;
#pragma unroll
            for (int i = 0; i < NS; i++) {                                    // :403
                if (!((sc.emissive_mask >> i) & 1u)) continue;                // :407 (uniform)
                const float* ls = uobj + 12 * (NP + i);
                const float lr2 = sc.r2[i];
                v3 le{ls[4], ls[5], ls[6]};
                v3 xc = v3{ls[0], ls[1], ls[2]} - x;                          // :408
                float occ_x[3];
#pragma unroll
                for (int q = 0; q < 3; q++) { v3 oq = v3{hot.c[q][0], hot.c[q][1], hot.c[q][2]} - x; occ_x[q] = dot(oq, oq); }
                const float xcc = dot(xc, xc);
                v3 sw = xc * dm::inversesqrt<Fast>(xcc);                      // :409 normalize(xc)
                v3 su = tangent_u<Fast>(sw);
                v3 sv = cross(sw, su);
                float cos_a_max = dm::fsqrt<Fast>(1.0f - dm::fdiv<Fast>(lr2, xcc));   // :410
                float cos_a = (1.0f - P.rx) + P.rx * cos_a_max;               // :411
                float sin_a = dm::fsqrt<Fast>(1.0f - cos_a * cos_a);
                float phi = (2.0f * kPi) * P.ry;                              // :412
                float sphi, cphi;
                dm::sincos_angle<Fast>(phi, P.ry, sphi, cphi);
                v3 l = normalize<Fast>(((su * cphi) * sin_a + (sv * sphi) * sin_a) + sw * cos_a);   // :413
                float tne;
                const bool reached = sc.nee_skip_planes != 0u ? shadow_reaches_sphere<Fast>(hot, x, l, i, xc, occ_x)   // :420 shadow ray
                                                               : intersect_slab<Fast>(hot, x, l, tne, false) == NP + i;
                if (reached) {
                    float omega = (2.0f * kPi) * (1.0f - cos_a_max);          // :421
                    P.accrad = P.accrad + ((divs_recip<Fast>(P.accmat, kPi, kInvPi) * dm::gmax(dot(l, nl), 0.0f)) * le) * omega;   // :422
                }
            }
            float r1 = (2.0f * kPi) * P.rx, r2 = P.ry, r2s = dm::fsqrt<Fast>(r2);   // :426
            v3 w = nl;
            v3 u = tangent_u<Fast>(w);   // :427
            v3 vv = cross(w, u);
            float s1, c1;
            dm::sincos_angle<Fast>(r1, P.rx, s1, c1);
            P.V = normalize<Fast>(((u * c1) * r2s + (vv * s1) * r2s) + w * dm::fsqrt<Fast>(1.0f - r2));   // :428
        }
        if (kind == RG_M) {                                                   // :432 mirror
            MC_REGION(5);
            P.V = reflect(P.V, P.N);
        }
        if (kind == RG_G) {                                                   // :437 glass
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
        }

        MC_TIME_MARK(1);   // heads
        // ================================================================== intersect + bounce prologue
        if (kind != RG_EMPTY && !fin) {
            MC_REGION(1);
            const uint32_t gx = P.pixkey & 0xffffu, gy = P.pixkey >> 16;
            const uint32_t samp = a.sample_begin + rg_item(P.meta) / G::PB;
            const v3 ro = P.X, rd = P.V;
            float t;
            const int id = intersect_slab<Fast>(hot, ro, rd, t, false);
            if (id < 0) {
                fin = true;   // :369 `continue` with an unchanged ray misses again at every later depth: the path is over
            } else {
                v3 x = ro + rd * t;                                           // :374
                const float* obj = lds_obj + 12 * id;                         // per-lane fetch from LDS
                const bool is_sphere = id >= NP;
                v3 geo{obj[0], obj[1], obj[2]};
                v3 col{obj[8], obj[9], obj[10]};
                const int mat = (int)obj[11];                                 // :378/:384, derived in stage_records
                const float p = obj[7];                                       // :394, derived in stage_records
                v3 n = is_sphere ? normalize<Fast>(x - geo) : geo;            // :381/:387
                v3 nl = dot(n, rd) < 0.0f ? n : -n;                           // :390
                if (__ballot(obj[3] != 0.0f) != 0ull) {                       // :391 (skipped when no lane hit an emitter)
                    v3 emi{obj[4], obj[5], obj[6]};
                    P.accrad = P.accrad + (P.accmat * emi) * emissive;
                }
                P.accmat = P.accmat * col;                                    // :392
                v3 rnd = rand01(gx, gy, samp * a.max_depth + depth);       // :393
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

        MC_TIME_MARK(2);   // intersect + prologue
        // ================================================================== finished samples -> reorder ring
        if (kind != RG_EMPTY && fin) {
            MC_REGION(10);
            const uint32_t item = rg_item(P.meta), slot = item % G::WIN;
            sh.ring[0][slot] = P.accrad.x; sh.ring[1][slot] = P.accrad.y; sh.ring[2][slot] = P.accrad.z;   // divided by spp at the fold
            rg_order();
            __hip_atomic_store(&sh.ring_tag[slot], (uint8_t)(item / G::WIN + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            kind = RG_EMPTY;
        }
        MC_TIME_MARK(3);   // retire
    }

    MC_TIME_FLUSH;
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
