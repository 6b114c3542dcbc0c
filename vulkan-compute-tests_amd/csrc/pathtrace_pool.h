// The sample-pool path tracer kernels for gfx950 — closed-box scenes (SceneArgs::box_ok); fast math (included by pathtrace_fast.hip),
// its careful tier (pathtrace_careful.hip; `Fast` is the math tier: 0 strict, 1 fast, 2 careful) and strict math (pathtrace_strict.hip).
//
// Why: the round-synchronous kernels (pathtrace_kernel.h) step all 64 lanes of a wave through the depths of ONE sample each;
// Russian roulette (pathTracer.comp:395-397) thins the wave from depth 6 on and the round still costs its longest path, so the
// intersection + prologue code runs with 52 of 64 lanes and `VALUUtilization` sits at 59 % (profiles/r02_pt_fast_pmc_summary.json).
// gfx950 issues every VALU instruction in ~2.4 cycles whatever its class or EXEC mask (profiles/r03_valu_microbench3.txt), so
// what is left is the number of lanes an issued instruction serves.
//
// What: a wave owns 64/S pixels; the S lanes of a pixel share ALL its samples as a pool.  Camera rays and their first
// intersection (:357-362, :316-341) are produced one batch — S consecutive samples of every pixel, 64 rays — at a time at full
// width into a stash in LDS (two batches deep); a lane whose path ended takes its pixel's next stash entry at the top of the
// next iteration, so lanes sit at different depths of different samples and the wave stays full until the pool runs dry.
// No path state ever moves between lanes, there are no queues between waves, no reorder ring: a lane adds the radiance it
// gathers (:391, :422) to its own register accumulator, and at the end the S partial sums of a pixel are added in lane order,
// scaled by 1/spp and :453 is applied.
// (First version: any lane took any pixel's sample and added its radiance to the pixel's accumulator in LDS with ds_add_f32.
// gfx950 executes an LDS float atomic at ~2 cycles per LANE — 131 LDS cycles per wave instruction, conflict or not — and the
// kernel became LDS-bound: 33.6 ms.  profiles/r03_pool_v1_lds_atomics_pmc.txt.)
//
// STRICT variant (Fast = 0; bit-identical to the oracle like every strict kernel): the per-pixel sum must be the reference's —
// samples added in sample order (:451-:452).  A lane therefore keeps a per-PATH accrad, writes accrad / spp into its sample's slot
// of a result ring in LDS (three batches per pixel for the reference scene, four beyond three spheres) when the path ends, and the pixel's 16 lanes add a batch's 16 results in sample
// order — the fold of the round-synchronous kernels, fed from LDS — once every sample of the batch has ended (oldest live sample
// of the pixel by a 4-step butterfly, only when a new batch is wanted).  Same decisions, same operations, same order: exact.
//
// Parity (fast): a sample follows the fast closed-box round-synchronous kernel's arithmetic (the same inlined functions) except for
// the cheaper equivalent forms this kernel alone uses — the bounce off a wall as a signed permutation (the same values), shadow rays
// decided by comparing squares and centre projections (disjoint spheres: the host's premise for selecting it), |c - x|^2 - r^2 and
// colour / p formed once — DESIGN.md §3.3; beyond that what differs is the ORDER in which a pixel's fp32 contributions are added — the
// reference adds accrad/spp sample by sample (:451-:452), here every lane sums the contributions of the samples it happened to trace and the S partial sums are added at
// the end — a reassociation, relative 1e-6 of the pixel value, far inside the fast-math tolerance; the strict variant
// above keeps the reference's order.  It is deterministic: a wave's schedule depends on nothing outside the wave, and
// the pixels a wave owns are the same for every tiling the host selects it for (pathtrace.hip).
#pragma once
#include <type_traits>
#include "pathtrace_kernel.h"

#ifndef MC_PT_POOL_WAVES
#define MC_PT_POOL_WAVES 7   // waves per SIMD the register budget is set for (72 VGPRs; 6: 18.29 ms, 7: 18.11, 8: 18.68 at K2)
#endif
#ifndef MC_PT_POOL_STRICT_WAVES
#define MC_PT_POOL_STRICT_WAVES 6   // strict kernel: 80 VGPRs (three values spilled, none reloaded inside an iteration), 6 blocks of 26 KB per CU
#endif
#ifndef MC_PT_POOL_KEEP_VALID   // the pixel's validity kept across the loop (a lane mask) instead of re-derived per batch
#define MC_PT_POOL_KEEP_VALID 1
#endif
#ifndef MC_PT_POOL_LANE_REGS   // the refill's lane / sub kept in registers (two instructions an iteration less; both kernels have the room)
#define MC_PT_POOL_LANE_REGS 1
#endif
#ifndef MC_PT_POOL_HOT_W
#define MC_PT_POOL_HOT_W true
#endif
#ifndef MC_PT_POOL_HOT_VGPR
#define MC_PT_POOL_HOT_VGPR false
#endif

namespace mc {
namespace pt {

constexpr uint32_t kPoolEntryFloats = 8;     // {rd.x, rd.y, rd.z, t | id (-1: nothing to trace), key0 = samp * maxDepth, rnd.x, rnd.y of key0}
constexpr uint32_t kPoolRecordStride = 16;   // floats per staged record: {geo.xyz, p (fast: 1 / p) | colour.rgb, material + 256 * emits | emission.xyz, RN(1 / p) (fast: p) | fast: colour.rgb / p, the same integer bits}
template <int NS> constexpr uint32_t pool_record_floats() { return (6u + (uint32_t)NS) * kPoolRecordStride; }   // 6 planes + NS spheres
constexpr uint32_t kPoolStashFloats = 128u * kPoolEntryFloats;     // per wave: 64/S pixels x 2 batches x S entries
// Strict: the result ring holds this many batches per pixel (3 planes x, y, z).  Three for the reference's scene: with the 23 + 3 KB of a
// block six blocks fit a CU's 160 KB, and 6 waves per SIMD with a 3-batch ring beat 5 waves with a 4-batch ring (32.7 against 33.1 ms at K2;
// 2 batches stall the production: 34.3; profiles/r04_strict_occupancy.txt).  Scenes with more spheres run at 4 waves per SIMD: 4 batches.
#ifndef MC_PT_POOL_RESULT_BATCHES
#define MC_PT_POOL_RESULT_BATCHES 0   // 0: automatic (by sphere count)
#endif
template <int NS> constexpr uint32_t pool_result_batches() { return MC_PT_POOL_RESULT_BATCHES ? MC_PT_POOL_RESULT_BATCHES : (NS <= 3 ? 3u : 4u); }
template <int NS> constexpr uint32_t pool_result_floats() { return 64u * pool_result_batches<NS>() * 3u; }   // per wave: 64/S pixels x batches x S samples x 3
template <int Fast, int NS> constexpr uint32_t pool_wave_lds_floats() { return kPoolStashFloats + (Fast ? 0u : pool_result_floats<NS>()); }
template <int Fast, int NS> constexpr size_t pool_block_lds_bytes() { return (pool_record_floats<NS>() + 4u * pool_wave_lds_floats<Fast, NS>()) * sizeof(float); }
// Waves per SIMD the register budget is set for: 7 / 6 for the reference's three spheres (72 / 80 VGPRs); every further sphere
// keeps five more values live across a bounce (c_i - x, |c_i - x|^2 and its r^2-reduced form), so the budget widens with the count.
template <int Fast, int NS> constexpr int pool_waves() {
    return Fast ? (NS <= 3 ? MC_PT_POOL_WAVES : NS <= 5 ? 6 : NS <= 6 ? 5 : 4) : (NS <= 3 ? MC_PT_POOL_STRICT_WAVES : NS <= 6 ? 4 : 3);
}

// Disjoint (fast math): the host proved the spheres pairwise disjoint — shadow rays are decided without square roots
// (shadow_visible_disjoint); false: overlapping spheres, the root form (shadow_reaches_sphere).  The strict kernel has one form.
template <int Fast, int S, int NS, bool Disjoint = true>
__global__ void __launch_bounds__(256, (pool_waves<Fast, NS>())) pathtrace_pool_kernel(PTArgs a) {
    constexpr uint32_t kPoolRecordFloats = pool_record_floats<NS>();
    extern __shared__ float lds_dyn[];
    float* lds_obj = lds_dyn;
    // The 9 records, re-packed for two 16-byte reads per bounce at address id << 6: the plane normal / sphere centre with the
    // roulette probability max(max(c.x, c.y), c.z) (:394), the colour with the material code int(floor(m + 0.5)) (:378/:384)
    // and an "emits" flag as integer bits — the same fp32 operations on the same operands as evaluating them at every bounce.
    // Fast math: slot 3 holds v_rcp_f32(p), the factor :397's division multiplies by — formed once per block instead of once
    // per bounce and lane (a transcendental blocks the SIMD for 8 cycles); p itself, which :396 compares with, is in slot 11.
    // Strict: slot 3 holds p and slot 11 the correctly rounded 1 / p that the short division of :397 starts from (dm::div3).
    if (threadIdx.x < 6u + (uint32_t)NS) {
        const float* o = a.scene.obj + 12u * threadIdx.x;
        float* r = lds_obj + kPoolRecordStride * threadIdx.x;
        const float p = dm::gmax(dm::gmax(o[8], o[9]), o[10]);
        r[0] = o[0]; r[1] = o[1]; r[2] = o[2]; r[3] = Fast ? dm::fdiv<Fast>(1.0f, p) : p;
        r[4] = o[8]; r[5] = o[9]; r[6] = o[10];
        const uint32_t emits = (o[4] != 0.0f || o[5] != 0.0f || o[6] != 0.0f) ? 256u : 0u;
        r[7] = dm::as_float((uint32_t)(int)__builtin_floorf(o[11] + 0.5f) | emits);
        r[8] = o[4]; r[9] = o[5]; r[10] = o[6]; r[11] = Fast ? p : dm::rcp_short(p);
        // fast math: the colour already divided by p (:392 and :397 as ONE multiplication past depth 5), with the same integer bits
        if constexpr (Fast) { r[12] = o[8] * r[3]; r[13] = o[9] * r[3]; r[14] = o[10] * r[3]; r[15] = r[7]; }
    }
    __syncthreads();
    // fast math: the sphere tests of a bounce (three of the shadow ray, three of the next ray, all from the hit point x) read
    // |c_i - x|^2 - r_i^2, formed once, instead of each adding r_i^2 to its b^2 - |c_i - x|^2
    // (fast tier only.  The careful tier keeps the reference's order (b^2 - |c - x|^2) + r^2: of the identities of exact arithmetic fast
    // math does not execute, this one and the un-normalised directions are the two whose forked samples do not balance — more of them
    // lose radiance than gain it; with both in the reference's form gains and losses cancel: profiles/r05_fork_bias_identities.txt)
    constexpr bool kOccR2 = Fast == 1 && MC_PT_FAST_OCC_MINUS_R2;
    constexpr uint32_t TW = WaveTile<S>::w, TH = WaveTile<S>::h, Ring = 2u * (uint32_t)S, RRing = pool_result_batches<NS>() * (uint32_t)S;
    HotSlabN<NS> hot;
    hot.template load<MC_PT_POOL_HOT_VGPR, MC_PT_POOL_HOT_W>(a.scene);   // (uniform operands: this kernel has no vector registers to spare for copies)
    // A lane's pixel (pix = lane / S of the wave tile) and slot of a batch (sub = lane % S) never change.  What derives from them
    // and is needed only now and then — the stash base, the tile row, the validity — is derived afresh from an opaque copy of
    // the thread id where it is used, so that it does not occupy registers across the bounce loop (80 VGPRs = 6 waves per SIMD).
    struct Lane { uint32_t lane, pix, sub, ty; bool valid; };
    auto my_lane = [&](bool with_row) {
        uint32_t tid = threadIdx.x;
        // (LANE_REGS: the refill's lane and sub stay in registers — the fast and careful kernels have the room.  The strict kernel at its
        //  80-register budget has not: kept "in registers" they were spilled, and the lanes-below-me mask was reloaded from scratch in
        //  the stash pick-up of nearly every iteration; there they are re-derived from the thread index, two instructions.)
        if (!(MC_PT_POOL_LANE_REGS && Fast != 0 && !with_row)) asm volatile("" : "+v"(tid));
        Lane q;
        q.lane = tid & 63u; q.pix = q.lane / (uint32_t)S; q.sub = q.lane % (uint32_t)S;
        const uint32_t wave = tid >> 6;
        q.ty = 0u; q.valid = false;
        if (with_row) {   // pathTracer.comp:348
            q.ty = blockIdx.y * (2u * TH) + (wave >> 1) * TH + q.pix / TW;
            const uint32_t gxx = blockIdx.x * (2u * TW) + (wave & 1u) * TW + q.pix % TW;
            q.valid = gxx < a.W && tile_row_to_storage(q.ty, a.row_begin, a.row_block, a.row_stride) < a.row_end;
        }
        return q;
    };
    // my pixel's coordinates (:349) stay: the RNG key of every bounce needs them; so does the base of its stash
    uint32_t gx, gy;
    bool pixel_valid;         // my pixel lies in the image and in the tile (:348)
    float* const gstash = lds_dyn + kPoolRecordFloats + (threadIdx.x >> 6) * pool_wave_lds_floats<Fast, NS>() +
                          ((threadIdx.x & 63u) / (uint32_t)S) * (Ring * kPoolEntryFloats);
    // strict: my pixel's result ring [3][RRing] (x, y, z planes) behind the wave's stash
    float* const gres = lds_dyn + kPoolRecordFloats + (threadIdx.x >> 6) * pool_wave_lds_floats<Fast, NS>() + kPoolStashFloats +
                        ((threadIdx.x & 63u) / (uint32_t)S) * (3u * RRing);
    // (re-deriving THIS address where it is used — path ends, commits — instead of keeping it was measured: + 1.1 ms at K2 strict)
    {
        const uint32_t tid = threadIdx.x, wave = tid >> 6, pix = (tid & 63u) / (uint32_t)S;
        gx = blockIdx.x * (2u * TW) + (wave & 1u) * TW + pix % TW;
        const uint32_t ty = blockIdx.y * (2u * TH) + (wave >> 1) * TH + pix / TW;
        const uint32_t srow = tile_row_to_storage(ty, a.row_begin, a.row_block, a.row_stride);
        pixel_valid = gx < a.W && srow < a.row_end;
        gy = a.H - 1u - (pixel_valid ? srow : 0u);
    }
    const uint32_t n_batches = (a.sample_end - a.sample_begin + (uint32_t)S - 1u) / (uint32_t)S;
    uint32_t batch = 0u;      // wave-uniform: batches produced; every pixel's stash has received batch * S entries
    uint32_t ghead = 0u;      // entries my pixel's lanes have taken (the same value in all S lanes)
    // ---- the state of a lane's path, and the radiance its paths have gathered
    v3 ro{0, 0, 0}, rd{0, 0, 1}, accmat{1, 1, 1}, acc{0, 0, 0};   // acc: fast = this lane's partial sum; strict = the pixel's ordered sum
    v3 accrad{0, 0, 0};       // strict: the radiance of the current path (:391, :422)
    if constexpr (!Fast) {
        // strict: a later sample range continues the ordered sum the caller passes back in (the samps.x protocol, :451-:452) —
        // every lane of the pixel starts from the stored value and adds the range's samples in order, as one launch would have
        if (a.sample_begin > 0u) {
            const Lane me = my_lane(true);
            if (me.valid) { const float4 prev = a.out[(size_t)me.ty * a.W + gx]; acc = v3{prev.x, prev.y, prev.z}; }
        }
    }
    uint32_t cur = 0u;        // strict: index (within my pixel) of the sample this lane traces; its result slot is cur % RRing
    uint32_t committed = 0u;  // strict: samples of my pixel already added to acc (the same value in all S lanes)
    // (wave-uniform: held in a scalar register — as a vector value the strict kernel spilled it and reloaded it at every path end)
    const float fspp = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)a.spp)));
    float emissive = 1.0f, t = 0.0f, occ[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) occ[i] = 0.0f;
    int id = 0;
    uint32_t key = 0u, kend = 0u, krr = 0u;   // key0 + depth; key0 + maxDepth; key0 + 5 (:395: roulette while key > krr)
    // rand01 of the bounce about to be traced (:393).  It is drawn at the END of the previous bounce, together with the Russian
    // roulette decision of :396 (the hit — hence p — is known there: the loop is rotated), so that a path the roulette ends
    // frees its lane at the end of an iteration, not a third of the way into the next one.
    float rx = 0.0f, ry = 0.0f;
    bool alive = false;
    // every iteration either traces a bounce of a live lane, or consumes stash entries, or produces a batch: bounded
    unsigned long long max_iters = (unsigned long long)n_batches * S * (a.max_depth + 2ull) + 64ull;
#ifdef MC_PT_POOL_TEST_BOUND   // diagnostic build only (tests/test_gpu_pool.py): a bound the loop must trip
    max_iters = MC_PT_POOL_TEST_BOUND;
#endif
    bool finished = false;
    for (unsigned long long it = 0; it < max_iters; it++) {
        // ---- lanes whose path ended take their pixel's next camera rays
        // (MC_REGION: diagnostic build only — make stats, tools/pool_region_stats.py — executions and active lanes per block)
        MC_REGION(0);    // an iteration
        const unsigned long long deadm = __ballot(!alive);
        if (deadm != 0ull) {
            MC_REGION(1);    // refill bookkeeping
            const Lane me = my_lane(false);
            // the dead lanes of my pixel: S bits of the ballot starting at my pixel's first lane
            const uint32_t gbits = (uint32_t)(deadm >> (me.lane - me.sub)) & (S == 32 ? ~0u : ((1u << (S & 31)) - 1u));
            const uint32_t need = (uint32_t)__builtin_popcount(gbits);
            uint32_t avail = batch * (uint32_t)S - ghead;
            bool want = batch < n_batches && __ballot(need > avail) != 0ull && __ballot(avail > (uint32_t)S) == 0ull;
            if constexpr (!Fast) {
                if (want) {   // (uniform) add the batches whose samples have all ended, in order; then: is the result ring free?
                    uint32_t oldest = alive ? cur : 0xffffffffu;                      // oldest sample a lane of my pixel still traces
                    if constexpr (S == 16) {
                        // The minimum over the pixel's 16 lanes = one DPP row: four rotations within the row (row_ror 8, 4, 2, 1), each
                        // folded into its v_min_u32 — no LDS round trip and no address registers.  (The butterfly through ds_bpermute kept
                        // four lane-address VGPRs live across the whole kernel; at the strict kernel's 80-register budget they were
                        // spilled and RELOADED FROM SCRATCH here, inside the loop, each load waited for.)  Integer minimum: same value.
                        auto rot_min = [](uint32_t v, auto ctrl) {
                            const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, decltype(ctrl)::value, 0xf, 0xf, false);
                            return o < v ? o : v;
                        };
                        oldest = rot_min(oldest, std::integral_constant<int, 0x128>{});   // row_ror:8
                        oldest = rot_min(oldest, std::integral_constant<int, 0x124>{});   // row_ror:4
                        oldest = rot_min(oldest, std::integral_constant<int, 0x122>{});   // row_ror:2
                        oldest = rot_min(oldest, std::integral_constant<int, 0x121>{});   // row_ror:1
                    } else {
#pragma unroll
                        for (uint32_t o = 1; o < (uint32_t)S; o <<= 1) {
                            const uint32_t other = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((me.lane ^ o) << 2), (int)oldest);
                            oldest = other < oldest ? other : oldest;
                        }
                    }
                    const uint32_t fin = oldest < ghead ? oldest : ghead;             // samples [0, fin) are taken and ended
                    while (committed + (uint32_t)S <= fin) {
                        for (uint32_t k = 0; k < (uint32_t)S; k++) {                  // :452 acc += accrad / spp, in sample order
                            const uint32_t slot = (committed + k) % RRing;
                            acc.x += gres[slot]; acc.y += gres[RRing + slot]; acc.z += gres[2u * RRing + slot];
                        }
                        committed += (uint32_t)S;
                    }
                    want = __ballot((batch + 1u) * (uint32_t)S - committed > RRing) == 0ull;   // every pixel's ring has room
                }
            }
            // one more batch when a pixel wants more rays than it has — and every pixel's ring has room for S more
            if (want) {
                MC_REGION(2);    // a batch of camera rays
                const Lane g = my_lane(MC_PT_POOL_KEEP_VALID ? false : true);
                const uint32_t samp = a.sample_begin + batch * (uint32_t)S + g.sub;
                // (strict: the pixel coordinates as fresh values — otherwise float(gy) of :359 is hoisted out of the loop, and at this
                //  kernel's register budget "out of the loop" meant spilled and reloaded from scratch at the end of every iteration)
                uint32_t cgx = gx, cgy = gy;
                if constexpr (!Fast) asm volatile("" : "+v"(cgx), "+v"(cgy));
                const v3 crd = camera_ray<Fast>(a, cgx, cgy, samp);
                v3 oc0[NS];
                float occ0[NS], ct;
#pragma unroll
                for (int i = 0; i < NS; i++) {
                    oc0[i] = v3{a.cam_oc[i][0], a.cam_oc[i][1], a.cam_oc[i][2]};
                    occ0[i] = kOccR2 ? a.cam_occ[i] - hot.r2[i] : a.cam_occ[i];
                }
                int cid = intersect_slab<Fast, (Fast != 0), kOccR2, 22>(hot, a.lc, crd, ct, false, occ0, oc0);
                // nothing to trace: a pixel outside the tile, a sample beyond the range; a camera ray that misses everything (:369)
                // gathers nothing either
                if (!((MC_PT_POOL_KEEP_VALID ? pixel_valid : g.valid) && samp < a.sample_end)) cid = -1;
                if constexpr (!Fast) {   // such a sample's result is a zero (adding +0 changes no bit of the sum)
                    const uint32_t slot = (batch * (uint32_t)S + g.sub) % RRing;
                    if (cid < 0) { gres[slot] = 0.0f; gres[RRing + slot] = 0.0f; gres[2u * RRing + slot] = 0.0f; }
                }
                float4* e = reinterpret_cast<float4*>(gstash + ((batch * (uint32_t)S + g.sub) & (Ring - 1u)) * kPoolEntryFloats);
                e[0] = make_float4(crd.x, crd.y, crd.z, ct);
                const v3 r0 = rand01(gx, gy, samp * a.max_depth);           // :393 at depth 0 (no roulette there: z unused)
                e[1] = make_float4(dm::as_float((uint32_t)cid), dm::as_float(samp * a.max_depth), r0.x, r0.y);
                batch++;
                avail += (uint32_t)S;
            }
            const uint32_t rank = (uint32_t)__builtin_popcount(gbits & ((1u << me.sub) - 1u));   // dead lanes of my pixel below me
            if (!alive && rank < avail) {
                MC_REGION(3);    // lanes taking a stash entry
                const float4* q = reinterpret_cast<const float4*>(gstash + ((ghead + rank) & (Ring - 1u)) * kPoolEntryFloats);
                const float4 q0 = q[0], q1 = q[1];
                rd = v3{q0.x, q0.y, q0.z}; t = q0.w;
                if constexpr (!Fast) { cur = ghead + rank; accrad = v3{0.0f, 0.0f, 0.0f}; }   // :361
                id = (int)dm::as_uint(q1.x);
                key = dm::as_uint(q1.y); kend = key + a.max_depth; krr = key + 5u;
                rx = q1.z; ry = q1.w;
                ro = a.lc;                                                  // :362
                accmat = v3{1.0f, 1.0f, 1.0f}; emissive = 1.0f;             // :361, :365
                alive = id >= 0;
            }
            ghead += need < avail ? need : avail;
        }
        // pool exhausted and every path ended?  (no lane alive but entries left: only empty entries were taken — go round again)
        if (__ballot(alive) == 0ull && batch >= n_batches && __ballot(ghead != batch * (uint32_t)S) == 0ull) { finished = true; break; }
        // ---- one bounce of every live lane: prologue, material, intersection of the next depth (the rotated loop of trace_sample)
        // (structured ifs, no break / continue: every extra exit edge of this block cost a dozen register copies at its merge)
        if (alive) {
            {
                MC_REGION(4);    // a bounce: prologue
                v3 x = ro + rd * t;                                               // :374
                v3 xoc[NS];                                                       // c_i - x (:317 at the next depth, :408 now)
#pragma unroll
                for (int i = 0; i < NS; i++) { xoc[i] = v3{hot.c[i][0], hot.c[i][1], hot.c[i][2]} - x; occ[i] = dot(xoc[i], xoc[i]); }
                float xcc[NS];                                                    // |c_i - x|^2 itself (:410 needs it)
#pragma unroll
                for (int i = 0; i < NS; i++) { xcc[i] = occ[i]; if constexpr (kOccR2) occ[i] = occ[i] - hot.r2[i]; }
                const float4* obj = reinterpret_cast<const float4*>(lds_obj + kPoolRecordStride * (uint32_t)id);   // per-lane fetch
                // (fast: past depth 5 the colour row is the one already divided by the roulette probability)
                const float4 o0 = obj[0], o1 = obj[(Fast && MC_PT_FAST_COLOUR_OVER_P && key > krr) ? 3 : 1];
                const bool is_sphere = id >= 6;
                v3 geo{o0.x, o0.y, o0.z};
                v3 col{o1.x, o1.y, o1.z};
                const uint32_t mbits = dm::as_uint(o1.w);
                const int mat = (int)(mbits & 255u);                              // :378/:384
                const float p = o0.w;                                             // :394 (fast: its reciprocal)
#ifdef MC_PT_REGION_STATS
                if (is_sphere) MC_REGION(5);    // sphere normal
#endif
                v3 n = is_sphere ? normalize<Fast>(x - geo) : geo;                // :381/:387
                const float dot_n_rd = dot(n, rd);
                v3 nl;                                                            // :390 nl = dot(n, rd) < 0 ? n : -n
                if constexpr (Fast) {   // (the sign bit of the dot product, inverted, flips n — trace_sample)
                    const uint32_t flip = ~dm::as_uint(dot_n_rd) & 0x80000000u;
                    nl = v3{dm::as_float(dm::as_uint(n.x) ^ flip), dm::as_float(dm::as_uint(n.y) ^ flip), dm::as_float(dm::as_uint(n.z) ^ flip)};
                } else {
                    nl = dot_n_rd < 0.0f ? n : -n;
                }
                v3& rad = Fast ? acc : accrad;                                    // where gathered radiance goes (see the header)
                if (__ballot(mbits >= 256u) != 0ull) {                            // :391 (non-emitters add a zero: box_ok)
                    MC_REGION(12);   // emission of a hit
                    const float4 o2 = obj[2];
                    rad = rad + (accmat * v3{o2.x, o2.y, o2.z}) * emissive;
                }
                accmat = accmat * col;                                            // :392
                const v3 rnd{rx, ry, 0.0f};                                       // :393 (drawn at the end of the previous bounce)
                if constexpr (Fast) { if (!MC_PT_FAST_COLOUR_OVER_P) accmat = accmat * (key > krr ? p : 1.0f); }   // :395, :397 (:396 was decided there too)
                else if (key > krr) accmat = divs_recip<Fast>(accmat, p, obj[2].w);
                bool go = true;
                {
                ro = x;                                                           // :429, :434, :447
                if (mat == 1) {                                                   // :400 diffuse
                    MC_REGION(13);   // diffuse: light sample + shadow test
                    v3 accmat_over_pi{0.0f, 0.0f, 0.0f};                          // strict: :422's accmat / pi, once per bounce
                    if constexpr (!Fast) accmat_over_pi = divs_recip<Fast>(accmat, kPi, kInvPi);
#pragma unroll
                    for (int i = 0; i < NS; i++) {                                // :403
                        if (!((a.scene.emissive_mask >> i) & 1u)) continue;       // :407 (uniform)
                        const float* ls = a.scene.obj + 12 * (6 + i);
                        v3 le{ls[4], ls[5], ls[6]};
                        float cos_a_max;
                        v3 l = light_sample_direction<Fast>(xoc[i], xcc[i], hot.r2[i], rnd, cos_a_max);   // :408-:413
                        bool lit;                                                                          // :420
                        if constexpr (Fast && Disjoint) lit = shadow_visible_disjoint<kOccR2>(hot, l, i, xoc, occ);
                        else lit = shadow_reaches_sphere<Fast, kOccR2>(hot, x, l, i, xoc[i], occ);
                        if (lit) {
                            MC_REGION(8);    // light contribution
                            if constexpr (Fast) {
                                const float scale = __builtin_fmaxf(dot(l, nl), 0.0f) * (2.0f - (cos_a_max + cos_a_max));   // :421-:422
                                rad = rad + (accmat * le) * scale;
                            } else {
                                const float omega = (2.0f * kPi) * (1.0f - cos_a_max);                     // :421
                                rad = rad + ((accmat_over_pi * dm::gmax(dot(l, nl), 0.0f)) * le) * omega;  // :422
                            }
                        }
                    }
                    // :426-:428 (uniform: no lane of the wave bounces off a diffuse SPHERE — the light — in almost every iteration)
                    if (__ballot(is_sphere) == 0ull) { MC_REGION(9); rd = cosine_bounce_wall<Fast>(id, rnd); }
                    else { MC_REGION(10); rd = cosine_bounce<Fast, true>(nl, rnd); }
                    emissive = 0.0f;                                              // :429
                } else {                                                          // :432 mirror, :437 glass (box_ok: 2 or 3)
                    MC_REGION(11);   // mirror / glass
                    if constexpr (Fast) rd = specular_bounce_fast<Fast>(mat, rd, n, dot_n_rd, rnd.x, accmat);
                    else rd = specular_bounce_general<false, true>(mat, rd, n, nl, dot_n_rd, rnd.x, accmat);
                    emissive = 1.0f;                                              // :447
                }
                key++;
                go = key != kend;                                                 // :367 depth limit
                if (go) {
                    MC_REGION(14);   // intersection of the next depth
                    id = intersect_slab<Fast, (Fast != 0), kOccR2, 16>(hot, ro, rd, t, false, occ, xoc);
                    go = id >= 0;                                                 // :369
                }
                if (go) {                                                         // the next bounce's random numbers and roulette
                    const v3 rn = rand01(gx, gy, key);                            // :393 (key = samp * maxDepth + depth)
                    rx = rn.x; ry = rn.y;
                    if (key > krr) {                                              // :395 depth > 5
                        MC_REGION(15);   // roulette
                        const float4* nobj = reinterpret_cast<const float4*>(lds_obj + kPoolRecordStride * (uint32_t)id);
                        go = !(rn.z >= nobj[Fast ? 2 : 0].w);                     // :396
                        // a path the roulette ends has still gathered the emission of this hit (:391 precedes :396)
                        if (__ballot(!go && dm::as_uint(nobj[1].w) >= 256u) != 0ull) {
                            const float4 o2 = nobj[2];
                            if (!go) rad = rad + (accmat * v3{o2.x, o2.y, o2.z}) * emissive;
                        }
                    }
                }
                }
                if constexpr (!Fast) {
                    if (!go) {   // the path has ended: :452's accrad / samps.y into the sample's slot of the result ring
                        const v3 q = divs_recip<Fast>(accrad, fspp, a.inv_spp);
                        const uint32_t slot = cur % RRing;
                        gres[slot] = q.x; gres[RRing + slot] = q.y; gres[2u * RRing + slot] = q.z;
                    }
                }
                alive = go;
            }
        }
    }
    // the bound tripped (a scheduling defect): say so — the image is incomplete (mc_context::check_status -> MC_ERR_HIP)
    if (!finished && a.status && (threadIdx.x & 63u) == 0u) atomicOr(a.status, 2u);
    float4 sum = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const Lane fin = my_lane(true);
    if constexpr (Fast) {
        // ---- :452-:453: the S partial sums of a pixel, added in lane order by each of its lanes (all S copies identical)
        auto from_lane = [](float v, uint32_t src) {
            return __int_as_float(__builtin_amdgcn_ds_bpermute((int)(src << 2), __float_as_int(v)));
        };
        for (uint32_t k = 0; k < (uint32_t)S; k++) {
            const uint32_t src = fin.lane - fin.sub + k;
            sum.x += from_lane(acc.x, src); sum.y += from_lane(acc.y, src); sum.z += from_lane(acc.z, src);
        }
        sum.x *= a.inv_spp; sum.y *= a.inv_spp; sum.z *= a.inv_spp;
        // a later sample range of a progressive render (the samps.x protocol, :451-:452): the range's share is added to the stored
        // accumulator.  (Fast math has no ordered-sum contract: 5 ranges and one launch agree within the tolerance, not bit for bit;
        // the SAME split composes identically on every tiling.)
        if (a.sample_begin > 0u && fin.valid) {
            const float4 prev = a.out[(size_t)fin.ty * a.W + gx];
            sum.x += prev.x; sum.y += prev.y; sum.z += prev.z;
        }
    } else {
        // ---- every sample of the pool has ended: the batches not yet added, in sample order (entries beyond the sample range hold zeros)
        while (committed < n_batches * (uint32_t)S) {
            for (uint32_t k = 0; k < (uint32_t)S; k++) {
                const uint32_t slot = (committed + k) % RRing;
                acc.x += gres[slot]; acc.y += gres[RRing + slot]; acc.z += gres[2u * RRing + slot];
            }
            committed += (uint32_t)S;
        }
        sum = make_float4(acc.x, acc.y, acc.z, 0.0f);
    }
    if (a.sample_end == a.spp) {                                                // :453 after sample spp-1
        sum.x = dm::fpow<Fast>(dm::gmin(dm::gmax(sum.x, 0.0f), 1.0f), 0.45f) * 255.0f + 0.5f;
        sum.y = dm::fpow<Fast>(dm::gmin(dm::gmax(sum.y, 0.0f), 1.0f), 0.45f) * 255.0f + 0.5f;
        sum.z = dm::fpow<Fast>(dm::gmin(dm::gmax(sum.z, 0.0f), 1.0f), 0.45f) * 255.0f + 0.5f;
    }
    if (fin.valid && fin.sub == 0u) a.out[(size_t)fin.ty * a.W + gx] = sum;
}

template <int Fast, int S, int NS> inline int launch_pool_one(const PTArgs& a, uint32_t tile_rows, hipStream_t s) {
    dim3 grid((a.W + block_w<S>() - 1u) / block_w<S>(), (tile_rows + block_h<S>() - 1u) / block_h<S>());
    constexpr size_t lds = pool_block_lds_bytes<Fast, NS>();
    if (Fast && (!a.scene.spheres_disjoint || MC_PT_FAST_NO_DISJOINT)) hipLaunchKernelGGL((pathtrace_pool_kernel<Fast, S, NS, false>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((pathtrace_pool_kernel<Fast, S, NS, true>), grid, dim3(256), lds, s, a);
    return MC_OK;
}
// variant 4 of launch_fast / launch_strict: 16 lanes per pixel and batch (the host never passes anything else); one instantiation
// per sphere count 1 .. kMaxSlabSpheres
template <int Fast> inline int launch_pool(const PTArgs& a, int S, uint32_t tile_rows, hipStream_t s) {
    if (S != 16) return MC_ERR_INVALID_ARGUMENT;
    switch (a.scene.n_spheres) {
        case 1: return launch_pool_one<Fast, 16, 1>(a, tile_rows, s);
        case 2: return launch_pool_one<Fast, 16, 2>(a, tile_rows, s);
        case 3: return launch_pool_one<Fast, 16, 3>(a, tile_rows, s);
        case 4: return launch_pool_one<Fast, 16, 4>(a, tile_rows, s);
        case 5: return launch_pool_one<Fast, 16, 5>(a, tile_rows, s);
        case 6: return launch_pool_one<Fast, 16, 6>(a, tile_rows, s);
        case 7: return launch_pool_one<Fast, 16, 7>(a, tile_rows, s);
        case 8: return launch_pool_one<Fast, 16, 8>(a, tile_rows, s);
        default: return MC_ERR_INVALID_ARGUMENT;
    }
}

}  // namespace pt
}  // namespace mc
