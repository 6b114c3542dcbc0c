// Scheduling simulator for the path tracer's sample-pool kernel (csrc/pathtrace_pool.h) — a design tool, not product code.
//
// Traces real paths of the default scene with the CPU oracle (TEST INFRASTRUCTURE, oracle/oracle_core.h), records for every sample
// the sequence of events "bounce k hit material m / was terminated by Russian roulette", and replays those sequences through models
// of the kernels' lane schedulers, counting wave ITERATIONS (one bounce of every live lane) and the lanes that are inside a bounce:
//   rounds     : the round-synchronous kernels (a wave = 4 pixels x 16 samples, all lanes step together, a round costs its longest path)
//   pool       : the sample-pool kernel as built — a wave owns a 2 x 2 tile; the 16 lanes of a pixel share its samples; camera rays
//                are produced one batch (16 samples x 4 pixels) at a time into a two-batch stash; a batch is produced when some pixel
//                wants more entries than it has and no pixel has more than 16 waiting; a free lane takes its pixel's next entry
//   continuous : VERDICT r3 item 1 — the same wave walks a SEQUENCE of L tiles; when a tile's pool runs dry the free lanes of a pixel
//                start the same pixel of the next tile instead of idling (fill / drain paid once per sequence instead of per tile)
// and the schedulers round 2 / 3 measured and rejected are gone with their kernels (profiles/r02*_regroup_*, r03_sched_sim_pool.txt).
// Prints iterations per 64 samples and the lanes in a bounce — the two numbers tools/pool_region_stats.py measures on the GPU
// (K2: 294.5 iterations per wave and 61.7 of 64 lanes, profiles/r03_pool_region_stats.txt) — and the iterations a perfect schedule
// would need, so that the gain of a scheduling change is known before it is built.
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -o tools/bin/sched_sim tools/sched_sim.cpp
//   tools/bin/sched_sim [--spp 500] [--seqs 48] [--seed 1]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../oracle/oracle_core.h"

namespace {

thread_local std::vector<uint8_t>* g_events = nullptr;
struct TracePolicy : oracle::PlainPolicy {
    static inline void c_bounce() { g_events->push_back(0); }             // a hit; the material bits stay 0 if RR terminates it
    static inline void c_hit(int type) { g_events->back() |= (uint8_t)(type ? 4 : 0); }   // bit 2: the object is a sphere
    static inline void c_material(int m) { g_events->back() |= (uint8_t)m; }
};

const float kPlanes[6 * 12] = {
    -1.0f, +0.0f, +0.0f, +2.6f, 0, 0, 0, 0, .85f, .25f, .25f, 1, +1.0f, +0.0f, +0.0f, +2.6f, 0, 0, 0, 0, .25f, .35f, .85f, 1,
    +0.0f, +1.0f, +0.0f, +2.0f, 0, 0, 0, 0, .75f, .75f, .75f, 1, +0.0f, -1.0f, +0.0f, +2.0f, 0, 0, 0, 0, .75f, .75f, .75f, 1,
    +0.0f, +0.0f, -1.0f, +2.8f, 0, 0, 0, 0, .85f, .85f, .25f, 1, +0.0f, +0.0f, +1.0f, +7.9f, 0, 0, 0, 0, 0.1f, 0.7f, 0.7f, 1,
};
const float kSpheres[3 * 12] = {
    -1.3f, -1.2f, -1.3f, 0.8f, 0, 0, 0, 0, .999f, .999f, .999f, 2, 1.3f, -1.2f, -0.2f, 0.8f, 0, 0, 0, 0, .999f, .999f, .999f, 3,
    0, 1.6f, 0, 0.2f, 100, 100, 100, 0, 0, 0, 0, 1,
};

// Bounce iterations a path spends in the pool kernel: one per executed material.  (Russian roulette is decided at the END of the
// previous bounce there, so the hit it terminates costs no iteration of its own; a path that reaches the depth limit ends after
// its last material.)
int path_iterations(const std::vector<uint8_t>& ev) {
    int n = 0;
    for (uint8_t e : ev) n += (e & 3) != 0;
    return n;
}

struct Stats {
    double iters = 0, lane_bounces = 0, batches = 0;
    long samples = 0;
    void print(const char* name, double ideal_iters) const {
        printf("%-46s iterations / 64 samples %6.3f (%.3fx the perfect schedule)   lanes in a bounce %5.2f   batches %.0f\n", name,
               iters * 64.0 / samples, iters / ideal_iters, lane_bounces / iters, batches);
    }
};

// A sequence of `L` tiles, each 4 pixels x spp samples: len[tile][pixel][sample] = bounce iterations of that sample's path.
using Seq = std::vector<std::vector<std::vector<int>>>;

// The round-synchronous kernels: per round of 16 samples of each of the 4 pixels, as many iterations as the longest path has hits
// (a path Russian roulette ends still sits through the intersection + prologue of that hit: +1 for those).
void sim_rounds(const std::vector<std::vector<std::vector<std::vector<uint8_t>>>>& ev, int spp, Stats& st) {
    for (auto& tile : ev)
        for (int base = 0; base < spp; base += 16) {
            size_t longest = 0;
            for (int p = 0; p < 4; p++)
                for (int s = base; s < std::min(spp, base + 16); s++) {
                    longest = std::max(longest, tile[p][s].size());
                    st.lane_bounces += path_iterations(tile[p][s]);
                    st.samples++;
                }
            st.iters += (double)longest;
        }
}

// The pool kernel (continuous = false: every tile on its own, as built; true: the sequence as ONE pool per pixel slot).
// Exactly the control flow of pathtrace_pool_kernel: refill, produce, take, exit test, one bounce.
void sim_pool(const Seq& seq, int spp, bool continuous, Stats& st) {
    const int S = 16;
    const int per_tile = ((spp + S - 1) / S) * S;                  // stash entries per pixel and tile (the ragged tail holds empty entries)
    const size_t groups = continuous ? 1 : seq.size();
    for (size_t g = 0; g < groups; g++) {
        const size_t t0 = continuous ? 0 : g, t1 = continuous ? seq.size() : g + 1;
        const long n_batches = (long)(t1 - t0) * (per_tile / S);
        long batch = 0, ghead[4] = {0, 0, 0, 0};
        int left[64];                                               // bounce iterations the lane's path still needs (0: free)
        for (int& l : left) l = 0;
        for (;;) {
            int need[4] = {0, 0, 0, 0};
            bool any_dead = false;
            for (int l = 0; l < 64; l++) if (!left[l]) { need[l / S]++; any_dead = true; }
            if (any_dead) {
                bool some_wants = false, some_full = false;
                for (int p = 0; p < 4; p++) {
                    const long avail = batch * S - ghead[p];
                    some_wants |= need[p] > avail;
                    some_full |= avail > S;
                }
                if (batch < n_batches && some_wants && !some_full) { batch++; st.batches++; }
                for (int p = 0; p < 4; p++) {
                    const long avail = batch * S - ghead[p];
                    long rank = 0;
                    for (int j = 0; j < S; j++) {
                        const int l = p * S + j;
                        if (left[l]) continue;
                        if (rank < avail) {
                            const long e = ghead[p] + rank;         // entry index of this pixel slot: tile e / per_tile, sample e % per_tile
                            const size_t tile = t0 + (size_t)(e / per_tile);
                            const int s = (int)(e % per_tile);
                            if (s < spp) { left[l] = seq[tile][p][s]; st.samples++; }   // (an empty entry is taken and the lane stays free)
                        }
                        rank++;
                    }
                    ghead[p] += std::min<long>(need[p], avail);
                }
            }
            int alive = 0;
            for (int l = 0; l < 64; l++) alive += left[l] > 0;
            bool drained = batch >= n_batches;
            for (int p = 0; p < 4; p++) drained &= ghead[p] == batch * S;
            if (!alive && drained) break;
            if (alive) {
                st.iters += 1;
                st.lane_bounces += alive;
                for (int l = 0; l < 64; l++) if (left[l]) left[l]--;
            }
        }
    }
}


// ---- a STRUCTURAL scheme, priced before it is built ------------------------------------------------------------
// The pool kernel's iteration costs ~375 instruction-equivalents, ~60 of which — sphere normal, mirror / glass — serve the 8 of 64
// lanes that sit on a sphere, in 98 % of the iterations (profiles/r03_pool_region_stats.txt).  "split": a lane whose path's NEXT hit is
// a sphere PARKS that path (16 dwords through LDS, one slot per lane) and takes its pixel's next sample; iterations then bounce off
// walls only.  When `threshold` lanes hold a parked (or stalled: slot occupied) sphere path, ONE sphere phase runs those bounces at
// full width: the parked path changes places with the lane's active one, which waits in the slot as "ready" and is taken back before
// any new sample.  No path state moves between lanes; a wave's schedule still depends on nothing outside the wave.
// Both schedulers are priced with the SAME per-block costs (static VALU instructions x runs, transcendentals at 5: the table in
// profiles/r03_pool_region_stats.txt), a block being paid whenever ANY lane of the wave needs it — as the hardware does.
struct Costs {
    double control = 16, take = 12, batch = 174, prologue = 40, normal = 14, specular = 45, diffuse = 92, wall_bounce = 32,
           general_bounce = 32, intersect = 45, roots = 28, rand = 20, roulette = 10;
    double park = 14, unpark = 14, swap = 30;   // split only: LDS traffic + address / flag bookkeeping, per block execution (--park / --swap)
};
struct Path { const std::vector<uint8_t>* ev; size_t pos; };   // pos: index of the next EXECUTED bounce (events with material bits)
static bool path_done(const Path& p) { return !p.ev || p.pos >= p.ev->size() || ((*p.ev)[p.pos] & 3) == 0; }
static bool wants_sphere(const Path& p) { return !path_done(p) && ((*p.ev)[p.pos] & 4); }

struct SplitStats { double cost = 0, iters = 0, phases = 0, lane_work = 0, stalled = 0; long samples = 0; };

// events[pixel][sample]; split = false prices the pool kernel as built (one block schedule, every block paid when any lane needs it)
void sim_cost(const std::vector<std::vector<std::vector<uint8_t>>>& tile, int spp, bool split, int threshold, const Costs& c, SplitStats& st) {
    const int S = 16;
    const long n_batches = (spp + S - 1) / S;
    long batch = 0, ghead[4] = {0, 0, 0, 0};
    Path act[64], slot[64];
    int slot_state[64];   // 0 empty, 1 parked (wants a sphere phase), 2 ready
    for (int l = 0; l < 64; l++) { act[l] = Path{nullptr, 0}; slot[l] = Path{nullptr, 0}; slot_state[l] = 0; }
    auto free_lane = [&](int l) { return path_done(act[l]); };
    for (long guard = 0; guard < 100000000; guard++) {
        double cost = c.control;
        // ---- refill: a ready path first, else the pixel's next stash entry
        bool any_unpark = false, any_take = false;
        if (split)
            for (int l = 0; l < 64; l++)
                if (free_lane(l) && slot_state[l] == 2) { act[l] = slot[l]; slot_state[l] = 0; any_unpark = true; }
        int need[4] = {0, 0, 0, 0};
        bool any_dead = false;
        for (int l = 0; l < 64; l++) if (free_lane(l)) { need[l / S]++; any_dead = true; }
        if (any_dead) {
            bool some_wants = false, some_full = false;
            for (int p = 0; p < 4; p++) {
                const long avail = batch * S - ghead[p];
                some_wants |= need[p] > avail;
                some_full |= avail > S;
            }
            if (batch < n_batches && some_wants && !some_full) { batch++; cost += c.batch; }
            for (int p = 0; p < 4; p++) {
                const long avail = batch * S - ghead[p];
                long rank = 0;
                for (int j = 0; j < S; j++) {
                    const int l = p * S + j;
                    if (!free_lane(l)) continue;
                    if (rank < avail) {
                        const long e = ghead[p] + rank;
                        if (e < spp) { act[l] = Path{&tile[p][(size_t)e], 0}; st.samples++; any_take = true; }
                    }
                    rank++;
                }
                ghead[p] += std::min<long>(need[p], avail);
            }
            cost += c.take * (any_take ? 1 : 0);
        }
        if (any_unpark) cost += c.unpark;
        // ---- anything left?
        int alive = 0, parked = 0;
        for (int l = 0; l < 64; l++) { alive += !free_lane(l); parked += slot_state[l] != 0; }
        bool drained = batch >= n_batches;
        for (int p = 0; p < 4; p++) drained &= ghead[p] == batch * S;
        if (!alive && !parked && drained) break;
        auto bounce_blocks = [&](bool sphere_phase, int& served) {   // prices one bounce of the lanes that take part; advances them
            bool any_sphere = false, any_spec = false, any_diffuse = false, any_wall = false, any_gen = false, any_rr = false;
            served = 0;
            for (int l = 0; l < 64; l++) {
                if (free_lane(l)) continue;
                const bool sph = wants_sphere(act[l]);
                if (split && sph != sphere_phase) continue;      // not this kind of iteration: the lane idles (stalled)
                const uint8_t e = (*act[l].ev)[act[l].pos];
                const int mat = e & 3;
                any_sphere |= sph; any_spec |= mat != 1; any_diffuse |= mat == 1; any_wall |= mat == 1 && !sph; any_gen |= mat == 1 && sph;
                any_rr |= act[l].pos >= 5;
                act[l].pos++;
                served++;
            }
            double k = c.prologue + c.intersect + c.roots + c.rand;
            if (any_sphere) k += c.normal;
            if (any_spec) k += c.specular;
            if (any_diffuse) k += c.diffuse;
            if (any_wall) k += c.wall_bounce;
            if (any_gen) k += c.general_bounce;
            if (any_rr) k += c.roulette;
            return k;
        };
        int served = 0;
        if (!split) {
            if (alive) { cost += bounce_blocks(false, served); st.iters++; st.lane_work += served; }
        } else {
            int want = 0, wall = 0;
            for (int l = 0; l < 64; l++) {
                if (slot_state[l] == 1) want++;
                if (!free_lane(l)) { if (wants_sphere(act[l])) want++; else wall++; }
            }
            const bool phase = want > 0 && (want >= threshold || wall == 0);
            if (phase) {
                // a parked path changes places with the lane's active one (which waits as "ready"); a stalled one is in place already
                for (int l = 0; l < 64; l++)
                    if (slot_state[l] == 1) {
                        if (!free_lane(l) && wants_sphere(act[l])) continue;   // both want the phase: the active one first, the slot stays
                        const Path a = act[l];
                        act[l] = slot[l];
                        if (!path_done(a)) { slot[l] = a; slot_state[l] = 2; } else slot_state[l] = 0;
                    }
                cost += c.swap + bounce_blocks(true, served);
                st.phases++; st.lane_work += served;
            } else if (wall) {
                cost += bounce_blocks(false, served);
                st.iters++; st.lane_work += served;
            }
            // a path whose next hit is a sphere parks as soon as its lane's slot is free (end of the iteration: the hit is known)
            bool any_park = false;
            for (int l = 0; l < 64; l++)
                if (!free_lane(l) && wants_sphere(act[l])) {
                    if (slot_state[l] == 0) { slot[l] = act[l]; slot_state[l] = 1; act[l] = Path{nullptr, 0}; any_park = true; }
                    else st.stalled++;
                }
            if (any_park) cost += c.park;
        }
        st.cost += cost;
    }
}


// "two paths per lane" (the form VERDICT r4 item 6 names): a lane holds TWO paths in registers and advances, in every iteration, one
// that fits the iteration's kind — a wall iteration (no sphere blocks) or, when `threshold` lanes hold a path whose next hit is a
// sphere, a sphere iteration.  A lane idles only when neither of its paths fits.  `select` = what choosing the working path costs an
// iteration (the state of both paths lives in registers: a select per state register read, or a masked swap).
void sim_two_paths(const std::vector<std::vector<std::vector<uint8_t>>>& tile, int spp, int threshold, const Costs& c, double select, SplitStats& st,
                   bool mixed = false) {   // mixed: in a sphere iteration the other lanes bounce off walls (every block paid, as today)
    const int S = 16;
    const long n_batches = (spp + S - 1) / S;
    long batch = 0, ghead[4] = {0, 0, 0, 0};
    Path pa[64][2];
    for (auto& q : pa) q[0] = q[1] = Path{nullptr, 0};
    for (long guard = 0; guard < 100000000; guard++) {
        double cost = c.control + select;
        int need[4] = {0, 0, 0, 0};
        bool any_dead = false;
        for (int l = 0; l < 64; l++) for (int k = 0; k < 2; k++) if (path_done(pa[l][k])) { need[l / S]++; any_dead = true; }
        if (any_dead) {
            bool some_wants = false, some_full = false, any_take = false;
            for (int p = 0; p < 4; p++) {
                const long avail = batch * S - ghead[p];
                some_wants |= need[p] > avail;
                some_full |= avail > S;
            }
            if (batch < n_batches && some_wants && !some_full) { batch++; cost += c.batch; }
            for (int p = 0; p < 4; p++) {
                const long avail = batch * S - ghead[p];
                long rank = 0;
                for (int j = 0; j < S; j++)
                    for (int k = 0; k < 2; k++) {
                        const int l = p * S + j;
                        if (!path_done(pa[l][k])) continue;
                        if (rank < avail) {
                            const long e = ghead[p] + rank;
                            if (e < spp) { pa[l][k] = Path{&tile[p][(size_t)e], 0}; st.samples++; any_take = true; }
                        }
                        rank++;
                    }
                ghead[p] += std::min<long>(need[p], avail);
            }
            if (any_take) cost += c.take;
        }
        int alive = 0, want = 0, wall = 0;
        for (int l = 0; l < 64; l++) {
            bool w = false, s = false;
            for (int k = 0; k < 2; k++) if (!path_done(pa[l][k])) { alive++; (wants_sphere(pa[l][k]) ? s : w) = true; }
            want += s; wall += w;
        }
        bool drained = batch >= n_batches;
        for (int p = 0; p < 4; p++) drained &= ghead[p] == batch * S;
        if (!alive && drained) break;
        const bool phase = want > 0 && (want >= threshold || wall == 0);
        bool any_spec = false, any_diffuse = false, any_wall = false, any_gen = false, any_rr = false;
        int served = 0;
        for (int l = 0; l < 64; l++) {
            int pick = -1;
            for (int k = 0; k < 2; k++) if (!path_done(pa[l][k]) && wants_sphere(pa[l][k]) == phase) { pick = k; break; }
            if (pick < 0 && mixed && phase)
                for (int k = 0; k < 2; k++) if (!path_done(pa[l][k])) { pick = k; break; }
            if (pick < 0) { if (!path_done(pa[l][0]) || !path_done(pa[l][1])) st.stalled++; continue; }
            Path& q = pa[l][pick];
            const int mat = (*q.ev)[q.pos] & 3;
            const bool sph = wants_sphere(q);
            any_spec |= mat != 1; any_diffuse |= mat == 1; any_wall |= mat == 1 && !sph; any_gen |= mat == 1 && sph; any_rr |= q.pos >= 5;
            q.pos++;
            served++;
        }
        double k = c.prologue + c.intersect + c.roots + c.rand;
        if (phase) k += c.normal;
        if (any_spec) k += c.specular;
        if (any_diffuse) k += c.diffuse;
        if (any_wall) k += c.wall_bounce;
        if (any_gen) k += c.general_bounce;
        if (any_rr) k += c.roulette;
        cost += k;
        (phase ? st.phases : st.iters)++;
        st.lane_work += served;
        st.cost += cost;
    }
}


// "sphere service" (the one scheme that moves path state BETWEEN WAVES of a block, CHANGELOG round-3 §9.3): a block = NW wall waves,
// each a sample pool as built but bouncing off walls only, + ONE service wave that owns no pixels.  A wall-wave lane whose path's next
// hit is a sphere pushes the path (16 dwords) into the block's LDS queue and takes its pixel's next sample; the service wave pops up to
// 64 queued paths and bounces them at full width (normal + mirror / glass + intersection), keeps those that hit a sphere again, and
// returns the others to their home wave, whose free lanes take returned paths before new samples.  Every wave runs on its own SIMD at
// the same rate: a discrete-event replay, every wave with its own clock in instruction-equivalents; the cost compared is the SUM over
// the waves (an idle wave issues nothing).  `xfer` = what one block execution of push / pop costs (LDS traffic + bookkeeping).
void sim_service(const std::vector<std::vector<std::vector<std::vector<uint8_t>>>>& tiles, size_t t0, int NW, int spp, int threshold,
                 const Costs& c, double xfer, double poll, SplitStats& st) {
    const int S = 16;
    const long n_batches = (spp + S - 1) / S;
    struct Wall {
        long batch = 0, ghead[4] = {0, 0, 0, 0};
        Path act[64];
        std::vector<Path> returned;      // paths the service wave sent home (next hit a wall)
        long outstanding = 0;            // paths of this wave at the service wave
        double clock = 0;
        bool done = false;
    };
    std::vector<Wall> w((size_t)NW);
    for (auto& q : w) for (auto& p : q.act) p = Path{nullptr, 0};
    struct Queued { Path p; int home; double ready; };
    std::vector<Queued> queue;           // the block's sphere queue (FIFO)
    Path sact[64]; int shome[64];
    for (auto& p : sact) p = Path{nullptr, 0};
    double sclock = 0;
    auto wall_step = [&](int wi) {
        Wall& q = w[(size_t)wi];
        const auto& tile = tiles[t0 + (size_t)wi];
        double cost = c.control + poll;
        bool any_ret = false, any_take = false;
        for (int l = 0; l < 64 && !q.returned.empty(); l++)
            if (path_done(q.act[l])) { q.act[l] = q.returned.back(); q.returned.pop_back(); any_ret = true; }
        int need[4] = {0, 0, 0, 0};
        bool any_dead = false;
        for (int l = 0; l < 64; l++) if (path_done(q.act[l])) { need[l / S]++; any_dead = true; }
        if (any_dead) {
            bool some_wants = false, some_full = false;
            for (int p = 0; p < 4; p++) { const long avail = q.batch * S - q.ghead[p]; some_wants |= need[p] > avail; some_full |= avail > S; }
            if (q.batch < n_batches && some_wants && !some_full) { q.batch++; cost += c.batch; }
            for (int p = 0; p < 4; p++) {
                const long avail = q.batch * S - q.ghead[p];
                long rank = 0;
                for (int j = 0; j < S; j++) {
                    const int l = p * S + j;
                    if (!path_done(q.act[l])) continue;
                    if (rank < avail) {
                        const long e = q.ghead[p] + rank;
                        if (e < spp) { q.act[l] = Path{&tile[(size_t)p][(size_t)e], 0}; st.samples++; any_take = true; }
                    }
                    rank++;
                }
                q.ghead[p] += std::min<long>(need[p], avail);
            }
        }
        if (any_take) cost += c.take;
        if (any_ret) cost += xfer;
        // paths whose (next) hit is a sphere go to the service wave
        bool any_push = false;
        for (int l = 0; l < 64; l++)
            if (wants_sphere(q.act[l])) { queue.push_back(Queued{q.act[l], wi, q.clock}); q.act[l] = Path{nullptr, 0}; q.outstanding++; any_push = true; }
        int alive = 0;
        for (int l = 0; l < 64; l++) alive += !path_done(q.act[l]);
        bool drained = q.batch >= n_batches;
        for (int p = 0; p < 4; p++) drained &= q.ghead[p] == q.batch * S;
        if (!alive && drained && !q.outstanding && q.returned.empty()) { q.done = true; q.clock += cost; st.cost += cost; return; }
        if (alive) {
            bool any_rr = false;
            int served = 0;
            for (int l = 0; l < 64; l++)
                if (!path_done(q.act[l])) { any_rr |= q.act[l].pos >= 5; q.act[l].pos++; served++; }
            cost += c.prologue + c.diffuse + c.wall_bounce + c.intersect + c.roots + c.rand + (any_rr ? c.roulette : 0.0);
            st.iters++; st.lane_work += served;
            for (int l = 0; l < 64; l++)   // the next hit is known at the end of the bounce: push at once
                if (wants_sphere(q.act[l])) { queue.push_back(Queued{q.act[l], wi, q.clock + cost}); q.act[l] = Path{nullptr, 0}; q.outstanding++; any_push = true; }
        } else {
            cost += 40.0;   // nothing to bounce: waiting for the service wave (a short sleep; counted)
            st.stalled++;
        }
        if (any_push) cost += xfer;
        q.clock += cost; st.cost += cost;
    };
    auto service_step = [&]() -> bool {   // returns false when there is nothing it could do yet
        int occupied = 0;
        for (int l = 0; l < 64; l++) occupied += !path_done(sact[l]);
        bool any_pop = false;
        size_t k = 0;
        for (int l = 0; l < 64 && k < queue.size(); l++)
            if (path_done(sact[l])) {
                while (k < queue.size() && queue[k].ready > sclock) k++;
                if (k >= queue.size()) break;
                sact[l] = queue[k].p; shome[l] = queue[k].home; queue.erase(queue.begin() + (long)k); occupied++; any_pop = true;
            }
        bool walls_waiting = true;
        for (auto& q : w) if (!q.done) { int a = 0; for (int l = 0; l < 64; l++) a += !path_done(q.act[l]); if (a || q.batch < n_batches) walls_waiting = false; }
        if (!occupied) return false;
        if (occupied < threshold && !walls_waiting) { if (any_pop) { sclock += xfer; st.cost += xfer; } return false; }
        double cost = c.control + poll + (any_pop ? xfer : 0.0) + c.prologue + c.normal + c.intersect + c.roots + c.rand;
        bool any_spec = false, any_diffuse = false, any_rr = false, any_ret = false;
        int served = 0;
        for (int l = 0; l < 64; l++)
            if (!path_done(sact[l])) {
                const int mat = (*sact[l].ev)[sact[l].pos] & 3;
                any_spec |= mat != 1; any_diffuse |= mat == 1; any_rr |= sact[l].pos >= 5;
                sact[l].pos++; served++;
                if (!wants_sphere(sact[l])) {    // a wall next, or the path has ended (its radiance goes home with it)
                    Wall& h = w[(size_t)shome[l]];
                    if (!path_done(sact[l])) h.returned.push_back(sact[l]);
                    h.outstanding--;
                    sact[l] = Path{nullptr, 0};
                    any_ret = true;
                }
            }
        if (any_spec) cost += c.specular;
        if (any_diffuse) cost += c.diffuse + c.general_bounce;
        if (any_rr) cost += c.roulette;
        if (any_ret) cost += xfer;
        st.phases++; st.lane_work += served;
        sclock += cost; st.cost += cost;
        return true;
    };
    for (long guard = 0; guard < 50000000; guard++) {
        // the wave with the smallest clock runs next
        int pick = -1;
        double best = 1e300;
        for (int i = 0; i < NW; i++) if (!w[(size_t)i].done && w[(size_t)i].clock < best) { best = w[(size_t)i].clock; pick = i; }
        bool all_done = pick < 0;
        if (all_done) break;
        if (sclock <= best) {
            if (!service_step()) sclock = best + 1e-9;   // nothing to do yet: sleep until the next wall wave has moved
        } else {
            wall_step(pick);
        }
    }
}

}  // namespace

int main(int argc, char** argv) {
    int W = 900, H = 600, spp = 500, max_depth = 12, n_seqs = 48, Lmax = 8;
    unsigned seed = 1;
    Costs costs;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--spp")) spp = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--seqs")) n_seqs = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--seed")) seed = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--park")) costs.park = costs.unpark = atof(argv[i + 1]);
        if (!strcmp(argv[i], "--swap")) costs.swap = atof(argv[i + 1]);
    }
    oracle::PT<TracePolicy> pt;
    pt.planes = kPlanes; pt.nPlanes = 6; pt.spheres = kSpheres; pt.nSpheres = 3; pt.mathMode = oracle::MATH_LIBM;
    std::mt19937 rng(seed);
    // n_seqs horizontal runs of Lmax consecutive 2 x 2 tiles at random places of the image (a sequence of the continuous kernel
    // is a function of global coordinates inside one tile row, so that every row tiling walks the same sequences)
    std::vector<std::vector<std::vector<std::vector<std::vector<uint8_t>>>>> ev(n_seqs);
    double hits = 0, nsamp = 0;
    for (auto& sq : ev) {
        const int x0 = (int)(rng() % ((W - 2 * Lmax) / 2)) * 2, y0 = (int)(rng() % (H / 2)) * 2;
        sq.resize(Lmax);
        for (int k = 0; k < Lmax; k++) {
            sq[k].resize(4);
            for (int p = 0; p < 4; p++) {
                sq[k][p].resize(spp);
                const int gx = x0 + 2 * k + p % 2, gy = y0 + p / 2;
                for (int s = 0; s < spp; s++) {
                    g_events = &sq[k][p][s];
                    pt.sample(gx, gy, W, H, s, max_depth);
                    hits += sq[k][p][s].size(); nsamp++;
                }
            }
        }
    }
    printf("traced %d sequences x %d tiles x 4 pixels x %d spp: %.2f hits per sample\n", n_seqs, Lmax, spp, hits / nsamp);
    double lane_iters = 0;
    std::vector<Seq> seqs(n_seqs);
    for (int q = 0; q < n_seqs; q++) {
        seqs[q].resize(Lmax);
        for (int k = 0; k < Lmax; k++) {
            seqs[q][k].resize(4);
            for (int p = 0; p < 4; p++) {
                seqs[q][k][p].resize(spp);
                for (int s = 0; s < spp; s++) { seqs[q][k][p][s] = path_iterations(ev[q][k][p][s]); lane_iters += seqs[q][k][p][s]; }
            }
        }
    }
    const double ideal = lane_iters / 64.0;   // every lane inside a bounce in every iteration
    printf("bounce iterations per sample %.3f; a perfect schedule needs %.3f iterations per 64 samples\n", lane_iters / nsamp, ideal * 64.0 / nsamp);
    Stats r;
    for (auto& sq : ev) sim_rounds(sq, spp, r);
    r.print("rounds (round-synchronous kernels)", ideal);
    Stats p1;
    for (auto& sq : seqs) sim_pool(sq, spp, false, p1);
    p1.print("pool, one tile per wave (as built)", ideal);
    for (int L : {2, 4, 8}) {
        Stats c;
        for (auto& sq : seqs)
            for (int k = 0; k + L <= Lmax; k += L) sim_pool(Seq(sq.begin() + k, sq.begin() + k + L), spp, true, c);
        char name[96];
        snprintf(name, sizeof name, "continuous, %d tiles per wave", L);
        c.print(name, ideal);
        printf("%-46s -> %.2f %% fewer iterations than one tile per wave\n", "", 100.0 * (1.0 - c.iters / p1.iters));
    }
    // ---- the split scheme (sphere bounces batched into phases) against the pool kernel as built, both priced per block
    {
        SplitStats base;
        for (auto& sq : ev) for (auto& tile : sq) sim_cost(tile, spp, false, 0, costs, base);
        printf("\ncost model, instruction-equivalents per sample (park / unpark %.0f, swap %.0f per block execution):\n", costs.park, costs.swap);
        printf("  %-44s %8.2f per sample, %6.1f per iteration, %5.2f lanes served (%ld samples)\n", "pool kernel as built",
               base.cost / base.samples, base.cost / base.iters, base.lane_work / base.iters, base.samples);
        for (int T : {8, 16, 24, 32, 48}) {
            SplitStats sp;
            for (auto& sq : ev) for (auto& tile : sq) sim_cost(tile, spp, true, T, costs, sp);
            printf("  split, sphere phase at %2d waiting lanes       %8.2f per sample (%+6.2f %%): %.1f wall iterations + %.1f phases per 64 samples, "
                   "%.2f lanes per bounce block, %.2f stalled lanes per block\n", T, sp.cost / sp.samples,
                   100.0 * (sp.cost / sp.samples / (base.cost / base.samples) - 1.0), sp.iters * 64.0 / sp.samples, sp.phases * 64.0 / sp.samples,
                   sp.lane_work / (sp.iters + sp.phases), sp.stalled / (sp.iters + sp.phases));
        }
    }
    {
        SplitStats base;
        for (auto& sq : ev) for (auto& tile : sq) sim_cost(tile, spp, false, 0, costs, base);
        for (double select : {0.0, 20.0})
            for (int T : {8, 16, 24, 32, 48}) {
                SplitStats sp;
                for (auto& sq : ev) for (auto& tile : sq) sim_two_paths(tile, spp, T, costs, select, sp);
                printf("  two paths per lane, select %2.0f, sphere iteration at %2d lanes %8.2f per sample (%+6.2f %%): %.1f wall + %.1f sphere iterations per 64 "
                       "samples, %.2f lanes per block, %.2f idle lanes\n", select, T, sp.cost / sp.samples,
                       100.0 * (sp.cost / sp.samples / (base.cost / base.samples) - 1.0), sp.iters * 64.0 / sp.samples, sp.phases * 64.0 / sp.samples,
                       sp.lane_work / (sp.iters + sp.phases), sp.stalled / (sp.iters + sp.phases));
            }
    }
    {
        SplitStats base;
        for (auto& sq : ev) for (auto& tile : sq) sim_cost(tile, spp, false, 0, costs, base);
        for (double select : {0.0, 20.0})
            for (int T : {4, 8, 16, 24, 32}) {
                SplitStats sp;
                for (auto& sq : ev) for (auto& tile : sq) sim_two_paths(tile, spp, T, costs, select, sp, true);
                printf("  two paths per lane, MIXED sphere iterations, select %2.0f, at %2d lanes %8.2f per sample (%+6.2f %%): %.1f wall-only + %.1f mixed iterations "
                       "per 64 samples, %.2f lanes per block, %.2f idle lanes\n", select, T, sp.cost / sp.samples,
                       100.0 * (sp.cost / sp.samples / (base.cost / base.samples) - 1.0), sp.iters * 64.0 / sp.samples, sp.phases * 64.0 / sp.samples,
                       sp.lane_work / (sp.iters + sp.phases), sp.stalled / (sp.iters + sp.phases));
            }
    }
    {
        SplitStats base;
        for (auto& sq : ev) for (auto& tile : sq) sim_cost(tile, spp, false, 0, costs, base);
        printf("\n  sphere service: a block = NW wall waves + 1 service wave (paths handed over through an LDS queue)\n");
        for (int NW : {3, 7})
            for (double xfer : {14.0, 30.0})
                for (int T : {16, 32, 48}) {
                    SplitStats sp;
                    for (auto& sq : ev) for (size_t t0 = 0; t0 + (size_t)NW <= sq.size(); t0 += (size_t)NW) sim_service(sq, t0, NW, spp, T, costs, xfer, 6.0, sp);
                    printf("  %d wall waves, transfer %2.0f, service at %2d lanes: %8.2f per sample (%+6.2f %%): %.1f wall iterations + %.1f service iterations per 64 samples, "
                           "%.2f lanes per bounce block, %.2f waiting wall iterations\n", NW, xfer, T, sp.cost / sp.samples,
                           100.0 * (sp.cost / sp.samples / (base.cost / base.samples) - 1.0), sp.iters * 64.0 / sp.samples, sp.phases * 64.0 / sp.samples,
                           sp.lane_work / (sp.iters + sp.phases), sp.stalled * 64.0 / sp.samples);
                }
    }
    return 0;
}
