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
    static inline void c_bounce() { g_events->push_back(0); }             // a hit; stays 0 if RR terminates it
    static inline void c_material(int m) { g_events->back() = (uint8_t)m; }
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
    for (uint8_t e : ev) n += e != 0;
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

}  // namespace

int main(int argc, char** argv) {
    int W = 900, H = 600, spp = 500, max_depth = 12, n_seqs = 48, Lmax = 8;
    unsigned seed = 1;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--spp")) spp = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--seqs")) n_seqs = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--seed")) seed = atoi(argv[i + 1]);
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
    return 0;
}
