// Scheduling simulator for the path tracer's lane-regrouping kernel — a design tool, not product code.
//
// Traces real paths of the default scene with the CPU oracle (TEST INFRASTRUCTURE, oracle/oracle_core.h), records for
// every sample the sequence of events "bounce k hit material m / was terminated by Russian roulette", and replays those
// sequences through models of the kernels' lane schedulers with a VALU-issue cost per code block:
//   rounds : the round-synchronous kernel of round 1 (a wave = 4 pixels x 16 samples, all lanes step together)
//   regroup: the round-2 scheduler — workgroup-shared FIFOs of parked paths; a wave keeps its diffuse paths in registers,
//            parks specular ones, refills vacated lanes with diffuse-ready paths, and when it cannot, spills its lanes and
//            runs a full-width batch of {parked specular paths + fresh camera samples}.
// Prints issued-instruction cost per sample, active-lane fraction, queue high-water marks and the reorder window, so that
// thresholds / capacities can be chosen before spending GPU time.
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -o tools/bin/sched_sim tools/sched_sim.cpp
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

// ---- cost model: VALU wave-instructions per execution of a code block (DESIGN.md §3.3 region counters) ----
struct Cost {
    double ip = 170, d = 190, g = 60, m = 20, cam = 70;   // fast kernel, measured (r02b): rounds = 12x170 + 11.9x190 + 10.4x60 + 8.5x20
    double ov_iter = 40;      // swap-point bookkeeping per scheduler iteration (ballots, prefix sums, LDS addressing)
    double ov_batch = 60;     // extra per S-batch (spill + reload addressing)
    double fold_round = 96;   // rounds kernel: ordered fold of 16 samples x 4 pixels
    double commit_item = 1.5; // regroup kernel: ordered commit, per item
};

static Cost g_cost;

struct Path {
    const uint8_t* ev;   // events: per bounce 0 = RR-terminated after intersect+prologue, 1/2/3 = material executed
    int n;               // number of hits recorded (a miss ends the list early)
    int max_depth;
};

struct Tile { std::vector<std::vector<uint8_t>> samples; std::vector<uint32_t> gx, gy; };   // [pixel * spp + s]

struct Stats {
    double cost = 0, useful = 0;          // issued wave-instructions, lane-weighted useful share (x64)
    double iters = 0;
    long n_samples = 0;
    int max_dq = 0, max_sq = 0, max_window = 0;
    double stall_lane_iters = 0;
};

// ---------------------------------------------------------------- rounds kernel (round 1)
void sim_rounds(const Tile& t, int pixels, int spp, int max_depth, const Cost& c, Stats& st) {
    // a wave = 4 pixels x 16 samples; a block's 16 pixels are 4 such waves
    for (int p0 = 0; p0 < pixels; p0 += 4) {
        for (int base = 0; base < spp; base += 16) {
            int pos[64], alive[64], kind[64];
            const std::vector<uint8_t>* ev[64];
            int n = 0;
            for (int l = 0; l < 64; l++) {
                int p = p0 + l / 16, s = base + l % 16;
                alive[l] = s < spp;
                ev[l] = alive[l] ? &t.samples[(size_t)p * spp + s] : nullptr;
                pos[l] = 0;
                n += alive[l];
            }
            st.cost += c.cam + c.fold_round; st.useful += n * (c.cam + c.fold_round) / 64.0 * 64.0 / 64.0 * 1.0;
            st.n_samples += n;
            for (int depth = 0; depth < max_depth; depth++) {
                int na = 0, nd = 0, ng = 0, nm = 0;
                for (int l = 0; l < 64; l++) {
                    kind[l] = -1;
                    if (!alive[l]) continue;
                    na++;
                    if (pos[l] >= (int)ev[l]->size()) { alive[l] = 0; continue; }   // miss: intersect ran, nothing else
                    int e = (*ev[l])[pos[l]++];
                    kind[l] = e;
                    if (e == 0) alive[l] = 0;
                    else if (e == 1) nd++;
                    else if (e == 2) nm++;
                    else ng++;
                }
                if (!na) break;
                st.iters += 1;
                st.cost += c.ip + (nd ? c.d : 0) + (ng ? c.g : 0) + (nm ? c.m : 0);
                st.useful += (na * c.ip + nd * c.d + ng * c.g + nm * c.m) / 64.0;
            }
        }
    }
}

// ---------------------------------------------------------------- regroup kernel (round 2)
struct Rec { int item; int pos; int kind; };   // kind: 1 D-ready, 2/3 specular pending, 4 camera (fresh)

struct RegroupCfg {
    int waves = 4;          // waves per workgroup sharing the queues
    int dq_cap = 128, sq_cap = 128;
    int window = 1024;      // reorder ring entries per workgroup (items in flight)
    int s_batch_min = 48;   // run an S-batch only if parked specular + fresh camera items reach this many
    bool share = true;
};

void sim_regroup(const Tile& t, int pix0, int pixels, int spp, int max_depth, const Cost& c, const RegroupCfg& cfg, Stats& st) {
    const int total_items = pixels * spp;   // item k -> pixel k % pixels, sample k / pixels
    int next_item = 0;
    std::vector<char> done(total_items, 0);
    int committed = 0;                       // all items < committed are done (conservative window base)
    std::vector<Rec> dq, sq;                 // FIFOs
    struct Wave { Rec lane[64]; double clock = 0; bool finished = false; };
    std::vector<Wave> W(cfg.waves);
    for (auto& w : W) for (auto& l : w.lane) l.kind = 0;
    auto path = [&](int item) -> const std::vector<uint8_t>& { return t.samples[(size_t)(pix0 + item % pixels) * spp + item / pixels]; };
    auto retire = [&](int item) {
        done[item] = 1;
        while (committed < total_items && done[committed]) committed++;
    };
    int live_waves = cfg.waves;
    while (live_waves) {
        // the wave with the smallest clock runs next (asynchronous waves of one workgroup)
        int wi = -1;
        for (int i = 0; i < cfg.waves; i++) if (!W[i].finished && (wi < 0 || W[i].clock < W[wi].clock)) wi = i;
        Wave& w = W[wi];
        // ---- swap point: classify, park specular lanes, fill vacancies
        int nD = 0, nS = 0, nE = 0;
        for (auto& l : w.lane) { if (l.kind == 1) nD++; else if (l.kind == 2 || l.kind == 3) nS++; else if (l.kind == 0) nE++; }
        // park specular lanes (if the queue has room; otherwise they stay and the iteration is mixed)
        for (auto& l : w.lane)
            if ((l.kind == 2 || l.kind == 3) && (int)sq.size() < cfg.sq_cap) { sq.push_back(l); l.kind = 0; nS--; nE++; }
        st.max_sq = std::max(st.max_sq, (int)sq.size());
        const int cam_avail = std::max(0, std::min(total_items, committed + cfg.window) - next_item);
        bool batch = false;
        if ((int)dq.size() >= nE) {
            for (auto& l : w.lane) if (l.kind == 0 && !dq.empty()) { l = dq.front(); dq.erase(dq.begin()); }
        } else if ((int)sq.size() + cam_avail >= cfg.s_batch_min && (int)dq.size() + nD <= cfg.dq_cap) {
            // S-batch: spill the D lanes, take parked specular paths + fresh camera items
            batch = true;
            for (auto& l : w.lane) if (l.kind == 1) { dq.push_back(l); l.kind = 0; }
            st.max_dq = std::max(st.max_dq, (int)dq.size());
            for (auto& l : w.lane) {
                if (l.kind != 0) continue;
                if (!sq.empty()) { l = sq.front(); sq.erase(sq.begin()); }
                else if (next_item < std::min(total_items, committed + cfg.window)) { l.item = next_item++; l.pos = 0; l.kind = 4; }
            }
        } else {
            // not enough of anything for a uniform iteration: take what there is (tail of the workgroup's work)
            for (auto& l : w.lane) if (l.kind == 0 && !dq.empty()) { l = dq.front(); dq.erase(dq.begin()); }
            for (auto& l : w.lane) {
                if (l.kind != 0) continue;
                if (!sq.empty()) { l = sq.front(); sq.erase(sq.begin()); }
                else if (next_item < std::min(total_items, committed + cfg.window)) { l.item = next_item++; l.pos = 0; l.kind = 4; }
            }
        }
        st.max_window = std::max(st.max_window, next_item - committed);
        // ---- one iteration: heads by kind, then intersect + prologue for every lane that continues
        int n1 = 0, n2 = 0, n3 = 0, n4 = 0, nip = 0;
        for (auto& l : w.lane) {
            if (l.kind == 0) continue;
            if (l.kind == 1) n1++; else if (l.kind == 2) n2++; else if (l.kind == 3) n3++; else n4++;
        }
        if (n1 + n2 + n3 + n4 == 0) {
            if (next_item >= total_items && sq.empty() && dq.empty()) { w.finished = true; live_waves--; continue; }
            // waiting for the window (or for another wave's parked paths): idle spin
            w.clock += 50; st.cost += 0; st.stall_lane_iters += 64;
            // guard: if every wave is idle and nothing can progress, bail out
            bool any = false;
            for (auto& o : W) for (auto& l : o.lane) if (l.kind) any = true;
            if (!any && sq.empty() && dq.empty() && cam_avail == 0 && next_item < total_items) { fprintf(stderr, "deadlock\n"); exit(1); }
            continue;
        }
        for (auto& l : w.lane) {
            if (l.kind == 0) continue;
            const auto& ev = path(l.item);
            // head executed (material of event pos-1 for kinds 1..3, camera for 4); does the path go on to intersect?
            bool cont = true;
            if (l.kind != 4 && l.pos >= max_depth) cont = false;            // the material block of the last depth
            if (!cont) { retire(l.item); l.kind = 0; st.n_samples++; continue; }
            nip++;
            if (l.pos >= (int)ev.size()) { retire(l.item); l.kind = 0; st.n_samples++; continue; }   // miss
            int e = ev[l.pos++];
            if (e == 0) { retire(l.item); l.kind = 0; st.n_samples++; }
            else l.kind = e;
        }
        double cst = c.ov_iter + (batch ? c.ov_batch : 0) + (n1 ? c.d : 0) + (n3 ? c.g : 0) + (n2 ? c.m : 0) + (n4 ? c.cam : 0) + (nip ? c.ip : 0);
        st.cost += cst;
        st.useful += (n1 * c.d + n3 * c.g + n2 * c.m + n4 * c.cam + nip * c.ip) / 64.0;
        st.iters += 1;
        w.clock += cst;
    }
    st.cost += c.commit_item * total_items / 1.0 * (1.0);   // ordered commit (lanes of one wave, amortised)
}

// ---------------------------------------------------------------- two path slots per lane, the parked one in LDS
// Lane-private: no queues, no atomics — a lane exchanges its register path with the one in its own LDS slot
// (ds_wrxchg_rtn_b32).  Every iteration the wave votes a mode: D (lanes that hold a diffuse-ready path run the diffuse head)
// or S (lanes that hold a specular-pending path or can start a camera sample run those heads); intersect + prologue follows.
struct TwoSlotCfg { int window = 256; int s_threshold = 40; int pixels = 4; };
void sim_twoslot(const Tile& t, int pix0, int spp, int max_depth, const Cost& c, const TwoSlotCfg& cfg, Stats& st) {
    const int pixels = cfg.pixels;
    const int total_items = pixels * spp;
    int next_item = 0, committed = 0;
    std::vector<char> done(total_items, 0);
    Rec A[64], B[64];
    for (int l = 0; l < 64; l++) A[l].kind = B[l].kind = 0;
    auto path = [&](int item) -> const std::vector<uint8_t>& { return t.samples[(size_t)(pix0 + item % pixels) * spp + item / pixels]; };
    auto retire = [&](int item) { done[item] = 1; while (committed < total_items && done[committed]) committed++; st.n_samples++; };
    for (;;) {
        // empty slots take camera items (inside the window)
        for (int l = 0; l < 64; l++)
            for (Rec* r : {&A[l], &B[l]})
                if (r->kind == 0 && next_item < std::min(total_items, committed + cfg.window)) { r->item = next_item++; r->pos = 0; r->kind = 4; }
        st.max_window = std::max(st.max_window, next_item - committed);
        int nD = 0, nS = 0, nAny = 0;
        for (int l = 0; l < 64; l++) {
            bool d = A[l].kind == 1 || B[l].kind == 1;
            bool s2 = (A[l].kind >= 2) || (B[l].kind >= 2);
            nD += d; nS += s2; nAny += (A[l].kind || B[l].kind);
        }
        if (!nAny) break;
        const bool modeS = nS >= cfg.s_threshold || nD == 0 || (nS > nD);
        int n1 = 0, n2 = 0, n3 = 0, n4 = 0, nip = 0;
        for (int l = 0; l < 64; l++) {
            // bring the path to run into A
            auto runnable = [&](const Rec& r) { return modeS ? r.kind >= 2 : r.kind == 1; };
            if (!runnable(A[l])) { if (runnable(B[l])) std::swap(A[l], B[l]); else continue; }
            Rec& r = A[l];
            if (r.kind == 1) n1++; else if (r.kind == 2) n2++; else if (r.kind == 3) n3++; else n4++;
            const auto& ev = path(r.item);
            if (r.kind != 4 && r.pos >= max_depth) { retire(r.item); r.kind = 0; continue; }
            nip++;
            if (r.pos >= (int)ev.size()) { retire(r.item); r.kind = 0; continue; }
            int e = ev[r.pos++];
            if (e == 0) { retire(r.item); r.kind = 0; } else r.kind = e;
        }
        double cst = c.ov_iter + (n1 ? c.d : 0) + (n3 ? c.g : 0) + (n2 ? c.m : 0) + (n4 ? c.cam : 0) + (nip ? c.ip : 0);
        st.cost += cst;
        st.useful += (n1 * c.d + n3 * c.g + n2 * c.m + n4 * c.cam + nip * c.ip) / 64.0;
        st.iters += 1;
    }
    st.cost += c.commit_item * total_items;
}


// ---------------------------------------------------------------- lane-private sample streams (path regeneration)
// A wave owns 64/S pixels for the whole sample range.  Lane (pixel, j) traces the samples j, j + S, j + 2S, ... of its pixel one
// after the other and starts the next one the moment a path ends — no round barrier.  Camera rays come from a per-lane stash
// that is topped up to `stash` rays for all lanes at once (wave-uniform point, full width) whenever some lane runs dry.
// S = 1: the lane IS the pixel and adds its samples in order by itself (bit-exact, no exchange).  S > 1: results go through a
// ring of `lookahead` rounds in LDS; a round is committed (ordered fold) when all 64 lanes have finished it, and a lane
// may run at most `lookahead` rounds ahead of the last committed one.
struct StreamCfg { int S = 1; int lookahead = 1 << 30; int stash = 4; double fin = 20, ov = 6, fold = 96; };
void sim_streams(const Tile& t, int pix0, int spp, int max_depth, const Cost& c, const StreamCfg& g, Stats& st) {
    const int S = g.S, per_lane = (spp + S - 1) / S;
    int k[64], pos[64], stash[64], active[64];   // k: index of the sample in flight (or next); active: a path is in flight
    for (int l = 0; l < 64; l++) { k[l] = 0; pos[l] = 0; stash[l] = 0; active[l] = 0; }
    auto sample_of = [&](int l, int kk) { return kk * S + l % S; };
    auto path = [&](int l, int kk) -> const std::vector<uint8_t>& { return t.samples[(size_t)(pix0 + l / S) * spp + sample_of(l, kk)]; };
    auto exists = [&](int l, int kk) { return kk < per_lane && sample_of(l, kk) < spp; };
    int committed = 0;   // rounds folded so far (S > 1)
    for (;;) {
        // lanes without a path start their next sample if the window allows it; an empty stash triggers a wave-wide refill
        bool need_refill = false;
        for (int l = 0; l < 64; l++)
            if (!active[l] && exists(l, k[l]) && k[l] < committed + g.lookahead && stash[l] == 0) need_refill = true;
        if (need_refill) {
            int most = 0; double sum = 0;
            for (int l = 0; l < 64; l++) {
                int remaining = 0;
                for (int kk = k[l] + (active[l] ? 1 : 0); kk < per_lane && exists(l, kk); kk++) remaining++;
                int want = std::min(g.stash, remaining) - stash[l];
                if (want < 0) want = 0;
                stash[l] += want; most = std::max(most, want); sum += want;
            }
            st.cost += most * c.cam; st.useful += sum * c.cam / 64.0;
        }
        int nstart = 0;
        for (int l = 0; l < 64; l++)
            if (!active[l] && exists(l, k[l]) && k[l] < committed + g.lookahead && stash[l] > 0) { stash[l]--; active[l] = 1; pos[l] = 0; nstart++; }
        int na = 0, nd = 0, ng = 0, nm = 0, nfin = 0;
        for (int l = 0; l < 64; l++) {
            if (!active[l]) continue;
            na++;
            const auto& ev = path(l, k[l]);
            bool fin = false;
            if (pos[l] >= (int)ev.size()) fin = true;                       // miss: intersect ran, nothing else
            else { int e = ev[pos[l]++]; if (e == 0) fin = true; else { if (e == 1) nd++; else if (e == 2) nm++; else ng++; if (pos[l] >= max_depth) fin = true; } }
            if (fin) { active[l] = 0; k[l]++; nfin++; st.n_samples++; }
        }
        if (!na) {
            bool any_left = false;
            for (int l = 0; l < 64; l++) if (exists(l, k[l])) any_left = true;
            if (!any_left) break;
            // every lane waits for the window: cannot happen (the slowest lane is always inside it)
        }
        st.iters += na ? 1 : 0;
        st.cost += (na ? c.ip : 0) + (nd ? c.d : 0) + (ng ? c.g : 0) + (nm ? c.m : 0) + g.ov + ((nfin || nstart) ? g.fin : 0);
        st.useful += (na * c.ip + nd * c.d + ng * c.g + nm * c.m + (nfin + nstart) * g.fin / 2) / 64.0;
        if (S > 1) {
            int mn = 1 << 30;
            for (int l = 0; l < 64; l++) mn = std::min(mn, exists(l, k[l]) || active[l] ? k[l] : per_lane);
            while (committed < mn) { committed++; st.cost += g.fold; st.useful += g.fold; }
        }
    }
}


// ---------------------------------------------------------------- rounds kernel + ONE survivor merge per round
// The round-synchronous kernel as it is (4 waves of a block, each 4 pixels x 16 samples), plus: at the start of depth `md` the
// waves of a pair (w, w^1) exchange path states through LDS so that the survivors of both fill wave w first; wave w^1 keeps
// only the overflow (usually nothing) and sleeps at the round's barrier.  One exchange of a 19-dword state per round and
// migrating path instead of one per bounce (the regrouping kernel of round 2).  Results return through LDS for the ordered fold.
void sim_merge_rounds(const Tile& t, int pixels, int spp, int max_depth, const Cost& c, int md, double ov_merge, Stats& st) {
    for (int p0 = 0; p0 + 16 <= pixels; p0 += 16) {          // a block = 16 pixels = 4 waves
        for (int base = 0; base < spp; base += 16) {
            struct L { const std::vector<uint8_t>* ev; int pos; bool alive; };
            std::vector<L> wave[4];
            for (int w = 0; w < 4; w++)
                for (int l = 0; l < 64; l++) {
                    int p = p0 + 4 * w + l / 16, s2 = base + l % 16;
                    L x; x.alive = s2 < spp; x.ev = x.alive ? &t.samples[(size_t)p * spp + s2] : nullptr; x.pos = 0;
                    wave[w].push_back(x);
                    if (x.alive) st.n_samples++;
                }
            for (int w = 0; w < 4; w++) { st.cost += c.cam + c.fold_round; st.useful += c.cam + c.fold_round; }
            for (int depth = 0; depth < max_depth; depth++) {
                if (depth == md) {
                    for (int w = 0; w < 4; w += 2) {
                        std::vector<L> surv;
                        for (int ww = w; ww < w + 2; ww++) for (auto& x : wave[ww]) if (x.alive) surv.push_back(x);
                        for (int ww = w; ww < w + 2; ww++) { wave[ww].assign(64, L{nullptr, 0, false}); }
                        for (size_t i = 0; i < surv.size(); i++) wave[w + (i >= 64)][i % 64] = surv[i];
                        st.cost += 2 * ov_merge;
                    }
                }
                for (int w = 0; w < 4; w++) {
                    int na = 0, nd = 0, ng = 0, nm = 0;
                    for (auto& x : wave[w]) {
                        if (!x.alive) continue;
                        na++;
                        if (x.pos >= (int)x.ev->size()) { x.alive = false; continue; }
                        int e = (*x.ev)[x.pos++];
                        if (e == 0) x.alive = false; else if (e == 1) nd++; else if (e == 2) nm++; else ng++;
                    }
                    if (!na) continue;
                    st.iters += 0.25;
                    st.cost += c.ip + (nd ? c.d : 0) + (ng ? c.g : 0) + (nm ? c.m : 0);
                    st.useful += (na * c.ip + nd * c.d + ng * c.g + nm * c.m) / 64.0;
                }
            }
        }
    }
}

// ---------------------------------------------------------------- rounds kernel with samples SORTED by path length
// Within a window of `window` consecutive samples of each pixel the samples are traced in order of (predicted) path length,
// 16 per pixel and round, so that a round's lanes finish together; the ordered fold then needs the window's results in
// LDS.  noise = 0: the true length (an oracle no kernel has): 1.14x at a 64-sample window, 1.22x over all 500.  noise < 0:
// the predictor a kernel could afford — Russian roulette's own random numbers, "terminates at the first depth >= 6 whose
// rnd.z >= -noise" — is right for 68 % of the samples and buys NOTHING (1.00x): one mispredicted long path among the 64
// lanes keeps the whole round alive, and a guaranteed bound (rnd.z >= 0.999, the spheres' survival probability) almost
// never triggers.  Idea rejected on the simulator, before any kernel was written.
void sim_sorted_rounds(const Tile& t, int pixels, int spp, int max_depth, const Cost& c, int window, double noise, unsigned seed, Stats& st) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    for (int p0 = 0; p0 < pixels; p0 += 4) {
        for (int w0 = 0; w0 < spp; w0 += window) {
            const int wn = std::min(window, spp - w0);
            // per pixel: order of the window's samples by key
            std::vector<int> order[4];
            for (int pp = 0; pp < 4; pp++) {
                std::vector<std::pair<double, int>> keys;
                for (int k = 0; k < wn; k++) {
                    const auto& ev = t.samples[(size_t)(p0 + pp) * spp + w0 + k];
                    double len = (double)ev.size();
                    if (noise < 0.0) {   // the predictor a kernel can afford: Russian roulette's own random numbers (pathTracer.comp:393-396)
                        len = (double)max_depth;
                        for (int d = 6; d < max_depth; d++) {
                            oracle::v3 r = oracle::rand01<oracle::PlainPolicy>(t.gx[p0 + pp], t.gy[p0 + pp], (uint32_t)(w0 + k) * (uint32_t)max_depth + (uint32_t)d);
                            if (r.z >= -noise) { len = d + 1; break; }
                        }
                        keys.push_back({len, k});
                    } else
                    keys.push_back({len + noise * (U(rng) - 0.5) * 6.0, k});
                }
                std::sort(keys.begin(), keys.end());
                for (auto& kv : keys) order[pp].push_back(kv.second);
            }
            for (int base = 0; base < wn; base += 16) {
                const std::vector<uint8_t>* ev[64]; int pos[64], alive[64];
                int n = 0;
                for (int l = 0; l < 64; l++) {
                    int pp = l / 16, k = base + l % 16;
                    alive[l] = k < wn;
                    ev[l] = alive[l] ? &t.samples[(size_t)(p0 + pp) * spp + w0 + order[pp][k]] : nullptr;
                    pos[l] = 0; n += alive[l];
                }
                st.cost += c.cam + c.fold_round + 24; st.n_samples += n;   // + deposit / key bookkeeping
                for (int depth = 0; depth < max_depth; depth++) {
                    int na = 0, nd = 0, ng = 0, nm = 0;
                    for (int l = 0; l < 64; l++) {
                        if (!alive[l]) continue;
                        na++;
                        if (pos[l] >= (int)ev[l]->size()) { alive[l] = 0; continue; }
                        int e = (*ev[l])[pos[l]++];
                        if (e == 0) alive[l] = 0; else if (e == 1) nd++; else if (e == 2) nm++; else ng++;
                    }
                    if (!na) break;
                    st.iters += 1;
                    st.cost += c.ip + (nd ? c.d : 0) + (ng ? c.g : 0) + (nm ? c.m : 0);
                    st.useful += (na * c.ip + nd * c.d + ng * c.g + nm * c.m) / 64.0;
                }
            }
        }
    }
}

}  // namespace


// ---------------------------------------------------------------- sample-pool kernel (round 3)
// One wave owns P = 64/S pixels and ALL their samples as one pool in batch order (batch b = samples b*S .. b*S+S-1 of every
// pixel).  Camera rays (+ their first intersection) are generated 64 at a time at full width into a stash; a lane whose path
// ended pops the next stash entry at the top of the next iteration — of any pixel of the wave.  No exchange of path state
// between lanes, no ordered commit (radiance goes to per-pixel LDS accumulators as it arises: fast math only).
// Cost per iteration: the blocks that have a lane (as `rounds`) + ov_pop when a lane pops + ov_lane (per-lane depth tests that
// were scalar in the round-synchronous kernel); per batch: cam + first intersection.
struct PoolCfg { int S = 16; double ov_pop = 16, ov_lane = 7, batch = 150; int pop_every = 1; };
void sim_pool(const Tile& t, int pix0, int spp, int max_depth, const Cost& c, const PoolCfg& g, Stats& st) {
    const int P = 64 / g.S;
    const int batches = (spp + g.S - 1) / g.S;
    long generated = 0, popped = 0;          // entries; entry e -> batch e / 64, pixel (e % 64) / S, sample batch * S + e % S
    const long total = (long)batches * 64;
    const std::vector<uint8_t>* ev[64];
    int pos[64], alive[64];
    for (int l = 0; l < 64; l++) { alive[l] = 0; ev[l] = nullptr; pos[l] = 0; }
    long iter = 0;
    for (;;) {
        int dead = 0;
        for (int l = 0; l < 64; l++) dead += !alive[l];
        bool pop_now = dead > 0 && (iter % g.pop_every == 0 || dead == 64);
        if (pop_now) {
            if (generated - popped < dead && generated < total) { generated += 64; st.cost += g.batch; st.useful += g.batch; }
            bool any = false;
            for (int l = 0; l < 64 && popped < generated; l++) {
                if (alive[l]) continue;
                long e = popped++;
                int b = (int)(e / 64), p = (int)(e % 64) / g.S, s = b * g.S + (int)(e % g.S);
                if (s >= spp) continue;       // ragged last batch: entry dropped at generation (compacted)
                ev[l] = &t.samples[(size_t)(pix0 + p) * spp + s]; pos[l] = 0; alive[l] = 1; any = true;
                st.n_samples++;
            }
            if (any) { st.cost += g.ov_pop; }
        }
        int na = 0, nd = 0, ng = 0, nm = 0;
        for (int l = 0; l < 64; l++) {
            if (!alive[l]) continue;
            na++;
            if (pos[l] >= (int)ev[l]->size() || pos[l] >= max_depth) { alive[l] = 0; continue; }
            int e = (*ev[l])[pos[l]++];
            if (e == 0) alive[l] = 0;
            else if (e == 1) nd++;
            else if (e == 2) nm++;
            else ng++;
            if (pos[l] >= max_depth) alive[l] = 0;   // depth limit: ends after its material block
        }
        if (!na) { if (popped >= total) break; iter++; continue; }
        st.iters += 1;
        st.cost += c.ip + g.ov_lane + (nd ? c.d : 0) + (ng ? c.g : 0) + (nm ? c.m : 0);
        st.useful += (na * c.ip + nd * c.d + ng * c.g + nm * c.m) / 64.0;
        iter++;
    }
}

int main(int argc, char** argv) {
    int W = 900, H = 600, spp = 500, max_depth = 12, n_tiles = 32, pixels = 64;
    bool only_streams = false, only_pool = false;
    unsigned seed = 1;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--spp")) spp = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--tiles")) n_tiles = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--seed")) seed = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--only")) { only_streams = !strcmp(argv[i + 1], "streams"); only_pool = !strcmp(argv[i + 1], "pool"); }
        if (!strcmp(argv[i], "--cost")) { /* current kernel (r03 PMC: 3943 wave-instructions per round): ip d g m cam */
            sscanf(argv[i + 1], "%lf,%lf,%lf,%lf,%lf", &g_cost.ip, &g_cost.d, &g_cost.g, &g_cost.m, &g_cost.cam); }
    }
    oracle::PT<TracePolicy> pt;
    pt.planes = kPlanes; pt.nPlanes = 6; pt.spheres = kSpheres; pt.nSpheres = 3; pt.mathMode = oracle::MATH_LIBM;
    std::mt19937 rng(seed);
    std::vector<Tile> tiles(n_tiles);
    double bounces = 0, nsamp = 0, hist[4] = {0, 0, 0, 0};
    for (auto& t : tiles) {
        int x0 = (rng() % (W / 8)) * 8, y0 = (rng() % (H / 8)) * 8;
        t.samples.resize((size_t)pixels * spp);
        t.gx.resize(pixels); t.gy.resize(pixels);
        // 64 pixels = an 8 x 8 square made of four 4 x 4 blocks (pixels 0..15, 16..31, ...): a wave of the S = 1 stream kernel,
        // or 4 blocks of the S = 16 kernels
        auto px = [&](int p) { return x0 + (p % 16) % 4 + 4 * ((p / 16) % 2); };
        auto py = [&](int p) { return y0 + (p % 16) / 4 + 4 * (p / 32); };
        for (int p = 0; p < pixels; p++) { t.gx[p] = px(p); t.gy[p] = py(p); }
        for (int p = 0; p < pixels; p++)
            for (int s = 0; s < spp; s++) {
                auto& ev = t.samples[(size_t)p * spp + s];
                g_events = &ev;
                pt.sample(px(p), py(p), W, H, s, max_depth);
                bounces += ev.size(); nsamp++;
                for (auto e : ev) hist[e]++;
            }
    }
    printf("traced %d tiles x %d pixels x %d spp: %.2f hits/sample; events: RR-terminated %.3f diffuse %.3f mirror %.3f glass %.3f\n",
           n_tiles, pixels, spp, bounces / nsamp, hist[0] / bounces, hist[1] / bounces, hist[2] / bounces, hist[3] / bounces);
    Cost c = g_cost;
    Stats base;
    for (auto& t : tiles) sim_rounds(t, pixels, spp, max_depth, c, base);
    printf("%-58s cost/sample %7.1f  lanes %.3f  iters/64samples %.2f\n", "rounds (round 1 kernel)", base.cost / base.n_samples,
           base.useful / base.cost, base.iters * 64.0 / base.n_samples);
    for (int S : {16, 4, 1})
        for (double ovp : {16.0, 30.0, 45.0})
            for (int every : {1, 2}) {
                PoolCfg g; g.S = S; g.ov_pop = ovp; g.pop_every = every;
                Stats s;
                for (auto& t : tiles) for (int p0 = 0; p0 + 64 / S <= pixels; p0 += 64 / S) sim_pool(t, p0, spp, max_depth, c, g, s);
                printf("pool S=%2d pop overhead %2.0f every %d iteration(s) : cost/sample %7.1f (%.3fx)  lanes %.3f  iters/64samples %.2f\n", S, ovp, every,
                       s.cost / s.n_samples, (base.cost / base.n_samples) / (s.cost / s.n_samples), s.useful / s.cost, s.iters * 64.0 / s.n_samples);
            }
    if (only_pool) return 0;
    for (int S : {1, 4, 16})
        for (int la : {1, 2, 3, 4, 6, 8, 1 << 30})
            for (int stash : {2, 4}) {
                if (S == 1 && la != (1 << 30)) continue;
                StreamCfg g; g.S = S; g.lookahead = la; g.stash = stash; g.fold = S == 1 ? 0 : 96.0 * S / 16;
                Stats s;
                for (auto& t : tiles) for (int p0 = 0; p0 + 64 / S <= pixels; p0 += 64 / S) sim_streams(t, p0, spp, max_depth, c, g, s);
                printf("streams S=%2d lookahead=%10d stash=%d : cost/sample %7.1f (%.3fx)  lanes %.3f  iters/64samples %.2f\n", S, la, stash,
                       s.cost / s.n_samples, (base.cost / base.n_samples) / (s.cost / s.n_samples), s.useful / s.cost, s.iters * 64.0 / s.n_samples);
            }
    for (int md : {7, 8, 9, 10, 11})
        for (double ov : {0.0, 60.0, 100.0}) {
            Stats s;
            for (auto& t : tiles) sim_merge_rounds(t, pixels, spp, max_depth, c, md, ov, s);
            printf("rounds + pair merge at depth %2d, overhead %3.0f/wave : cost/sample %7.1f (%.3fx)  lanes %.3f  iters/64samples %.2f\n", md, ov,
                   s.cost / s.n_samples, (base.cost / base.n_samples) / (s.cost / s.n_samples), s.useful / s.cost, s.iters * 64.0 / s.n_samples);
        }
    if (only_streams) return 0;
    struct Named { const char* name; RegroupCfg cfg; };
    std::vector<Named> cfgs;
    for (int waves : {1, 4, 8})
        for (int win : {64 * waves, 128 * waves, 256 * waves})
            for (int dqc : {64, 128})
                for (int sqc : {64, 128}) {
                    RegroupCfg g; g.waves = waves; g.window = win; g.s_batch_min = 48; g.dq_cap = dqc; g.sq_cap = sqc;
                    cfgs.push_back({"", g});
                }
    for (auto& n : cfgs) {
        Stats s;
        const int ppb = 4 * n.cfg.waves;   // pixels per workgroup: 4 per wave (as the rounds kernel at S = 16)
        for (auto& t : tiles) for (int p0 = 0; p0 < pixels; p0 += ppb) sim_regroup(t, p0, ppb, spp, max_depth, c, n.cfg, s);
        printf("regroup waves=%d window=%4d batch_min=%2d dq_cap=%3d sq_cap=%3d : cost/sample %7.1f (%.3fx)  lanes %.3f  maxDq %3d maxSq %3d maxWin %4d idle %.0f\n",
               n.cfg.waves, n.cfg.window, n.cfg.s_batch_min, n.cfg.dq_cap, n.cfg.sq_cap, s.cost / s.n_samples,
               (base.cost / base.n_samples) / (s.cost / s.n_samples), s.useful / s.cost, s.max_dq, s.max_sq, s.max_window, s.stall_lane_iters / 64);
    }
    for (int win : {32, 64, 128, 256, 500})
        for (double noise : {0.0, -0.70, -0.75, -0.80, -0.85}) {
            Stats s;
            for (auto& t : tiles) sim_sorted_rounds(t, pixels, spp, max_depth, c, win, noise, 7, s);
            printf("sorted rounds window=%3d noise=%.1f : cost/sample %7.1f (%.3fx)  lanes %.3f  iters/64samples %.2f\n", win, noise,
                   s.cost / s.n_samples, (base.cost / base.n_samples) / (s.cost / s.n_samples), s.useful / s.cost, s.iters * 64.0 / s.n_samples);
        }
    for (int ov : {10, 20})
        for (int win : {64, 128, 256, 512})
            for (int thr : {24, 32, 40, 48}) {
                Cost c2 = c; c2.ov_iter = ov; c2.commit_item = 1.0;
                TwoSlotCfg g; g.window = win; g.s_threshold = thr;
                Stats s;
                for (auto& t : tiles) for (int p0 = 0; p0 < pixels; p0 += g.pixels) sim_twoslot(t, p0, spp, max_depth, c2, g, s);
                printf("twoslot ov=%2d window=%4d s_thr=%2d : cost/sample %7.1f (%.3fx)  lanes %.3f  iters/64 %.2f maxWin %4d\n", ov, win, thr,
                       s.cost / s.n_samples, (base.cost / base.n_samples) / (s.cost / s.n_samples), s.useful / s.cost, s.iters * 64.0 / s.n_samples, s.max_window);
            }
    return 0;
}
