// Global data association: exact min-cost flow on the unit-capacity tracking network (host C++).
//
// Replaces libmot.data_association.MinCostFlowTracker.compute_trajectories() as driven by the reference at
// axtrack/AxonDetections.py:663-690 (libmot itself is absent from the reference tree; formulation restated from
// Zhang/Li/Nevatia CVPR'08 -- DESIGN.md "Unpinned third-party semantics").
//
// Network: source S, sink T, per detection k an observation arc u_k -> v_k (cost obs[k]), an entry arc S -> u_k,
// an exit arc v_k -> T, and transition arcs v_a -> u_b (CSR by a). All capacities are 1, all costs integers.
// The number of unit flows F in [min_flow, max_flow] with minimum total cost is wanted. The cost is convex
// in F, so successive shortest augmenting paths stop exactly at that optimum:
//   push while F < min_flow, or while F < max_flow and the next path has negative cost.
//
// Two exact solvers share this file:
//   * solve_lsap (fast path). Without the bound on F the problem is a sparse rectangular assignment: row k = the
//     "out slot" of detection k, matched to the in-slot R_b of a successor b (cost obs_k + trans_kb: b's entry
//     cost is refunded), to its own in-slot R_k (cost 0: k unused) or to a private exit column X_k
//     (cost obs_k + entry_k + exit_k). Rows are inserted in a shuffled order by shortest augmenting paths on reduced
//     costs (Jonker-Volgenant / Crouse); a search ends at the first free column and X_k is always free, so
//     searches stay local instead of sweeping the whole network. Its optimum is the flow optimum over all F.
//     Blocks of frames are solved concurrently on host threads and joined through the few rows between them
//     (Lsap::run); the optimum is unique, so the thread count does not change the result.
//   * solve_ssp (general path): successive shortest s-t paths with Johnson potentials (first potentials from one
//     DP sweep over the frame-ordered DAG). Used when the unconstrained optimum has F outside [min_flow, max_flow].
// Arc costs carry a 16-bit identity hash (axt_arc_cost_int) that makes the optimum unique, so both solvers (and
// any other exact solver) return the same trajectories.
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <queue>
#include <thread>
#include <vector>

#include "../../include/axtrack_hip.h"

void axt_set_error(const char *fmt, ...);

namespace {

constexpr int64_t INF = INT64_MAX / 4;
constexpr int32_t NONE = -2, TERMINAL = -1;   // pred: TERMINAL = fed by S; succ: TERMINAL = drains to T

struct Solver {
    int n;
    const int64_t *obs, *entry, *exitc, *row_ptr, *cost;
    const int32_t *col;
    std::vector<int32_t> pred, succ;      // arc index of the flow-carrying in/out transition, TERMINAL or NONE
    std::vector<int32_t> pred_tail;       // tail detection of pred[k] when pred[k] >= 0
    std::vector<int64_t> pi, dist;
    std::vector<int32_t> par;             // parent node on the shortest-path tree
    std::vector<int32_t> par_arc;         // transition arc index used (or -1)
    std::vector<int32_t> touched;
    int S, T;

    inline int U(int k) const { return 2 * k; }
    inline int V(int k) const { return 2 * k + 1; }

    void init_potentials()
    {
        pi.assign(2 * n + 2, INF);
        pi[S] = 0;
        for (int k = 0; k < n; ++k) pi[U(k)] = entry[k];
        int64_t dT = INF;
        for (int k = 0; k < n; ++k) {
            const int64_t dv = pi[U(k)] + obs[k];
            pi[V(k)] = dv;
            for (int64_t e = row_ptr[k]; e < row_ptr[k + 1]; ++e) {
                const int b = col[e];
                if (dv + cost[e] < pi[U(b)]) pi[U(b)] = dv + cost[e];
            }
            if (dv + exitc[k] < dT) dT = dv + exitc[k];
        }
        pi[T] = dT;
    }

    // returns true if T was reached; dist/par filled for settled nodes
    bool dijkstra()
    {
        typedef std::pair<int64_t, int32_t> Item;
        std::priority_queue<Item, std::vector<Item>, std::greater<Item>> pq;
        for (int32_t x : touched) dist[x] = INF;
        touched.clear();
        auto relax = [&](int from, int to, int64_t c, int64_t d, int32_t arc) {
            const int64_t nd = d + c + pi[from] - pi[to];
            if (nd < dist[to]) {
                if (dist[to] == INF) touched.push_back(to);
                dist[to] = nd;
                par[to] = from;
                par_arc[to] = arc;
                pq.push(Item(nd, to));
            }
        };
        dist[S] = 0;
        touched.push_back(S);
        for (int k = 0; k < n; ++k)
            if (pred[k] != TERMINAL) relax(S, U(k), entry[k], 0, -1);
        while (!pq.empty()) {
            const Item it = pq.top();
            pq.pop();
            const int x = it.second;
            const int64_t d = it.first;
            if (d > dist[x]) continue;
            if (x == T) return true;
            const int k = x >> 1;
            if ((x & 1) == 0) {   // u_k
                if (pred[k] == NONE) relax(x, V(k), obs[k], d, -1);
                else if (pred[k] >= 0) {
                    const int32_t e = pred[k];
                    relax(x, V(pred_tail[k]), -cost[e], d, e);
                }
            } else {              // v_k
                if (succ[k] != TERMINAL) relax(x, T, exitc[k], d, -1);
                const int32_t busy = succ[k];
                for (int64_t e = row_ptr[k]; e < row_ptr[k + 1]; ++e)
                    if ((int32_t)e != busy) relax(x, U(col[e]), cost[e], d, (int32_t)e);
                if (pred[k] != NONE) relax(x, U(k), -obs[k], d, -1);
            }
        }
        return false;
    }

    void augment()
    {
        // collect the path T <- ... <- S, then apply removals before additions
        std::vector<std::pair<int32_t, int32_t>> steps;   // (from, to)
        std::vector<int32_t> arcs;
        for (int x = T; x != S; x = par[x]) {
            steps.push_back(std::make_pair(par[x], x));
            arcs.push_back(par_arc[x]);
        }
        for (size_t i = 0; i < steps.size(); ++i) {       // removals: backward transition arcs u_b -> v_a
            const int from = steps[i].first, to = steps[i].second;
            if (from != S && to != T && (from & 1) == 0 && (to & 1) == 1 && (from >> 1) != (to >> 1)) {
                succ[to >> 1] = NONE;
                pred[from >> 1] = NONE;
            }
        }
        for (size_t i = 0; i < steps.size(); ++i) {       // additions
            const int from = steps[i].first, to = steps[i].second;
            if (from == S) pred[to >> 1] = TERMINAL;
            else if (to == T) succ[from >> 1] = TERMINAL;
            else if ((from & 1) == 1 && (to & 1) == 0 && (from >> 1) != (to >> 1)) {
                succ[from >> 1] = arcs[i];
                pred[to >> 1] = arcs[i];
                pred_tail[to >> 1] = from >> 1;
            }
        }
    }
};


// ---------------------------------------------------------------------------------------------------------
// Sparse rectangular assignment by shortest augmenting paths. Columns: [0,n) = R_b, [n,2n) = X_k.
// Rows are inserted in a fixed pseudo-random order: in frame order every insertion grabs the best free successor
// and later frames keep displacing those choices, which costs re-routing searches that sweep thousands of rows
// (378 k rows scanned on the C3 network against 230 k in shuffled order; the optimum is unique, the order only
// changes the work). Per-column search state sits in one 32-byte record and is invalidated by a per-search stamp
// instead of being reset.
// ---------------------------------------------------------------------------------------------------------
static double now_ms()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

struct Lsap {
    int n;
    const int64_t *obs, *entry, *exitc, *row_ptr, *cost;
    const int32_t *col;
    struct Col {
        int64_t v;            // dual (<= 0; 0 while unmatched)
        int64_t spc;          // shortest path cost in the current search (valid if stamp matches)
        int32_t row;          // matched row or -1
        int32_t pred_row;
        int32_t pred_arc;
        uint32_t stamp;       // 2 * search id (+1 once the column is in the shortest-path tree)
    };
    struct Row {
        int64_t u;            // dual
        int64_t base;         // obs + entry: cost of using the detection, before the transition
        int64_t own;          // obs + entry + exit: a track of its own / the end of a track
        int64_t arc_begin;
        int32_t degree;
        int32_t col;          // matched column or -1
        int32_t arc;          // transition arc used (original index) or -1
        int32_t pad;
    };
    struct Arc {              // the arcs of every row sorted by w
        int64_t w;            // cost[e] - entry[head]: the successor's entry cost is refunded
        int32_t head;
        int32_t id;           // original arc index
    };
    std::vector<Col> c;
    std::vector<Row> rw;
    std::vector<Arc> arcs;
    // Two phases (a continuation in the entry / exit cost). Ending a track costs entry + exit = 4 cost units against
    // 0.002-0.7 for a link: a row that has to end a track finds its nearest free column, the private exit, only after
    // a search ball of reduced radius ~4 -- on dense timelapses the whole neighbourhood -- and every track end near the
    // END of the timelapse pays that (the right spine of the time-block tree: 30 of 32 ms on config 3). The first phase
    // therefore solves the problem with the exits at a fifth of their price: balls stay small, the blocks of the tree
    // join in a few ms. For a fixed number of tracks the optimum does not depend on that price at all (it adds a
    // constant), so the second phase -- full price -- only has to reconsider the track ENDS: taken out and inserted
    // afresh, serially (config 3: 81 ends, of which 63 stay and 18 tracks dissolve; 152 k -> 101 k rows scanned).
    // Measured on the GPU box's host (16 threads), one phase -> two: config 3 (19 k detections) 33.5 -> 22.2 ms, one GPU's share
    // of config 4 (40 k) 265 -> 157 ms, of config 5 (20 k, masked) 70 -> 50 ms; a 309 k-detection timelapse 1.28 -> 1.65 s
    // (its 441 track ends, re-inserted serially, cost more than the tree saves): two phases up to 120 k detections.
    bool two_phase = false;
    inline int64_t discount(int k) const
    {
        if (!two_phase) return 0;
        const int64_t whole = ((entry[k] + exitc[k]) >> 16) / 5 * 4;      // four fifths of entry + exit, in whole cost units
        return whole > 0 ? whole << 16 : 0;
    }
    size_t stat_rows = 0, stat_relax = 0, stat_push = 0, stat_died = 0;
    std::mutex stat_lock;
    // Concurrent insertions on SHARED rows and columns (the second phase; Search::insert_row with a ticket): a search
    // claims every column record before it reads it (a row is only ever reached through its matched column, so the
    // column's claim covers the row), keeps its claims until its duals and its augmentation are written, and lets go.
    // Two searches that never meet run side by side; when they meet, the older one (smaller ticket) waits for the
    // record and the younger one gives up everything it holds and starts again ("wait-die": nobody waits for an older
    // search, so there is no cycle, and the oldest search always finishes). Every search therefore sees the state some
    // serial order of the insertions would have shown it; row insertion reaches the one optimum in ANY order, so the
    // result does not depend on the interleaving.
    std::unique_ptr<std::atomic<int32_t>[]> owner;        // per column: 0 = free, else the ticket of the holder
    std::atomic<uint32_t> next_stamp{0};

    // ---- one search on several threads (Search::finish_in_parallel) ---------------------------------------------------------
    // A handful of searches settle most of the network (config 4: one row of the root separator scans 244 k of the 319 k rows,
    // 97 searches hold 84 % of the rows scanned on the critical path), and they sit on the spine of the time-block tree, where
    // one thread works and the others have nothing left to do. A search that has grown past kParSwitch rows therefore goes on
    // as a label-correcting search in buckets of distance ("near-far") on a team of threads: the labels the serial phase has
    // settled stay, the open ones seed the first buckets, columns take (distance, predecessor row) in ONE 64-bit word by
    // atomic minimum, and the nearest free column is an atomic minimum too. It ends with exactly the distances Dijkstra
    // would have found for every column nearer than that free column -- which is all the dual update needs -- so the
    // result is the serial one, at any team size.
    static constexpr int kDistShift = 24;                 // label = distance << 24 | predecessor row (or column, for the bound)
    static constexpr uint64_t kNoLabel = ~0ull;
    std::unique_ptr<std::atomic<uint64_t>[]> plabel;      // per column, kNoLabel outside a parallel search
    std::unique_ptr<int64_t[]> pexpanded;                 // per column: the distance its row was last expanded with (INF = never)
    std::once_flag par_alloc;
    int par_switch = 8192, par_buckets = 48;
    struct HelperPool {
        std::mutex m;
        std::condition_variable cv;
        std::vector<std::thread> threads;
        struct Job { void *search; int tid; int epoch; };
        std::deque<Job> jobs;
        int idle = 0;
        bool quit = false;
    } pool;
    std::atomic<int> tasks_running{0};                   // tree tasks (leaves, separators, the second phase) at work right now
    void start_helpers()
    {
        if (!pool.threads.empty() || budget <= 1) return;
        if (getenv("AXT_MCF_NO_PAR_SEARCH")) return;
        if (const char *e = getenv("AXT_MCF_PAR_SWITCH")) par_switch = atoi(e) > 8 ? atoi(e) : 8;
        if (const char *e = getenv("AXT_MCF_PAR_BUCKETS")) par_buckets = atoi(e) > 1 ? atoi(e) : 1;
        if ((int64_t)n >= (1 << 23)) return;                 // predecessor rows and columns must fit the label's low bits
        // Measured on the GPU box's host (16 threads, profiles/r04d_par_search.log): config 4 (319 k detections) 1 191 -> 725 ms of
        // insertions, the root separator 484 -> 202 ms; config 3 (19 k) 18.7 -> 21-26 ms: its largest search is 14 k rows and a
        // team costs more than it saves there. So: only networks of config 4's scale.
        int min_n = 100000;
        if (const char *e = getenv("AXT_MCF_PAR_MIN_N")) min_n = atoi(e);
        if (n < min_n) return;
        for (int t = 0; t < budget - 1; ++t) {
            try { pool.threads.emplace_back([this] { helper_main(); }); } catch (...) { break; }
        }
    }
    void stop_helpers()
    {
        {
            std::lock_guard<std::mutex> g(pool.m);
            pool.quit = true;
        }
        pool.cv.notify_all();
        for (std::thread &t : pool.threads) t.join();
        pool.threads.clear();
        pool.quit = false;
    }
    void helper_main();
    // up to `want` idle helpers for one search; they run Search::team_member(tid), tid = 1 .. returned count
    int recruit(void *search, int want, int epoch)
    {
        if (pool.threads.empty() || want <= 0) return 0;
        std::lock_guard<std::mutex> g(pool.m);
        // threads that are busy with tree tasks count against the budget: a search only takes what would otherwise idle
        const int spare = budget - tasks_running.load(std::memory_order_relaxed) - ((int)pool.threads.size() - pool.idle);
        int k = std::min(std::min(want, pool.idle), spare);
        if (k <= 0) return 0;
        for (int t = 1; t <= k; ++t) pool.jobs.push_back(HelperPool::Job{search, t, epoch});
        pool.idle -= k;
        pool.cv.notify_all();
        return k;
    }
    ~Lsap() { stop_helpers(); }

    typedef std::pair<int64_t, int32_t> Item;
    // One thread's search state. Several of them work on the shared rows / columns at the same time, on index ranges
    // that cannot meet (see run()).
    struct Search {
    Lsap &L;
    std::vector<int32_t> sr_rows, sc_cols;
    uint32_t search;                                  // stamps above every stamp already left in this task's columns
    size_t stat_rows = 0, stat_relax = 0, stat_push = 0, stat_died = 0;
    int32_t ticket = 0;                               // > 0: concurrent mode (claims), smaller = older
    std::vector<int32_t> claimed;
    int32_t blocked_col = -1, blocked_by = 0;         // the claim that made this search give up
    Search(Lsap &l, uint32_t first) : L(l), search(first) {}
    ~Search()
    {
        std::lock_guard<std::mutex> g(L.stat_lock);
        L.stat_rows += stat_rows; L.stat_relax += stat_relax; L.stat_push += stat_push; L.stat_died += stat_died;
    }
    // Claim column j for this search. false: an older search holds it -- this one has to give up.
    inline bool claim(int j)
    {
        std::atomic<int32_t> &o = L.owner[j];
        int32_t cur = o.load(std::memory_order_relaxed);
        if (cur == ticket) return true;
        for (;;) {
            if (cur == 0) {
                if (o.compare_exchange_weak(cur, ticket, std::memory_order_acq_rel, std::memory_order_relaxed)) {
                    claimed.push_back(j);
                    return true;
                }
                continue;
            }
            if (cur < ticket) { blocked_col = j; blocked_by = cur; return false; }
            __builtin_ia32_pause();                    // a younger search holds it: it will finish or give up
            cur = o.load(std::memory_order_acquire);
        }
    }
    inline bool try_claim(int j)                       // no waiting, no giving up (after the augmentation)
    {
        std::atomic<int32_t> &o = L.owner[j];
        int32_t cur = o.load(std::memory_order_relaxed);
        if (cur == ticket) return true;
        if (cur != 0 || !o.compare_exchange_strong(cur, ticket, std::memory_order_acq_rel, std::memory_order_relaxed)) return false;
        claimed.push_back(j);
        return true;
    }
    void release()
    {
        for (int32_t j : claimed) L.owner[j].store(0, std::memory_order_release);
        claimed.clear();
    }
    // Monotone radix heap with lazy deletion (stale entries are skipped when popped). A Dijkstra search never pushes a key
    // below the last one it popped (reduced costs are >= 0), so an entry can be filed by the highest bit in which its key
    // differs from that last key: a push is one append, a pop takes from the bucket of equal keys and, when that is empty,
    // spreads the lowest occupied bucket over the buckets below it around its smallest key. Keys pushed before the first pop
    // (the fresh row's own options, whose distances may lie below zero) wait in a staging bucket.
    struct RadixHeap {
        static constexpr int kStage = 65;
        std::vector<Item> b[66];
        uint64_t occupied = 0;                 // bit k: b[k] is not empty (k = 0..64 -> bits 0..63 for k <= 63; b[64] tracked apart)
        bool top_bucket = false;               // b[64] not empty
        uint64_t last = 0;
        bool started = false;
        size_t count = 0;
        static inline uint64_t ukey(int64_t k) { return (uint64_t)k ^ ((uint64_t)1 << 63); }
        inline bool empty() const { return count == 0; }
        void clear()
        {
            if (count || !b[kStage].empty()) {
                uint64_t m = occupied;
                while (m) { const int k = __builtin_ctzll(m); m &= m - 1; b[k].clear(); }
                b[64].clear();
                b[kStage].clear();
            }
            occupied = 0; top_bucket = false; started = false; count = 0;
        }
        inline void file(const Item &it)
        {
            const uint64_t x = ukey(it.first) ^ last;
            const int k = x ? 64 - __builtin_clzll(x) : 0;
            b[k].push_back(it);
            if (k < 64) occupied |= (uint64_t)1 << k; else top_bucket = true;
        }
        inline void push(const Item &it)
        {
            ++count;
            if (!started) { b[kStage].push_back(it); return; }
            if (ukey(it.first) < last) { b[0].push_back(it); occupied |= 1; return; }      // (cannot happen with feasible duals)
            file(it);
        }
        Item pop()
        {
            if (!started) {
                started = true;
                uint64_t mn = ~(uint64_t)0;
                for (const Item &it : b[kStage]) if (ukey(it.first) < mn) mn = ukey(it.first);
                last = mn;
                for (const Item &it : b[kStage]) file(it);
                b[kStage].clear();
            }
            if (b[0].empty()) {
                occupied &= ~(uint64_t)1;
                const int k = occupied ? __builtin_ctzll(occupied) : 64;
                std::vector<Item> &src = b[k];
                uint64_t mn = ~(uint64_t)0;
                for (const Item &it : src) if (ukey(it.first) < mn) mn = ukey(it.first);
                last = mn;
                if (k < 64) occupied &= ~((uint64_t)1 << k); else top_bucket = false;
                spill.swap(src);
                for (const Item &it : spill) file(it);
                spill.clear();
            }
            const Item top = b[0].back();
            b[0].pop_back();
            if (b[0].empty()) occupied &= ~(uint64_t)1;
            --count;
            return top;
        }
        std::vector<Item> spill;
        template <typename F> void for_each(F f) const
        {
            for (int k = 0; k < 66; ++k) for (const Item &it : b[k]) f(it);
        }
    } heap;
    inline void heap_push(const Item &it) { heap.push(it); }
    inline Item heap_pop() { return heap.pop(); }

    // false (concurrent mode only): the search met an older one and gave up; nothing was changed, insert the row again
    bool insert_row(int i) { return ticket > 0 ? insert_row_t<true>(i) : insert_row_t<false>(i); }

    template <bool PAR>                       // (two instantiations: the serial one carries none of the claim logic)
    bool insert_row_t(int i)
    {
        std::vector<Col> &c = L.c;
        std::vector<Row> &rw = L.rw;
        const std::vector<Arc> &arcs = L.arcs;
        const int n = L.n;
        heap.clear();
        constexpr bool par = PAR;
        bool dead = false;
        if (par) search = L.next_stamp.fetch_add(2, std::memory_order_relaxed);
        else search += 2;
        const uint32_t open = search, closed = search + 1;
        int64_t minVal = 0;
        int64_t best_free = INF;          // shortest distance to a FREE column seen so far: nothing at or beyond it
        int cur = i, sink = -1;           // can be part of the shortest augmenting path, so it is not even queued
        sr_rows.clear();
        sc_cols.clear();
        in_team = false;
        par_next_try = (size_t)L.par_switch;
        while (sink < 0) {
            if (!par && sr_rows.size() >= par_next_try && !sc_cols.empty() && best_free - c[sc_cols[0]].spc < ((int64_t)1 << 39)) {
                // this search has grown large: the rest of it on whatever threads are idle (Lsap: "one search on several threads")
                const int helpers = L.recruit(this, L.budget - 1, team.epoch.load(std::memory_order_relaxed) + 1);
                if (helpers > 0) {
                    sink = finish_in_parallel(i, helpers, minVal, best_free, open);
                    in_team = true;
                    break;
                }
                par_next_try = sr_rows.size() * 2;
            }
            sr_rows.push_back(cur);
            const Row &rc = rw[cur];
            const int64_t off = minVal - rc.u;
            auto relax = [&](int j, int64_t w, int32_t arc) {
                if (par && !claim(j)) { dead = true; return; }
                Col &cj = c[j];
                ++stat_relax;
                if (cj.stamp == closed) return;
                const int64_t r = off + w - cj.v;
                if (r >= best_free) return;
                if (cj.row < 0) best_free = r;
                if (cj.stamp != open || r < cj.spc) {
                    cj.stamp = open;
                    cj.spc = r;
                    cj.pred_row = cur;
                    cj.pred_arc = arc;
                    ++stat_push;
                    heap_push(Item(r, j));
                }
            };
            relax(cur, 0, -1);                  // stay unused
            if (!dead) relax(n + cur, rc.own, -1);         // a track of its own / track end
            if (dead) { release(); ++stat_died; return false; }
            // arcs sorted by cost: column duals are <= 0, so once the bare cost reaches best_free the rest cannot matter
            const Arc *ap = arcs.data() + rc.arc_begin;
            // the column records are visited in arc order, i.e. at random: ask for them a few arcs ahead
            for (int k = 0; k < rc.degree && k < 6; ++k) __builtin_prefetch(&c[ap[k].head]);
            for (int k = 0; k < rc.degree; ++k) {
                const int64_t w = rc.base + ap[k].w;
                if (off + w >= best_free) break;
                if (k + 6 < rc.degree) __builtin_prefetch(&c[ap[k + 6].head]);
                relax(ap[k].head, w, ap[k].id);
                if (dead) break;
            }
            if (dead) { release(); ++stat_died; return false; }
            int j = -1;
            while (!heap.empty()) {
                const Item it = heap_pop();
                const Col &cj = c[it.second];
                if (cj.stamp != open || it.first > cj.spc) continue;
                j = it.second;
                minVal = it.first;
                break;
            }
            c[j].stamp = closed;             // X_i is always reachable, so a column is always found
            sc_cols.push_back(j);
            if (c[j].row < 0) sink = j;
            else cur = c[j].row;
        }
        // dual update (Crouse 2016, Alg. 1); unmatched columns keep v = 0, as the rectangular dual requires
        rw[i].u += minVal;
        for (size_t k = 1; k < sr_rows.size(); ++k) {
            Row &r = rw[sr_rows[k]];
            r.u += minVal - c[r.col].spc;
        }
        for (int32_t j : sc_cols) c[j].v -= minVal - c[j].spc;
        // augment
        int j = sink;
        for (;;) {
            const int r = c[j].pred_row;
            c[j].row = r;
            const int prev = rw[r].col;
            rw[r].col = j;
            rw[r].arc = c[j].pred_arc;
            if (r == i) break;
            j = prev;
        }
        stat_rows += sr_rows.size();
        // Reduction transfer (the dual tightening of Jonker-Volgenant's initialisation, applied after every search): a row
        // raises its dual to its SECOND-best reduced cost and hands the difference to its matched column, whose dual
        // drops -- still feasible (every other edge of the row keeps a non-negative reduced cost, every other row sees
        // the column get dearer), still tight on the matched edge. The shortest-path update above leaves duals as
        // shallow as feasibility allows, so columns keep looking cheap to rows that will never get them and later
        // searches wander through them; tightened duals stop those searches at the first look. The inserted row always;
        // every scanned row after a search that went further than a few rows (measured on the config-3 network, 16
        // threads: 43 -> 34 ms; 224 k -> 150 k rows scanned).
        auto transfer = [&](int r) {
            Row &rr = rw[r];
            const int m = rr.col;
            int64_t best2 = INF, cm = 0;
            bool all = true;
            auto see = [&](int j, int64_t w) {
                if (par && all && !try_claim(j)) all = false;       // somebody else is at that column: leave this row's duals as they are
                if (!all) return;
                if (j == m) cm = w;
                else { const int64_t k = w - c[j].v; if (k < best2) best2 = k; }
            };
            see(r, 0);
            see(n + r, rr.own);
            const Arc *ap = arcs.data() + rr.arc_begin;
            for (int k = 0; k < rr.degree; ++k) see(ap[k].head, rr.base + ap[k].w);
            if (all && best2 < INF && best2 > rr.u) { rr.u = best2; c[m].v = cm - best2; }
        };
        if (in_team) { team_resume(kTransfer); in_team = false; return true; }     // the transfer of all scanned rows, on the team; then the helpers go
        if (sr_rows.size() > 8) for (int32_t r : sr_rows) transfer(r);
        else transfer(i);
        if (par) release();
        return true;
    }

    // ---- this search on a team of threads (Lsap: "one search on several threads"); this thread is member 0 ----------------
    enum { kExpand = 0, kScanFar = 1, kRefill = 2, kWriteBack = 3, kOwnerStep = 4, kTransfer = 5, kLeave = 6 };
    struct Team {
        int size = 1;
        std::atomic<int> arrived{0}, phase{0};
        std::atomic<int> epoch{0}, gone{0};      // epoch: the team of search number `epoch` is set up; gone: helpers that have left it
        int start_phase = 0;
        std::atomic<size_t> cursor{0};
        std::atomic<uint64_t> bound{0};          // distance << 24 | column: the nearest free column so far
        int mode = kExpand, pending = kLeave;
        int64_t theta = 0, delta = 1, min_val = 0;
        int64_t bias = 0;                        // labels hold distance - bias: a fresh row's distances start below zero (its dual starts at 0)
        // a bucket is drained without a barrier per hop: every member works off its own `near` list (what it pushes stays with
        // it), a member that runs dry says so and takes from `pool`, into which busy members put half of a long list when
        // somebody is hungry; the bucket is empty when all members are hungry and the pool is empty
        std::mutex pool_m;
        std::vector<int32_t> pool;
        std::atomic<size_t> pool_size{0};        // (what a dry member peeks at without the lock)
        std::atomic<int> hungry{0};
        std::atomic<bool> drained{false};
        struct alignas(64) Slot {
            std::vector<int32_t> near, far, touched, scanned;
            size_t relax = 0, rows = 0, push = 0;
            int64_t far_min = INF;
        };
        std::vector<Slot> slot;
    } team;
    inline uint64_t pack(int64_t d, int x) const { return ((uint64_t)(d - team.bias) << Lsap::kDistShift) | (uint32_t)x; }
    inline int64_t dist_of(uint64_t lab) const { return (int64_t)(lab >> Lsap::kDistShift) + team.bias; }
    void team_barrier(int &my_phase)
    {
        const int target = ++my_phase;
        if (team.arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == team.size) {
            team.arrived.store(0, std::memory_order_relaxed);
            team.phase.store(target, std::memory_order_release);
        } else {
            int spins = 0;
            while (team.phase.load(std::memory_order_acquire) - target < 0)
                if (++spins > 4096) { std::this_thread::yield(); spins = 0; } else __builtin_ia32_pause();
        }
    }
    static inline int64_t v_of(const Col &cj) { return __atomic_load_n(&cj.v, __ATOMIC_RELAXED); }
    // the row of column j goes out to its options with the distance j holds now
    void expand(int j, Team::Slot &me)
    {
        std::vector<Col> &c = L.c;
        const std::vector<Row> &rw = L.rw;
        const int n = L.n;
        const int64_t d = dist_of(L.plabel[j].load(std::memory_order_relaxed));
        int64_t bd = dist_of(team.bound.load(std::memory_order_relaxed));
        if (d >= bd || d >= __atomic_load_n(&L.pexpanded[j], __ATOMIC_RELAXED)) return;
        __atomic_store_n(&L.pexpanded[j], d, __ATOMIC_RELAXED);      // (two members may both get here for one column: the work is done twice, nothing else)
        const int r = c[j].row;
        const Row &rr = rw[r];
        ++me.rows;
        const int64_t off = d - rr.u;
        auto relax = [&](int k, int64_t w) {
            ++me.relax;
            const int64_t rc = off + w - c[k].v;
            if (rc >= bd) return;
            const uint64_t lab = pack(rc, r);
            std::atomic<uint64_t> &slot = L.plabel[k];
            uint64_t old = slot.load(std::memory_order_relaxed);
            for (;;) {                           // only a strictly smaller DISTANCE replaces a label: predecessor chains cannot close through arcs of reduced cost 0
                if (old != Lsap::kNoLabel && rc >= dist_of(old)) return;
                if (slot.compare_exchange_weak(old, lab, std::memory_order_relaxed)) break;
            }
            if (old == Lsap::kNoLabel) me.touched.push_back(k);
            if (c[k].row < 0) {                  // a free column: the search ends there unless a nearer one turns up
                const uint64_t b = pack(rc, k);
                uint64_t ob = team.bound.load(std::memory_order_relaxed);
                while (b < ob && !team.bound.compare_exchange_weak(ob, b, std::memory_order_relaxed)) {}
                bd = std::min(bd, rc);
            } else {
                ++me.push;
                (rc < team.theta ? me.near : me.far).push_back(k);
            }
        };
        relax(r, 0);
        relax(n + r, rr.own);
        const Arc *ap = L.arcs.data() + rr.arc_begin;
        for (int k = 0; k < rr.degree; ++k) {
            const int64_t w = rr.base + ap[k].w;
            if (off + w >= bd) break;            // column duals are <= 0: the bare cost bounds the reduced one
            relax(ap[k].head, w);
        }
    }
    int32_t arc_to(int r, int j) const           // the original index of the arc r -> in-slot j (-1: its own in-slot or its exit)
    {
        const Row &rr = L.rw[r];
        if (j == r || j >= L.n) return -1;
        const Arc *ap = L.arcs.data() + rr.arc_begin;
        for (int k = 0; k < rr.degree; ++k) if (ap[k].head == j) return ap[k].id;
        return -1;
    }
    // reduction transfer of one row (see insert_row_t), safe beside other rows' transfers: a row writes its own dual and its
    // matched column's, and a stale (higher) dual read from another column only makes the raise smaller
    void transfer_row_shared(int r)
    {
        std::vector<Col> &c = L.c;
        Row &rr = L.rw[r];
        const int n = L.n, m = rr.col;
        int64_t best2 = INF, cm = 0;
        auto see = [&](int j, int64_t w) { if (j == m) cm = w; else { const int64_t k = w - v_of(c[j]); if (k < best2) best2 = k; } };
        see(r, 0);
        see(n + r, rr.own);
        const Arc *ap = L.arcs.data() + rr.arc_begin;
        for (int k = 0; k < rr.degree; ++k) see(ap[k].head, rr.base + ap[k].w);
        if (best2 < INF && best2 > rr.u) { rr.u = best2; __atomic_store_n(&c[m].v, cm - best2, __ATOMIC_RELAXED); }
    }
    // Every member of the team runs this, the owner as tid 0. A round = what team.mode says, a barrier, member 0 decides the
    // next mode, a barrier. In kOwnerStep the owner leaves the loop for its serial work (dual update, augmentation) while the
    // helpers wait at the round's first barrier; team_resume() brings it back.
    int owner_ph = 0, team_sink_pred = -1;
    bool in_team = false;
    size_t par_next_try = 0;
    // a helper: recruited before the owner knew the team's size, so it waits until the team is set up; says when it has left
    void team_helper(int tid, int epoch)
    {
        int spins = 0;
        while (team.epoch.load(std::memory_order_acquire) - epoch < 0)
            if (++spins > 4096) { std::this_thread::yield(); spins = 0; } else __builtin_ia32_pause();
        int ph = team.start_phase;
        team_rounds(tid, ph);
        team.gone.fetch_add(1, std::memory_order_release);
    }
    void team_rounds(int tid, int &ph)
    {
        Team::Slot &me = team.slot[tid];
        std::vector<Col> &c = L.c;
        constexpr size_t CH = 16;
        for (;;) {
            const int mode = team.mode;
            if (mode == kLeave) return;
            if (mode == kOwnerStep) {
                if (tid == 0) return;
            } else if (mode == kExpand) {
                // drain this bucket
                for (;;) {
                    size_t since = 0;
                    while (!me.near.empty()) {
                        const int32_t j = me.near.back();
                        me.near.pop_back();
                        expand(j, me);
                        if (++since >= 64) {
                            since = 0;
                            if (team.hungry.load(std::memory_order_relaxed) > 0 && me.near.size() >= 4 * CH) {
                                std::lock_guard<std::mutex> g(team.pool_m);
                                const size_t give = me.near.size() / 2;
                                team.pool.insert(team.pool.end(), me.near.begin(), me.near.begin() + give);
                                me.near.erase(me.near.begin(), me.near.begin() + give);
                                team.pool_size.store(team.pool.size(), std::memory_order_release);
                            }
                        }
                    }
                    // dry: take from the pool, or wait until somebody gives or everybody is dry
                    bool got = false;
                    {
                        std::unique_lock<std::mutex> g(team.pool_m);
                        if (!team.pool.empty()) {
                            const size_t take = std::min<size_t>(team.pool.size(), std::max<size_t>(CH, team.pool.size() / team.size));
                            me.near.assign(team.pool.end() - take, team.pool.end());
                            team.pool.resize(team.pool.size() - take);
                            team.pool_size.store(team.pool.size(), std::memory_order_release);
                            got = true;
                        } else if (team.hungry.fetch_add(1, std::memory_order_relaxed) + 1 == team.size)
                            team.drained.store(true, std::memory_order_release);
                    }
                    if (got) continue;
                    int spins = 0;
                    for (;;) {
                        if (team.drained.load(std::memory_order_acquire)) break;
                        if (team.pool_size.load(std::memory_order_acquire) > 0) {     // (a peek; checked again under the lock)
                            std::unique_lock<std::mutex> g(team.pool_m);
                            if (!team.pool.empty() && !team.drained.load(std::memory_order_relaxed)) {
                                const size_t take = std::min<size_t>(team.pool.size(), std::max<size_t>(CH, team.pool.size() / team.size));
                                me.near.assign(team.pool.end() - take, team.pool.end());
                                team.pool.resize(team.pool.size() - take);
                                team.pool_size.store(team.pool.size(), std::memory_order_release);
                                team.hungry.fetch_sub(1, std::memory_order_relaxed);
                                got = true;
                                break;
                            }
                        }
                        if (++spins > 2048) { std::this_thread::yield(); spins = 0; } else __builtin_ia32_pause();
                    }
                    if (!got) break;
                }
            } else if (mode == kScanFar || mode == kRefill) {
                // kScanFar: drop what cannot matter any more, find the nearest label left; kRefill: the next bucket moves to `near`
                const int64_t bd = dist_of(team.bound.load(std::memory_order_relaxed));
                size_t keep = 0;
                int64_t fmin = INF;
                for (int32_t j : me.far) {
                    const int64_t d = dist_of(L.plabel[j].load(std::memory_order_relaxed));
                    if (d >= bd || d >= __atomic_load_n(&L.pexpanded[j], __ATOMIC_RELAXED)) continue;
                    if (mode == kRefill && d < team.theta) { me.near.push_back(j); continue; }
                    me.far[keep++] = j;
                    if (d < fmin) fmin = d;
                }
                me.far.resize(keep);
                me.far_min = fmin;
            } else if (mode == kWriteBack) {
                // the labels of the columns nearer than the free column go where the dual update and the augmentation read them
                for (int32_t j : me.touched) {
                    const uint64_t lab = L.plabel[j].load(std::memory_order_relaxed);
                    const int64_t d = dist_of(lab);
                    const int pr = (int)(lab & ((1u << Lsap::kDistShift) - 1));
                    if (c[j].row >= 0 && d <= team.min_val) {        // (<=: the path may reach the free column over arcs of reduced cost 0)
                        c[j].spc = d; c[j].pred_row = pr; c[j].pred_arc = arc_to(pr, j);
                        me.scanned.push_back(j);
                    }
                    L.plabel[j].store(Lsap::kNoLabel, std::memory_order_relaxed);
                    L.pexpanded[j] = INF;
                }
                me.touched.clear();
            } else if (mode == kTransfer) {
                size_t idx;
                while ((idx = team.cursor.fetch_add(64, std::memory_order_relaxed)) < sr_rows.size())
                    for (size_t q = idx; q < std::min(idx + 64, sr_rows.size()); ++q) transfer_row_shared(sr_rows[q]);
            }
            team_barrier(ph);
            if (tid == 0) team_decide();
            team_barrier(ph);
        }
    }
    // the owner, back from its serial step: the team goes on with `next` (kTransfer, or kLeave: the helpers go home)
    void team_resume(int next)
    {
        team.pending = next;
        team.cursor.store(0, std::memory_order_relaxed);
        team_barrier(owner_ph);                  // the first barrier of the kOwnerStep round, where the helpers wait
        team_decide();
        team_barrier(owner_ph);
        team_rounds(0, owner_ph);
        // the team's fields are reused by this thread's next large search: not before every helper is out of the loop
        while (team.gone.load(std::memory_order_acquire) < team.size - 1) __builtin_ia32_pause();
    }
    // between the two barriers of a round, member 0 alone: what the team does next
    void team_decide()
    {
        static const bool trace = getenv("AXT_MCF_TRACE") != nullptr;
        if (trace) fprintf(stderr, "[par]   decide: mode %d theta %.4f bound %.4f\n", team.mode, (double)team.theta / 65536e6, (double)dist_of(team.bound.load()) / 65536e6);
        if (team.mode == kExpand) {
            team.mode = kScanFar;                // the bucket is empty: what is the nearest label left?
        } else if (team.mode == kRefill) {
            team.hungry.store(0, std::memory_order_relaxed);
            team.drained.store(false, std::memory_order_relaxed);
            team.mode = kExpand;
        } else if (team.mode == kScanFar) {
            int64_t gmin = INF;
            for (const Team::Slot &sl : team.slot) gmin = std::min(gmin, sl.far_min);
            const int64_t bd = dist_of(team.bound.load(std::memory_order_relaxed));
            if (gmin >= bd) {                    // nothing left that is nearer than the free column: the search is over
                team.min_val = bd;
                const int sink = (int)(team.bound.load(std::memory_order_relaxed) & ((1u << Lsap::kDistShift) - 1));
                team_sink_pred = (int)(L.plabel[sink].load(std::memory_order_relaxed) & ((1u << Lsap::kDistShift) - 1));
                team.mode = kWriteBack;
            } else {
                team.theta = gmin + team.delta;
                team.mode = kRefill;
            }
        } else if (team.mode == kWriteBack) {
            team.mode = kOwnerStep;
        } else if (team.mode == kOwnerStep) {
            team.mode = team.pending;
        } else if (team.mode == kTransfer) {
            team.mode = kLeave;
        }
    }

    // The search for row i has settled the columns sc_cols (the row of the last one, `cur`, not expanded yet) at distances up to
    // minVal and holds open labels in its heap: the rest of it on a team of 1 + helpers threads. Returns the sink; minVal, the
    // labels (Col::spc / pred_row / pred_arc), sr_rows and sc_cols are left as the serial search would have left them.
    int finish_in_parallel(int i, int helpers, int64_t &minVal, int64_t best_free, uint32_t open)
    {
        std::vector<Col> &c = L.c;
        const int n = L.n;
        std::call_once(L.par_alloc, [&] {
            L.plabel.reset(new std::atomic<uint64_t>[2 * (size_t)n]);
            L.pexpanded.reset(new int64_t[2 * (size_t)n]);
            for (size_t j = 0; j < 2 * (size_t)n; ++j) { L.plabel[j].store(Lsap::kNoLabel, std::memory_order_relaxed); L.pexpanded[j] = INF; }
        });
        team.size = 1 + helpers;
        team.gone.store(0, std::memory_order_relaxed);
        team.start_phase = owner_ph;             // (the barrier counters run on from search to search)
        if ((int)team.slot.size() < team.size) team.slot.resize(team.size);
        for (Team::Slot &sl : team.slot) { sl.near.clear(); sl.far.clear(); sl.touched.clear(); sl.scanned.clear(); sl.far_min = INF; }
        team.pool.clear();
        team.pool_size.store(0, std::memory_order_relaxed);
        team.hungry.store(0, std::memory_order_relaxed);
        team.drained.store(false, std::memory_order_relaxed);
        Team::Slot &me = team.slot[0];
        team.bias = c[sc_cols[0]].spc;           // the first column Dijkstra settled: no label is nearer
        team.delta = std::max<int64_t>(1, (best_free - minVal) / L.par_buckets);
        team.theta = minVal + team.delta;
        uint64_t bound = Lsap::kNoLabel;
        auto seed = [&](int j, bool expanded) {
            const Col &cj = c[j];
            L.plabel[j].store(pack(cj.spc, cj.pred_row), std::memory_order_relaxed);
            L.pexpanded[j] = expanded ? cj.spc : INF;
            me.touched.push_back(j);
            if (expanded) return;
            if (cj.row < 0) bound = std::min(bound, pack(cj.spc, j));
            else (cj.spc < team.theta ? me.near : me.far).push_back(j);
        };
        for (size_t k = 0; k < sc_cols.size(); ++k) seed(sc_cols[k], k + 1 < sc_cols.size());
        heap.for_each([&](const Item &it) {
            const Col &cj = c[it.second];
            if (cj.stamp == open && it.first == cj.spc && L.plabel[it.second].load(std::memory_order_relaxed) == Lsap::kNoLabel) seed(it.second, false);
        });
        heap.clear();
        team.bound.store(bound, std::memory_order_relaxed);
        team.cursor.store(0);
        team.mode = kExpand;
        if (me.near.size() > 64) {               // the open labels of the serial phase: shared out from the start
            team.pool.assign(me.near.begin() + 32, me.near.end());
            me.near.resize(32);
            team.pool_size.store(team.pool.size(), std::memory_order_relaxed);
        }
        if (getenv("AXT_MCF_TRACE")) fprintf(stderr, "[par] row %d: team %d, closed %zu, near %zu far %zu, bound %.3f minVal %.3f theta %.3f\n", i, team.size, sc_cols.size(), me.near.size() + team.pool.size(), me.far.size(), (double)dist_of(bound) / 65536e6, (double)minVal / 65536e6, (double)team.theta / 65536e6);
        team.epoch.fetch_add(1, std::memory_order_release);          // the helpers recruited for this search may come in
        team_rounds(0, owner_ph);                // returns in kOwnerStep: the labels are written back, the helpers wait
        minVal = team.min_val;
        const int sink = (int)(team.bound.load(std::memory_order_relaxed) & ((1u << Lsap::kDistShift) - 1));
        {   // the sink's label (it is not matched: the write-back skipped it; its label was reset there, so the predecessor
            // comes from the bound's twin kept by the member that set it -- simplest: look it up among the scanned rows)
            Col &cs = c[sink];
            cs.spc = minVal;
            cs.pred_row = team_sink_pred;
            cs.pred_arc = arc_to(team_sink_pred, sink);
        }
        sc_cols.clear();
        sr_rows.clear();
        sr_rows.push_back(i);
        for (Team::Slot &sl : team.slot) {
            stat_relax += sl.relax; stat_push += sl.push; sl.relax = sl.push = sl.rows = 0;
            for (int32_t j : sl.scanned) { sc_cols.push_back(j); sr_rows.push_back(c[j].row); }
            sl.scanned.clear();
        }
        sc_cols.push_back(sink);
        return sink;
    }
    // rows in a fixed pseudo-random order (xorshift64 seeded by the first row: the same at any thread count)
    void insert_shuffled(int lo, int hi)
    {
        const int m = hi - lo;
        if (m <= 0) return;
        std::vector<int32_t> order(m);
        for (int i = 0; i < m; ++i) order[i] = lo + i;
        uint64_t x = 88172645463325252ull ^ ((uint64_t)lo * 0x9e3779b97f4a7c15ull);
        if (!x) x = 88172645463325252ull;
        for (int i = m - 1; i > 0; --i) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            std::swap(order[i], order[x % (uint64_t)(i + 1)]);
        }
        for (int i = 0; i < m; ++i) insert_row(order[i]);
    }
    };  // struct Search

    // Insert `rows` (in this order of seniority) on `threads` host threads that share every row and column.
    void insert_concurrently(const std::vector<int32_t> &rows, int threads)
    {
        if (rows.empty()) return;
        if (threads > (int)rows.size()) threads = (int)rows.size();
        if (threads <= 1) {
            Search w(*this, next_stamp.load());
            for (int32_t k : rows) w.insert_row(k);
            next_stamp.store(w.search + 2);
            return;
        }
        if (!owner) {
            owner.reset(new std::atomic<int32_t>[2 * (size_t)n]);
            for (size_t j = 0; j < 2 * (size_t)n; ++j) owner[j].store(0, std::memory_order_relaxed);
        }
        std::atomic<size_t> next{0};
        auto work = [&]() {
            Search w(*this, 0);
            for (;;) {
                const size_t q = next.fetch_add(1, std::memory_order_relaxed);
                if (q >= rows.size()) break;
                w.ticket = (int32_t)q + 1;
                while (!w.insert_row(rows[q])) {
                    // start again once the older search that was in the way has let go of that column
                    int spins = 0;
                    while (owner[w.blocked_col].load(std::memory_order_acquire) == w.blocked_by)
                        if (++spins > 64) { std::this_thread::yield(); spins = 0; } else __builtin_ia32_pause();
                }
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; ++t) {
            try { pool.emplace_back(work); } catch (...) { break; }
        }
        work();
        for (std::thread &t : pool) t.join();
    }

    // ---- time blocks ---------------------------------------------------------------------------------------
    // Row k only reaches columns R_k, X_k and R_b of its successors b > k, all within reach[k] = max b. Cut the rows at
    // p: with q = 1 + max reach of the rows before p, the rows [0,p) use columns below q and the rows [q,n) columns
    // from q on -- two independent assignment problems as long as the rows [p,q) (two frames of detections for the
    // tracker's networks) stay out. So: leaves of a binary tree = blocks of rows solved concurrently, inner nodes =
    // the separator rows between two finished halves, inserted by ordinary augmenting-path searches (which cannot
    // leave the two halves: the next separator's rows are not in yet). Row insertion reaches the one optimum in any
    // order, so the result does not depend on the number of threads.
    std::vector<int32_t> cut_p, cut_q;        // leaf i = rows [cut_q[i], cut_p[i+1]), separator i = rows [cut_p[i], cut_q[i])

    // Frame-sharded ranks (axt_mcf_shard_*): the leaves are dealt to the ranks in equal runs of `group` leaves; a rank solves
    // its own run -- a subtree -- from the replicated network, the states of all subtrees are exchanged (one all-gather)
    // and every rank then joins them: solve_tree with solved_group = group skips the subtrees and inserts the separators
    // above them. The optimum is unique, so this is the single-process result.
    int solved_group = 0;
    static constexpr uint32_t kJoinStamp = 0x20000000u;          // above every stamp a subtree can have left in its columns

    uint32_t solve_tree(int a, int b)
    {
        if (solved_group > 0 && b - a == solved_group && a % solved_group == 0) return kJoinStamp;
        if (b - a == 1) {
            const double t0 = now_ms();
            Search w(*this, 0);
            tasks_running.fetch_add(1, std::memory_order_relaxed);
            w.insert_shuffled(cut_q[a], cut_p[a + 1]);
            tasks_running.fetch_sub(1, std::memory_order_relaxed);
            if (getenv("AXT_MCF_DEBUG")) fprintf(stderr, "  leaf %d: rows [%d,%d) scanned %zu, %.1f ms\n", a, cut_q[a], cut_p[a + 1], w.stat_rows, now_ms() - t0);
            return w.search + 2;
        }
        const int m = (a + b) / 2;
        uint32_t left = 0, right = 0;
        bool spawned = false;
        std::thread t;
        try {
            t = std::thread([&] { left = solve_tree(a, m); });
            spawned = true;
        } catch (...) {
        }
        if (!spawned) left = solve_tree(a, m);
        right = solve_tree(m, b);
        if (spawned) t.join();
        const double t0 = now_ms();
        Search w(*this, left > right ? left : right);
        tasks_running.fetch_add(1, std::memory_order_relaxed);
        w.insert_shuffled(cut_p[m], cut_q[m]);
        tasks_running.fetch_sub(1, std::memory_order_relaxed);
        if (getenv("AXT_MCF_DEBUG")) fprintf(stderr, "  separator %d (of leaves %d..%d): rows [%d,%d) scanned %zu, %.1f ms\n", m, a, b, cut_p[m], cut_q[m], w.stat_rows, now_ms() - t0);
        return w.search + 2;
    }

    static int thread_budget()
    {
        if (const char *e = getenv("AXT_MCF_THREADS")) {
            const int v = atoi(e);
            if (v >= 1) return v < 256 ? v : 256;
        }
        const unsigned hc = std::thread::hardware_concurrency();
        return hc == 0 ? 1 : (hc > 16 ? 16 : (int)hc);        // a GPU's share of the host's cores
    }

    int leaves = 1, budget = 1;
    double t_prep0 = 0, t_prep1 = 0;

    // setup of the rows and the cuts of the time-block tree; `world` > 1: a leaf count that the ranks can share evenly
    void prepare(int world = 1)
    {
        const double t0 = now_ms();
        t_prep0 = t0;
        budget = thread_budget();
        two_phase = getenv("AXT_MCF_TWO_PHASE") ? true : getenv("AXT_MCF_ONE_PHASE") ? false : n <= 120000;
        c.assign(2 * (size_t)n, Col{0, 0, -1, -1, -1, 0});
        rw.resize(n);
        arcs.resize((size_t)row_ptr[n]);
        std::vector<int32_t> reach(n);
        auto setup = [&](int k0, int k1) {
            for (int k = k0; k < k1; ++k) {
                const int64_t lo = row_ptr[k], hi = row_ptr[k + 1];
                rw[k] = Row{0, obs[k] + entry[k], obs[k] + entry[k] + exitc[k] - discount(k), lo, (int32_t)(hi - lo), -1, -1, 0};
                int32_t far = k;
                for (int64_t e = lo; e < hi; ++e) {                             // insertion sort: rows are short
                    const int64_t w = cost[e] - entry[col[e]];
                    int64_t q = e;
                    while (q > lo && arcs[q - 1].w > w) { arcs[q] = arcs[q - 1]; --q; }
                    arcs[q] = Arc{w, col[e], (int32_t)e};
                    if (col[e] > far) far = col[e];
                }
                reach[k] = far;
            }
        };
        {
            const int parts = n >= 4096 ? budget : 1;
            std::vector<std::thread> pool;
            for (int t = 1; t < parts; ++t) {
                const int k0 = (int)((int64_t)n * t / parts), k1 = (int)((int64_t)n * (t + 1) / parts);
                try { pool.emplace_back(setup, k0, k1); } catch (...) { setup(k0, k1); }
            }
            setup(0, (int)((int64_t)n / parts));
            for (std::thread &t : pool) t.join();
        }
        // leaves: a power of two, at least kMinLeaf rows each, separators that do not run into the next cut
        int kMinLeaf = 1024;
        if (const char *e = getenv("AXT_MCF_MIN_LEAF")) kMinLeaf = atoi(e) >= 1 ? atoi(e) : 1;    // tests: force the tree on small networks
        leaves = 1;
        // shared between ranks: every rank must build the SAME tree, so nothing of this process (its CPU set, its environment's
        // thread count) may enter the leaf count -- a fixed 16 leaves per rank, the thread budget of a one-GPU share
        int cap_leaves = world > 1 ? 16 * world : budget;
        if (const char *e = getenv("AXT_MCF_LEAVES")) cap_leaves = atoi(e) >= 1 ? atoi(e) : 1;       // (experiments: more leaves than threads)
        while (leaves * 2 <= cap_leaves && n / (leaves * 2) >= kMinLeaf) leaves *= 2;
        for (; leaves > 1; leaves /= 2) {
            cut_p.assign(leaves + 1, 0);
            cut_q.assign(leaves + 1, 0);
            cut_p[leaves] = cut_q[leaves] = n;
            std::vector<int32_t> want(leaves);
            for (int i = 1; i < leaves; ++i) want[i] = (int32_t)((int64_t)n * i / leaves);
            bool ok = true;
            int32_t far = -1;
            int next = 1;
            for (int k = 0; k < n && next < leaves; ++k) {
                while (next < leaves && k == want[next]) {
                    cut_p[next] = k;
                    cut_q[next] = far + 1 > k ? far + 1 : k;
                    ++next;
                }
                if (reach[k] > far) far = reach[k];
            }
            for (int i = 1; i < leaves && ok; ++i)
                ok = cut_q[i] <= cut_p[i + 1] && cut_q[i] - cut_p[i] < (cut_p[i + 1] - cut_p[i]) / 2;
            if (ok) break;
        }
        if (leaves <= 1) {
            leaves = 1;
            cut_p.assign(2, 0); cut_q.assign(2, 0);
            cut_p[1] = cut_q[1] = n;
        }
        start_helpers();
        t_prep1 = now_ms();
    }

    // ---- state of a subtree of leaves [a, b): its rows [cut_q[a], cut_p[b]) and the columns they can reach, in-slots
    // [cut_q[a], cut_q[b]) and the rows' private exits. Layout: rows {u i64, col i32, arc i32}, in-slots {v i64, row i32, pad},
    // exits {v i64, row i32, pad}.
    // (a 16-byte header first: a hash of the cuts of the whole tree, n and the leaf count -- a rank whose tree differs, e.g.
    // through a different AXT_MCF_MIN_LEAF, is told so instead of importing a state laid out for other cuts)
    uint64_t tree_hash() const
    {
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](uint64_t v) { h ^= v; h *= 1099511628211ull; };
        mix((uint64_t)n); mix((uint64_t)leaves);
        for (int32_t v : cut_p) mix((uint64_t)(uint32_t)v);
        for (int32_t v : cut_q) mix((uint64_t)(uint32_t)v);
        return h;
    }
    int64_t state_bytes(int a, int b) const
    {
        const int64_t rows = cut_p[b] - cut_q[a], cols = cut_q[b] - cut_q[a];
        return 16 + 16 * (rows + cols + rows);
    }
    bool state_matches(const unsigned char *in) const
    {
        struct Head { uint64_t h; int32_t n, leaves; };
        const Head *hd = reinterpret_cast<const Head *>(in);
        return hd->h == tree_hash() && hd->n == n && hd->leaves == leaves;
    }
    void export_state(int a, int b, unsigned char *out) const
    {
        struct Head { uint64_t h; int32_t n, leaves; };
        *reinterpret_cast<Head *>(out) = Head{tree_hash(), n, leaves};
        out += 16;
        struct Rec { int64_t x; int32_t y, z; };
        Rec *o = reinterpret_cast<Rec *>(out);
        for (int k = cut_q[a]; k < cut_p[b]; ++k) *o++ = Rec{rw[k].u, rw[k].col, rw[k].arc};
        for (int j = cut_q[a]; j < cut_q[b]; ++j) *o++ = Rec{c[j].v, c[j].row, 0};
        for (int k = cut_q[a]; k < cut_p[b]; ++k) *o++ = Rec{c[n + k].v, c[n + k].row, 0};
    }
    void import_state(int a, int b, const unsigned char *in)
    {
        in += 16;
        struct Rec { int64_t x; int32_t y, z; };
        const Rec *o = reinterpret_cast<const Rec *>(in);
        for (int k = cut_q[a]; k < cut_p[b]; ++k, ++o) { rw[k].u = o->x; rw[k].col = o->y; rw[k].arc = o->z; }
        for (int j = cut_q[a]; j < cut_q[b]; ++j, ++o) { c[j].v = o->x; c[j].row = o->y; c[j].stamp = 0; }
        for (int k = cut_q[a]; k < cut_p[b]; ++k, ++o) { c[n + k].v = o->x; c[n + k].row = o->y; c[n + k].stamp = 0; }
    }

    void run()
    {
        prepare();
        solve_tree(0, leaves);
        second_phase();
    }

    void second_phase()
    {
        const double t0 = t_prep0, t1 = t_prep1;
        const double t2 = now_ms();
        size_t ends = 0;
        if (two_phase) {
            // second phase: the exits get their full price back. Nothing else changes, so every dual stays feasible and
            // every matched edge tight except those of the rows that sit on their exit column: those are taken out (their
            // private column is free again) and inserted afresh.
            std::vector<int32_t> again;
            for (int k = 0; k < n; ++k) {
                const bool at_exit = rw[k].col == n + k;
                rw[k].own += discount(k);
                if (at_exit && discount(k) > 0) again.push_back(k);
            }
            two_phase = false;
            for (int32_t k : again) { c[n + k].row = -1; c[n + k].v = 0; rw[k].col = -1; rw[k].arc = -1; rw[k].u = 0; }
            next_stamp.store(0x7f000000u);
            // side by side only where the searches have room to miss each other: on config 3's 19 k detections every
            // search covers a sixth of the timelapse and the younger ones would mostly wait
            int par = n >= 50000 ? budget : 1;
            if (const char *e = getenv("AXT_MCF_ENDS_THREADS")) par = atoi(e) >= 1 ? atoi(e) : 1;
            if (par > 1) {
                // searches that run side by side should be far apart in time: deal the ends (which are in time order) so
                // that consecutive tickets come from `par` different stretches of the timelapse
                std::vector<int32_t> dealt;
                dealt.reserve(again.size());
                const size_t per = (again.size() + par - 1) / par;
                for (size_t q = 0; q < per; ++q)
                    for (int t = 0; t < par; ++t)
                        if (t * per + q < again.size() && t * per + q < (t + 1) * per) dealt.push_back(again[t * per + q]);
                again.swap(dealt);
            }
            tasks_running.fetch_add(par, std::memory_order_relaxed);
            insert_concurrently(again, par);
            tasks_running.fetch_sub(par, std::memory_order_relaxed);
            ends = again.size();
        }
        if (getenv("AXT_MCF_DEBUG"))
            fprintf(stderr, "lsap: second phase %zu track ends, %.1f ms; ", ends, now_ms() - t2);
        if (getenv("AXT_MCF_DEBUG"))
            fprintf(stderr, "lsap: n=%d threads<=%d leaves=%d rows scanned=%zu relax=%zu push=%zu gave up=%zu  setup %.1f ms, insertions %.1f ms\n", n,
                    budget, leaves, stat_rows, stat_relax, stat_push, stat_died, t1 - t0, now_ms() - t1);
    }
};

void Lsap::helper_main()
{
    for (;;) {
        HelperPool::Job job;
        {
            std::unique_lock<std::mutex> lk(pool.m);
            ++pool.idle;
            pool.cv.wait(lk, [&] { return pool.quit || !pool.jobs.empty(); });
            if (pool.jobs.empty()) { --pool.idle; return; }
            job = pool.jobs.front();
            pool.jobs.pop_front();
        }
        static_cast<Search *>(job.search)->team_helper(job.tid, job.epoch);
    }
}

}  // namespace

// The whole solve; `solved`: an assignment-form solver that has already run (the frame-sharded path), or null.
// h_pot_u / h_pot_v / pot_t (all three or none): node potentials of the optimum (include/axtrack_hip.h, axt_mcf_solve_duals).
static int mcf_solve_impl(int n_det, const int64_t *h_obs, const int64_t *h_entry, const int64_t *h_exit,
                          const int64_t *h_row_ptr, const int32_t *h_col, const int64_t *h_cost, int min_flow,
                          int max_flow, int32_t *h_next, int32_t *h_track, int *n_tracks, int64_t *total_cost, Lsap *solved,
                          int64_t *h_pot_u = nullptr, int64_t *h_pot_v = nullptr, int64_t *pot_t = nullptr)
{
    if (n_det < 0 || !h_row_ptr || !n_tracks || !total_cost || (n_det > 0 && (!h_obs || !h_entry || !h_exit || !h_next || !h_track))) {
        axt_set_error("axt_mcf_solve: null or negative argument");
        return AXT_EINVAL;
    }
    if (min_flow < 0 || max_flow < min_flow) {
        axt_set_error("axt_mcf_solve: need 0 <= min_flow <= max_flow (got %d, %d)", min_flow, max_flow);
        return AXT_EINVAL;
    }
    for (int k = 0; k < n_det; ++k)
        for (int64_t e = h_row_ptr[k]; e < h_row_ptr[k + 1]; ++e)
            if (h_col[e] <= k || h_col[e] >= n_det) {
                axt_set_error("axt_mcf_solve: arc %lld of detection %d does not point forward in time", (long long)e, k);
                return AXT_EINVAL;
            }
    *n_tracks = 0;
    *total_cost = 0;
    Solver s;
    s.n = n_det;
    s.obs = h_obs; s.entry = h_entry; s.exitc = h_exit; s.row_ptr = h_row_ptr; s.cost = h_cost; s.col = h_col;
    s.S = 2 * n_det; s.T = 2 * n_det + 1;
    s.pred.assign(n_det, NONE);
    s.succ.assign(n_det, NONE);
    s.pred_tail.assign(n_det, -1);
    int F = 0;
    int64_t total = 0;
    bool done = false;
    if (n_det > 0 && (solved || !getenv("AXT_MCF_FORCE_SSP"))) {
        // fast path: optimum over all flow counts as an assignment problem
        Lsap local;
        Lsap &a = solved ? *solved : local;
        if (!solved) {
            a.n = n_det;
            a.obs = h_obs; a.entry = h_entry; a.exitc = h_exit; a.row_ptr = h_row_ptr; a.cost = h_cost; a.col = h_col;
            a.run();
        }
        for (int k = 0; k < n_det; ++k) {
            const int j = a.rw[k].col;
            if (j == k) continue;                                   // unused
            total += h_obs[k];
            if (j >= n_det) { s.succ[k] = TERMINAL; total += h_exit[k]; }
            else { s.succ[k] = a.rw[k].arc; s.pred[j] = a.rw[k].arc; s.pred_tail[j] = k; total += h_cost[a.rw[k].arc]; }
        }
        for (int k = 0; k < n_det; ++k)
            if (a.rw[k].col != k && s.pred[k] == NONE) { s.pred[k] = TERMINAL; total += h_entry[k]; ++F; }
        done = F >= min_flow && F <= max_flow;
        if (done && h_pot_u) {
            // The assignment duals as node potentials of the flow network, pi(S) = pi(T) = 0: pi(u_k) = entry_k + v(R_k),
            // pi(v_k) = obs_k + entry_k - u(row k). Then the reduced cost of a transition arc is its row/column slack, of an
            // entry arc -v(R_k) >= 0 (= 0 where R_k is unmatched: a track starts), of an exit arc the slack of X_k (<= 0 only
            // where the row sits on it), of an observation arc u + v of (row k, R_k) <= 0 (= 0 where the detection is unused).
            for (int k = 0; k < n_det; ++k) {
                h_pot_u[k] = h_entry[k] + a.c[k].v;
                h_pot_v[k] = h_obs[k] + h_entry[k] - a.rw[k].u;
            }
            *pot_t = 0;
        }
        if (!done) {
            std::fill(s.pred.begin(), s.pred.end(), NONE);
            std::fill(s.succ.begin(), s.succ.end(), NONE);
            F = 0;
            total = 0;
        }
    }
    if (n_det > 0 && !done) {
        s.dist.assign(2 * n_det + 2, INF);
        s.par.assign(2 * n_det + 2, -1);
        s.par_arc.assign(2 * n_det + 2, -1);
        s.init_potentials();
        while (F < max_flow) {
            if (!s.dijkstra()) {
                // no further path: the nodes the search reached move down together (reduced costs among them and out of
                // the unreached part stay >= 0, no residual arc leaves them), which puts pi(T) - pi(S) at or above 0
                if (h_pot_u) {
                    int64_t far = 0;
                    for (int32_t x : s.touched) if (s.dist[x] > far) far = s.dist[x];
                    for (int32_t x : s.touched) s.pi[x] += s.dist[x] - far;
                    const int64_t gap = s.pi[s.T] - s.pi[s.S];
                    if (gap < 0) for (int32_t x : s.touched) s.pi[x] += gap;
                }
                break;
            }
            const int64_t dT = s.dist[s.T];
            const int64_t path_cost = dT + s.pi[s.T] - s.pi[s.S];
            if (F >= min_flow && path_cost >= 0) {
                // optimal. Potentials for the certificate: pi += min(dist, theta) - theta keeps every residual reduced cost
                // >= 0 for any 0 <= theta <= dT; theta = -(pi(T) - pi(S)) (the cost of the last path pushed, if negative)
                // puts pi(T) - pi(S) at exactly 0, as an arc T -> S strictly inside its bounds requires.
                if (h_pot_u) {
                    int64_t theta = -(s.pi[s.T] - s.pi[s.S]);
                    if (theta < 0) theta = 0;
                    if (theta > dT) theta = dT;
                    for (int32_t x : s.touched) s.pi[x] += (s.dist[x] < theta ? s.dist[x] : theta) - theta;
                }
                break;
            }
            // Johnson update pi'[x] = pi[x] + min(dist[x], dT) keeps every residual reduced cost >= 0. Adding
            // the same constant to all potentials changes no reduced cost, so instead of +dT on the (many)
            // nodes the search never touched, the touched ones get min(dist, dT) - dT <= 0.
            if (getenv("AXT_MCF_SSP_STATS")) {          // how much of the shortest-path tree an augmentation invalidates (the work muSSP cannot avoid)
                size_t changed = 0, settled = 0;
                for (int32_t x : s.touched) { if (s.dist[x] <= dT) { ++settled; if (s.dist[x] > 0) ++changed; } }
                fprintf(stderr, "ssp path %d: cost %.3f, touched %zu, settled %zu, distance changed %zu of %d nodes\n", F, (double)(path_cost >> 16) * 1e-6, s.touched.size(), settled, changed, 2 * n_det + 2);
            }
            for (int32_t x : s.touched) s.pi[x] += (s.dist[x] < dT ? s.dist[x] : dT) - dT;
            s.augment();
            total += path_cost;
            ++F;
        }
    }
    if (F < min_flow) {
        for (int k = 0; k < n_det; ++k) { h_next[k] = -1; h_track[k] = -1; }
        return AXT_INFEASIBLE;
    }
    int id = 0;
    for (int k = 0; k < n_det; ++k) { h_next[k] = -1; h_track[k] = -1; }
    for (int k = 0; k < n_det; ++k) {
        if (s.pred[k] != TERMINAL) continue;
        int x = k;
        for (;;) {
            h_track[x] = id;
            const int32_t e = s.succ[x];
            if (e < 0) break;
            h_next[x] = h_col[e];
            x = h_col[e];
        }
        ++id;
    }
    *n_tracks = id;
    *total_cost = total;
    if (h_pot_u && !done) {
        for (int k = 0; k < n_det; ++k) { h_pot_u[k] = s.pi[s.U(k)] - s.pi[s.S]; h_pot_v[k] = s.pi[s.V(k)] - s.pi[s.S]; }
        *pot_t = s.pi[s.T] - s.pi[s.S];
    }
    return AXT_OK;
}

extern "C" int axt_mcf_solve(int n_det, const int64_t *h_obs, const int64_t *h_entry, const int64_t *h_exit,
                             const int64_t *h_row_ptr, const int32_t *h_col, const int64_t *h_cost, int min_flow,
                             int max_flow, int32_t *h_next, int32_t *h_track, int *n_tracks, int64_t *total_cost)
{
    return mcf_solve_impl(n_det, h_obs, h_entry, h_exit, h_row_ptr, h_col, h_cost, min_flow, max_flow, h_next, h_track, n_tracks,
                          total_cost, nullptr);
}

extern "C" int axt_mcf_solve_duals(int n_det, const int64_t *h_obs, const int64_t *h_entry, const int64_t *h_exit,
                                   const int64_t *h_row_ptr, const int32_t *h_col, const int64_t *h_cost, int min_flow,
                                   int max_flow, int32_t *h_next, int32_t *h_track, int *n_tracks, int64_t *total_cost,
                                   int64_t *h_pot_u, int64_t *h_pot_v, int64_t *pot_t)
{
    if (n_det > 0 && (!h_pot_u || !h_pot_v || !pot_t)) { axt_set_error("axt_mcf_solve_duals: null potential array"); return AXT_EINVAL; }
    int64_t dummy = 0;
    return mcf_solve_impl(n_det, h_obs, h_entry, h_exit, h_row_ptr, h_col, h_cost, min_flow, max_flow, h_next, h_track, n_tracks,
                          total_cost, nullptr, h_pot_u ? h_pot_u : &dummy, h_pot_v ? h_pot_v : &dummy, pot_t ? pot_t : &dummy);
}

// ---- frame-sharded solve (include/axtrack_hip.h: axt_mcf_shard_*) ---------------------------------------------------
struct axt_mcf_shard {
    Lsap a;
    int rank = 0, world = 1, group = 0;          // group = leaves per rank (0: the network is too small to share: every rank solves it whole)
};

extern "C" int axt_mcf_shard_begin(int n_det, const int64_t *h_obs, const int64_t *h_entry, const int64_t *h_exit,
                                   const int64_t *h_row_ptr, const int32_t *h_col, const int64_t *h_cost, int rank, int world,
                                   axt_mcf_shard **out, int64_t *state_bytes)
{
    if (n_det < 0 || !h_row_ptr || !out || !state_bytes || world < 1 || rank < 0 || rank >= world || (world & (world - 1)) ||
        (n_det > 0 && (!h_obs || !h_entry || !h_exit))) {
        axt_set_error("axt_mcf_shard_begin: bad argument (world must be a power of two, 0 <= rank < world)");
        return AXT_EINVAL;
    }
    for (int k = 0; k < n_det; ++k)
        for (int64_t e = h_row_ptr[k]; e < h_row_ptr[k + 1]; ++e)
            if (h_col[e] <= k || h_col[e] >= n_det) {
                axt_set_error("axt_mcf_shard_begin: arc %lld of detection %d does not point forward in time", (long long)e, k);
                return AXT_EINVAL;
            }
    axt_mcf_shard *sh = new axt_mcf_shard;
    sh->rank = rank; sh->world = world;
    Lsap &a = sh->a;
    a.n = n_det;
    a.obs = h_obs; a.entry = h_entry; a.exitc = h_exit; a.row_ptr = h_row_ptr; a.cost = h_cost; a.col = h_col;
    *state_bytes = 0;
    if (n_det > 0) {
        a.prepare(world);
        sh->group = a.leaves >= world ? a.leaves / world : 0;
        if (sh->group > 0) {
            a.solve_tree(rank * sh->group, (rank + 1) * sh->group);           // this rank's run of leaves and the separators inside it
            *state_bytes = a.state_bytes(rank * sh->group, (rank + 1) * sh->group);
        }
    }
    *out = sh;
    return AXT_OK;
}

extern "C" int axt_mcf_shard_export(const axt_mcf_shard *sh, void *h_state)
{
    if (!sh || (sh->group > 0 && !h_state)) { axt_set_error("axt_mcf_shard_export: null argument"); return AXT_EINVAL; }
    if (sh->group > 0) sh->a.export_state(sh->rank * sh->group, (sh->rank + 1) * sh->group, static_cast<unsigned char *>(h_state));
    return AXT_OK;
}

static int shard_finish_impl(axt_mcf_shard *sh, const void *const *h_states, const int64_t *h_state_bytes, int min_flow,
                             int max_flow, int32_t *h_next, int32_t *h_track, int *n_tracks, int64_t *total_cost,
                             int64_t *h_pot_u, int64_t *h_pot_v, int64_t *pot_t)
{
    if (!sh || !n_tracks || !total_cost) { axt_set_error("axt_mcf_shard_finish: null argument"); return AXT_EINVAL; }
    Lsap &a = sh->a;
    if (a.n > 0) {
        if (sh->group > 0) {
            for (int r = 0; r < sh->world; ++r) {
                if (r == sh->rank) continue;
                if (!h_states || !h_states[r] || !h_state_bytes || h_state_bytes[r] != a.state_bytes(r * sh->group, (r + 1) * sh->group)) {
                    axt_set_error("axt_mcf_shard_finish: the state of rank %d is missing or has the wrong size", r);
                    return AXT_EINVAL;
                }
                if (!a.state_matches(static_cast<const unsigned char *>(h_states[r]))) {
                    axt_set_error("axt_mcf_shard_finish: rank %d built another time-block tree (different AXT_MCF_MIN_LEAF?): its state does not fit", r);
                    return AXT_EINVAL;
                }
                a.import_state(r * sh->group, (r + 1) * sh->group, static_cast<const unsigned char *>(h_states[r]));
            }
            a.solved_group = sh->group;
        }
        a.solve_tree(0, a.leaves);               // the separators above the ranks' subtrees (everything, if the network was not shared)
        a.solved_group = 0;
        a.second_phase();
    }
    return mcf_solve_impl(a.n, a.obs, a.entry, a.exitc, a.row_ptr, a.col, a.cost, min_flow, max_flow, h_next, h_track, n_tracks,
                          total_cost, a.n > 0 ? &a : nullptr, h_pot_u, h_pot_v, pot_t);
}

extern "C" int axt_mcf_shard_finish(axt_mcf_shard *sh, const void *const *h_states, const int64_t *h_state_bytes, int min_flow,
                                    int max_flow, int32_t *h_next, int32_t *h_track, int *n_tracks, int64_t *total_cost)
{
    return shard_finish_impl(sh, h_states, h_state_bytes, min_flow, max_flow, h_next, h_track, n_tracks, total_cost, nullptr, nullptr, nullptr);
}

extern "C" int axt_mcf_shard_finish_duals(axt_mcf_shard *sh, const void *const *h_states, const int64_t *h_state_bytes, int min_flow,
                                          int max_flow, int32_t *h_next, int32_t *h_track, int *n_tracks, int64_t *total_cost,
                                          int64_t *h_pot_u, int64_t *h_pot_v, int64_t *pot_t)
{
    if (sh && sh->a.n > 0 && (!h_pot_u || !h_pot_v || !pot_t)) { axt_set_error("axt_mcf_shard_finish_duals: null potential array"); return AXT_EINVAL; }
    return shard_finish_impl(sh, h_states, h_state_bytes, min_flow, max_flow, h_next, h_track, n_tracks, total_cost, h_pot_u, h_pot_v, pot_t);
}

extern "C" void axt_mcf_shard_free(axt_mcf_shard *sh) { delete sh; }
