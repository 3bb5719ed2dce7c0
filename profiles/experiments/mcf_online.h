// Two-sided, time-ordered assignment solver for the tracker's network (host C++; included by mcf.cpp only).
//
// Same optimum as Lsap (mcf.cpp) -- the problem is the same linear program and its optimum is unique -- reached in another
// order. Lsap inserts the ROWS (out-slots of the detections) in a shuffled order while every column (in-slot) is there from
// the start at price 0; rows grab free columns cheaply, prices end as shallow as feasibility allows, and the rows that come
// late have to flood (profiles/experiments/README_mcf_round4.md: one search settles 244 k of 319 k rows; with the DEEPEST
// feasible prices the same insertions scan 1.6 rows each). Here BOTH sides are inserted, frame by frame:
//     for every frame t:   the in-slots of frame t  (each looks BACK at the out-slots of frames t-1, t-2),
//                          then the out-slots of frame t (their successors do not exist yet: each takes its private exit,
//                          or its own in-slot = "unused").
// An out-slot at the front of the sweep therefore waits on its exit with its dual at the exit's cost -- the highest dual it
// can ever have -- and an in-slot that arrives finds the rows that want it one hop away, all indifferent to letting go of
// their exits: prices are deep by construction and searches end after a hop or two.
//
// Linear program (symmetric): every out-slot i is matched to the in-slot of a successor (cost obs_i + transition), to its
// own in-slot (0: the detection is unused) or to its private exit X_i (obs_i + exit_i); every in-slot j to the out-slot of
// a predecessor, to its own out-slot, or to its private entry E_j (entry_j). X and E are optional partners (dual <= 0, 0
// when unused). Either side's vertices are inserted by the same shortest-augmenting-path step (Search::insert<SIDE>).
// Blocks of frames are swept concurrently with the arcs between blocks hidden; a join then shows the hidden arcs of one
// boundary and re-inserts the few in-slots (and the out-slots they were matched to) whose dual no longer respects them.
#pragma once

struct Online {
    int n = 0;
    const int64_t *obs, *entry, *exitc, *row_ptr, *cost;
    const int32_t *col;
    static constexpr int32_t ABSENT = -2, PRIVATE = -1;
    // side 0 = out-slots (rows), side 1 = in-slots (columns)
    std::vector<int64_t> dual[2];
    std::vector<int32_t> match[2];                    // partner on the other side, PRIVATE (exit / entry), ABSENT (not inserted yet)
    std::vector<int64_t> in_ptr, in_cost;             // CSC of the transition arcs
    std::vector<int32_t> in_row, in_arc;
    std::vector<int32_t> block;                       // per detection: its block of frames
    std::vector<char> opened;                         // per IN-SLOT: the arcs into it from the block before its own are visible
    std::vector<int32_t> blk_lo;                      // first detection of every block (+ n at the end)
    std::vector<int32_t> blk_far;                     // per block b: the last in-slot (of block b + 1) its out-slots reach
    // search state per vertex and side (labels of the OTHER side's vertices are keyed by their own index)
    struct Mark { int64_t d; int32_t pred; uint32_t stamp; };
    std::vector<Mark> mark[2];
    std::atomic<uint32_t> next_stamp{2};
    size_t stat_scanned = 0, stat_relax = 0, stat_searches = 0, stat_repairs = 0;
    std::mutex stat_lock;
    int budget = 1;

    inline int64_t priv_cost(int side, int x) const { return side == 0 ? obs[x] + exitc[x] : entry[x]; }
    inline bool visible(int i, int j) const { return block[i] == block[j] || opened[j]; }

    typedef std::pair<int64_t, int32_t> Item;
    struct Search {
        Online &P;
        std::vector<Item> heap;
        std::vector<int32_t> scanned_same, closed_other;   // vertices of the root's side that were expanded / of the other side that were settled
        size_t scanned = 0, relax = 0, searches = 0;
        explicit Search(Online &p) : P(p) {}
        ~Search()
        {
            std::lock_guard<std::mutex> g(P.stat_lock);
            P.stat_scanned += scanned; P.stat_relax += relax; P.stat_searches += searches;
        }
        void push(Item it)
        {
            size_t k = heap.size();
            heap.push_back(it);
            while (k > 0) {
                const size_t p = (k - 1) >> 2;
                if (heap[p].first <= it.first) break;
                heap[k] = heap[p];
                k = p;
            }
            heap[k] = it;
        }
        Item pop()
        {
            const Item top = heap[0], last = heap.back();
            heap.pop_back();
            const size_t m = heap.size();
            if (m) {
                size_t k = 0;
                for (;;) {
                    const size_t c0 = 4 * k + 1;
                    if (c0 >= m) break;
                    size_t best = c0;
                    const size_t ce = c0 + 4 < m ? c0 + 4 : m;
                    for (size_t q = c0 + 1; q < ce; ++q)
                        if (heap[q].first < heap[best].first) best = q;
                    if (last.first <= heap[best].first) break;
                    heap[k] = heap[best];
                    k = best;
                }
                heap[k] = last;
            }
            return top;
        }

        // Insert vertex x of side S (0: out-slot, 1: in-slot), which is ABSENT or unmatched: one shortest augmenting path on
        // reduced costs from x to a vertex that can let go of an optional partner -- a vertex of the other side that sits on
        // its private partner, or a vertex of x's side on the path that takes its own private partner.
        template <int S>
        void insert(int x)
        {
            constexpr int O = 1 - S;
            const int n = P.n;
            std::vector<int64_t> &dS = P.dual[S], &dO = P.dual[O];
            std::vector<int32_t> &mS = P.match[S], &mO = P.match[O];
            std::vector<Mark> &mk = P.mark[O];            // labels of the other side's vertices; index n + z = the private partner of z (side S)
            const uint32_t open = P.next_stamp.fetch_add(2, std::memory_order_relaxed), closed = open + 1;
            ++searches;
            heap.clear();
            scanned_same.clear();
            closed_other.clear();
            mS[x] = PRIVATE;                              // (present from now on; overwritten below)
            dS[x] = 0;
            int cur = x;
            int64_t minVal = 0;
            int32_t sink = -1;                            // y < n: other-side vertex y that sits on its private partner; n + z: z takes its private partner
            // private partners' labels live in a small side table (a vertex of side S is expanded at most once per search)
            while (sink < 0) {
                scanned_same.push_back(cur);
                ++scanned;
                const int64_t off = minVal - dS[cur];
                auto relax = [&](int y, int64_t w, int32_t pred) {
                    ++relax_count();
                    Mark &m = mk[y];
                    if (m.stamp == closed) return;
                    const int64_t r = off + w - (y < n ? dO[y] : 0);
                    if (m.stamp != open || r < m.d) {
                        m.stamp = open; m.d = r; m.pred = pred;
                        push(Item(r, y));
                    }
                };
                relax(n + cur, P.priv_cost(S, cur), cur);                        // cur's own private partner
                if (mO[cur] != ABSENT) relax(cur, 0, cur);                       // the other slot of the same detection ("unused")
                if (S == 0) {
                    for (int64_t e = P.row_ptr[cur]; e < P.row_ptr[cur + 1]; ++e) {
                        const int j = P.col[e];
                        if (mO[j] != ABSENT && P.visible(cur, j)) relax(j, P.obs[cur] + P.cost[e], cur);
                    }
                } else {
                    for (int64_t q = P.in_ptr[cur]; q < P.in_ptr[cur + 1]; ++q) {
                        const int i = P.in_row[q];
                        if (mO[i] != ABSENT && P.visible(i, cur)) relax(i, P.obs[i] + P.in_cost[q], cur);
                    }
                }
                int y = -1;
                while (!heap.empty()) {
                    const Item it = pop();
                    const Mark &m = mk[it.second];
                    if (m.stamp != open || it.first > m.d) continue;
                    y = it.second;
                    minVal = it.first;
                    break;
                }
                mk[y].stamp = closed;                     // (cur's private partner is always reachable: y >= 0)
                closed_other.push_back(y);
                if (y >= n || mO[y] == PRIVATE) sink = y;
                else cur = mO[y];
            }
            static const int wr = getenv("AXT_WATCH_ROW") ? atoi(getenv("AXT_WATCH_ROW")) : -1, wc = getenv("AXT_WATCH_COL") ? atoi(getenv("AXT_WATCH_COL")) : -1;
            const int64_t w_before_r = wr >= 0 ? P.dual[0][wr] : 0, w_before_c = wc >= 0 ? P.dual[1][wc] : 0;
            // dual update (Jonker-Volgenant / Crouse, both sides alike)
            dS[x] += minVal;
            for (size_t k = 1; k < scanned_same.size(); ++k) {
                const int z = scanned_same[k];
                dS[z] += minVal - mk[mS[z]].d;            // z was reached through its partner mS[z]
            }
            for (int32_t y : closed_other)
                if (y < n) dO[y] -= minVal - mk[y].d;
            if (wr >= 0 && (P.dual[0][wr] != w_before_r || P.dual[1][wc] != w_before_c))
                fprintf(stderr, "  search S=%d root %d (scanned %zu, minVal %lld): row %d dual %lld -> %lld, col %d dual %lld -> %lld\n", S, x, scanned_same.size(), (long long)minVal,
                        wr, (long long)w_before_r, (long long)P.dual[0][wr], wc, (long long)w_before_c, (long long)P.dual[1][wc]);
            if (const char *hp_ = getenv("AXT_MCF_ONLINE_HIST")) {
                static std::atomic<size_t> hist[2][8];
                static std::atomic<size_t> total{0};
                int bkt = 0; for (size_t v = scanned_same.size(); v > 1 && bkt < 7; v >>= 2) ++bkt;
                hist[S][bkt]++;
                if (++total == (size_t)atol(hp_)) for (int sd = 0; sd < 2; ++sd) { fprintf(stderr, "side %d:", sd); for (int b = 0; b < 8; ++b) fprintf(stderr, " <%d:%zu", 1 << (2 * b + 1), hist[sd][b].load()); fprintf(stderr, "\n"); }
            }
            // augment
            int y = sink;
            for (;;) {
                const int z = mk[y].pred;                 // the vertex of side S that takes y
                const int prev = mS[z];
                if (y >= n) mS[z] = PRIVATE;
                else { mS[z] = y; mO[y] = z; }
                if (z == x) break;
                y = prev;
            }
        }
        size_t &relax_count() { return relax; }
    };

    void build_in()
    {
        in_ptr.assign((size_t)n + 1, 0);
        for (int64_t e = 0; e < row_ptr[n]; ++e) in_ptr[col[e] + 1]++;
        for (int j = 0; j < n; ++j) in_ptr[j + 1] += in_ptr[j];
        in_row.resize((size_t)row_ptr[n]);
        in_cost.resize((size_t)row_ptr[n]);
        in_arc.resize((size_t)row_ptr[n]);
        std::vector<int64_t> fill(in_ptr.begin(), in_ptr.end() - 1);
        for (int k = 0; k < n; ++k)
            for (int64_t e = row_ptr[k]; e < row_ptr[k + 1]; ++e) {
                const int64_t q = fill[col[e]]++;
                in_row[q] = k; in_cost[q] = cost[e]; in_arc[q] = (int32_t)e;
            }
    }

    // Frames are not given: a "frame" boundary is where the successors of a row begin. Rows are numbered in frame order and
    // reach at most two frames ahead; the sweep only needs an order in which every in-slot comes after the out-slots that
    // reach it and before its own out-slot -- the detection order itself: in-slot k, then out-slot k.
    void sweep(int lo, int hi)
    {
        Search w(*this);
        for (int k = lo; k < hi; ++k) {
            w.insert<1>(k);
            w.insert<0>(k);
        }
    }

    // the arcs from block b into block b + 1 become visible: every in-slot of block b + 1 that such an arc reaches and whose
    // dual no longer respects it is taken out (with the out-slot it was matched to) and inserted again
    void join(int b, Search &w)
    {
        // one in-slot at a time: the arcs into the in-slots that have not had their turn stay hidden, so that every search runs
        // on non-negative reduced costs
        const int lo = blk_lo[b + 1], hi = std::min<int>(blk_lo[b + 2], blk_far[b] + 1);
        for (int j = lo; j < hi; ++j) {
            opened[j] = 1;
            bool violated = false;
            for (int64_t q = in_ptr[j]; q < in_ptr[j + 1] && !violated; ++q) {
                const int i = in_row[q];
                violated = block[i] == b && dual[0][i] + dual[1][j] > obs[i] + in_cost[q];
            }
            if (!violated) continue;
            ++stat_repairs;
            const int p = match[1][j];
            match[1][j] = ABSENT;
            if (p >= 0) match[0][p] = ABSENT;
            w.insert<1>(j);
            static const bool dbg = getenv("AXT_MCF_ONLINE_CHECK") != nullptr;

            if (p >= 0 && match[0][p] == ABSENT) w.insert<0>(p);
            if (dbg) fprintf(stderr, "  repaired col %d (was with %d): now with %d; row %d now with %d\n", j, p, match[1][j], p, p >= 0 ? match[0][p] : -9);
        }
    }

    // debugging aid: the first arc (between inserted vertices, visible) whose reduced cost is negative or whose matched pair is not tight
    long check(const char *when) const
    {
        long bad = 0;
        for (int i = 0; i < n; ++i) {
            if (match[0][i] == ABSENT) continue;
            if (dual[0][i] > obs[i] + exitc[i] && match[0][i] != PRIVATE) { if (!bad++) fprintf(stderr, "[%s] row %d above its exit\n", when, i); }
            if (match[1][i] != ABSENT && dual[0][i] + dual[1][i] > 0) { if (!bad++) fprintf(stderr, "[%s] own slots of %d: %lld\n", when, i, (long long)(dual[0][i] + dual[1][i])); }
            for (int64_t e = row_ptr[i]; e < row_ptr[i + 1]; ++e) {
                const int j = col[e];
                if (match[1][j] == ABSENT || !visible(i, j)) continue;
                const int64_t rc = obs[i] + cost[e] - dual[0][i] - dual[1][j];
                if (rc < 0 || (match[0][i] == j && rc != 0)) { if (!bad++) fprintf(stderr, "[%s] arc %d -> %d rc %lld (match of %d: %d, of col %d: %d)\n", when, i, j, (long long)rc, i, match[0][i], j, match[1][j]); }
            }
        }
        for (int j = 0; j < n; ++j)
            if (match[1][j] != ABSENT && match[1][j] != PRIVATE && dual[1][j] > entry[j]) { if (!bad++) fprintf(stderr, "[%s] col %d above its entry\n", when, j); }
        return bad;
    }

    void run(int threads, int min_block)
    {
        budget = threads;
        for (int s = 0; s < 2; ++s) {
            dual[s].assign(n, 0);
            match[s].assign(n, ABSENT);
            mark[s].assign(2 * (size_t)n, Mark{0, -1, 0});
        }
        build_in();
        // blocks: cut where no arc crosses more than into the next block; a block must hold at least the reach of its rows
        int blocks = 1;
        while (blocks * 2 <= threads && n / (blocks * 2) >= min_block) blocks *= 2;
        std::vector<int32_t> reach(n);
        for (int k = 0; k < n; ++k) {
            int32_t far = k;
            for (int64_t e = row_ptr[k]; e < row_ptr[k + 1]; ++e) far = std::max(far, col[e]);
            reach[k] = far;
        }
        for (; blocks > 1; blocks /= 2) {
            blk_lo.assign(blocks + 1, 0);
            blk_lo[blocks] = n;
            for (int b = 1; b < blocks; ++b) blk_lo[b] = (int32_t)((int64_t)n * b / blocks);
            bool ok = true;
            for (int b = 0; b + 1 < blocks && ok; ++b) {           // rows of block b must not reach beyond block b + 1
                int32_t far = 0;
                for (int k = blk_lo[b]; k < blk_lo[b + 1]; ++k) far = std::max(far, reach[k]);
                ok = far < blk_lo[b + 2];
            }
            if (ok) break;
        }
        if (blocks <= 1) { blocks = 1; blk_lo.assign(2, 0); blk_lo[1] = n; }
        block.resize(n);
        blk_far.assign(blocks, 0);
        for (int b = 0; b < blocks; ++b)
            for (int k = blk_lo[b]; k < blk_lo[b + 1]; ++k) { block[k] = b; blk_far[b] = std::max(blk_far[b], reach[k]); }
        opened.assign(n, 0);
        std::vector<std::thread> pool;
        if (getenv("AXT_MCF_ONLINE_SERIAL")) { for (int b = 1; b < blocks; ++b) sweep(blk_lo[b], blk_lo[b + 1]); }
        else
        for (int b = 1; b < blocks; ++b) {
            try { pool.emplace_back([this, b] { sweep(blk_lo[b], blk_lo[b + 1]); }); } catch (...) { sweep(blk_lo[b], blk_lo[b + 1]); }
        }
        sweep(blk_lo[0], blk_lo[1]);
        for (std::thread &t : pool) t.join();
        Search w(*this);
        if (getenv("AXT_MCF_ONLINE_CHECK") && check("after the sweeps")) exit(3);
        for (int b = 0; b + 1 < blocks; ++b) join(b, w);
        if (getenv("AXT_MCF_ONLINE_CHECK")) fprintf(stderr, "final check: %ld violations\n", check("final"));
    }
};
