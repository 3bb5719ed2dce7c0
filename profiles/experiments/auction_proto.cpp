// Host prototype (serial simulation of SYNCHRONOUS rounds) of an epsilon-scaling forward/reverse auction for the tracker's
// assignment form -- counts the rounds a GPU kernel would need. Not part of the library.
//   g++ -O2 -shared -fPIC -o /tmp/auction_proto.so profiles/experiments/auction_proto.cpp
//
// Rows i (out-slot of detection i) must all be assigned, to: X_i (private exit, cost own_i, price pinned at 0), the in-slot
// R_i of itself (cost 0: unused) or R_b of a successor (cost base_i + w). In-slot columns are optional and carry a price
// p_j >= 0; an unassigned column must end at price 0 (forward auction cannot lower prices: reverse steps do).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

typedef int64_t i64;
static const i64 INF = INT64_MAX / 4;

extern "C" int auction_solve(int n, const i64 *obs, const i64 *entry, const i64 *exitc, const i64 *row_ptr, const int32_t *col,
                             const i64 *cost, int32_t *out_col /* n: assigned column: i = unused, n+i = exit, else successor */,
                             i64 *out_price, int theta, int verbose, i64 eps_final_shift, i64 *stats /* [4] */)
{
    const i64 S = (i64)n + 1;
    std::vector<i64> own(n), base(n);
    std::vector<i64> w((size_t)row_ptr[n]);
    i64 cmax = 0;
    for (int i = 0; i < n; ++i) {
        base[i] = (obs[i] + entry[i]) * S;
        own[i] = (obs[i] + entry[i] + exitc[i]) * S;
        for (i64 e = row_ptr[i]; e < row_ptr[i + 1]; ++e) {
            w[e] = base[i] + (cost[e] - entry[col[e]]) * S;
            cmax = std::max<i64>(cmax, llabs(w[e]));
        }
        cmax = std::max<i64>(cmax, llabs(own[i]));
    }
    // in-arcs (CSC)
    std::vector<i64> in_ptr(n + 1, 0);
    for (i64 e = 0; e < row_ptr[n]; ++e) in_ptr[col[e] + 1]++;
    for (int j = 0; j < n; ++j) in_ptr[j + 1] += in_ptr[j];
    std::vector<int32_t> in_row((size_t)row_ptr[n]);
    std::vector<i64> in_w((size_t)row_ptr[n]);
    {
        std::vector<i64> fill(in_ptr.begin(), in_ptr.end() - 1);
        for (int i = 0; i < n; ++i)
            for (i64 e = row_ptr[i]; e < row_ptr[i + 1]; ++e) { const i64 q = fill[col[e]]++; in_row[q] = i; in_w[q] = w[e]; }
    }
    std::vector<i64> p(n, 0);
    std::vector<int32_t> rcol(n, -1), owner(n, -1);
    std::vector<i64> rval(n, 0);                 // value of the row's current assignment: w + p at the time... (kept exact: recomputed)
    auto value_of = [&](int i) -> i64 {          // current value of row i's assignment
        const int j = rcol[i];
        if (j == n + i) return own[i];
        if (j == i) return p[i];
        for (i64 e = row_ptr[i]; e < row_ptr[i + 1]; ++e) if (col[e] == j) return w[e] + p[j];
        return INF;
    };
    size_t rounds_f = 0, rounds_r = 0, bids = 0, offers = 0;
    std::vector<int32_t> active, next_active, bid_col(n);
    std::vector<i64> bid_val(n), best_bid(n);
    std::vector<int32_t> best_row(n);
    i64 eps = cmax / 2;
    if (eps < 1) eps = 1;
    int phase = 0;
    const i64 eps_final = eps_final_shift > 0 ? ((i64)1 << eps_final_shift) : 1;
    for (;;) {
        ++phase;
        // ---- phase start: empty assignment, prices kept
        std::fill(rcol.begin(), rcol.end(), -1);
        std::fill(owner.begin(), owner.end(), -1);
        active.resize(n);
        for (int i = 0; i < n; ++i) active[i] = i;
        size_t rf0 = rounds_f, rr0 = rounds_r, b0 = bids, o0 = offers;
        // ---- forward rounds
        std::vector<int32_t> touched_cols;
        while (!active.empty()) {
            ++rounds_f;
            touched_cols.clear();
            for (int32_t i : active) {
                i64 m1 = own[i], m2 = INF; int j1 = n + i;
                auto see = [&](int j, i64 v) { if (v < m1) { m2 = m1; m1 = v; j1 = j; } else if (v < m2) m2 = v; };
                see(i, p[i]);
                for (i64 e = row_ptr[i]; e < row_ptr[i + 1]; ++e) see(col[e], w[e] + p[col[e]]);
                ++bids;
                if (j1 == n + i) { rcol[i] = n + i; bid_col[i] = -1; continue; }
                bid_col[i] = j1;
                bid_val[i] = p[j1] + (m2 - m1) + eps;
            }
            for (int32_t i : active) {
                const int j = bid_col[i];
                if (j < 0) continue;
                if (std::find(touched_cols.begin(), touched_cols.end(), j) == touched_cols.end() && true) {}
            }
            // resolve: highest bid per column (ties: lowest row)
            next_active.clear();
            for (int32_t i : active) { const int j = bid_col[i]; if (j >= 0) { best_bid[j] = -1; best_row[j] = -1; } }
            for (int32_t i : active) {
                const int j = bid_col[i];
                if (j < 0) continue;
                if (bid_val[i] > best_bid[j] || (bid_val[i] == best_bid[j] && i < best_row[j])) { best_bid[j] = bid_val[i]; best_row[j] = i; }
            }
            for (int32_t i : active) {
                const int j = bid_col[i];
                if (j < 0) continue;
                if (best_row[j] != i) { next_active.push_back(i); continue; }
                if (owner[j] >= 0) { rcol[owner[j]] = -1; next_active.push_back(owner[j]); }
                owner[j] = i; rcol[i] = j; p[j] = best_bid[j];
            }
            active.swap(next_active);
        }
        // ---- reverse rounds: unassigned in-slot columns with a positive price
        std::vector<int32_t> cols, next_cols;
        for (int j = 0; j < n; ++j) if (owner[j] < 0 && p[j] > 0) cols.push_back(j);
        std::vector<i64> off_price(n), acc_val(n);
        std::vector<int32_t> off_row(n), acc_col(n, -1);
        while (!cols.empty()) {
            ++rounds_r;
            next_cols.clear();
            std::vector<int32_t> offered_rows;
            for (int32_t j : cols) {
                ++offers;
                i64 b1 = -INF, b2 = -INF; int i1 = -1;
                auto see = [&](int i, i64 wij) { const i64 b = value_of(i) - wij; if (b > b1) { b2 = b1; b1 = b; i1 = i; } else if (b > b2) b2 = b; };
                see(j, 0);
                for (i64 q = in_ptr[j]; q < in_ptr[j + 1]; ++q) see(in_row[q], in_w[q]);
                if (b1 <= eps) { p[j] = 0; off_row[j] = -1; continue; }
                i64 pn = b2 - eps; if (pn < 0) pn = 0;
                off_row[j] = i1; off_price[j] = pn;
            }
            for (int32_t j : cols) {            // rows accept the best offer
                const int i = off_row[j];
                if (i < 0) continue;
                const i64 wij = (i == j) ? 0 : [&]{ for (i64 e = row_ptr[i]; e < row_ptr[i + 1]; ++e) if (col[e] == j) return w[e]; return INF; }();
                const i64 v = wij + off_price[j];
                if (acc_col[i] < 0) offered_rows.push_back(i);
                if (acc_col[i] < 0 || v < acc_val[i]) { acc_col[i] = j; acc_val[i] = v; }
            }
            for (int32_t j : cols) {
                const int i = off_row[j];
                if (i < 0) continue;
                if (acc_col[i] != j) { next_cols.push_back(j); continue; }
            }
            for (int32_t i : offered_rows) {
                const int j = acc_col[i];
                acc_col[i] = -1;
                const int old = rcol[i];
                if (old >= 0 && old < n) { owner[old] = -1; if (p[old] > 0) next_cols.push_back(old); }
                rcol[i] = j; owner[j] = i; p[j] = off_price[j];
            }
            cols.swap(next_cols);
        }
        if (verbose) fprintf(stderr, "phase %2d eps 2^%.1f: forward rounds %zu (bids %zu), reverse rounds %zu (offers %zu)\n", phase, __builtin_log2((double)eps),
                             rounds_f - rf0, bids - b0, rounds_r - rr0, offers - o0);
        if (eps <= eps_final) break;
        eps /= theta;
        if (eps < eps_final) eps = eps_final;
    }
    for (int i = 0; i < n; ++i) out_col[i] = rcol[i];
    if (out_price) for (int j = 0; j < n; ++j) out_price[j] = p[j];
    stats[0] = rounds_f; stats[1] = rounds_r; stats[2] = bids; stats[3] = offers;
    return phase;
}
