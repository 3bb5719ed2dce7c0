// Frame-to-frame Hungarian association on gfx950 (BASELINE config 3: "detection + cost-matrix + Hungarian
// association"). A build-side variant: the reference itself only ever runs the global min-cost-flow tracker
// (AxonDetections.py:663-690), whose cost model this variant reuses unchanged:
//   * cost of linking detection a (frame t) to b (frame t+g) = transition_model(D(a,b), g)
//     (mincostflow_models.py:67-119), admitted when < MCF_EDGE_COST_THR, i.e. D <= dmax[g-1];
//   * leaving a detection without a successor costs the threshold itself, so any admitted link is preferred.
// Pass 1 solves, independently for every frame pair (t, t+1), the rectangular assignment
//     min sum_{matched} c(a,b) + sum_{unmatched rows} U(a)
// exactly (Jonker-Volgenant shortest augmenting paths on integer costs); pass 2 does the same for gap 2 between
// detections of t that found no successor in t+1 and detections of t+2 that found no predecessor in t+1
// (MCF_MAX_NUM_MISSES = 1). A final sweep numbers the chains in (first frame, index) order.
//
// One wavefront per frame pair: each lane owns columns j = lane, lane+64, ...; the cost matrix is never stored --
// c(a,b) is recomputed from the anchors (closed-form path length, table look-up, identity hash), which is cheaper
// than reading 8 bytes. Costs are the same 64-bit integers as the flow network's (axt_arc_cost_int), so the
// optimum is unique and any exact LSAP solver (e.g. SciPy's linear_sum_assignment) returns the same matching.
#include "axt_common.h"

#include <type_traits>

namespace {

constexpr long HINF = 0x3fffffffffffffffL;

__device__ __forceinline__ long h_arc_cost_int(long units, int kind, long a, long b)
{
    unsigned long x = ((unsigned long)kind << 60) ^ ((unsigned long)a << 30) ^ (unsigned long)b;
    x += 0x9E3779B97F4A7C15ul;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ul;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBul;
    x ^= x >> 31;
    return units * 65536 + (long)(x & 0xFFFFul);
}

__device__ __forceinline__ int h_path_len_open(int xa, int ya, int xb, int yb, int H, int W, int max_dist, int conn8)
{
    const int dx = abs(xa - xb), dy = abs(ya - yb);
    const long d2 = (long)dx * dx + (long)dy * dy;
    const int len = (conn8 ? max(dx, dy) : dx + dy) + 1;
    const bool inb = xa >= 0 && xa < W && ya >= 0 && ya < H && xb >= 0 && xb < W && yb >= 0 && yb < H;
    return (d2 < (long)max_dist * max_dist && len <= max_dist && inb) ? len : max_dist;
}

// wave-wide argmin of (key, idx): smallest key, ties to the smallest idx. Key (< 2^52) and index (< 2^11) travel as
// one 64-bit word; inside each row of 16 lanes the minimum is formed with DPP moves (quad swaps, half-row and row
// mirrors: 2-cycle VALU operations instead of ~100-cycle LDS-crossbar shuffles), the four row minima meet through
// v_readlane.
__device__ __forceinline__ unsigned long dpp_min_step(unsigned long v, unsigned long o) { return o < v ? o : v; }
#define AXT_DPP_MIN(v, ctrl)                                                                              \
    {                                                                                                     \
        const unsigned lo_ = (unsigned)(v), hi_ = (unsigned)((v) >> 32);                                  \
        const unsigned ol_ = __builtin_amdgcn_update_dpp(lo_, lo_, ctrl, 0xF, 0xF, false);                \
        const unsigned oh_ = __builtin_amdgcn_update_dpp(hi_, hi_, ctrl, 0xF, 0xF, false);                \
        (v) = dpp_min_step(v, ((unsigned long)oh_ << 32) | ol_);                                          \
    }
__device__ __forceinline__ void wave_argmin(long &key, int &idx)
{
    unsigned long v = key >= (1l << 51) ? ~0ul : (((unsigned long)key << 11) | (unsigned)idx);   // keys are >= 0
    AXT_DPP_MIN(v, 0xB1);        // quad_perm [1,0,3,2]
    AXT_DPP_MIN(v, 0x4E);        // quad_perm [2,3,0,1]
    AXT_DPP_MIN(v, 0x141);       // row_half_mirror
    AXT_DPP_MIN(v, 0x140);       // row_mirror: every lane of a row now holds the row's minimum
    unsigned long best = ~0ul;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, r * 16), hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), r * 16);
        best = dpp_min_step(best, ((unsigned long)hi << 32) | lo);
    }
    if (best == ~0ul) { key = HINF; idx = 0x7fffffff; }
    else { key = (long)(best >> 11); idx = (int)(best & 2047u); }
}

// One wave per frame pair (t, t+gap). succ/pred are per detection slot [n_frames*cap]:
//   GAP == 1: writes succ1[t*cap+i] = j or -1 and pred1[(t+1)*cap+j] = i or -1 for every slot of the pair.
//   GAP == 2: rows = detections of t with succ1 < 0, columns = detections of t+2 with pred1 < 0;
//             writes succ2 / pred2 the same way.
// NC > 0 (3 or 9): frames with at most 64*NC detection slots -- the per-column search state (dual, distance, predecessor, matched
// row, flags) lives in NC registers per lane instead of LDS, so that one search step costs one LDS round trip (the
// cost row and the row dual) instead of a dozen dependent ones. NC == 0: any cap, column state in LDS.
// NC > 0: launched with 256 threads. All four waves share the initialisation (every row's cheapest option: n x m link
// costs at ~1 000 cycles each -- the 64-bit identity hash -- were 40 % of the kernel on one wave), then waves 1-3 leave
// and wave 0 runs the searches alone.
// Diagnostic build only (-DAXT_HUNG_STATS, profiles/hungarian_stats.py): per frame pair of the last gap-1 launch the
// s_memtime ticks of the initialisation and of the searches, the number of searches and of search steps, n and m.
#ifdef AXT_HUNG_STATS
__device__ unsigned long long g_hung_stats[4096 * 8];
#define HUNG_STAMP() __builtin_amdgcn_s_memtime()
#endif

template <int GAP, int NC>
__global__ __launch_bounds__(NC > 0 ? 256 : 64) void hungarian_pair_kernel(
    const int *__restrict__ x, const int *__restrict__ y, const int *__restrict__ count,
    const int *__restrict__ frame_off, int n_frames, int cap, int H, int W, int max_dist, int conn8,
    int dmax, const long *__restrict__ units, long thr_units,
    const int *__restrict__ succ1, const int *__restrict__ pred1,
    int *__restrict__ succ_out, int *__restrict__ pred_out, int cdim, int t_first,
    const short *__restrict__ dtab, int tab_gaps, const long *__restrict__ ctab)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
    const int t = t_first + blockIdx.x, tb = t + GAP, lane = threadIdx.x, nthr = NC > 0 ? 256 : 64;
    if (tb >= n_frames) return;
    const int n = min(count[t], cap), m = min(count[tb], cap);
    long *v = reinterpret_cast<long *>(hsm);            // [cap] column duals
    long *spc = v + cap;                                // [cap]
    long *u = spc + cap;                                // [cap] row duals
    int *row4col = reinterpret_cast<int *>(u + cap);    // [cap]
    int *col4row = row4col + cap;                       // [cap]  (-1 unassigned, -2 own dummy)
    int *pred = col4row + cap;                          // [cap]
    int *sr = pred + cap;                               // [cap] rows scanned in this search
    int *xs = sr + cap, *ys = xs + cap;                 // [cap] column anchors
    int *xr = ys + cap, *yr = xr + cap;                 // [cap] row anchors
    long *lunits = reinterpret_cast<long *>(yr + cap + (cap & 1));            // [dmax+1] cost table of this gap
    long *ccache = lunits + dmax + 1;                   // [cdim][cdim] cost matrix of the pair, if it fits
    unsigned char *in_sc = reinterpret_cast<unsigned char *>(ccache + (long)cdim * cdim);   // [cap]
    unsigned char *col_ok = in_sc + cap;                // [cap] column takes part
    long *dummy = reinterpret_cast<long *>(col_ok + cap + ((8 - (2 * cap) % 8) % 8));     // [cap] cost of leaving a row unlinked
    long *pbest = dummy + cap;                          // [4 cap] (NC > 0) the initialisation's partial minima ...
    int *pbj = reinterpret_cast<int *>(pbest + 4 * cap);    // [4 cap] ... and their columns
    const bool cached = n <= cdim && m <= cdim;

#ifdef AXT_HUNG_STATS
    const unsigned long long st0 = HUNG_STAMP();
    unsigned long long st1 = 0, n_search = 0, n_step = 0;
#endif
    const long a0 = frame_off[t], b0 = frame_off[tb];
    for (int j = lane; j < m; j += nthr) {
        v[j] = 0;
        row4col[j] = -1;
        xs[j] = x[(long)tb * cap + j];
        ys[j] = y[(long)tb * cap + j];
        col_ok[j] = (GAP == 1) ? 1 : (pred1[(long)tb * cap + j] < 0);
    }
    for (int i = lane; i < n; i += nthr) {
        u[i] = 0;
        col4row[i] = -1;
        xr[i] = x[(long)t * cap + i];
        yr[i] = y[(long)t * cap + i];
    }
    for (int d = lane; d <= dmax; d += nthr) lunits[d] = units[d];
    __syncthreads();

    // cost of linking row i to column j (HINF: not admitted)
    auto link_cost = [&](int i, int j) -> long {
        if (!col_ok[j]) return HINF;
        if (ctab)                            // link costs given per pair (the appearance term: axt_hungarian_pairs_costs)
            return ctab[(((long)t * cap + i) * tab_gaps + (GAP - 1)) * cap + j];
        int d;
        if (dtab) {                          // masked grid: path lengths from the arc builder's searches (<= 0: none)
            d = dtab[(((long)t * cap + i) * tab_gaps + (GAP - 1)) * cap + j];
            if (d <= 0) return HINF;
        } else {
            d = h_path_len_open(xr[i], yr[i], xs[j], ys[j], H, W, max_dist, conn8);
        }
        return d <= dmax ? h_arc_cost_int(lunits[d], 3, a0 + i, b0 + j) : HINF;
    };

    // ---- initialisation (Jonker-Volgenant): every row gets u = its cheapest option (a column or its own dummy),
    // which keeps all reduced costs >= 0 with v = 0, and is assigned to that option if it is still free (tight
    // pair). Only rows that lose such a contest need an augmenting search. Lanes work on different rows here.
    if constexpr (NC > 0) {
        // (row, quarter of the columns) per thread; the four partial minima of a row are combined in column order, so the
        // result is the sequential scan's (first column among equal costs; the dummy wins a tie with a column)
        for (int idx = lane; idx < 4 * n; idx += nthr) {
            const int i = idx >> 2, qq = idx & 3;
            const bool active = (GAP == 1) || (succ1[(long)t * cap + i] < 0);
            long best = HINF;
            int bj = -1;
            if (active) {
                const int j1 = (int)(((long)m * (qq + 1)) >> 2);
                for (int j = (int)(((long)m * qq) >> 2); j < j1; ++j) {
                    const long c = link_cost(i, j);
                    if (cached) ccache[i * cdim + j] = c;
                    if (c < best) { best = c; bj = j; }
                }
            }
            pbest[idx] = best;
            pbj[idx] = bj;
            if (qq == 0) dummy[i] = h_arc_cost_int(thr_units, 1, a0 + i, 0);
        }
        __syncthreads();
        for (int i = lane; i < n; i += nthr) {
            long best = dummy[i];
            int bj = -2;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
                if (pbest[4 * i + qq] < best) { best = pbest[4 * i + qq]; bj = pbj[4 * i + qq]; }
            u[i] = best;
            pred[i] = bj;                // pred[] is free until the first search: holds the row's preferred column
        }
        __syncthreads();
        if (lane >= 64) return;          // the searches are one wave's work (a barrier no longer counts a wave that has ended)
    } else {
        for (int i = lane; i < n; i += 64) {
            const bool active = (GAP == 1) || (succ1[(long)t * cap + i] < 0);
            long best = dummy[i] = h_arc_cost_int(thr_units, 1, a0 + i, 0);
            int bj = -2;
            if (active) {
                for (int j = 0; j < m; ++j) {
                    const long c = link_cost(i, j);
                    if (cached) ccache[i * cdim + j] = c;
                    if (c < best) { best = c; bj = j; }
                }
            }
            u[i] = best;
            pred[i] = bj;                // pred[] is free until the first search: holds the row's preferred column
        }
        __syncthreads();
    }
    if (lane == 0) {
        for (int i = 0; i < n; ++i) {
            if (GAP == 2 && succ1[(long)t * cap + i] >= 0) continue;
            const int bj = pred[i];
            if (bj == -2) col4row[i] = -2;
            else if (row4col[bj] < 0) { row4col[bj] = i; col4row[i] = bj; }
        }
    }
    __syncthreads();

#ifdef AXT_HUNG_STATS
    st1 = HUNG_STAMP();
#endif
    if constexpr (NC > 0) {
        // ---- register-resident column state: column j = lane + 64*k lives in slot k of lane j % 64. The number of slots a search
        // step walks is chosen per pair from its column count (block-uniform): a pair of 75 detections under cap = 144 pays for
        // two slots, not three.
        auto resident = [&](auto ns_tag) {
        constexpr int NS = decltype(ns_tag)::value;
        long v_r[NS], spc_r[NS];
        int pred_r[NS], rc_r[NS];            // predecessor row in the search / matched row
        bool ok_r[NS], sc_r[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int j = lane + 64 * k;
            v_r[k] = 0;
            ok_r[k] = j < m && col_ok[j];
            rc_r[k] = j < m ? row4col[j] : -1;
        }
        // The searches, once per way of getting a link cost: from the pair's cost matrix in LDS (the usual case) or computed
        // (pairs too large for it). Two copies of the loop rather than a choice per column: with the choice inside, hipcc
        // branches around every column slot and waits for each slot's LDS read on its own -- ~400 instructions and three
        // LDS round trips per search step, 1 700 cycles (profiles/hungarian_stats.py).
        auto searches = [&](auto cached_tag) {
        constexpr bool CACHED = decltype(cached_tag)::value;
        for (int i = 0; i < n; ++i) {
            if (GAP == 2 && succ1[(long)t * cap + i] >= 0) continue;      // wave-uniform
            if (col4row[i] != -1) continue;                               // settled by the initialisation
#pragma unroll
            for (int k = 0; k < NS; ++k) { spc_r[k] = HINF; sc_r[k] = false; pred_r[k] = -1; }
            long minVal = 0, best_dummy = HINF;
            int cur = i, dummy_row = -1, n_sr = 0, sink = -1;             // sink >= 0: real column; -2: dummy of dummy_row
#ifdef AXT_HUNG_STATS
            ++n_search;
#endif
            for (;;) {
#ifdef AXT_HUNG_STATS
                ++n_step;
#endif
                if (lane == 0) sr[n_sr] = cur;
                ++n_sr;
                const long ucur = u[cur];
                const long rd = minVal + dummy[cur] - ucur;
                if (rd < best_dummy) { best_dummy = rd; dummy_row = cur; }
                long bkey = HINF;
                int bidx = 0x7fffffff;
                // branch-free over the lane's columns: the cost reads go out together, everything else is selects
                long cst[NS];
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const int j = lane + 64 * k;
                    if constexpr (CACHED) {
                        const long c = ccache[cur * cdim + min(j, cdim - 1)];     // columns >= m: not ok, whatever is read
                        cst[k] = ok_r[k] ? c : HINF;
                    } else {
                        cst[k] = !ok_r[k] ? HINF : link_cost(cur, j);
                    }
                }
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const bool open = ok_r[k] && !sc_r[k];
                    const long r = minVal + cst[k] - ucur - v_r[k];
                    const bool upd = open && cst[k] != HINF && r < spc_r[k];
                    spc_r[k] = upd ? r : spc_r[k];
                    pred_r[k] = upd ? cur : pred_r[k];
                    const long cand = open ? spc_r[k] : HINF;
                    const bool better = cand < bkey;
                    bkey = better ? cand : bkey;
                    bidx = better ? lane + 64 * k : bidx;
                }
                wave_argmin(bkey, bidx);
                if (best_dummy <= bkey) { sink = -2; minVal = best_dummy; break; }
                minVal = bkey;
                // the winning column (wave-uniform index) joins the tree; its owner tells everyone the matched row
                int matched = -1;
#pragma unroll
                for (int k = 0; k < NS; ++k)
                    if ((bidx >> 6) == k) {
                        if (lane == (bidx & 63)) sc_r[k] = true;
                        matched = __builtin_amdgcn_readlane(rc_r[k], bidx & 63);
                    }
                if (matched < 0) { sink = bidx; break; }
                cur = matched;
            }
            // publish what the row update and the augmentation read: distances and predecessors of the tree's columns
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int j = lane + 64 * k;
                if (j < m && sc_r[k]) { spc[j] = spc_r[k]; pred[j] = pred_r[k]; }
                if (j < m && sc_r[k] && j != sink) v_r[k] -= minVal - spc_r[k];      // dual update of the columns
            }
            __syncthreads();
            // dual update (Crouse 2016, Alg. 1): rows of SR
            for (int k = lane; k < n_sr; k += 64) {
                const int r = sr[k];
                u[r] += (k == 0) ? minVal : minVal - spc[col4row[r]];
            }
            __syncthreads();
            // augment back to row i
            if (lane == 0) {
                int r, jnew;
                if (sink == -2) { r = dummy_row; jnew = -2; }
                else { r = pred[sink]; jnew = sink; }
                for (;;) {
                    const int jprev = col4row[r];
                    col4row[r] = jnew;
                    if (jnew >= 0) row4col[jnew] = r;
                    if (r == i) break;
                    jnew = jprev;
                    r = pred[jnew];
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int j = lane + 64 * k;
                if (j < m) rc_r[k] = row4col[j];
            }
        }
        };
        if (cached) searches(std::true_type{});
        else searches(std::false_type{});
        };
        if constexpr (NC == 3) {
            if (m <= 64) resident(std::integral_constant<int, 1>{});
            else if (m <= 128) resident(std::integral_constant<int, 2>{});
            else resident(std::integral_constant<int, 3>{});
        } else {
            resident(std::integral_constant<int, (NC > 0 ? NC : 1)>{});
        }
    } else {
    for (int i = 0; i < n; ++i) {
            if (GAP == 2 && succ1[(long)t * cap + i] >= 0) continue;      // wave-uniform
            if (col4row[i] != -1) continue;                               // settled by the initialisation
            for (int j = lane; j < m; j += 64) { spc[j] = HINF; in_sc[j] = 0; }
            __syncthreads();
            long minVal = 0, best_dummy = HINF;
            int cur = i, dummy_row = -1, n_sr = 0, sink = -1;             // sink >= 0: real column; -2: dummy of dummy_row
            for (;;) {
                if (lane == 0) sr[n_sr] = cur;
                ++n_sr;
                const long ucur = u[cur];
                const long rd = minVal + dummy[cur] - ucur;
                if (rd < best_dummy) { best_dummy = rd; dummy_row = cur; }
                long bkey = HINF;
                int bidx = 0x7fffffff;
                for (int j = lane; j < m; j += 64) {
                    if (in_sc[j] || !col_ok[j]) continue;
                    const long c = cached ? ccache[cur * cdim + j] : link_cost(cur, j);
                    long s = spc[j];
                    if (c != HINF) {
                        const long r = minVal + c - ucur - v[j];
                        if (r < s) { s = r; spc[j] = r; pred[j] = cur; }
                    }
                    if (s < bkey) { bkey = s; bidx = j; }
                }
                wave_argmin(bkey, bidx);
                if (best_dummy <= bkey) { sink = -2; minVal = best_dummy; break; }
                minVal = bkey;
                if (lane == 0) in_sc[bidx] = 1;
                __syncthreads();
                if (row4col[bidx] < 0) { sink = bidx; break; }
                cur = row4col[bidx];
            }
            __syncthreads();
            // dual update (Crouse 2016, Alg. 1): rows of SR, columns of SC
            for (int k = lane; k < n_sr; k += 64) {
                const int r = sr[k];
                u[r] += (k == 0) ? minVal : minVal - spc[col4row[r]];
            }
            for (int j = lane; j < m; j += 64)
                if (in_sc[j] && j != sink) v[j] -= minVal - spc[j];
            __syncthreads();
            // augment back to row i
            if (lane == 0) {
                int r, jnew;
                if (sink == -2) { r = dummy_row; jnew = -2; }
                else { r = pred[sink]; jnew = sink; }
                for (;;) {
                    const int jprev = col4row[r];
                    col4row[r] = jnew;
                    if (jnew >= 0) row4col[jnew] = r;
                    if (r == i) break;
                    jnew = jprev;
                    r = pred[jnew];
                }
            }
            __syncthreads();
        }
    }
#ifdef AXT_HUNG_STATS
    if (GAP == 1 && lane == 0 && blockIdx.x < 4096) {
        unsigned long long *o = g_hung_stats + blockIdx.x * 8;
        o[0] = st1 - st0; o[1] = HUNG_STAMP() - st1; o[2] = n_search; o[3] = n_step; o[4] = n; o[5] = m;
    }
#endif
    for (int i = lane; i < n; i += 64) {
        const bool active = (GAP == 1) || (succ1[(long)t * cap + i] < 0);
        succ_out[(long)t * cap + i] = (active && col4row[i] >= 0) ? col4row[i] : -1;
    }
    for (int j = lane; j < m; j += 64) pred_out[(long)tb * cap + j] = row4col[j];
}

// chains -> track ids, numbered by (first frame, index), in parallel:
//   root[k]  = first detection of k's chain, by pointer jumping over the predecessor links (ceil(log4 F) rounds);
//   id[root] = rank of the root among all roots in slot order (one block-wide scan over the slots).
// (four hops per launch: every launch costs ~5 us of queue time whatever it does, and a hop is one cached load)
__device__ __forceinline__ int chain_parent(long s, const int *__restrict__ count, int cap, const int *__restrict__ pred1,
                                            const int *__restrict__ pred2)
{
    const int t = s / cap, i = s - (long)t * cap;
    if (i >= min(count[t], cap)) return -1;              // slots beyond count
    const int p1 = (t >= 1) ? pred1[s] : -1;
    const int p2 = (t >= 2) ? pred2[s] : -1;
    return (p1 >= 0) ? (t - 1) * cap + p1 : (p2 >= 0) ? (t - 2) * cap + p2 : (int)s;
}

__global__ void chain_init_kernel(const int *__restrict__ count, int n_frames, int cap, const int *__restrict__ pred1,
                                  const int *__restrict__ pred2, int *__restrict__ root)
{
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= (long)n_frames * cap) return;
    int r = chain_parent(s, count, cap, pred1, pred2);
#pragma unroll
    for (int hop = 0; hop < 3; ++hop)
        if (r >= 0) r = chain_parent(r, count, cap, pred1, pred2);
    root[s] = r;
}

__global__ void chain_jump_kernel(const int *__restrict__ in, int *__restrict__ out, long n)
{
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    int r = in[s];
#pragma unroll
    for (int hop = 0; hop < 3; ++hop)
        if (r >= 0) r = in[r];
    out[s] = r;
}

// ids of roots = exclusive prefix count of (root[s] == s) in slot order. Single block, 1024 slots per round in slot order
// (coalesced): a wave ballots its roots, the sixteen wave counts go through LDS, the running total carries over. The loads of
// sixteen rounds are issued together (one memory latency per 16 k slots instead of one per round).
__global__ __launch_bounds__(1024) void chain_rank_kernel(const int *__restrict__ root, long n, int *__restrict__ rank,
                                                          int *__restrict__ n_tracks)
{
    constexpr int R = 16;                                    // rounds per batch (even: the two sets of counts alternate)
    __shared__ int wsum[2][16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int base = 0;
    for (long b0 = 0; b0 < n; b0 += 1024 * R) {
        int r[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const long s = b0 + q * 1024 + threadIdx.x;
            r[q] = s < n ? root[s] : -2;                     // (-2 beyond n, -1 in empty slots: never equal to a slot)
        }
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const long s = b0 + q * 1024 + threadIdx.x;
            if (b0 + q * 1024 < n) {                         // (block-uniform: the rounds beyond n are skipped by everybody)
                const bool is_root = r[q] == (int)s;
                const unsigned long long m = __ballot(is_root);
                if (lane == 0) wsum[q & 1][w] = __popcll(m);
                __syncthreads();                             // (two sets of counts: one barrier per round)
                int off = 0, tot = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k) { const int v = wsum[q & 1][k]; off += k < w ? v : 0; tot += v; }
                if (is_root) rank[s] = base + off + __popcll(m & ((1ull << lane) - 1));
                base += tot;
            }
        }
        __syncthreads();                                     // (a batch may end on either set)
    }
    if (threadIdx.x == 0) *n_tracks = base;
}

__global__ void chain_assign_kernel(const int *__restrict__ root, const int *__restrict__ rank, int *__restrict__ track, long n)
{
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int r = root[s];
    track[s] = (r < 0) ? -1 : rank[r];
}

// The whole chain numbering in ONE launch for small problems (n <= 8 k slots; at 36 k slots the single workgroup took 152 us against 55 us for the launches): one workgroup walks the slots in
// strides, workgroup barriers separate the pointer-doubling rounds (a CU's L1 is coherent for its own workgroup).
// Replaces 3 + log2(frames) launches whose run time was mostly launch gaps.
__global__ __launch_bounds__(1024) void chain_small_kernel(const int *__restrict__ count, int n_frames, int cap,
                                                           const int *__restrict__ pred1, const int *__restrict__ pred2,
                                                           int *__restrict__ ra, int *__restrict__ rb, int *__restrict__ track,
                                                           int *__restrict__ n_tracks)
{
    __shared__ int cnt[1024];
    const long n = (long)n_frames * cap;
    for (long s = threadIdx.x; s < n; s += 1024) {
        const int t = s / cap, i = s - (long)t * cap;
        int r = -1;
        if (i < min(count[t], cap)) {
            const int p1 = (t >= 1) ? pred1[s] : -1;
            const int p2 = (t >= 2) ? pred2[s] : -1;
            r = (p1 >= 0) ? (t - 1) * cap + p1 : (p2 >= 0) ? (t - 2) * cap + p2 : (int)s;
        }
        ra[s] = r;
    }
    __syncthreads();
    int *in = ra, *out = rb;
    for (int span = 1; span < n_frames; span *= 2) {
        for (long s = threadIdx.x; s < n; s += 1024) {
            const int r = in[s];
            out[s] = (r < 0) ? -1 : in[r];
        }
        __syncthreads();
        int *tmp = in; in = out; out = tmp;
    }
    // rank of the roots in slot order, then every slot takes its root's rank
    const long per = (n + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    int c = 0;
    for (long s = lo; s < hi; ++s) c += in[s] == (int)s;
    cnt[threadIdx.x] = c;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int t = threadIdx.x >= o ? cnt[threadIdx.x - o] : 0;
        __syncthreads();
        cnt[threadIdx.x] += t;
        __syncthreads();
    }
    int id = cnt[threadIdx.x] - c;
    for (long s = lo; s < hi; ++s)
        if (in[s] == (int)s) out[s] = id++;
    if (threadIdx.x == 1023) *n_tracks = cnt[1023];
    __syncthreads();
    for (long s = threadIdx.x; s < n; s += 1024) {
        const int r = in[s];
        track[s] = (r < 0) ? -1 : out[r];
    }
}

__global__ void fill2_int_kernel(int *p, int *q, long n, int v)
{
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) { p[s] = v; q[s] = v; }
}

}  // namespace

int axt_frame_offsets(const int32_t *d_count, int n_frames, int cap, int32_t *d_off, hipStream_t st);

// Pass 1 / pass 2 for the source frames [t_begin, t_end): links into d_pred = pred1 | pred2, each i32 [n_frames*cap]
// (predecessor index in frame t-1 / t-2, or -1). Frame-sharded runs give every rank its own range and combine the
// arrays with one element-wise MAX all-reduce (entries not owned stay -1; the one redundant boundary pair is
// deterministic, so equal on both ranks). d_work i32 [2*n_frames*cap + n_frames + 1].
struct axt_grid;
int axt_masked_distance_table(const axt_grid *g, const int32_t *d_x, const int32_t *d_y, const int32_t *d_count,
                              const int32_t *d_src_count, int n_frames, int cap, int max_dist, int max_gap,
                              const int32_t *h_dmax, const int32_t *d_dmax, int16_t *d_Dtmp, hipStream_t st);

namespace {
__global__ void range_count_kernel(const int *__restrict__ count, int *__restrict__ out, int n, int a, int b)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (i >= a && i < b) ? count[i] : 0;
}
}  // namespace

static int hungarian_pairs_impl(const int32_t *d_x, const int32_t *d_y, const int32_t *d_count, int n_frames, int cap,
                                const axt_grid *grid, int H, int W, int max_dist, int conn8, int max_gap,
                                const int32_t *h_dmax, const int64_t *d_cost_units, int64_t thr_units, int t_begin,
                                int t_end, int32_t *d_pred, int32_t *d_work, void *stream, const int64_t *d_ctab)
{
    AXT_REQUIRE(d_x && d_y && d_count && h_dmax && d_cost_units && d_work && d_pred, "null argument");
    AXT_REQUIRE(n_frames >= 1 && cap >= 1 && cap <= 2048, "axt_hungarian_pairs: cap %d out of range [1,2048]", cap);
    AXT_REQUIRE(max_gap == 1 || max_gap == 2, "axt_hungarian_pairs: max_gap must be 1 or 2");
    AXT_REQUIRE(t_begin >= 0 && t_begin <= t_end && t_end <= n_frames, "axt_hungarian_pairs: bad frame range [%d,%d)", t_begin, t_end);
    hipStream_t st = (hipStream_t)stream;
    const long slots = (long)n_frames * cap;
    int *pred1 = d_pred, *pred2 = d_pred + slots;
    int *succ1 = d_work, *succ2 = succ1 + slots, *frame_off = succ2 + slots;
    hipLaunchKernelGGL(fill2_int_kernel, dim3((unsigned)((2 * slots + 255) / 256)), dim3(256), 0, st, succ1, pred1, 2 * slots, -1);
    AXT_LAUNCH_CHECK();
    int rc = axt_frame_offsets(d_count, n_frames, cap, frame_off, st);
    if (rc) return rc;
    const size_t lds_base = (size_t)cap * (3 * 8 + 8 * 4 + 2) + 8 + (size_t)(max_dist + 2) * 8 + 8 +
                            (size_t)cap * 8 + (cap <= 576 ? (size_t)cap * 48 : 0);       // + dummy costs, + the initialisation's partial minima
    AXT_REQUIRE(lds_base <= 160 * 1024, "axt_hungarian_pairs: cap %d needs %zu bytes of LDS", cap, lds_base);
    // pairs with at most cdim x cdim detections keep their cost matrix in LDS (72 KiB) instead of recomputing it
    int cdim = cap < 96 ? cap : 96;
    if (lds_base + (size_t)cdim * cdim * 8 > 160 * 1024) cdim = 0;
    const size_t lds = lds_base + (size_t)cdim * cdim * 8;
    static AxtOncePerDevice once;                 // (per device: see axt_common.h)
    if (once.pending()) {
        AXT_CHECK_HIP(hipFuncSetAttribute((const void *)hungarian_pair_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AXT_CHECK_HIP(hipFuncSetAttribute((const void *)hungarian_pair_kernel<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AXT_CHECK_HIP(hipFuncSetAttribute((const void *)hungarian_pair_kernel<2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AXT_CHECK_HIP(hipFuncSetAttribute((const void *)hungarian_pair_kernel<1, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AXT_CHECK_HIP(hipFuncSetAttribute((const void *)hungarian_pair_kernel<2, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AXT_CHECK_HIP(hipFuncSetAttribute((const void *)hungarian_pair_kernel<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        once.mark();
    }
    // pass 2 of source frame t needs pass 1 of the pairs (t, t+1) and (t+1, t+2): pass 1 runs one frame further
    const int e1 = (max_gap == 2 ? t_end + 1 : t_end) < n_frames - 1 ? (max_gap == 2 ? t_end + 1 : t_end) : n_frames - 1;
    // masked grid: the path lengths of the source frames' detections to the next max_gap frames, from the searches of the
    // arc builder (path_bfs.hip), read by the pair kernels instead of the closed form
    short *dtab = nullptr;
    int *aux = nullptr;
    if (grid && !d_ctab && e1 > t_begin) {
        AXT_CHECK_HIP(hipMallocAsync((void **)&dtab, sizeof(short) * (size_t)n_frames * cap * max_gap * cap, st));
        AXT_CHECK_HIP(hipMallocAsync((void **)&aux, sizeof(int) * ((size_t)n_frames + max_gap), st));
        int *src_count = aux, *dmax_dev = aux + n_frames;
        hipLaunchKernelGGL(range_count_kernel, dim3((n_frames + 255) / 256), dim3(256), 0, st, d_count, src_count, n_frames, t_begin, e1);
        AXT_LAUNCH_CHECK();
        AXT_CHECK_HIP(hipMemcpyAsync(dmax_dev, h_dmax, sizeof(int) * max_gap, hipMemcpyHostToDevice, st));
        rc = axt_masked_distance_table(grid, d_x, d_y, d_count, src_count, n_frames, cap, max_dist, max_gap, h_dmax, dmax_dev, dtab, st);
        if (rc) return rc;
    }
    if (e1 > t_begin) {
        hipLaunchKernelGGL((cap <= 192 ? hungarian_pair_kernel<1, 3> : cap <= 576 ? hungarian_pair_kernel<1, 9> : hungarian_pair_kernel<1, 0>), dim3(e1 - t_begin), dim3(cap <= 576 ? 256 : 64), lds, st, d_x, d_y, d_count, frame_off,
                           n_frames, cap, H, W, max_dist, conn8, h_dmax[0], (const long *)d_cost_units, (long)thr_units,
                           (const int *)nullptr, (const int *)nullptr, succ1, pred1, cdim, t_begin, (const short *)dtab, max_gap,
                           (const long *)d_ctab);
        AXT_LAUNCH_CHECK();
    }
    const int e2 = t_end < n_frames - 2 ? t_end : n_frames - 2;
    if (max_gap == 2 && e2 > t_begin) {
        hipLaunchKernelGGL((cap <= 192 ? hungarian_pair_kernel<2, 3> : cap <= 576 ? hungarian_pair_kernel<2, 9> : hungarian_pair_kernel<2, 0>), dim3(e2 - t_begin), dim3(cap <= 576 ? 256 : 64), lds, st, d_x, d_y, d_count, frame_off,
                           n_frames, cap, H, W, max_dist, conn8, h_dmax[1],
                           (const long *)d_cost_units + (max_dist + 1), (long)thr_units, (const int *)succ1,
                           (const int *)pred1, succ2, pred2, cdim, t_begin, (const short *)dtab, max_gap, (const long *)d_ctab);
        AXT_LAUNCH_CHECK();
    }
    if (dtab) {
        AXT_CHECK_HIP(hipFreeAsync(dtab, st));
        AXT_CHECK_HIP(hipFreeAsync(aux, st));
    }
    return AXT_OK;
}

extern "C" int axt_hungarian_pairs_grid(const int32_t *d_x, const int32_t *d_y, const int32_t *d_count, int n_frames, int cap,
                                        const axt_grid *grid, int H, int W, int max_dist, int conn8, int max_gap,
                                        const int32_t *h_dmax, const int64_t *d_cost_units, int64_t thr_units, int t_begin,
                                        int t_end, int32_t *d_pred, int32_t *d_work, void *stream)
{
    return hungarian_pairs_impl(d_x, d_y, d_count, n_frames, cap, grid, H, W, max_dist, conn8, max_gap, h_dmax, d_cost_units,
                                thr_units, t_begin, t_end, d_pred, d_work, stream, nullptr);
}

// The same passes with the link costs GIVEN: d_ctab i64 [n_frames][cap][max_gap][cap], entry (t, i, g-1, j) = the integer cost
// (axt_arc_cost_int, kind 3, global detection indices) of linking detection i of frame t to detection j of frame t+g, or
// 0x3fffffffffffffff where the link is not admitted. Used for cost models the closed form does not cover (the appearance
// term: the costs of axt_build_arcs_vis scattered into the table).
extern "C" int axt_hungarian_pairs_costs(const int32_t *d_x, const int32_t *d_y, const int32_t *d_count, int n_frames, int cap,
                                         int max_gap, const int64_t *d_ctab, int64_t thr_units, int t_begin, int t_end,
                                         int32_t *d_pred, int32_t *d_work, void *stream)
{
    AXT_REQUIRE(d_ctab != nullptr, "axt_hungarian_pairs_costs: null cost table");
    const int32_t dmax0[2] = {0, 0};
    return hungarian_pairs_impl(d_x, d_y, d_count, n_frames, cap, nullptr, 1 << 20, 1 << 20, 1, 0, max_gap, dmax0,
                                (const int64_t *)d_ctab, thr_units, t_begin, t_end, d_pred, d_work, stream, d_ctab);
}

extern "C" int axt_hungarian_pairs(const int32_t *d_x, const int32_t *d_y, const int32_t *d_count, int n_frames, int cap,
                                   int H, int W, int max_dist, int conn8, int max_gap, const int32_t *h_dmax,
                                   const int64_t *d_cost_units, int64_t thr_units, int t_begin, int t_end,
                                   int32_t *d_pred, int32_t *d_work, void *stream)
{
    return axt_hungarian_pairs_grid(d_x, d_y, d_count, n_frames, cap, nullptr, H, W, max_dist, conn8, max_gap, h_dmax,
                                    d_cost_units, thr_units, t_begin, t_end, d_pred, d_work, stream);
}

// Chains of links -> trajectory ids numbered by (first frame, index). d_pred as written by axt_hungarian_pairs
// (after the all-reduce in sharded runs); d_work i32 [2*n_frames*cap]; d_track i32 [n_frames*cap]; d_n_tracks i32 [1].
extern "C" int axt_chain_tracks(const int32_t *d_count, int n_frames, int cap, const int32_t *d_pred, int32_t *d_work,
                                int32_t *d_track, int32_t *d_n_tracks, void *stream)
{
    AXT_REQUIRE(d_count && d_pred && d_work && d_track && d_n_tracks && n_frames >= 1 && cap >= 1, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    const long slots = (long)n_frames * cap;
    const int *pred1 = d_pred, *pred2 = d_pred + slots;
    const unsigned nb = (unsigned)((slots + 255) / 256);
    int *ra = d_work, *rb = d_work + slots;
    if (slots <= 8192) {
        hipLaunchKernelGGL(chain_small_kernel, dim3(1), dim3(1024), 0, st, d_count, n_frames, cap, pred1, pred2, ra, rb, d_track,
                           d_n_tracks);
        AXT_LAUNCH_CHECK();
        return AXT_OK;
    }
    hipLaunchKernelGGL(chain_init_kernel, dim3(nb), dim3(256), 0, st, d_count, n_frames, cap, pred1, pred2, ra);
    AXT_LAUNCH_CHECK();
    for (long span = 4; span < n_frames; span *= 4) {        // (chain_init_kernel has gone four hops already)
        hipLaunchKernelGGL(chain_jump_kernel, dim3(nb), dim3(256), 0, st, (const int *)ra, rb, slots);
        AXT_LAUNCH_CHECK();
        int *tmp = ra; ra = rb; rb = tmp;
    }
    hipLaunchKernelGGL(chain_rank_kernel, dim3(1), dim3(1024), 0, st, (const int *)ra, slots, rb, d_n_tracks);
    AXT_LAUNCH_CHECK();
    hipLaunchKernelGGL(chain_assign_kernel, dim3(nb), dim3(256), 0, st, (const int *)ra, (const int *)rb, d_track, slots);
    AXT_LAUNCH_CHECK();
    return AXT_OK;
}

// Both steps for a whole timelapse on one GPU. d_work i32 [4*n_frames*cap + n_frames + 1].
extern "C" int axt_hungarian_assoc(const int32_t *d_x, const int32_t *d_y, const int32_t *d_count, int n_frames, int cap,
                                   int H, int W, int max_dist, int conn8, int max_gap, const int32_t *h_dmax,
                                   const int64_t *d_cost_units, int64_t thr_units, int32_t *d_work, int32_t *d_track,
                                   int32_t *d_n_tracks, void *stream)
{
    AXT_REQUIRE(d_work && n_frames >= 1 && cap >= 1, "bad argument");
    const long slots = (long)n_frames * cap;
    int32_t *pred = d_work, *work = d_work + 2 * slots;
    int rc = axt_hungarian_pairs(d_x, d_y, d_count, n_frames, cap, H, W, max_dist, conn8, max_gap, h_dmax, d_cost_units,
                                 thr_units, 0, n_frames, pred, work, stream);
    if (rc) return rc;
    return axt_chain_tracks(d_count, n_frames, cap, pred, work, d_track, d_n_tracks, stream);
}

#ifdef AXT_HUNG_STATS
// diagnostic build: the statistics of the last gap-1 launch, u64 [4096 frame pairs][8]
extern "C" int axt_debug_hung_stats(unsigned long long *h_out)
{
    return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_hung_stats), sizeof(unsigned long long) * 4096 * 8) == hipSuccess ? 0 : -1;
}
#endif
