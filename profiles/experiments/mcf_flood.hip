// EXPERIMENT, NOT PART OF THE LIBRARY (round 3). Measured exact and SLOWER than the host searches:
// config 3 (81 insertions) 47-75 ms against 19 ms on one host thread, the moving-cone scene 137-162 ms against 45 ms
// (profiles/r03n_device_floods.log). One workgroup per search pays 8-25 us per search round (global atomics, three
// barriers, one CU's share of the L2) where the host pays 0.35 us per scanned row; with bucketed (near-far) ordering the
// relaxation count is Dijkstra's (1.95 M against 1.8 M) but a search needs 60-230 rounds. It was wired in as
// axt_mcf_solve_device (arc list in HBM, state uploaded after the host's first phase) and returned the host solver's
// trajectories and cost on every network of the tests before it was taken out again.
// Second phase of the flow solve on the GPU: the re-insertion of the track ends (row a-12; libmot call sites
// axtrack/AxonDetections.py:663-690, the solver itself is csrc/mcf.cpp).
//
// After the first phase of the assignment-form solver (exits at a fifth of their price) the rows that sit on their exit
// column are taken out and inserted afresh at the full price (Lsap::second_phase). An insertion is a shortest
// augmenting path search on reduced costs (Jonker-Volgenant / Crouse); the searches that DISSOLVE a track have to settle
// every column within the price difference of the track end -- thousands of rows, 20 ms of the 26 ms solve of config 3
// on one host thread, and they overlap too much to run side by side on host threads. Here one 1 024-thread workgroup runs
// them one after the other, each as a PARALLEL label-correcting search:
//   * frontier = rows whose column got a shorter distance; one wavefront per frontier row relaxes its arcs (coalesced
//     reads of the CSR row built by axt_build_arcs, which is already in HBM): free columns and the row's private exit
//     lower `best` (64-bit atomic min on distance << 21 | column), matched columns lower dist[column] (64-bit atomic min)
//     and put the column's row on the next frontier; everything at or beyond `best` is pruned. Reduced costs are >= 0,
//     so at the fixed point every distance below `best` is exact -- what Dijkstra would have settled -- and `best` is the
//     length of the shortest augmenting path and names its sink (ties by column index: any shortest path gives an
//     optimal assignment, and the optimum is unique).
//   * a column's distance and its predecessor row travel in ONE 64-bit word (distance << 20 | row) that is replaced only
//     by a strictly smaller distance (compare-and-swap): the predecessor always belongs to the distance, and predecessor
//     chains cannot close into cycles through arcs of reduced cost 0 (a distance is lowered TO a given value only once);
//   * dual update (Crouse 2016, Alg. 1) for the columns with dist < best and their rows; the sink's predecessor is the
//     visited row whose relaxation equals `best`; augmentation by one lane.
// State (duals, matching) lives in HBM for the whole batch; the host uploads it after its first phase and reads it back.
// Exactness does not depend on anything here being in the host solver's order: row insertion reaches the one optimum.
#include "axt_common.h"

namespace {

constexpr unsigned long long FINF = ~0ull;
constexpr int TAG_BITS = 21;                       // columns 0 .. 2n-1 < 2^21
#ifndef AXT_FLOOD_SHIFT
#define AXT_FLOOD_SHIFT 6
#endif
constexpr int PRED_BITS = 20;                      // rows 0 .. n-1 < 2^20; distances < 2^43

struct FloodArgs {
    int n;
    const long *row_ptr;                           // arcs, CSR by tail (axt_build_arcs)
    const int *col;
    const long *cost;
    const long *base, *own, *entry;                // per row: obs + entry, obs + entry + exit (full price); per detection: entry
    long *u, *v;                                   // duals: rows [n], in-slot columns [n]
    int *row4col, *col4row;                        // in-slot column -> row (-1 free); row -> column (j < n in-slot, n + k exit, -1 none)
    unsigned long long *dist;                      // [n] (search distance << 20 | predecessor row) of the in-slot columns (FINF = untouched)
    int *flag, *front_a, *front_b, *near, *touched;   // [n] each
    const int *ends;                               // rows to insert, in order
    int n_ends;
    int *stats;                                    // [8]: relaxations / 1024, frontier rows, iterations, floods, error (0 = none)
};

__global__ __launch_bounds__(1024) void mcf_floods_kernel(FloodArgs a)
{
    __shared__ unsigned long long best;            // (distance << 21) | sink column  (sink: j < n free in-slot, n + r exit of row r)
    __shared__ int n_front, n_next, n_touched, sink_pred, n_near;
    __shared__ unsigned long long min_far;
    __shared__ long u_src;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.n;
    int epoch = 1;
    long st_relax = 0, st_rows = 0, st_iter = 0;
    for (int q = 0; q < a.n_ends; ++q) {
        const int e = a.ends[q];
        // ---- the source row's own dual: its cheapest option, so that every distance from it is >= 0
        if (tid == 0) { best = FINF; n_front = 0; n_next = 0; n_touched = 0; u_src = 0x3fffffffffffffffL; }
        __syncthreads();
        {
            long m = 0x3fffffffffffffffL;
            const long lo = a.row_ptr[e], hi = a.row_ptr[e + 1];
            for (long k = lo + tid; k < hi; k += 1024) {
                const int j = a.col[k];
                const long w = a.base[e] + a.cost[k] - a.entry[j] - a.v[j];
                m = w < m ? w : m;
            }
            if (tid == 0) {
                const long w0 = 0 - a.v[e];                       // its own in-slot ("unused")
                m = w0 < m ? w0 : m;
                m = a.own[e] < m ? a.own[e] : m;                  // its private exit (free: dual 0)
            }
            for (int o = 32; o > 0; o >>= 1) { const long t = __shfl_down(m, o); m = t < m ? t : m; }
            if (lane == 0) atomicMin((long long *)&u_src, (long long)m);
        }
        __syncthreads();
        const long ue = u_src;
        if (tid == 0) { a.front_a[0] = e; n_front = 1; }
        __syncthreads();
        int *front = a.front_a, *next = a.front_b;
        // ---- label-correcting search (every distance only ever decreases, so it ends; the cap is a guard against a state
        // that is not dual feasible -- a caller's bug must not leave waves spinning on the GPU)
        long iters = 0;
        bool failed = false;
        // Near-far ordering: a round expands only the frontier rows within `tau` of the source and defers the rest, so that
        // rows are expanded roughly in order of distance (a plain label-correcting sweep re-expanded every row ~13 times on the
        // config-3 network: 23 M relaxations against Dijkstra's 1.8 M). tau moves on to the nearest deferred row plus 1/64 of the
        // current search radius (the distance of the best free column seen so far) whenever nothing is near.
        long tau = 0;
        for (;;) {
            const int nf = n_front;
            if (nf == 0) break;
            if (++iters > 64l * n + 4096) { failed = true; break; }
            ++epoch;
            ++st_iter;
            // ---- classify: near rows to `near`, the rest to the next frontier; rows at or beyond `best` are dropped
            if (tid == 0) { n_near = 0; min_far = FINF; }
            __syncthreads();
            const unsigned long long best_d = best >> TAG_BITS;
            for (int f = tid; f < nf; f += 1024) {
                const int r = front[f];
                unsigned long long d = 0;
                if (r != e) {
                    const int mc = a.col4row[r];
                    if (mc < 0 || mc >= n) { if (a.stats) a.stats[4] = 5; continue; }        // never on a consistent matching
                    d = a.dist[mc] >> PRED_BITS;
                }
                if (d > best_d) continue;
                if ((long)d <= tau) { const int at = atomicAdd(&n_near, 1); if (at < n) a.near[at] = r; }
                else {
                    atomicMin(&min_far, d);
                    if (atomicExch(&a.flag[r], epoch) != epoch) { const int at = atomicAdd(&n_next, 1); if (at < n) next[at] = r; }
                }
            }
            __syncthreads();
            const int nn = n_near;
            if (nn == 0) {
                // nothing near: move the horizon on to the nearest deferred row (+ delta) and look again
                if (min_far != FINF) tau = (long)min_far + (long)(best_d >> AXT_FLOOD_SHIFT) + 65536;
                __syncthreads();
                if (tid == 0) { n_front = n_next; n_next = 0; }
                int *t = front; front = next; next = t;
                __syncthreads();
                continue;
            }
            for (int f = wave; f < nn; f += 16) {
                const int r = a.near[f];
                const int mcol = r == e ? -1 : a.col4row[r];
                const unsigned long long dr = r == e ? 0ull : a.dist[mcol] >> PRED_BITS;
                const long ur = r == e ? ue : a.u[r];
                const long off = (long)dr - ur;
                const long lo = a.row_ptr[r], hi = a.row_ptr[r + 1];
                st_rows += lane == 0;
                // lanes 0 .. deg-1: the arcs; two more lanes: own in-slot, private exit
                for (long k = lo + lane - 2; k < hi; k += 64) {
                    int j;
                    long w;
                    bool is_exit = false;
                    if (k == lo - 2) { j = r; w = 0; }
                    else if (k == lo - 1) { j = n + r; w = a.own[r]; is_exit = true; }
                    else if (k >= lo) { j = a.col[k]; w = a.base[r] + a.cost[k] - a.entry[j]; }
                    else continue;
                    if (j == mcol) continue;                              // the tree edge the row was reached through
                    ++st_relax;
                    const long rc = off + w - (is_exit ? 0 : a.v[j]);     // >= 0
                    if (rc < 0 || rc >= (1l << 42)) continue;             // (beyond anything a track end can be worth)
                    const unsigned long long key = ((unsigned long long)rc << TAG_BITS) | (unsigned)j;
                    if (key >= best) continue;
                    const int owner = is_exit ? -1 : a.row4col[j];
                    if (owner < 0) atomicMin(&best, key);
                    else {
                        // (distance, predecessor) replaced only by a strictly smaller distance
                        const unsigned long long mine = ((unsigned long long)rc << PRED_BITS) | (unsigned)r;
                        unsigned long long old = a.dist[j];
                        bool won = false;
                        while ((old >> PRED_BITS) > (unsigned long long)rc) {
                            const unsigned long long seen = atomicCAS(&a.dist[j], old, mine);
                            if (seen == old) { won = true; break; }
                            old = seen;
                        }
                        if (won) {
                            if (old == FINF) { const int at = atomicAdd(&n_touched, 1); if (at < n) a.touched[at] = j; }
                            if (atomicExch(&a.flag[owner], epoch) != epoch) { const int at = atomicAdd(&n_next, 1); if (at < n) next[at] = owner; }
                        }
                    }
                }
            }
            __syncthreads();
            if (tid == 0) { n_front = n_next; n_next = 0; }
            int *t = front; front = next; next = t;
            __syncthreads();
        }
        if (failed || best == FINF) {                  // (uniform: every thread read the same shared words)
            if (tid == 0 && a.stats) a.stats[4] = failed ? 1 : 2;
            break;
        }
        // ---- the shortest augmenting path: length, sink; predecessors of the columns below it
        const unsigned long long bk = best;
        const long minVal = (long)(bk >> TAG_BITS);
        const int sink = (int)(bk & ((1u << TAG_BITS) - 1));
        const int nt = n_touched;
        // the sink's predecessor (a free in-slot has no distance word of its own): the visited row -- the source, or the row of
        // a column at distance <= best -- whose relaxation of the sink equals `best`
        if (tid == 0) sink_pred = -1;
        __syncthreads();
        if (sink < n) {
            for (int f = wave; f < nt + 1; f += 16) {
                int r, mcol;
                unsigned long long dr;
                if (f == nt) { r = e; dr = 0; mcol = -1; }
                else {
                    mcol = a.touched[f];
                    dr = a.dist[mcol] >> PRED_BITS;
                    if ((long)dr > minVal) continue;
                    r = a.row4col[mcol];
                }
                const long ur = r == e ? ue : a.u[r];
                const long off = (long)dr - ur;
                const long lo = a.row_ptr[r], hi = a.row_ptr[r + 1];
                for (long k = lo + lane - 1; k < hi; k += 64) {
                    int j;
                    long w;
                    if (k == lo - 1) { j = r; w = 0; }
                    else { j = a.col[k]; w = a.base[r] + a.cost[k] - a.entry[j]; }
                    if (j != sink || j == mcol) continue;
                    if (off + w - a.v[j] == minVal) atomicMax(&sink_pred, r);
                }
            }
        }
        __syncthreads();
        // ---- dual update: the source, the columns of the tree and their rows
        for (int f = tid; f < nt; f += 1024) {
            const int j = a.touched[f];
            const long d = (long)(a.dist[j] >> PRED_BITS);
            if (d < minVal) {
                a.v[j] -= minVal - d;
                a.u[a.row4col[j]] += minVal - d;
            }
        }
        if (tid == 0) a.u[e] = ue + minVal;
        __syncthreads();
        // ---- augment back to the source; then forget the search
        if (tid == 0) {
            int j = sink, r = sink >= n ? sink - n : sink_pred;
            for (int steps = 0;; ++steps) {
                if (r < 0 || r >= n || j < 0 || j >= 2 * n || steps > n) { if (a.stats) { a.stats[4] = 3; a.stats[5] = q; a.stats[6] = j; a.stats[7] = r; } break; }     // never on a consistent state
                const int prev = a.col4row[r];
                a.col4row[r] = j;
                if (j < n) a.row4col[j] = r;
                if (r == e) break;
                j = prev;
                if (j < 0 || j >= n || a.dist[j] == FINF) { if (a.stats) { a.stats[4] = 4; a.stats[5] = q; a.stats[6] = j; a.stats[7] = r * 1000 + steps; } break; }
                r = (int)(a.dist[j] & ((1u << PRED_BITS) - 1));
            }
        }
        __syncthreads();                                   // (the walk reads the distance words)
        for (int f = tid; f < nt; f += 1024) a.dist[a.touched[f]] = FINF;
        __syncthreads();
    }
    // statistics (diagnostics only)
    if (a.stats) {
        atomicAdd(&a.stats[0], (int)(st_relax >> 10));
        atomicAdd(&a.stats[1], (int)st_rows);
        if (tid == 0) { a.stats[2] = (int)st_iter; a.stats[3] = a.n_ends; }
    }
}

__global__ void fill_u64_kernel(unsigned long long *p, long n, unsigned long long v)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace

// Host side: state up, one launch, state down. All h_* arrays have n entries (h_ends: n_ends); d_row_ptr [n+1], d_col,
// d_cost: the arc list in HBM (the tensors axt_build_arcs filled). h_col4row uses n + k for row k's private exit.
// h_stats (optional, 4 ints): relaxations / 1024, frontier rows, search iterations, floods.
int axt_mcf_device_floods(int n, const int64_t *d_row_ptr, const int32_t *d_col, const int64_t *d_cost, const int64_t *h_base,
                          const int64_t *h_own, const int64_t *h_entry, int64_t *h_u, int64_t *h_v, int32_t *h_row4col,
                          int32_t *h_col4row, const int32_t *h_ends, int n_ends, int32_t *h_stats, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    AXT_REQUIRE(n >= 1 && n < (1 << PRED_BITS) && 2l * n < (1l << TAG_BITS) && n_ends >= 0 && d_row_ptr && d_col && d_cost, "axt_mcf_device_floods: bad argument (n = %d)", n);
    if (n_ends == 0) return AXT_OK;
    const size_t n8 = sizeof(long) * (size_t)n, n4 = sizeof(int) * (size_t)n;
    char *blk = nullptr;
    const size_t bytes = 6 * n8 + 7 * n4 + sizeof(int) * (size_t)n_ends + 32 + 16 * 16;      // every piece is rounded up to 16 bytes
    AXT_CHECK_HIP(hipMallocAsync((void **)&blk, bytes, st));
    char *p = blk;
    auto take = [&](size_t b) { char *q = p; p += (b + 15) & ~(size_t)15; return q; };
    FloodArgs a;
    a.n = n; a.row_ptr = (const long *)d_row_ptr; a.col = d_col; a.cost = (const long *)d_cost;
    long *d_base = (long *)take(n8), *d_own = (long *)take(n8), *d_entry = (long *)take(n8);
    a.u = (long *)take(n8); a.v = (long *)take(n8);
    a.dist = (unsigned long long *)take(n8);
    a.row4col = (int *)take(n4); a.col4row = (int *)take(n4);
    a.flag = (int *)take(n4); a.front_a = (int *)take(n4); a.front_b = (int *)take(n4); a.near = (int *)take(n4); a.touched = (int *)take(n4);
    int *d_ends = (int *)take(sizeof(int) * (size_t)n_ends);
    a.stats = (int *)take(32);
    a.base = d_base; a.own = d_own; a.entry = d_entry; a.ends = d_ends; a.n_ends = n_ends;
    AXT_REQUIRE((size_t)(p - blk) <= bytes, "axt_mcf_device_floods: workspace layout");
    AXT_CHECK_HIP(hipMemcpyAsync(d_base, h_base, n8, hipMemcpyHostToDevice, st));
    AXT_CHECK_HIP(hipMemcpyAsync(d_own, h_own, n8, hipMemcpyHostToDevice, st));
    AXT_CHECK_HIP(hipMemcpyAsync(d_entry, h_entry, n8, hipMemcpyHostToDevice, st));
    AXT_CHECK_HIP(hipMemcpyAsync(a.u, h_u, n8, hipMemcpyHostToDevice, st));
    AXT_CHECK_HIP(hipMemcpyAsync(a.v, h_v, n8, hipMemcpyHostToDevice, st));
    AXT_CHECK_HIP(hipMemcpyAsync(a.row4col, h_row4col, n4, hipMemcpyHostToDevice, st));
    AXT_CHECK_HIP(hipMemcpyAsync(a.col4row, h_col4row, n4, hipMemcpyHostToDevice, st));
    AXT_CHECK_HIP(hipMemcpyAsync(d_ends, h_ends, sizeof(int) * (size_t)n_ends, hipMemcpyHostToDevice, st));
    AXT_CHECK_HIP(hipMemsetAsync(a.flag, 0, n4, st));
    AXT_CHECK_HIP(hipMemsetAsync(a.stats, 0, 32, st));
    hipLaunchKernelGGL(fill_u64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.dist, (long)n, FINF);
    AXT_LAUNCH_CHECK();
    hipLaunchKernelGGL(mcf_floods_kernel, dim3(1), dim3(1024), 0, st, a);
    AXT_LAUNCH_CHECK();
    AXT_CHECK_HIP(hipMemcpyAsync(h_u, a.u, n8, hipMemcpyDeviceToHost, st));
    AXT_CHECK_HIP(hipMemcpyAsync(h_v, a.v, n8, hipMemcpyDeviceToHost, st));
    AXT_CHECK_HIP(hipMemcpyAsync(h_row4col, a.row4col, n4, hipMemcpyDeviceToHost, st));
    AXT_CHECK_HIP(hipMemcpyAsync(h_col4row, a.col4row, n4, hipMemcpyDeviceToHost, st));
    int stats[8] = {};
    AXT_CHECK_HIP(hipMemcpyAsync(stats, a.stats, 32, hipMemcpyDeviceToHost, st));
    AXT_CHECK_HIP(hipStreamSynchronize(st));
    AXT_CHECK_HIP(hipFreeAsync(blk, st));
    if (h_stats) memcpy(h_stats, stats, 16);
    if (stats[4]) {
        axt_set_error("axt_mcf_device_floods: the search state was inconsistent (code %d; insertion %d of %d, column %d, row/steps %d, n %d)", stats[4],
                      stats[5], n_ends, stats[6], stats[7], n);
        return AXT_ERUNTIME;
    }
    return AXT_OK;
}
