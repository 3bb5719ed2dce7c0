// Path lengths on a masked grid: weights {1 on mask, 65536 off} (reference AxonDetections.py:598), the A* of
// utils.py:379 (pyastar2d, absent from the reference tree -- convention in DESIGN.md).
//
// For one source detection the lengths to ALL targets come from a single-source search, instead of one A* per
// (source, target) pair as the reference does (AxonDetections.py:570-576): one workgroup per source.
//
// Definition (identical to the CPU checker's): on the whole grid, find the minimum-cost 4-/8-connected path to
// each target, cost of a move = weight of the cell moved into.
// With weights {1, 65536} and fewer than 65536 on-mask moves the cost order equals the lexicographic order of
// (off-mask cells entered, moves), packed here as the 64-bit key off << 32 | moves (moves can exceed 16 bits in
// a 1001^2 window). The result is moves + 1 cells, or max_dist ("None")
// when that exceeds max_dist, the euclidean gate fails or an end point lies outside the grid.
//
// Search: frontier label-correcting (parallel Bellman-Ford over worklists). Every cell of the current frontier
// relaxes its neighbours with atomicMin on the 64-bit key; improved cells enter the next frontier once
// (stamp array). Keys only decrease and the iteration ends when no key changes, so the fixed point is the exact
// shortest-path key of every cell, independent of scheduling. The frontiers live in HBM and each step costs two
// workgroup barriers: this is the exact, general search (any target, paths across off-mask cells, the paths
// themselves); the arc builder's hot path is the bit-parallel LDS search further down and falls back to this one
// only for the rare targets that one cannot decide.
#include "axt_common.h"

#include <new>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

struct axt_grid;
extern "C" void axt_grid_destroy(axt_grid *g);

namespace {

typedef unsigned long long u64;

struct Scratch {
    u64 *key;          // [n_src][win_cells]
    int *stamp;        // [n_src][win_cells]
    int *list_a;       // [n_src][win_cells]
    int *list_b;       // [n_src][win_cells]
};

__global__ __launch_bounds__(256) void path_sssp_kernel(
    const int *__restrict__ xa, const int *__restrict__ ya, int na,
    const int *__restrict__ xb, const int *__restrict__ yb, int nb,
    const unsigned char *__restrict__ mask, int H, int W, int max_dist, int conn8, long win_cells_cap,
    Scratch sc, int *__restrict__ D)
{
    const int src = blockIdx.x, tid = threadIdx.x;
    int *Drow = D + (long)src * nb;
    const int sx = xa[src], sy = ya[src];
    const bool src_in = sx >= 0 && sx < W && sy >= 0 && sy < H;
    if (!src_in) {
        for (int j = tid; j < nb; j += 256) Drow[j] = max_dist;
        return;
    }
    const int y0 = 0, y1 = H - 1, x0 = 0, x1 = W - 1;      // the search runs on the whole grid
    const int wh = y1 - y0 + 1, ww = x1 - x0 + 1;
    const long ncell = (long)wh * ww;
    u64 *key = sc.key + (long)src * win_cells_cap;
    int *stamp = sc.stamp + (long)src * win_cells_cap;
    int *cur = sc.list_a + (long)src * win_cells_cap;
    int *nxt = sc.list_b + (long)src * win_cells_cap;
    for (long c = tid; c < ncell; c += 256) { key[c] = ~0ull; stamp[c] = -1; }
    __shared__ int n_cur, n_nxt;
    __syncthreads();
    if (tid == 0) {
        const int s = (sy - y0) * ww + (sx - x0);
        key[s] = 0;
        cur[0] = s;
        n_cur = 1;
        n_nxt = 0;
    }
    __threadfence_block();
    __syncthreads();
    const int nn = conn8 ? 8 : 4;
    const int dy8[8] = {-1, 1, 0, 0, -1, -1, 1, 1}, dx8[8] = {0, 0, -1, 1, -1, 1, -1, 1};
    // ends when the frontier is empty; keys strictly decrease, so at most ncell rounds can do work
    for (long iter = 0; iter < ncell + 8; ++iter) {
        const int n = n_cur;
        if (n == 0) break;
        for (int e = tid; e < n; e += 256) {
            const int c = cur[e];
            const u64 k = __hip_atomic_load(&key[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int cy = c / ww, cx = c - cy * ww;
            for (int d = 0; d < nn; ++d) {
                const int ny = cy + dy8[d], nx = cx + dx8[d];
                if (ny < 0 || ny >= wh || nx < 0 || nx >= ww) continue;
                const bool on = mask[(long)(ny + y0) * W + (nx + x0)] == 1;
                const u64 nk = k + 1ull + (on ? 0ull : (1ull << 32));
                const int nc = ny * ww + nx;
                const u64 old = atomicMin(&key[nc], nk);
                if (nk < old) {
                    if (atomicExch(&stamp[nc], (int)iter) != (int)iter) nxt[atomicAdd(&n_nxt, 1)] = nc;
                }
            }
        }
        __threadfence_block();
        __syncthreads();
        if (tid == 0) { n_cur = n_nxt; n_nxt = 0; }
        int *t = cur; cur = nxt; nxt = t;
        __syncthreads();
    }
    __syncthreads();
    for (int j = tid; j < nb; j += 256) {
        const int tx = xb[j], ty = yb[j];
        int out = max_dist;
        const long dx = tx - sx, dy = ty - sy;
        if (tx >= x0 && tx <= x1 && ty >= y0 && ty <= y1 && dx * dx + dy * dy < (long)max_dist * max_dist) {
            const u64 k = key[(long)(ty - y0) * ww + (tx - x0)];
            if (k != ~0ull) {
                const u64 moves = k & 0xffffffffull;
                if (moves + 1 <= (u64)max_dist) out = (int)moves + 1;
            }
        }
        Drow[j] = out;
    }
}

// The cells of the paths whose lengths the search above found: walk from the target back to the source, at every
// cell c to the first neighbour n (order: up, down, left, right, then the diagonals) with key[n] + weight(c) == key[c]
// -- one exists at the fixed point of the search. cells[(src*nb + j)*max_dist + k] = y*W + x of the k-th cell, source
// first, for k < D[src][j]; pairs without a path (D == max_dist) are left alone. Which of several equally cheap
// paths the reference's A* returns is not pinned: this is one of them.
__global__ __launch_bounds__(64) void path_backtrack_kernel(
    const int *__restrict__ xb, const int *__restrict__ yb, int nb, const unsigned char *__restrict__ mask, int H, int W,
    int max_dist, int conn8, long win_cells_cap, const u64 *__restrict__ key_base, const int *__restrict__ D,
    int *__restrict__ cells)
{
    const int src = blockIdx.y, j = blockIdx.x * 64 + threadIdx.x;
    if (j >= nb) return;
    const int d = D[(long)src * nb + j];
    if (d >= max_dist) return;
    const u64 *key = key_base + (long)src * win_cells_cap;
    int *out = cells + ((long)src * nb + j) * max_dist;
    const int nn = conn8 ? 8 : 4;
    const int dy8[8] = {-1, 1, 0, 0, -1, -1, 1, 1}, dx8[8] = {0, 0, -1, 1, -1, 1, -1, 1};
    int cx = xb[j], cy = yb[j];
    for (int k = d - 1; k >= 0; --k) {
        const long c = (long)cy * W + cx;
        out[k] = (int)c;
        if (k == 0) break;
        const u64 want = key[c] - (1ull + (mask[c] == 1 ? 0ull : (1ull << 32)));
        int found = -1;
        for (int q = 0; q < nn && found < 0; ++q) {
            const int ny = cy + dy8[q], nx = cx + dx8[q];
            if (ny < 0 || ny >= H || nx < 0 || nx >= W) continue;
            if (key[(long)ny * W + nx] == want) found = q;
        }
        if (found < 0) {                                   // cannot happen at the fixed point; leave a visible mark
            for (int r = 0; r < k; ++r) out[r] = -1;
            break;
        }
        cy += dy8[found];
        cx += dx8[found];
    }
}

}  // namespace

int axt_path_cost_masked(const int32_t *d_xa, const int32_t *d_ya, int na, const int32_t *d_xb, const int32_t *d_yb,
                         int nb, const uint8_t *d_mask, int H, int W, int max_dist, int conn8, int32_t *d_D,
                         hipStream_t st, int32_t *d_cells)
{
    const long win = (long)H * W;
    // sources are processed in batches so that the HBM scratch (24 bytes per window cell and source) stays bounded
    const long bytes_per_src = win * (8 + 4 + 4 + 4);
    long batch = (long)(8ll << 30) / bytes_per_src;     // <= 8 GiB of scratch
    if (batch < 1) batch = 1;
    if (batch > na) batch = na;
    unsigned char *raw = nullptr;
    AXT_CHECK_HIP(hipMallocAsync((void **)&raw, (size_t)(batch * bytes_per_src), st));
    Scratch sc;
    sc.key = reinterpret_cast<u64 *>(raw);
    sc.stamp = reinterpret_cast<int *>(raw + batch * win * 8);
    sc.list_a = sc.stamp + batch * win;
    sc.list_b = sc.list_a + batch * win;
    for (long s0 = 0; s0 < na; s0 += batch) {
        const int n = (int)((na - s0 < batch) ? na - s0 : batch);
        hipLaunchKernelGGL(path_sssp_kernel, dim3(n), dim3(256), 0, st, d_xa + s0, d_ya + s0, n, d_xb, d_yb, nb, d_mask, H,
                           W, max_dist, conn8, win, sc, d_D + s0 * nb);
        AXT_LAUNCH_CHECK();
        if (d_cells) {
            hipLaunchKernelGGL(path_backtrack_kernel, dim3((nb + 63) / 64, n), dim3(64), 0, st, d_xb, d_yb, nb, d_mask, H, W,
                               max_dist, conn8, win, (const u64 *)sc.key, (const int *)(d_D + s0 * nb), d_cells + s0 * nb * max_dist);
            AXT_LAUNCH_CHECK();
        }
    }
    AXT_CHECK_HIP(hipFreeAsync(raw, st));
    return AXT_OK;
}

// =====================================================================================================================
// Fast path for the arc builder on a masked grid.
//
// Only path lengths up to dmax (251 cells at gap 1, 86 at gap 2) can become arcs. For a target T that is reachable
// from the source S through on-mask cells only, the minimum-cost path IS the shortest on-mask path (every off-mask
// cell costs 65536), so its length is a plain breadth-first distance -- and whether such a path exists at all is a
// property of the mask's connected components, computed once per timelapse (axt_grid_create). Hence per source:
//   * level-0 targets (T == S, or T on the mask in a component S touches): BFS over on-mask cells, depth-limited to
//     dmax-1 moves; not reached within that depth => the optimum is longer than dmax cells => no arc;
//   * other targets (off-mask, or in a component S does not touch) close enough that a path of <= dmax cells could
//     exist: with the component fields (axt_grid::d_off / d_tight) they are served by the same search over "tight"
//     steps -- sources on the mask: one front; sources off it: two fronts (see mask_bfs_kernel) -- and only without
//     the fields (more than 64 components), or where two components tie, marked for the searches further down.
// The BFS is bit-parallel: the (2R+1)^2 window around S lives in LDS as three bitmaps (reached A/B, mask M); one BFS
// step is a 4-/8-neighbour dilation of whole 32-cell words, AND-ed with the mask. One workgroup per source serves
// the targets of BOTH following frames (gaps 1 and 2).
// =====================================================================================================================
struct axt_grid {
    int H = 0, W = 0, Ww = 0, conn8 = 0;
    unsigned char *d_mask = nullptr;     // [H][W] 0/1
    unsigned int *d_bits = nullptr;      // [H][Ww] bit x%32 of word x/32, zero-padded
    int *d_label = nullptr;              // [H][W] connected-component label >= 1 on the mask, 0 off it
    // [n_comp][H][W] u8: fewest off-mask cells any path from component `label` has to enter to reach the cell (the
    // cell itself included when it is off the mask), saturated at 255; NULL when the mask has too many components
    unsigned char *d_off = nullptr;
    int n_comp = 0;
    bool has_fields = false;             // d_off covers every component (trivially so for an empty mask)
    // [n_comp][4 or 8][H][Ww] bit rows: bit x of row y of direction d is set iff stepping INTO (y,x) from its
    // neighbour (y+oy[d], x+ox[d]) keeps the off-cell count minimal: d_off[A][(y,x)] == d_off[A][neighbour] + [cell off]
    unsigned int *d_tight = nullptr;
};

namespace {

constexpr int BFS_R = 250;                       // moves; cells <= 251
constexpr int BFS_WH = 2 * BFS_R + 1;            // window rows
constexpr int BFS_WW = 18;                       // window words per row: 501 bits + up to 31 bits of alignment + guard

__device__ __forceinline__ unsigned int dil_h(const unsigned int *row, int w)
{
    const unsigned int c = row[w];
    const unsigned int l = w > 0 ? row[w - 1] : 0u, r = w + 1 < BFS_WW ? row[w + 1] : 0u;
    return c | (c << 1) | (l >> 31) | (c >> 1) | (r << 31);
}

constexpr int BFS_THREADS = 1024;      // one workgroup per CU (LDS-bound): many waves hide the LDS latency of the sweeps

struct BfsGeo { int wy0, wx0, H, W, Ww, depth, sr, sc, n_targets; };

// Breadth-first search over the tight steps of one component's field (bit rows tg, axt_grid::d_tight), bit-parallel on
// the window bitmaps in LDS. Every thread owns the words e = tid + k * 1024 of the window for the whole search and
// keeps their tight-step rows (one word per direction) in registers: a move then costs LDS reads and one barrier, no
// global memory (with the rows fetched from L2 inside every move a move took ~4 us: several dependent L2 round trips).
//   off == false: source on the mask; front c0/c1 (ping-pong), seeded by the caller in c0.
//   off == true : source off the mask; the all-off front in c0/c1 (dilation over off-mask cells, seeded with the source)
//                 and, from move best_a + 1 on, the front over the mask in v0/v1 (tight steps of component best_comp,
//                 seeded with the mask cells next to the all-off front). A target is settled by the all-off front at
//                 move s iff s <= tkv (see the kernel), else by the other front.
// One barrier per move: per-move flags in LDS (three in rotation) carry "some word changed" and "some target is still
// open" (as of the previous move's checks), so the checks of move s run beside the dilation of move s + 1.
template <int NDIR>
__device__ __forceinline__ void bfs_owned(bool off, unsigned int *c0, unsigned int *c1, unsigned int *v0, unsigned int *v1,
                                          const unsigned int *__restrict__ tg, const unsigned int *__restrict__ bits,
                                          const BfsGeo &G, int *tpos, short *tres, const short *tkv, int (*s_step)[2],
                                          int best_comp, int best_a, int tid)
{
    constexpr int NWORDS = BFS_WH * BFS_WW;
    constexpr int KW = (NWORDS + BFS_THREADS - 1) / BFS_THREADS;           // 9
    constexpr int oy8[8] = {-1, 1, 0, 0, -1, -1, 1, 1}, ox8[8] = {0, 0, -1, 1, -1, 1, -1, 1};
    unsigned int tw[KW][NDIR], mbk[KW], okk[KW];
    const long dstride = (long)G.H * G.Ww;
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const int e = tid + k * BFS_THREADS;
        const int r = e / BFS_WW, w = e - r * BFS_WW;
        const int gy = G.wy0 + r, gw = (G.wx0 >> 5) + w;
        const bool ing = e < NWORDS && gy >= 0 && gy < G.H && gw >= 0 && gw < G.Ww;
#pragma unroll
        for (int d = 0; d < NDIR; ++d) tw[k][d] = (ing && tg) ? tg[d * dstride + (long)gy * G.Ww + gw] : 0u;
        const unsigned int mb = (ing && off) ? bits[(long)gy * G.Ww + gw] : 0u;
        const unsigned int valid = !ing ? 0u : (gw == G.Ww - 1 && (G.W & 31)) ? ((1u << (G.W & 31)) - 1u) : 0xffffffffu;
        mbk[k] = mb;
        okk[k] = ~mb & valid;
    }
    int my_open = 0;
    for (int e = tid; e < G.n_targets; e += BFS_THREADS) my_open |= (tpos[e] >= 0);
    unsigned int *fc = off ? v0 : c0, *fn = off ? v1 : c1;          // the front over tight steps
    unsigned int *oc = c0, *on = c1;                                 // (off) the all-off front
    for (int s = 1; s <= G.depth; ++s) {
        // after s moves only the cells within s of the source can be set: rows sr-s..sr+s, words of columns sc-s..sc+s
        const int rlo = max(0, G.sr - s), rhi = min(BFS_WH - 1, G.sr + s);
        const int wlo = max(0, (G.sc - s) >> 5), whi = min(BFS_WW - 1, (G.sc + s) >> 5);
        int changed = 0;
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            const int e = tid + k * BFS_THREADS;
            const int r = e / BFS_WW, w = e - r * BFS_WW;
            if (e >= NWORDS || r < rlo || r > rhi || w < wlo || w > whi) continue;
            unsigned int d_off_front = 0u;
            if (off) {
                const unsigned int *row = oc + r * BFS_WW;
                unsigned int d = dil_h(row, w);
                if (NDIR == 8) {
                    if (r > 0) d |= dil_h(row - BFS_WW, w);
                    if (r + 1 < BFS_WH) d |= dil_h(row + BFS_WW, w);
                } else {
                    if (r > 0) d |= row[w - BFS_WW];
                    if (r + 1 < BFS_WH) d |= row[w + BFS_WW];
                }
                const unsigned int vo = row[w] | (d & okk[k]);
                on[e] = vo;
                changed |= (vo != row[w]);
                d_off_front = d;
            }
            const unsigned int old = fc[e];
            unsigned int vv = old;
            if (off && s == best_a + 1) {
                if (best_comp > 0) vv |= d_off_front & mbk[k];               // the step from the all-off front onto the mask
            } else if (!off || (best_comp > 0 && s > best_a + 1)) {
#pragma unroll
                for (int d = 0; d < NDIR; ++d) {
                    const int rp = r + oy8[d];
                    if (rp < 0 || rp >= BFS_WH) continue;
                    const unsigned int *prow = fc + rp * BFS_WW;
                    const unsigned int p0 = prow[w];
                    unsigned int from;
                    if (ox8[d] < 0) from = (p0 << 1) | (w > 0 ? prow[w - 1] >> 31 : 0u);
                    else if (ox8[d] > 0) from = (p0 >> 1) | (w + 1 < BFS_WW ? prow[w + 1] << 31 : 0u);
                    else from = p0;
                    vv |= from & tw[k][d];
                }
            }
            fn[e] = vv;
            changed |= (vv != old);
        }
        if (changed) s_step[s % 3][0] = 1;
        if (my_open) s_step[s % 3][1] = 1;
        __syncthreads();
        const int both = s_step[s % 3][0] | (s_step[s % 3][1] << 1);
        if (tid == 0) s_step[(s + 2) % 3][0] = s_step[(s + 2) % 3][1] = 0;      // for the move after the next one
        if (!(both & 2)) break;                                        // every target was settled by the previous move
        my_open = 0;
        for (int e = tid; e < G.n_targets; e += BFS_THREADS) {
            const int pos = tpos[e];
            if (pos < 0) continue;
            const int r = pos / (BFS_WW * 32), c = pos - r * (BFS_WW * 32);
            const bool by_off = off && (on[r * BFS_WW + (c >> 5)] >> (c & 31) & 1u) && s <= (int)tkv[e];
            const bool by_front = fn[r * BFS_WW + (c >> 5)] >> (c & 31) & 1u;
            if (by_off || by_front) {
                tres[e] = (short)(s + 1);
                tpos[e] = -1;
            } else {
                my_open = 1;
            }
        }
        unsigned int *sw = fc; fc = fn; fn = sw;
        sw = oc; oc = on; on = sw;
        if (!(both & 1)) break;                                        // no front moved
    }
}

// Dtmp[((t*cap + i) * max_gap + g-1) * cap + j]: path length (cells) if <= dmax[g-1], 0 = no arc, -1 = needs exact search
__global__ __launch_bounds__(BFS_THREADS) void mask_bfs_kernel(
    const int *__restrict__ x, const int *__restrict__ y, const int *__restrict__ count, const int *__restrict__ src_count,
    int n_frames, int cap, const unsigned int *__restrict__ bits, const int *__restrict__ label, int H, int W, int Ww,
    int conn8, int max_dist, int max_gap, const int *__restrict__ dmax, short *__restrict__ Dtmp,
    const unsigned int *__restrict__ tight, const unsigned char *__restrict__ off_field, int n_comp, int off_mode_ok)
{
    extern __shared__ __attribute__((aligned(16))) unsigned int bsm[];
    unsigned int *A = bsm, *Bm = A + BFS_WH * BFS_WW, *M = Bm + BFS_WH * BFS_WW;
    // a fourth bitmap only when the launch provides it (off_mode_ok): sources off the mask run two wavefronts
    unsigned int *V2 = M + BFS_WH * BFS_WW;
    int *tpos = reinterpret_cast<int *>(M + (off_mode_ok ? 2 : 1) * BFS_WH * BFS_WW);      // [max_gap*cap] window bit position or -1
    short *tres = reinterpret_cast<short *>(tpos + max_gap * cap); // [max_gap*cap]
    short *tkv = tres + max_gap * cap;                             // [max_gap*cap] (off_mode_ok) off-cell count of the best path over the mask
    __shared__ int s_labels[8];
    __shared__ int n_slab, n_open;
    __shared__ int s_best_comp, s_best_a;
    __shared__ int s_step[3][2];          // per move (three in rotation): [0] some word changed, [1] some target is still open

    const int t = blockIdx.y, i = blockIdx.x, tid = threadIdx.x;
    if (i >= min(src_count[t], cap)) return;
    const int sx = x[(long)t * cap + i], sy = y[(long)t * cap + i];
    short *drow = Dtmp + ((long)t * cap + i) * max_gap * cap;
    const bool s_in = sx >= 0 && sx < W && sy >= 0 && sy < H;
    const int wy0 = sy - BFS_R, wx0 = ((sx - BFS_R) >> 5) << 5;     // window origin; columns word-aligned (floor)
    int depth = 0;
    for (int g = 0; g < max_gap; ++g) depth = max(depth, dmax[g] - 1);
    depth = min(depth, BFS_R);
    // A source ON the mask (component A) knows the fewest off-mask cells k(c) = d_off[A][c] of every cell, and the
    // minimum-cost paths are exactly the paths along which k grows by [cell off the mask] at every step ("tight"
    // steps: a prefix of an optimal path is optimal). Their fewest moves is a plain breadth-first search over tight
    // steps -- the same bit-parallel sweep, with one precomputed bit row per direction (axt_grid::d_tight) in place of
    // the mask -- and it serves every target, in whatever component or off the mask.
    const int ls_src = s_in ? label[(long)sy * W + sx] : 0;
    const bool tight_mode = tight != nullptr && ls_src > 0;
    const int ndir = conn8 ? 8 : 4;
    // A source OFF the mask (this round): the fewest off-mask cells of its paths are K(c) = min(P(c), min over
    // components A of a_A + d_off[A][c]) -- P = all-off paths (as many moves as off-mask cells), a_A = d_off[A][S] - 1 =
    // off-mask cells entered before the mask is first touched, in component A. Minimum-cost paths that touch the mask
    // first in A are: an all-off walk of exactly a_A cells, one step onto A, then tight steps of A's field. So two
    // wavefronts in the same loop: the all-off breadth-first search (dilation over off-mask cells), and from step
    // a_A + 1 on the tight-step search of the component A with the smallest a_A, seeded with the mask cells next to the
    // all-off front. A target is settled by the all-off front at step s iff s <= a_A + d_off[A][T] (an all-off path has
    // the fewest moves of all paths with that many off-mask cells), else by the other front. Targets for which another
    // component could do as well (a_B + d_off[B][T] <= a_A + d_off[A][T]) are left to the windowed search.
    const bool off_mode = off_mode_ok && tight != nullptr && off_field != nullptr && s_in && ls_src == 0;
    if (off_mode) {
        if (tid == 0) {
            int best = 0, ba = 0x7fffffff;
            for (int a = 0; a < n_comp; ++a) {
                const int v = off_field[((long)a * H + sy) * W + sx];
                if (v < 255 && v - 1 < ba) { ba = v - 1; best = a + 1; }
            }
            s_best_comp = best;
            s_best_a = ba;
        }
        __syncthreads();
    }
    const int best_comp = off_mode ? s_best_comp : 0, best_a = off_mode ? s_best_a : 0;
    const unsigned int *tg = tight_mode ? tight + (long)(ls_src - 1) * ndir * H * Ww
                             : (off_mode && best_comp > 0) ? tight + (long)(best_comp - 1) * ndir * H * Ww : nullptr;

    // ---- source labels
    if (tid == 0) {
        int n = 0;
        if (s_in) {
            const int ls = label[(long)sy * W + sx];
            if (ls) s_labels[n++] = ls;
            else {
                const int nn = conn8 ? 8 : 4;
                const int dy8[8] = {-1, 1, 0, 0, -1, -1, 1, 1}, dx8[8] = {0, 0, -1, 1, -1, 1, -1, 1};
                for (int d = 0; d < nn; ++d) {
                    const int ny = sy + dy8[d], nx = sx + dx8[d];
                    if (ny < 0 || ny >= H || nx < 0 || nx >= W) continue;
                    const int l = label[(long)ny * W + nx];
                    bool seen = l == 0;
                    for (int k = 0; k < n; ++k) seen |= s_labels[k] == l;
                    if (!seen) s_labels[n++] = l;
                }
            }
        }
        n_slab = n;
        n_open = 0;
        for (int k = 0; k < 3; ++k) s_step[k][0] = s_step[k][1] = 0;
    }
    // ---- window bitmaps
    for (int e = tid; e < BFS_WH * BFS_WW; e += BFS_THREADS) {
        const int r = e / BFS_WW, w = e - r * BFS_WW;
        const int gy = wy0 + r, gw = (wx0 >> 5) + w;
        M[e] = (off_mode || !(gy >= 0 && gy < H && gw >= 0 && gw < Ww)) ? 0u : bits[(long)gy * Ww + gw];   // off mode: M is the second front
        A[e] = 0u;
        Bm[e] = 0u;
        if (off_mode) V2[e] = 0u;
    }
    __syncthreads();
    // ---- targets of frames t+1 .. t+max_gap
    for (int e = tid; e < max_gap * cap; e += BFS_THREADS) {
        const int g = e / cap, j = e - g * cap, tb = t + g + 1;
        int pos = -1;
        short res = 0;
        if (tb < n_frames && j < min(count[tb], cap) && s_in) {
            const int tx = x[(long)tb * cap + j], ty = y[(long)tb * cap + j];
            const long dx = tx - sx, dy = ty - sy;
            const bool t_in = tx >= 0 && tx < W && ty >= 0 && ty < H;
            const int lim = dmax[g];
            const int lower = (conn8 ? (int)max(labs(dx), labs(dy)) : (int)(labs(dx) + labs(dy))) + 1;   // cells needed at least
            if (t_in && dx * dx + dy * dy < (long)max_dist * max_dist && lower <= lim) {
                if (dx == 0 && dy == 0) res = 1;
                else if (off_mode) {
                    int kv = 0x7fff;
                    bool ambiguous = false;
                    if (best_comp > 0) {
                        const int ft = off_field[((long)(best_comp - 1) * H + ty) * W + tx];
                        if (ft < 255) kv = best_a + ft;
                        for (int a = 0; a < n_comp && !ambiguous; ++a) {
                            if (a == best_comp - 1) continue;
                            const int fs = off_field[((long)a * H + sy) * W + sx], fb = off_field[((long)a * H + ty) * W + tx];
                            if (fs < 255 && fb < 255 && fs - 1 + fb <= kv) ambiguous = true;
                        }
                    }
                    if (ambiguous) res = -1;                          // the windowed search decides
                    else { pos = (ty - wy0) * (BFS_WW * 32) + (tx - wx0); atomicAdd(&n_open, 1); tkv[e] = (short)kv; }
                } else {
                    const int lt = label[(long)ty * W + tx];
                    bool level0 = tight_mode;
                    for (int k = 0; k < n_slab; ++k) level0 |= (lt != 0 && s_labels[k] == lt);
                    if (level0) { pos = (ty - wy0) * (BFS_WW * 32) + (tx - wx0); atomicAdd(&n_open, 1); }
                    else res = -1;                                    // exact search decides
                }
            }
        }
        tpos[e] = pos;
        tres[e] = res;
    }
    if (tid == 0 && s_in) {
        const int e = (sy - wy0) * BFS_WW + ((sx - wx0) >> 5);
        A[e] |= 1u << ((sx - wx0) & 31);
        if (!off_mode) M[e] |= 1u << ((sx - wx0) & 31);               // the seed counts even off the mask
    }
    __syncthreads();

    const int sr = BFS_R, sc = sx - wx0;                              // the source's window row / column
    if (off_mode || tight_mode) {
        // ---- searches over tight steps (source on the mask, or off it with the two fronts): every thread owns fixed words
        // of the window and keeps their tight-step rows in registers
        const BfsGeo geo{wy0, wx0, H, W, Ww, depth, sr, sc, max_gap * cap};
        if (conn8) bfs_owned<8>(off_mode, A, Bm, M, V2, tg, bits, geo, tpos, tres, tkv, s_step, best_comp, best_a, tid);
        else bfs_owned<4>(off_mode, A, Bm, M, V2, tg, bits, geo, tpos, tres, tkv, s_step, best_comp, best_a, tid);
        __syncthreads();
        for (int e = tid; e < max_gap * cap; e += BFS_THREADS) {
            const int g = e / cap;
            short r = tres[e];
            if (r > 0 && r > dmax[g]) r = 0;
            drow[e] = r;
        }
        return;
    }

    // ---- breadth-first search, one dilation per move
    unsigned int *cur = A, *nxt = Bm;
    int my_open = 0;                                                   // (one barrier per move, as above)
    for (int e = tid; e < max_gap * cap; e += BFS_THREADS) my_open |= (tpos[e] >= 0);
    for (int s = 1; s <= depth; ++s) {
        // after s moves only the cells within s of the source can be set: rows sr-s..sr+s, words of columns sc-s..sc+s
        const int rlo = max(0, sr - s), rhi = min(BFS_WH - 1, sr + s);
        const int wlo = max(0, (sc - s) >> 5), whi = min(BFS_WW - 1, (sc + s) >> 5), nw = whi - wlo + 1;
        int changed = 0;
        const float inv_nw = 1.0f / (float)nw;             // e < 9018, nw <= 18: (e + 0.5) * inv_nw truncates to e / nw exactly
        for (int e = tid; e < (rhi - rlo + 1) * nw; e += BFS_THREADS) {
            const int rr = (int)(((float)e + 0.5f) * inv_nw);
            const int r = rlo + rr, w = wlo + (e - rr * nw);
            const unsigned int *row = cur + r * BFS_WW;
            unsigned int v;
            if (conn8) {
                v = dil_h(row, w);
                if (r > 0) v |= dil_h(row - BFS_WW, w);
                if (r + 1 < BFS_WH) v |= dil_h(row + BFS_WW, w);
            } else {
                v = dil_h(row, w);
                if (r > 0) v |= row[w - BFS_WW];
                if (r + 1 < BFS_WH) v |= row[w + BFS_WW];
            }
            v &= M[r * BFS_WW + w];
            nxt[r * BFS_WW + w] = v;
            changed |= (v != row[w]);
        }
        if (changed) s_step[s % 3][0] = 1;
        if (my_open) s_step[s % 3][1] = 1;
        __syncthreads();
        const int both = s_step[s % 3][0] | (s_step[s % 3][1] << 1);
        if (tid == 0) s_step[(s + 2) % 3][0] = s_step[(s + 2) % 3][1] = 0;
        if (!(both & 2)) break;
        my_open = 0;
        for (int e = tid; e < max_gap * cap; e += BFS_THREADS) {
            const int pos = tpos[e];
            if (pos < 0) continue;
            const int r = pos / (BFS_WW * 32), c = pos - r * (BFS_WW * 32);
            if (nxt[r * BFS_WW + (c >> 5)] >> (c & 31) & 1u) {
                tres[e] = (short)(s + 1);
                tpos[e] = -1;
            } else {
                my_open = 1;
            }
        }
        unsigned int *sw = cur; cur = nxt; nxt = sw;
        if (!(both & 1)) break;
    }
    __syncthreads();
    for (int e = tid; e < max_gap * cap; e += BFS_THREADS) {
        const int g = e / cap;
        short r = tres[e];
        if (r > 0 && r > dmax[g]) r = 0;
        drow[e] = r;
    }
}

// ---- targets in another component of the mask -----------------------------------------------------------------------
// The minimum-cost path to such a target has to cross off-mask cells; its cost order is (off-mask cells entered,
// moves). The fewest off-mask cells k* any path needs is a property of the mask (axt_grid::d_off, one 0-1 search per
// component when the grid is created), so the per-source work shrinks to the (2R+1)^2 window that paths of <= dmax cells
// cannot leave: a label-correcting search on keys off << 16 | moves that does not expand cells at the move limit.
//   * optimum within the limit: every prefix of it is an optimal path of no more moves, inside the window -> the
//     search finds exactly its key, and its off-count equals k*;
//   * optimum longer than the limit: whatever the window search finds has a worse key but fewer moves, hence more
//     off-mask cells than k* -> recognised, no arc.
// k*: every path that touches the mask at all passes through some component A, so the fewest off-mask cells of such
// paths is kA = min over A of (d_off[A][S] - [S off the mask]) + d_off[A][T] (the first term by reversing the path
// A -> S); a source on the mask gives kA = k* = d_off[label(S)][T]. Paths that never touch the mask have as many moves
// as off-mask cells, so if one of them is the optimum it is short enough to be found by the window search itself.
// Hence: the window result (off, moves) is the global optimum iff off <= kA.
constexpr int CROSS_W = 2 * BFS_R + 1;
constexpr long CROSS_CELLS = (long)CROSS_W * CROSS_W;

__global__ __launch_bounds__(256) void mask_cross_kernel(
    const int *__restrict__ tasks, int n_tasks, const int *__restrict__ x, const int *__restrict__ y,
    const int *__restrict__ count, int n_frames, int cap, const unsigned char *__restrict__ mask,
    const int *__restrict__ label, const unsigned char *__restrict__ off_field, int n_comp, int H, int W, int conn8, int max_gap,
    const int *__restrict__ dmax, unsigned int *__restrict__ key_all, int *__restrict__ list_all,
    short *__restrict__ Dtmp)
{
    const int tid = threadIdx.x;
    unsigned int *key = key_all + (long)blockIdx.x * CROSS_CELLS;
    int *cur = list_all + (long)blockIdx.x * 2 * CROSS_CELLS, *nxt = cur + CROSS_CELLS;
    __shared__ int n_cur, n_nxt, overflow;
    __shared__ int off_s[64];                              // d_off[A][S] - [S off the mask] of the current source
    const int nn = conn8 ? 8 : 4;
    const int dy8[8] = {-1, 1, 0, 0, -1, -1, 1, 1}, dx8[8] = {0, 0, -1, 1, -1, 1, -1, 1};
    int depth = 0;
    for (int g = 0; g < max_gap; ++g) depth = max(depth, dmax[g] - 1);
    depth = min(depth, BFS_R);
    for (int task = blockIdx.x; task < n_tasks; task += gridDim.x) {
        const int src = tasks[task], t = src / cap;
        const int sx = x[src], sy = y[src];
        const int wy0 = sy - BFS_R, wx0 = sx - BFS_R;
        short *drow = Dtmp + (long)src * max_gap * cap;
        __syncthreads();                                   // the previous task's readers are done with key[]
        for (long c = tid; c < CROSS_CELLS; c += 256) key[c] = 0xffffffffu;
        __syncthreads();
        if (tid == 0) {
            const int s = BFS_R * CROSS_W + BFS_R;
            key[s] = 0;
            cur[0] = s;
            n_cur = 1;
            n_nxt = 0;
            overflow = 0;
        }
        __threadfence_block();
        __syncthreads();
        int *fa = cur, *fb = nxt;
        // keys only decrease, so the frontier empties; the cap is a belt-and-braces exit every wave reaches
        for (int iter = 0; iter < 4 * CROSS_W * CROSS_W; ++iter) {
            const int n = n_cur;
            if (n == 0) break;
            for (int e = tid; e < n; e += 256) {
                const int c = fa[e];
                const unsigned int k = __hip_atomic_load(&key[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if ((int)(k & 0xffffu) >= depth) continue;                       // at the move limit: not expanded
                const int cy = c / CROSS_W, cx = c - cy * CROSS_W;
                for (int d = 0; d < nn; ++d) {
                    const int ny = cy + dy8[d], nx = cx + dx8[d];
                    if (ny < 0 || ny >= CROSS_W || nx < 0 || nx >= CROSS_W) continue;
                    const int gy = wy0 + ny, gx = wx0 + nx;
                    if (gy < 0 || gy >= H || gx < 0 || gx >= W) continue;
                    const int nc = ny * CROSS_W + nx;
                    // most relaxations fail (a cell is improved once or twice but reached from every side): look first
                    const unsigned int seen = __hip_atomic_load(&key[nc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if ((seen >> 16) == 0u && seen <= k + 1u) continue;                  // cannot be beaten, whatever the cell is
                    const unsigned int nk = k + 1u + (mask[(long)gy * W + gx] == 1 ? 0u : 0x10000u);
                    if (seen <= nk) continue;
                    const unsigned int old = atomicMin(&key[nc], nk);
                    if (nk < old) {                  // a cell improved twice in one round is listed twice: harmless
                        const int slot = atomicAdd(&n_nxt, 1);
                        if (slot < (int)CROSS_CELLS) fb[slot] = nc; else overflow = 1;
                    }
                }
            }
            __threadfence_block();
            __syncthreads();
            if (tid == 0) { n_cur = min(n_nxt, (int)CROSS_CELLS); n_nxt = 0; }
            int *sw = fa; fa = fb; fb = sw;
            __syncthreads();
        }
        __syncthreads();
        if (tid < n_comp)
            off_s[tid] = (int)off_field[((long)tid * H + sy) * W + sx] - (mask[(long)sy * W + sx] == 1 ? 0 : 1);
        __syncthreads();
        for (int e = tid; e < max_gap * cap; e += 256) {
            if (drow[e] != -1 || overflow) continue;              // (a list overflowed: the general search takes the source)
            const int g = e / cap, j = e - g * cap, tb = t + g + 1;
            const int tx = x[(long)tb * cap + j], ty = y[(long)tb * cap + j];
            int ka = 0x7fffffff;
            for (int a = 0; a < n_comp; ++a) ka = min(ka, off_s[a] + (int)off_field[((long)a * H + ty) * W + tx]);
            const unsigned int k = key[(long)(ty - wy0) * CROSS_W + (tx - wx0)];
            short res = 0;
            if (k != 0xffffffffu && (int)(k >> 16) <= ka && (int)(k & 0xffffu) + 1 <= dmax[g]) res = (short)((k & 0xffffu) + 1);
            drow[e] = res;
        }
    }
}

// per source: does any target need the exact search?
__global__ void mask_flag_kernel(const short *__restrict__ Dtmp, const int *__restrict__ count, int n_frames, int cap,
                                 int max_gap, int *__restrict__ flags, int *__restrict__ n_flagged)
{
    const int t = blockIdx.y, i = blockIdx.x;
    if (i >= min(count[t], cap)) { if (threadIdx.x == 0) flags[(long)t * cap + i] = 0; return; }
    const short *drow = Dtmp + ((long)t * cap + i) * max_gap * cap;
    int any = 0;
    for (int e = threadIdx.x; e < max_gap * cap; e += blockDim.x) any |= (drow[e] < 0);
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) {
        flags[(long)t * cap + i] = any;
        if (any) atomicAdd(n_flagged, 1);
    }
}

// after the exact search of one source against one frame: replace the -1 marks
__global__ void mask_patch_kernel(short *__restrict__ drow, const int *__restrict__ Dexact, int nb, int lim)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nb) return;
    if (drow[j] < 0) drow[j] = (Dexact[j] <= lim) ? (short)Dexact[j] : (short)0;
}

}  // namespace

extern "C" int axt_grid_create(const uint8_t *h_mask, int H, int W, int conn8, axt_grid **out)
{
    AXT_REQUIRE(h_mask && out && H > 0 && W > 0, "bad argument");
    axt_grid *g = new (std::nothrow) axt_grid();
    if (!g) return AXT_ENOMEM;
    g->H = H; g->W = W; g->Ww = (W + 31) / 32; g->conn8 = conn8 ? 1 : 0;
    std::vector<unsigned int> bits((size_t)H * g->Ww, 0u);
    std::vector<int> label((size_t)H * W, 0);
    std::vector<unsigned char> m01((size_t)H * W);
    for (long k = 0; k < (long)H * W; ++k) m01[k] = h_mask[k] == 1;          // AxonDetections.py:598: mask == 1
    for (int yy = 0; yy < H; ++yy)
        for (int xx = 0; xx < W; ++xx)
            if (m01[(size_t)yy * W + xx]) bits[(size_t)yy * g->Ww + (xx >> 5)] |= 1u << (xx & 31);
    // connected components by flood fill
    std::vector<int> stack;
    int next = 0;
    const int nn = conn8 ? 8 : 4;
    const int dy8[8] = {-1, 1, 0, 0, -1, -1, 1, 1}, dx8[8] = {0, 0, -1, 1, -1, 1, -1, 1};
    for (long k = 0; k < (long)H * W; ++k) {
        if (!m01[k] || label[k]) continue;
        label[k] = ++next;
        stack.push_back((int)k);
        while (!stack.empty()) {
            const int c = stack.back();
            stack.pop_back();
            const int cy = c / W, cx = c % W;
            for (int d = 0; d < nn; ++d) {
                const int ny = cy + dy8[d], nx = cx + dx8[d];
                if (ny < 0 || ny >= H || nx < 0 || nx >= W) continue;
                const int n = ny * W + nx;
                if (m01[n] && !label[n]) { label[n] = next; stack.push_back(n); }
            }
        }
    }
    // fewest off-mask cells from every component to every cell (0-1 breadth-first search per component)
    g->n_comp = next;
    std::vector<unsigned char> off;
    constexpr int kMaxComp = 64;
    if (next >= 1 && next <= kMaxComp && (size_t)next * H * W <= ((size_t)256 << 20)) {
        off.assign((size_t)next * H * W, 255);
        std::vector<int> dist((size_t)H * W);
        std::vector<int> level, later, work;
        for (int L = 1; L <= next; ++L) {
            std::fill(dist.begin(), dist.end(), INT32_MAX);
            level.clear();
            for (long k = 0; k < (long)H * W; ++k)
                if (label[k] == L) { dist[k] = 0; level.push_back((int)k); }
            // Dial's buckets for weights {0, 1}: close the current level over the zero-weight (on-mask) moves, collect
            // the off-mask cells one level up; 255 levels are all a path of <= 251 cells can use
            for (int d = 0; d < 255 && !level.empty(); ++d) {
                work.swap(level);
                later.clear();
                while (!work.empty()) {
                    const int c = work.back();
                    work.pop_back();
                    if (dist[c] != d) continue;
                    const int cy = c / W, cx = c % W;
                    for (int q = 0; q < nn; ++q) {
                        const int ny = cy + dy8[q], nx = cx + dx8[q];
                        if (ny < 0 || ny >= H || nx < 0 || nx >= W) continue;
                        const int n = ny * W + nx;
                        const int nd = d + (m01[n] ? 0 : 1);
                        if (nd < dist[n]) {
                            dist[n] = nd;
                            if (nd == d) work.push_back(n); else later.push_back(n);
                        }
                    }
                }
                level.clear();
                for (int n : later)
                    if (dist[n] == d + 1) level.push_back(n);
            }
            unsigned char *o = off.data() + (size_t)(L - 1) * H * W;
            for (long k = 0; k < (long)H * W; ++k)
                if (dist[k] < 255) o[k] = (unsigned char)dist[k];
        }
    }
    g->has_fields = next == 0 || !off.empty();
    std::vector<unsigned int> tightv;
    if (!off.empty()) {
        const int Ww = g->Ww;
        const int oy8[8] = {-1, 1, 0, 0, -1, -1, 1, 1}, ox8[8] = {0, 0, -1, 1, -1, 1, -1, 1};
        tightv.assign((size_t)next * nn * H * Ww, 0u);
        for (int L = 0; L < next; ++L) {
            const unsigned char *o = off.data() + (size_t)L * H * W;
            for (int d = 0; d < nn; ++d) {
                unsigned int *tb = tightv.data() + ((size_t)L * nn + d) * H * Ww;
                for (int yy = 0; yy < H; ++yy) {
                    const int py = yy + oy8[d];
                    if (py < 0 || py >= H) continue;
                    for (int xx = 0; xx < W; ++xx) {
                        const int px = xx + ox8[d];
                        if (px < 0 || px >= W) continue;
                        const int kc = o[(size_t)yy * W + xx], kp = o[(size_t)py * W + px];
                        if (kc < 255 && kp < 255 && kc == kp + (m01[(size_t)yy * W + xx] ? 0 : 1))
                            tb[(size_t)yy * Ww + (xx >> 5)] |= 1u << (xx & 31);
                    }
                }
            }
        }
    }
    int rc = AXT_OK;
    if (!off.empty() && (hipMalloc((void **)&g->d_off, off.size()) != hipSuccess ||
                         hipMemcpy(g->d_off, off.data(), off.size(), hipMemcpyHostToDevice) != hipSuccess)) {
        axt_set_error("axt_grid_create: device allocation for the component distance fields failed");
        axt_grid_destroy(g);
        return AXT_ENOMEM;
    }
    if (!tightv.empty() && (hipMalloc((void **)&g->d_tight, tightv.size() * 4) != hipSuccess ||
                            hipMemcpy(g->d_tight, tightv.data(), tightv.size() * 4, hipMemcpyHostToDevice) != hipSuccess)) {
        axt_set_error("axt_grid_create: device allocation for the tight-step rows failed");
        axt_grid_destroy(g);
        return AXT_ENOMEM;
    }
    if (hipMalloc((void **)&g->d_mask, (size_t)H * W) != hipSuccess || hipMalloc((void **)&g->d_bits, bits.size() * 4) != hipSuccess ||
        hipMalloc((void **)&g->d_label, label.size() * 4) != hipSuccess) {
        axt_set_error("axt_grid_create: device allocation failed");
        rc = AXT_ENOMEM;
    } else if (hipMemcpy(g->d_mask, m01.data(), m01.size(), hipMemcpyHostToDevice) != hipSuccess ||
               hipMemcpy(g->d_bits, bits.data(), bits.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
               hipMemcpy(g->d_label, label.data(), label.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        axt_set_error("axt_grid_create: upload failed");
        rc = AXT_EHIP;
    }
    if (rc) { axt_grid_destroy(g); return rc; }
    *out = g;
    return AXT_OK;
}

extern "C" void axt_grid_destroy(axt_grid *g)
{
    if (!g) return;
    (void)hipFree(g->d_mask);
    (void)hipFree(g->d_bits);
    (void)hipFree(g->d_label);
    (void)hipFree(g->d_off);
    (void)hipFree(g->d_tight);
    delete g;
}

extern "C" const uint8_t *axt_grid_mask(const axt_grid *g) { return g ? g->d_mask : nullptr; }

// Fills Dtmp (layout above) for every source detection; exact-search fallback included. Synchronises the stream.
int axt_masked_distance_table(const axt_grid *g, const int32_t *d_x, const int32_t *d_y, const int32_t *d_count,
                              const int32_t *d_src_count, int n_frames, int cap, int max_dist, int max_gap,
                              const int32_t *h_dmax, const int32_t *d_dmax, int16_t *d_Dtmp, hipStream_t st)
{
    for (int k = 0; k < max_gap; ++k)
        AXT_REQUIRE(h_dmax[k] - 1 <= BFS_R, "masked arcs: dmax %d exceeds the BFS window (%d cells)", h_dmax[k], BFS_R + 1);
    // sources off the mask run two wavefronts (a fourth bitmap and a third per-target array) when that fits the LDS and
    // the mask has its component fields; otherwise they take the windowed search as before
    const size_t lds4 = (size_t)4 * BFS_WH * BFS_WW * 4 + (size_t)max_gap * cap * 8 + 16;
    const int off_mode_ok = (g->d_tight != nullptr && g->d_off != nullptr && g->n_comp >= 1 && lds4 <= 159 * 1024 &&
                             !getenv("AXT_PATH_NO_OFFMODE")) ? 1 : 0;
    const size_t lds = off_mode_ok ? lds4 : (size_t)3 * BFS_WH * BFS_WW * 4 + (size_t)max_gap * cap * 6 + 16;
    AXT_REQUIRE(lds <= 159 * 1024, "masked arcs: cap %d needs %zu bytes of LDS", cap, lds);
    static AxtOncePerDevice once;                 // (per device: see axt_common.h)
    if (int rc = axt_max_dynamic_lds(mask_bfs_kernel, 159 * 1024, once)) return rc;
    hipLaunchKernelGGL(mask_bfs_kernel, dim3(cap, n_frames), dim3(BFS_THREADS), lds, st, d_x, d_y, d_count, d_src_count, n_frames, cap, g->d_bits,
                       g->d_label, g->H, g->W, g->Ww, g->conn8, max_dist, max_gap, d_dmax, d_Dtmp, (const unsigned int *)g->d_tight,
                       (const unsigned char *)g->d_off, g->n_comp, off_mode_ok);
    AXT_LAUNCH_CHECK();
    int *flags = nullptr, *n_flagged = nullptr;
    AXT_CHECK_HIP(hipMallocAsync((void **)&flags, sizeof(int) * ((size_t)n_frames * cap + 1), st));
    n_flagged = flags + (size_t)n_frames * cap;
    AXT_CHECK_HIP(hipMemsetAsync(n_flagged, 0, sizeof(int), st));
    hipLaunchKernelGGL(mask_flag_kernel, dim3(cap, n_frames), dim3(64), 0, st, d_Dtmp, d_src_count, n_frames, cap, max_gap, flags,
                       n_flagged);
    AXT_LAUNCH_CHECK();
    int nf = 0;
    AXT_CHECK_HIP(hipMemcpyAsync(&nf, n_flagged, sizeof(int), hipMemcpyDeviceToHost, st));
    AXT_CHECK_HIP(hipStreamSynchronize(st));
    if (nf > 0) {
        std::vector<int> hf((size_t)n_frames * cap), hc(n_frames);
        AXT_CHECK_HIP(hipMemcpy(hf.data(), flags, hf.size() * 4, hipMemcpyDeviceToHost));
        AXT_CHECK_HIP(hipMemcpy(hc.data(), d_count, (size_t)n_frames * 4, hipMemcpyDeviceToHost));
        if (g->has_fields) {
            // targets in other components / off the mask: windowed search of every flagged source (resolves them all)
            std::vector<int> tasks;
            for (int t = 0; t < n_frames; ++t)
                for (int i = 0; i < hc[t] && i < cap; ++i)
                    if (hf[(size_t)t * cap + i]) tasks.push_back(t * cap + i);
            const int n_tasks = (int)tasks.size();
            if (getenv("AXT_PATH_DEBUG")) fprintf(stderr, "masked arcs: %d sources off the mask or without component fields (windowed search)\n", n_tasks);
            const int wgs = n_tasks < 512 ? n_tasks : 512;       // two per CU; more only thrash the caches (measured)
            int *d_tasks = nullptr;
            unsigned char *scratch = nullptr;
            AXT_CHECK_HIP(hipMallocAsync((void **)&d_tasks, sizeof(int) * n_tasks, st));
            AXT_CHECK_HIP(hipMallocAsync((void **)&scratch, (size_t)wgs * CROSS_CELLS * 12, st));
            AXT_CHECK_HIP(hipMemcpyAsync(d_tasks, tasks.data(), sizeof(int) * n_tasks, hipMemcpyHostToDevice, st));
            unsigned int *key = reinterpret_cast<unsigned int *>(scratch);
            int *lists = reinterpret_cast<int *>(scratch + (size_t)wgs * CROSS_CELLS * 4);
            const bool timed = getenv("AXT_PATH_DEBUG") != nullptr;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (timed) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, st); }
            hipLaunchKernelGGL(mask_cross_kernel, dim3(wgs), dim3(256), 0, st, (const int *)d_tasks, n_tasks, d_x, d_y, d_count,
                               n_frames, cap, g->d_mask, (const int *)g->d_label, (const unsigned char *)g->d_off, g->n_comp, g->H, g->W,
                               g->conn8, max_gap, d_dmax, key, lists, d_Dtmp);
            AXT_LAUNCH_CHECK();
            if (timed) {
                float ms = 0;
                (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
                fprintf(stderr, "masked arcs: windowed search of %d sources took %.1f ms\n", n_tasks, ms);
                (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
            }
            AXT_CHECK_HIP(hipMemsetAsync(n_flagged, 0, sizeof(int), st));
            hipLaunchKernelGGL(mask_flag_kernel, dim3(cap, n_frames), dim3(64), 0, st, d_Dtmp, d_src_count, n_frames, cap, max_gap, flags,
                               n_flagged);
            AXT_LAUNCH_CHECK();
            AXT_CHECK_HIP(hipMemcpyAsync(&nf, n_flagged, sizeof(int), hipMemcpyDeviceToHost, st));
            AXT_CHECK_HIP(hipStreamSynchronize(st));                 // also: tasks[] may go out of scope
            AXT_CHECK_HIP(hipFreeAsync(d_tasks, st));
            AXT_CHECK_HIP(hipFreeAsync(scratch, st));
            if (nf > 0) AXT_CHECK_HIP(hipMemcpy(hf.data(), flags, hf.size() * 4, hipMemcpyDeviceToHost));
        }
    }
    if (nf > 0) {
        if (getenv("AXT_PATH_DEBUG")) fprintf(stderr, "masked arcs: %d sources left for the general search\n", nf);
        std::vector<int> hf((size_t)n_frames * cap), hc(n_frames);
        AXT_CHECK_HIP(hipMemcpy(hf.data(), flags, hf.size() * 4, hipMemcpyDeviceToHost));
        AXT_CHECK_HIP(hipMemcpy(hc.data(), d_count, (size_t)n_frames * 4, hipMemcpyDeviceToHost));
        int *dex = nullptr;
        AXT_CHECK_HIP(hipMalloc((void **)&dex, sizeof(int) * cap));
        for (int t = 0; t < n_frames; ++t)
            for (int i = 0; i < hc[t] && i < cap; ++i) {
                if (!hf[(size_t)t * cap + i]) continue;
                for (int gp = 0; gp < max_gap; ++gp) {
                    const int tb = t + gp + 1;
                    if (tb >= n_frames) continue;
                    const int nb = hc[tb] < cap ? hc[tb] : cap;
                    if (nb == 0) continue;
                    int rc = axt_path_cost_masked(d_x + (size_t)t * cap + i, d_y + (size_t)t * cap + i, 1, d_x + (size_t)tb * cap,
                                                  d_y + (size_t)tb * cap, nb, g->d_mask, g->H, g->W, max_dist, g->conn8, dex, st, nullptr);
                    if (rc) { (void)hipFree(dex); return rc; }
                    hipLaunchKernelGGL(mask_patch_kernel, dim3((nb + 255) / 256), dim3(256), 0, st,
                                       d_Dtmp + (((size_t)t * cap + i) * max_gap + gp) * cap, (const int *)dex, nb, h_dmax[gp]);
                }
            }
        AXT_CHECK_HIP(hipStreamSynchronize(st));
        (void)hipFree(dex);
    }
    AXT_CHECK_HIP(hipFreeAsync(flags, st));
    return AXT_OK;
}
