// Path lengths on a masked grid: weights {1 on mask, 65536 off} (reference AxonDetections.py:598), the A* of
// utils.py:379 (pyastar2d, absent from the reference tree -- convention in DESIGN.md).
//
// For one source detection the lengths to ALL targets come from a single-source search, instead of one A* per
// (source, target) pair as the reference does (AxonDetections.py:570-576): one workgroup per source.
//
// Definition (identical to the CPU checker's): inside the window of half-width max_dist around the source,
// find the minimum-cost 4-/8-connected path to each target, cost of a move = weight of the cell moved into.
// With weights {1, 65536} and fewer than 65536 on-mask moves the cost order equals the lexicographic order of
// (off-mask cells entered, moves), packed here as the 64-bit key off << 32 | moves (moves can exceed 16 bits in
// a 1001^2 window). The result is moves + 1 cells, or max_dist ("None")
// when that exceeds max_dist, the euclidean gate fails or an end point lies outside the grid.
//
// Search: frontier label-correcting (parallel Bellman-Ford over worklists). Every cell of the current frontier
// relaxes its neighbours with atomicMin on the 64-bit key; improved cells enter the next frontier once
// (stamp array). Keys only decrease and the iteration ends when no key changes, so the fixed point is the exact
// shortest-path key of every cell, independent of scheduling. First correct version: the frontiers live in HBM
// and each step costs two workgroup barriers (a bit-parallel LDS variant is the planned optimisation).
#include "axt_common.h"

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
    const int y0 = max(sy - max_dist, 0), y1 = min(sy + max_dist, H - 1);
    const int x0 = max(sx - max_dist, 0), x1 = min(sx + max_dist, W - 1);
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

}  // namespace

int axt_path_cost_masked(const int32_t *d_xa, const int32_t *d_ya, int na, const int32_t *d_xb, const int32_t *d_yb,
                         int nb, const uint8_t *d_mask, int H, int W, int max_dist, int conn8, int32_t *d_D,
                         hipStream_t st)
{
    const long win = (long)(2 * max_dist + 1 < H ? 2 * max_dist + 1 : H) * (2 * max_dist + 1 < W ? 2 * max_dist + 1 : W);
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
    }
    AXT_CHECK_HIP(hipFreeAsync(raw, st));
    return AXT_OK;
}
