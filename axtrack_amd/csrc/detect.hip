// YOLO grid -> per-frame detections on gfx950, plus the tile-occupancy scan.
//
// Replaces the per-frame pandas pipeline of AxonDetections.detect_dataset
// (reference axtrack/AxonDetections.py:111-133):
//   _yolo_Y2pandas_det (:178-248)  ->  decode + threshold
//   Timelapse.stitch_tiles (Timelapse.py:166-197)  ->  + tile origin
//   _non_max_supression (:250-278)  ->  greedy centre-distance NMS
// One workgroup per frame; everything stays in LDS. All arithmetic that decides an integer
// (anchor rounding, the 0.55 cut, dx^2+dy^2 < 529) is done exactly as the reference does it.
#include "axt_common.h"

namespace {

struct TileOrigins { int n; short yx[2 * 256]; };

// ------------------------------------------------------------------------------------------------
// tile occupancy: occ[tile] = any(frames[:, tile] > 0)        (Timelapse.py:551-558)
// grid: (row groups of 64, T_all); block: 256 threads striding over x. A block returns at once when every
// tile of its tile row is already known to be occupied, so after the first few frames the scan costs nothing
// (in real data nearly every tile is occupied); the worst case reads the timelapse once.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_occupancy_kernel(const float *__restrict__ frames, int H, int W, int ntx,
                                                             unsigned int *__restrict__ occ_words, int t_first)
{
    constexpr int ROWS = 64;
    const int y0 = blockIdx.x * ROWS, t = t_first + blockIdx.y;
    const int ty = y0 / AXT_TILE;
    volatile unsigned int *occ = occ_words + ty * ntx;
    __shared__ int todo;
    if (threadIdx.x == 0) {
        int n = 0;
        for (int tx = 0; tx < ntx; ++tx) n += (occ[tx] == 0);
        todo = n;
    }
    __syncthreads();
    if (todo == 0) return;
    for (int tx = 0; tx < ntx; ++tx) {
        if (occ[tx]) continue;                       // racy read: a stale 0 only costs a redundant scan
        const int x0 = tx * AXT_TILE, x1 = min(W, x0 + AXT_TILE);
        int any = 0;
        for (int r = 0; r < ROWS && y0 + r < H; ++r) {
            const float *row = frames + ((long)t * H + y0 + r) * W;
            for (int x = x0 + threadIdx.x; x < x1; x += 256) any |= (row[x] > 0.f);
        }
        if (__any(any) && (threadIdx.x & 63) == 0) atomicOr(&occ_words[ty * ntx + tx], 1u);
    }
}

__global__ void occ_words_to_bytes(const unsigned int *__restrict__ w, uint8_t *__restrict__ out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = w[i] ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// decode + stitch + NMS, one workgroup (256 threads) per frame
// LDS (dynamic): cand conf/x/y [ncand], sorted conf/x/y [ncand], state [ncand]
// ------------------------------------------------------------------------------------------------
constexpr int ST_UNDECIDED = 0, ST_ALIVE = 1, ST_DEAD = 2;

__global__ __launch_bounds__(256) void decode_stitch_nms_kernel(
    const float *__restrict__ yolo, int n_tiles, TileOrigins tiles, float conf_thr, int thr2, int cap,
    float *__restrict__ o_conf, int *__restrict__ o_x, int *__restrict__ o_y, int *__restrict__ o_count)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int ncand = n_tiles * AXT_CELLS;
    float *c_conf = reinterpret_cast<float *>(smem_raw);
    int *c_x = reinterpret_cast<int *>(c_conf + ncand);
    int *c_y = c_x + ncand;
    float *s_conf = reinterpret_cast<float *>(c_y + ncand);
    int *s_x = reinterpret_cast<int *>(s_conf + ncand);
    int *s_y = s_x + ncand;
    int *state = s_y + ncand;
    __shared__ int n_kept;

    const int frame = blockIdx.x, tid = threadIdx.x;
    const float *yf = yolo + (long)frame * n_tiles * AXT_YOLO_FLOATS;
    if (tid == 0) n_kept = 0;

    // ---- decode (AxonDetections.py:192-210) and threshold (:212-220)
    for (int c = tid; c < ncand; c += 256) {
        const int k = c / AXT_CELLS, cell = c - k * AXT_CELLS;
        const int i = cell / AXT_S, j = cell - i * AXT_S;      // dim1 = x cell, dim2 = y cell
        const float conf = yf[c * 3 + 0], xin = yf[c * 3 + 1], yin = yf[c * 3 + 2];
        const bool zero = (conf == 0.f) && (xin == 0.f) && (yin == 0.f);
        // x = round_half_even(((x_in + i) * tilesize) / Sx), all in f32
        float xf = rintf(__fdiv_rn(__fmul_rn(__fadd_rn(xin, (float)i), (float)AXT_TILE), (float)AXT_S));
        float yf2 = rintf(__fdiv_rn(__fmul_rn(__fadd_rn(yin, (float)j), (float)AXT_TILE), (float)AXT_S));
        if (zero) { xf = 0.f; yf2 = 0.f; }
        c_conf[c] = conf;
        c_x[c] = (int)xf + tiles.yx[2 * k + 1] * AXT_TILE;      // stitch: Timelapse.py:191-192
        c_y[c] = (int)yf2 + tiles.yx[2 * k] * AXT_TILE;
    }
    __syncthreads();

    // ---- order by descending confidence; ties keep (tile, cell) order. rank by counting.
    for (int c = tid; c < ncand; c += 256) {
        const float conf = c_conf[c];
        if (!(conf >= conf_thr)) continue;                      // NaN fails the comparison, as in torch
        int rank = 0;
        for (int o = 0; o < ncand; ++o) {
            const float oc = c_conf[o];
            if (!(oc >= conf_thr)) continue;
            rank += (oc > conf) || (oc == conf && o < c);
        }
        s_conf[rank] = conf;
        s_x[rank] = c_x[c];
        s_y[rank] = c_y[c];
        state[rank] = ST_UNDECIDED;
        atomicAdd(&n_kept, 1);
    }
    __syncthreads();
    const int n = n_kept;

    // ---- greedy NMS (AxonDetections.py:261-274) resolved in parallel rounds: row i dies iff an
    // earlier ALIVE row lies within dx^2+dy^2 < thr2; it is alive once every earlier row within
    // that distance is known dead. The fixed point is unique (= the sequential result).
    for (int round = 0; round <= n; ++round) {
        int pending_any = 0;
        for (int i = tid; i < n; i += 256) {
            if (state[i] != ST_UNDECIDED) continue;
            const int xi = s_x[i], yi = s_y[i];
            int st = ST_ALIVE;
            for (int j = 0; j < i; ++j) {
                const int dx = s_x[j] - xi, dy = s_y[j] - yi;
                if (dx * dx + dy * dy < thr2) {
                    const int sj = state[j];
                    if (sj == ST_ALIVE) { st = ST_DEAD; break; }
                    if (sj == ST_UNDECIDED) st = ST_UNDECIDED;
                }
            }
            if (st == ST_UNDECIDED) pending_any = 1;
            else state[i] = st;
        }
        if (!__syncthreads_or(pending_any)) break;
    }
    __syncthreads();

    // ---- compact survivors in order (block-wide exclusive scan over `alive`)
    __shared__ int wave_tot[4];
    __shared__ int base_sh;
    if (tid == 0) base_sh = 0;
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        const int alive = (i < n) && (state[i] == ST_ALIVE);
        const unsigned long long m = __ballot(alive);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(m);
        __syncthreads();
        int off = base_sh;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        if (alive) {
            const long o = (long)frame * cap + off + before;
            o_conf[o] = s_conf[i];
            o_x[o] = s_x[i];
            o_y[o] = s_y[i];
        }
        __syncthreads();
        if (tid == 0) base_sh += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
    if (tid == 0) o_count[frame] = base_sh;
}

// ------------------------------------------------------------------------------------------------
// The same for frames of more than 28 kept tiles (4096 x 4096 = 64 tiles, up to the 256 the detector takes): the
// candidate arrays live in a per-frame workspace in HBM instead of LDS, and only the candidates that pass the
// threshold are ranked (the rank is a count over pairs: quadratic in what is ranked). Same order, same arithmetic, same
// result as the LDS kernel -- the reference has no size limit (AxonDetections.py:111-133).
// workspace per frame: kept conf/x/y/idx [ncand], sorted conf/x/y [ncand], state [ncand]  (8 x ncand x 4 bytes)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decode_stitch_nms_big_kernel(
    const float *__restrict__ yolo, int n_tiles, TileOrigins tiles, float conf_thr, int thr2, int cap,
    float *__restrict__ o_conf, int *__restrict__ o_x, int *__restrict__ o_y, int *__restrict__ o_count,
    unsigned char *__restrict__ work)
{
    const int ncand = n_tiles * AXT_CELLS;
    const int frame = blockIdx.x, tid = threadIdx.x;
    float *k_conf = reinterpret_cast<float *>(work + (size_t)frame * ncand * 32);
    int *k_x = reinterpret_cast<int *>(k_conf + ncand);
    int *k_y = k_x + ncand;
    int *k_idx = k_y + ncand;
    float *s_conf = reinterpret_cast<float *>(k_idx + ncand);
    int *s_x = reinterpret_cast<int *>(s_conf + ncand);
    int *s_y = s_x + ncand;
    int *state = s_y + ncand;
    __shared__ int wave_tot[4], base_sh;
    const float *yf = yolo + (long)frame * n_tiles * AXT_YOLO_FLOATS;
    const int lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_sh = 0;
    __syncthreads();

    // ---- decode, threshold, keep in (tile, cell) order (block-wide scan per 256 candidates)
    for (int c0 = 0; c0 < ncand; c0 += 256) {
        const int c = c0 + tid;
        float conf = 0.f;
        int px = 0, py = 0;
        bool keep = false;
        if (c < ncand) {
            const int k = c / AXT_CELLS, cell = c - k * AXT_CELLS;
            const int i = cell / AXT_S, j = cell - i * AXT_S;
            conf = yf[c * 3 + 0];
            const float xin = yf[c * 3 + 1], yin = yf[c * 3 + 2];
            const bool zero = (conf == 0.f) && (xin == 0.f) && (yin == 0.f);
            float xf = rintf(__fdiv_rn(__fmul_rn(__fadd_rn(xin, (float)i), (float)AXT_TILE), (float)AXT_S));
            float yf2 = rintf(__fdiv_rn(__fmul_rn(__fadd_rn(yin, (float)j), (float)AXT_TILE), (float)AXT_S));
            if (zero) { xf = 0.f; yf2 = 0.f; }
            px = (int)xf + tiles.yx[2 * k + 1] * AXT_TILE;
            py = (int)yf2 + tiles.yx[2 * k] * AXT_TILE;
            keep = conf >= conf_thr;
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wave_tot[wave] = __popcll(m);
        __syncthreads();
        int off = base_sh;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        if (keep) {
            const int o = off + __popcll(m & ((1ull << lane) - 1ull));
            k_conf[o] = conf; k_x[o] = px; k_y[o] = py; k_idx[o] = c;
        }
        __syncthreads();
        if (tid == 0) base_sh += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
    const int n = base_sh;
    __syncthreads();
    if (tid == 0) base_sh = 0;

    // ---- rank by descending confidence, ties in (tile, cell) order (kept candidates are already in that order)
    for (int c = tid; c < n; c += 256) {
        const float conf = k_conf[c];
        int rank = 0;
        for (int o = 0; o < n; ++o) {
            const float oc = k_conf[o];
            rank += (oc > conf) || (oc == conf && o < c);
        }
        s_conf[rank] = conf;
        s_x[rank] = k_x[c];
        s_y[rank] = k_y[c];
        state[rank] = ST_UNDECIDED;
    }
    __threadfence_block();
    __syncthreads();

    // ---- greedy NMS in parallel rounds (as above)
    for (int round = 0; round <= n; ++round) {
        int pending_any = 0;
        for (int i = tid; i < n; i += 256) {
            if (__hip_atomic_load(&state[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != ST_UNDECIDED) continue;
            const int xi = s_x[i], yi = s_y[i];
            int st = ST_ALIVE;
            for (int j = 0; j < i; ++j) {
                const int dx = s_x[j] - xi, dy = s_y[j] - yi;
                if (dx * dx + dy * dy < thr2) {
                    const int sj = __hip_atomic_load(&state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (sj == ST_ALIVE) { st = ST_DEAD; break; }
                    if (sj == ST_UNDECIDED) st = ST_UNDECIDED;
                }
            }
            if (st == ST_UNDECIDED) pending_any = 1;
            else __hip_atomic_store(&state[i], st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __threadfence_block();
        if (!__syncthreads_or(pending_any)) break;
    }
    __syncthreads();

    // ---- compact survivors in order
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        const int alive = (i < n) && (state[i] == ST_ALIVE);
        const unsigned long long m = __ballot(alive);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(m);
        __syncthreads();
        int off = base_sh;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        if (alive) {
            const long o = (long)frame * cap + off + before;
            o_conf[o] = s_conf[i];
            o_x[o] = s_x[i];
            o_y[o] = s_y[i];
        }
        __syncthreads();
        if (tid == 0) base_sh += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
    if (tid == 0) o_count[frame] = base_sh;
}

}  // namespace

extern "C" {

int axt_tile_occupancy(const float *d_frames, int T_all, int H, int W, uint8_t *d_occ, void *stream)
{
    AXT_REQUIRE(d_frames && d_occ, "null argument");
    AXT_REQUIRE(T_all > 0 && H > 0 && W > 0, "bad shape %dx%dx%d", T_all, H, W);
    hipStream_t st = (hipStream_t)stream;
    const int nty = axt_cdiv(H, AXT_TILE), ntx = axt_cdiv(W, AXT_TILE);
    unsigned int *words = nullptr;
    AXT_CHECK_HIP(hipMallocAsync((void **)&words, sizeof(unsigned int) * nty * ntx, st));
    AXT_CHECK_HIP(hipMemsetAsync(words, 0, sizeof(unsigned int) * nty * ntx, st));
    // two launches: the first few frames usually mark every tile, and the blocks of the second launch then
    // return at once (their early-exit test runs after the first launch has completed)
    const int t_head = T_all < 4 ? T_all : 4;
    hipLaunchKernelGGL(tile_occupancy_kernel, dim3(axt_cdiv(H, 64), t_head), dim3(256), 0, st, d_frames, H, W, ntx, words, 0);
    AXT_LAUNCH_CHECK();
    if (T_all > t_head) {
        hipLaunchKernelGGL(tile_occupancy_kernel, dim3(axt_cdiv(H, 64), T_all - t_head), dim3(256), 0, st, d_frames, H, W,
                           ntx, words, t_head);
        AXT_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(occ_words_to_bytes, dim3(axt_cdiv(nty * ntx, 256)), dim3(256), 0, st, words, d_occ, nty * ntx);
    AXT_LAUNCH_CHECK();
    AXT_CHECK_HIP(hipFreeAsync(words, st));
    return AXT_OK;
}

int axt_decode_stitch_nms(const float *d_yolo, int n_frames, int n_tiles, const int32_t *h_tile_yx, float conf_thr,
                          int min_dist, int cap, float *d_conf, int32_t *d_x, int32_t *d_y, int32_t *d_count,
                          void *stream)
{
    AXT_REQUIRE(d_yolo && h_tile_yx && d_conf && d_x && d_y && d_count, "null argument");
    AXT_REQUIRE(n_tiles >= 1 && n_tiles <= 256, "n_tiles %d out of range (1..256)", n_tiles);
    AXT_REQUIRE(cap >= n_tiles * AXT_CELLS, "cap %d < n_tiles*144 = %d", cap, n_tiles * AXT_CELLS);
    AXT_REQUIRE(min_dist >= 0 && min_dist < 32768, "min_dist out of range");
    if (n_frames <= 0) return AXT_OK;
    TileOrigins tiles;
    tiles.n = n_tiles;
    for (int k = 0; k < n_tiles; ++k) {
        tiles.yx[2 * k] = (short)h_tile_yx[2 * k];
        tiles.yx[2 * k + 1] = (short)h_tile_yx[2 * k + 1];
    }
    if (n_tiles > 28) {                         // candidates of a frame do not fit the LDS: workspace in HBM
        const size_t per_frame = (size_t)n_tiles * AXT_CELLS * 32;
        hipStream_t st = (hipStream_t)stream;
        unsigned char *work = nullptr;
        AXT_CHECK_HIP(hipMallocAsync((void **)&work, per_frame * (size_t)n_frames, st));
        hipLaunchKernelGGL(decode_stitch_nms_big_kernel, dim3(n_frames), dim3(256), 0, st, d_yolo, n_tiles, tiles, conf_thr,
                           min_dist * min_dist, cap, d_conf, d_x, d_y, d_count, work);
        AXT_LAUNCH_CHECK();
        AXT_CHECK_HIP(hipFreeAsync(work, st));
        return AXT_OK;
    }
    const size_t lds = (size_t)n_tiles * AXT_CELLS * 7 * 4;
    static AxtOncePerDevice once;                 // (per device: see axt_common.h)
    if (int rc = axt_max_dynamic_lds(decode_stitch_nms_kernel, 28 * AXT_CELLS * 7 * 4, once)) return rc;
    hipLaunchKernelGGL(decode_stitch_nms_kernel, dim3(n_frames), dim3(256), lds, (hipStream_t)stream, d_yolo, n_tiles,
                       tiles, conf_thr, min_dist * min_dist, cap, d_conf, d_x, d_y, d_count);
    AXT_LAUNCH_CHECK();
    return AXT_OK;
}

}  // extern "C"
