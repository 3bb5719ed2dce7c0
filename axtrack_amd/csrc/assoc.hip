// Association costs on gfx950: observation costs, path-length matrices and the admissible-arc list.
//
// Replaces, from the reference:
//   conf capping + observation_model      AxonDetections.py:655-659, mincostflow_models.py:6-27
//   _compute_detections_astar_paths / _get_astar_path_distances   AxonDetections.py:526-629,717-752
//   the tracker's edge admission (transition_model + cost_threshold)   mincostflow_models.py:67-119
//
// Path length convention (pyastar2d is absent from the reference tree): number of cells of a
// minimum-cost 4-connected (or 8-connected) path, both end points included; `max_dist` when the euclidean
// distance is >= max_dist, an end point lies outside the grid or the path needs more than max_dist cells.
// On an all-ones mask that is |dx|+|dy|+1 (max(|dx|,|dy|)+1) in closed form; masked grids run a
// breadth-first search per source (path_bfs.hip).
#include "axt_common.h"

// every rounding in this file is part of the contract with the CPU restatement: no fused multiply-adds
#pragma clang fp contract(off)

int axt_path_cost_masked(const int32_t *d_xa, const int32_t *d_ya, int na, const int32_t *d_xb, const int32_t *d_yb,
                         int nb, const uint8_t *d_mask, int H, int W, int max_dist, int conn8, int32_t *d_D,
                         hipStream_t st, int32_t *d_cells);
struct axt_grid;
extern "C" const uint8_t *axt_grid_mask(const axt_grid *g);
int axt_masked_distance_table(const axt_grid *g, const int32_t *d_x, const int32_t *d_y, const int32_t *d_count,
                              const int32_t *d_src_count, int n_frames, int cap, int max_dist, int max_gap,
                              const int32_t *h_dmax, const int32_t *d_dmax, int16_t *d_Dtmp, hipStream_t st);

namespace {

__device__ __forceinline__ int path_len_open(int xa, int ya, int xb, int yb, int H, int W, int max_dist, int conn8)
{
    const int dx = abs(xa - xb), dy = abs(ya - yb);
    // euclidean gate, AxonDetections.py:620-628: sqrt(dy^2+dx^2) < 500  <=>  dx^2+dy^2 < 500^2 (integers)
    const long d2 = (long)dx * dx + (long)dy * dy;
    const int len = (conn8 ? max(dx, dy) : dx + dy) + 1;
    const bool inb = xa >= 0 && xa < W && ya >= 0 && ya < H && xb >= 0 && xb < W && yb >= 0 && yb < H;
    return (d2 < (long)max_dist * max_dist && len <= max_dist && inb) ? len : max_dist;
}

__global__ void path_cost_open_kernel(const int *__restrict__ xa, const int *__restrict__ ya, int na,
                                      const int *__restrict__ xb, const int *__restrict__ yb, int nb, int H, int W,
                                      int max_dist, int conn8, int *__restrict__ D)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)na * nb) return;
    const int i = idx / nb, j = idx - (long)i * nb;
    D[idx] = path_len_open(xa[i], ya[i], xb[j], yb[j], H, W, max_dist, conn8);
}

// ---- observation costs -------------------------------------------------------------------------
__global__ void conf_max_kernel(const float *__restrict__ conf, const int *__restrict__ count, int cap,
                                unsigned int *__restrict__ max_bits)
{
    const int f = blockIdx.x, n = count[f];
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, conf[(long)f * cap + i]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    // confidences that passed the 0.55 floor are positive: their bit patterns order like the floats
    if ((threadIdx.x & 63) == 0) atomicMax(max_bits, __float_as_uint(m));
}

__global__ void obs_cost_kernel(const float *__restrict__ conf, const int *__restrict__ count, int cap, int method,
                                double max_cost, const unsigned int *__restrict__ max_bits, double *__restrict__ cost)
{
    const int f = blockIdx.x, n = count[f];
    const double cmax = (double)__uint_as_float(*max_bits);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double s = (double)conf[(long)f * cap + i];          // python float of the f32 value (:653)
        if (method == 0) s = s / cmax;                        // 'scale_to_max' (:658-659)
        else if (s > 1.0) s = 1.0;                            // 'ceil' (:656-657)
        double beta = (s - 1.0) * -1.0 + 1e-6;                // mincostflow_models.py:23
        double c = log(beta / (1.0 - beta));                  // :24
        if (c > max_cost) c = max_cost;                       // :25-26
        if (c < -max_cost) c = -max_cost;
        cost[(long)f * cap + i] = c;
    }
}

// ---- arc list ----------------------------------------------------------------------------------
// integer arc cost = cost_units << 16 | hash16(kind, a, b): same formula as axt_arc_cost_int (api.cpp)
__device__ __forceinline__ long arc_cost_int(long units, int kind, long a, long b)
{
    unsigned long x = ((unsigned long)kind << 60) ^ ((unsigned long)a << 30) ^ (unsigned long)b;
    x += 0x9E3779B97F4A7C15ul;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ul;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBul;
    x ^= x >> 31;
    return units * 65536 + (long)(x & 0xFFFFul);
}

// frame_off[t] = number of detections in frames < t (single block; n_frames is a few thousand at most)
__global__ void frame_offsets_kernel(const int *__restrict__ count, int n_frames, int cap, int *__restrict__ frame_off)
{
    __shared__ int carry;
    __shared__ int buf[1024];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n_frames; base += 1024) {
        const int i = base + threadIdx.x;
        int v = (i < n_frames) ? min(count[i], cap) : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            int t = (threadIdx.x >= o) ? buf[threadIdx.x - o] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n_frames) frame_off[i] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += buf[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) frame_off[n_frames] = carry;
}

// One wave per (source detection a of frame t, gap g): scans the detections of frame t+g, 64 per step.
// TABLE: path lengths come from the masked-grid BFS pass instead of the closed form.
// FILL == false: writes the number of admitted targets to cnt[(a_global)*max_gap + g-1].
// FILL == true : writes the arcs at row_ptr[a_global] + (arcs of smaller gaps) in ascending b.
// VIS (MCF_VIS_SIM_WEIGHT > 0): the transition cost is no longer a table of (D, gap) -- it is
// -log((1-w) * (1 - D/max) * miss^(gap-1) + w * (1 - bhattacharyya(hist_a, hist_b)) + 1e-6) per pair
// (mincostflow_models.py:100-118), computed here in f64 with the reference's operation order, and an arc is admitted
// iff that cost is below the edge threshold. dmax then only bounds the candidates (cost with similarity 1).
struct VisParams {
    const float *hist;      // [n_frames, cap, 180] min-max normalised histograms (axt_box_histograms)
    const double *hsum;     // [n_frames, cap] their bin sums
    double w, thr, mp[8];   // weight, edge cost threshold, miss_rate^(gap-1)
};

__device__ __forceinline__ double vis_transition_cost(const VisParams &vp, const float *ha, double sa, const float *hb,
                                                      double sb, int d, int g, int max_dist)
{
    // plain operators: fp contract is off in this file, every operation rounds once, in the reference's order
    double s12 = 0.0;
    for (int b = 0; b < 180; ++b) s12 += sqrt((double)ha[b] * (double)hb[b]);
    double s = sa * sb;
    s = fabs(s) > 1.1920928955078125e-07 ? 1.0 / sqrt(s) : 1.0;
    const double t = 1.0 - s12 * s;
    const double vs = 1.0 - sqrt(t > 0.0 ? t : 0.0);                                  // 1 - cv2.compareHist(...)
    const double dist = (((double)d / (double)max_dist) - 1.0) * -1.0;                // ((D / max) - 1) * -1
    if (dist == 0.0) return INFINITY;
    const double v = (1.0 - vp.w) * dist * vp.mp[g - 1] + vp.w * vs + 1e-6;
    return -log(v);
}

template <bool FILL, bool TABLE, bool VIS>
__global__ __launch_bounds__(256) void arcs_open_kernel(
    const int *__restrict__ x, const int *__restrict__ y, const int *__restrict__ count, const int *__restrict__ src_count,
    const int *__restrict__ frame_off, int n_frames, int cap, int H, int W, int max_dist, int conn8, int max_gap,
    const int *__restrict__ dmax, int *__restrict__ cnt, const long *__restrict__ row_ptr,
    int *__restrict__ col, short *__restrict__ len, unsigned char *__restrict__ gapv,
    const long *__restrict__ cost_units, long *__restrict__ cost, const short *__restrict__ Dtmp, VisParams vp)
{
    __shared__ float ha_s[VIS ? 4 : 1][VIS ? 180 : 1];
    const int t = blockIdx.x;
    const int na = min(src_count[t], cap);             // rows are built for these (a frame-sharded rank: its own frames)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    for (int w = blockIdx.y * waves + wave; w < na * max_gap; w += gridDim.y * waves) {
        const int i = w / max_gap, g = w % max_gap + 1;
        const int a = frame_off[t] + i;
        const int tb = t + g;
        int total = 0;
        if (tb < n_frames) {
            const int nb = min(count[tb], cap);
            const int xa = x[(long)t * cap + i], ya = y[(long)t * cap + i];
            const int lim = dmax[g - 1];
            long base = 0;
            if (FILL) {
                base = row_ptr[a];
                for (int gg = 1; gg < g; ++gg) base += cnt[(long)a * max_gap + gg - 1];
            }
            double sa = 0.0;
            if (VIS) {                      // this wave's source histogram, read by every lane for every target
                for (int b = lane; b < 180; b += 64) ha_s[wave][b] = vp.hist[((long)t * cap + i) * 180 + b];
                sa = vp.hsum[(long)t * cap + i];
            }
            for (int j0 = 0; j0 < nb; j0 += 64) {
                const int j = j0 + lane;
                int d = max_dist;
                if (j < nb) {
                    if (TABLE) {        // masked grid: path lengths were computed by the BFS pass (0 = no arc)
                        d = Dtmp[(((long)t * cap + i) * max_gap + (g - 1)) * cap + j];
                        if (d <= 0) d = max_dist;
                    } else {
                        d = path_len_open(xa, ya, x[(long)tb * cap + j], y[(long)tb * cap + j], H, W, max_dist, conn8);
                    }
                }
                bool ok = (j < nb) && (d <= lim);
                double c = 0.0;
                if (VIS && ok) {
                    c = vis_transition_cost(vp, ha_s[wave], sa, vp.hist + ((long)tb * cap + j) * 180,
                                            vp.hsum[(long)tb * cap + j], d, g, max_dist);
                    ok = c < vp.thr;
                }
                const unsigned long long m = __ballot(ok);
                if (FILL && ok) {
                    const long o = base + total + __popcll(m & ((1ull << lane) - 1ull));
                    col[o] = frame_off[tb] + j;
                    len[o] = (short)d;
                    gapv[o] = (unsigned char)g;
                    if (cost) {
                        const long units = VIS ? llrint(c * 1e6) : cost_units[(long)(g - 1) * (max_dist + 1) + d];
                        cost[o] = arc_cost_int(units, 3, a, frame_off[tb] + j);
                    }
                }
                total += __popcll(m);
            }
        }
        if (!FILL && lane == 0) cnt[(long)a * max_gap + g - 1] = total;
    }
}

// row_ptr = exclusive scan of the per-detection arc counts (single block, sequential over chunks)
__global__ void row_ptr_kernel(const int *__restrict__ cnt, const int *__restrict__ frame_off, int n_frames, int max_gap,
                               long *__restrict__ row_ptr)
{
    __shared__ long carry;
    __shared__ long buf[1024];
    const int n_det = frame_off[n_frames];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n_det; base += 1024) {
        const int i = base + threadIdx.x;
        long v = 0;
        if (i < n_det)
            for (int g = 0; g < max_gap; ++g) v += cnt[(long)i * max_gap + g];
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            long t = (threadIdx.x >= o) ? buf[threadIdx.x - o] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n_det) row_ptr[i] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += buf[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) row_ptr[n_det] = carry;
}

}  // namespace

int axt_frame_offsets(const int32_t *d_count, int n_frames, int cap, int32_t *d_off, hipStream_t st)
{
    hipLaunchKernelGGL(frame_offsets_kernel, dim3(1), dim3(1024), 0, st, d_count, n_frames, cap, d_off);
    AXT_LAUNCH_CHECK();
    return AXT_OK;
}

extern "C" {

int axt_obs_costs(const float *d_conf, const int32_t *d_count, int n_frames, int cap, int method, double max_conf_cost,
                  double *d_cost, void *stream)
{
    AXT_REQUIRE(d_conf && d_count && d_cost, "null argument");
    AXT_REQUIRE(method == 0 || method == 1, "method must be 0 (scale_to_max) or 1 (ceil)");
    if (n_frames <= 0) return AXT_OK;
    hipStream_t st = (hipStream_t)stream;
    unsigned int *mx = nullptr;
    AXT_CHECK_HIP(hipMallocAsync((void **)&mx, sizeof(unsigned int), st));
    AXT_CHECK_HIP(hipMemsetAsync(mx, 0, sizeof(unsigned int), st));
    hipLaunchKernelGGL(conf_max_kernel, dim3(n_frames), dim3(64), 0, st, d_conf, d_count, cap, mx);
    AXT_LAUNCH_CHECK();
    hipLaunchKernelGGL(obs_cost_kernel, dim3(n_frames), dim3(64), 0, st, d_conf, d_count, cap, method, max_conf_cost,
                       mx, d_cost);
    AXT_LAUNCH_CHECK();
    AXT_CHECK_HIP(hipFreeAsync(mx, st));
    return AXT_OK;
}

int axt_path_cost(const int32_t *d_xa, const int32_t *d_ya, int na, const int32_t *d_xb, const int32_t *d_yb, int nb,
                  const axt_grid *grid, int H, int W, int max_dist, int conn8, int32_t *d_D, void *stream)
{
    AXT_REQUIRE(na >= 0 && nb >= 0 && H > 0 && W > 0 && max_dist > 0 && max_dist < 32768, "bad argument");
    if ((long)na * nb == 0) return AXT_OK;
    AXT_REQUIRE(d_xa && d_ya && d_xb && d_yb && d_D, "null argument");
    hipStream_t st = (hipStream_t)stream;
    if (grid) return axt_path_cost_masked(d_xa, d_ya, na, d_xb, d_yb, nb, axt_grid_mask(grid), H, W, max_dist, conn8, d_D, st, nullptr);
    const long n = (long)na * nb;
    hipLaunchKernelGGL(path_cost_open_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_xa, d_ya, na, d_xb,
                       d_yb, nb, H, W, max_dist, conn8, d_D);
    AXT_LAUNCH_CHECK();
    return AXT_OK;
}

int axt_path_cells(const int32_t *d_xa, const int32_t *d_ya, int na, const int32_t *d_xb, const int32_t *d_yb, int nb,
                   const axt_grid *grid, int H, int W, int max_dist, int conn8, int32_t *d_D, int32_t *d_cells, void *stream)
{
    AXT_REQUIRE(na >= 0 && nb >= 0 && H > 0 && W > 0 && max_dist > 0 && max_dist < 32768, "bad argument");
    AXT_REQUIRE(grid, "axt_path_cells: needs a masked grid (on an all-ones mask every monotone staircase is a shortest path)");
    if ((long)na * nb == 0) return AXT_OK;
    AXT_REQUIRE(d_xa && d_ya && d_xb && d_yb && d_D && d_cells, "null argument");
    return axt_path_cost_masked(d_xa, d_ya, na, d_xb, d_yb, nb, axt_grid_mask(grid), H, W, max_dist, conn8, d_D,
                                (hipStream_t)stream, d_cells);
}

static int build_arcs_impl(const int32_t *d_x, const int32_t *d_y, const int32_t *d_count, int n_frames, int cap,
                           const axt_grid *grid, int H, int W, int max_dist, int conn8, int max_gap, const int32_t *h_dmax,
                           int64_t *d_row_ptr, int32_t *d_work, int32_t *d_col, int16_t *d_len, uint8_t *d_gap,
                           const int64_t *d_cost_units, int64_t *d_cost, int64_t *n_arcs, const VisParams *vis,
                           const int16_t *d_ext_table, void *stream, const int32_t *d_src_count = nullptr)
{
    const int32_t *src_count = d_src_count ? d_src_count : d_count;
    AXT_REQUIRE(d_x && d_y && d_count && h_dmax && d_row_ptr && d_work && n_arcs, "null argument");
    AXT_REQUIRE(n_frames >= 1 && cap >= 1 && max_gap >= 1 && max_gap <= 8, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    // d_work layout: cnt [n_frames*cap*max_gap] | frame_off [n_frames+1] | dmax [max_gap] | pad |
    //                masked grids only: Dtmp i16 [n_frames*cap*max_gap*cap]
    int *cnt = d_work;
    int *frame_off = d_work + (size_t)n_frames * cap * max_gap;
    int *dmax = frame_off + n_frames + 1;
    // path lengths from a table: the caller's (d_ext_table) or the one the masked-grid search fills in d_work
    const short *Dtmp = d_ext_table ? d_ext_table
                                    : reinterpret_cast<short *>(d_work + (((size_t)n_frames * cap * max_gap + n_frames + 1 + max_gap + 3) & ~(size_t)3));
    const bool table = grid != nullptr || d_ext_table != nullptr;
    const dim3 grid_dim(n_frames, 8), block(256);
    const bool fill = d_col != nullptr;
    const VisParams vp = vis ? *vis : VisParams{};
    // the four (table, appearance) variants of one pass
    auto launch = [&](auto kern) {
        hipLaunchKernelGGL(kern, grid_dim, block, 0, st, d_x, d_y, d_count, src_count, frame_off, n_frames, cap, H, W, max_dist, conn8,
                           max_gap, dmax, cnt, fill ? (const long *)d_row_ptr : (const long *)nullptr, fill ? d_col : (int *)nullptr,
                           fill ? d_len : (short *)nullptr, fill ? d_gap : (unsigned char *)nullptr,
                           fill ? (const long *)d_cost_units : (const long *)nullptr, fill ? (long *)d_cost : (long *)nullptr,
                           table ? Dtmp : (const short *)nullptr, vp);
    };
    if (!fill) {
        AXT_CHECK_HIP(hipMemcpyAsync(dmax, h_dmax, sizeof(int) * max_gap, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(frame_offsets_kernel, dim3(1), dim3(1024), 0, st, d_count, n_frames, cap, frame_off);
        AXT_LAUNCH_CHECK();
        if (grid && !d_ext_table) {
            const int rc = axt_masked_distance_table(grid, d_x, d_y, d_count, src_count, n_frames, cap, max_dist, max_gap, h_dmax,
                                                     dmax, const_cast<short *>(Dtmp), st);
            if (rc) return rc;
        }
        if (d_src_count)         // rows of other ranks' frames stay empty
            AXT_CHECK_HIP(hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)n_frames * cap * max_gap, st));
        if (table && vis) launch(arcs_open_kernel<false, true, true>);
        else if (table) launch(arcs_open_kernel<false, true, false>);
        else if (vis) launch(arcs_open_kernel<false, false, true>);
        else launch(arcs_open_kernel<false, false, false>);
        AXT_LAUNCH_CHECK();
        hipLaunchKernelGGL(row_ptr_kernel, dim3(1), dim3(1024), 0, st, cnt, frame_off, n_frames, max_gap,
                           (long *)d_row_ptr);
        AXT_LAUNCH_CHECK();
        int n_det = 0;
        AXT_CHECK_HIP(hipMemcpyAsync(&n_det, frame_off + n_frames, sizeof(int), hipMemcpyDeviceToHost, st));
        AXT_CHECK_HIP(hipStreamSynchronize(st));
        long total = 0;
        AXT_CHECK_HIP(hipMemcpyAsync(&total, (long *)d_row_ptr + n_det, sizeof(long), hipMemcpyDeviceToHost, st));
        AXT_CHECK_HIP(hipStreamSynchronize(st));
        *n_arcs = total;
        return AXT_OK;
    }
    AXT_REQUIRE(d_len && d_gap, "null argument");
    AXT_REQUIRE(vis || (d_cost == nullptr) == (d_cost_units == nullptr), "d_cost and d_cost_units go together");
    if (table && vis) launch(arcs_open_kernel<true, true, true>);
    else if (table) launch(arcs_open_kernel<true, true, false>);
    else if (vis) launch(arcs_open_kernel<true, false, true>);
    else launch(arcs_open_kernel<true, false, false>);
    AXT_LAUNCH_CHECK();
    return AXT_OK;
}

int axt_build_arcs(const int32_t *d_x, const int32_t *d_y, const int32_t *d_count, int n_frames, int cap,
                   const axt_grid *grid, int H, int W, int max_dist, int conn8, int max_gap, const int32_t *h_dmax,
                   int64_t *d_row_ptr, int32_t *d_work, int32_t *d_col, int16_t *d_len, uint8_t *d_gap,
                   const int64_t *d_cost_units, int64_t *d_cost, int64_t *n_arcs, void *stream)
{
    return build_arcs_impl(d_x, d_y, d_count, n_frames, cap, grid, H, W, max_dist, conn8, max_gap, h_dmax, d_row_ptr, d_work,
                           d_col, d_len, d_gap, d_cost_units, d_cost, n_arcs, nullptr, nullptr, stream);
}

int axt_build_arcs_from_lengths(const int16_t *d_len_table, const int32_t *d_x, const int32_t *d_y, const int32_t *d_count,
                                int n_frames, int cap, int max_dist, int max_gap, const int32_t *h_dmax,
                                int64_t *d_row_ptr, int32_t *d_work, int32_t *d_col, int16_t *d_len, uint8_t *d_gap,
                                const int64_t *d_cost_units, int64_t *d_cost, int64_t *n_arcs, void *stream)
{
    AXT_REQUIRE(d_len_table, "axt_build_arcs_from_lengths: null table");
    return build_arcs_impl(d_x, d_y, d_count, n_frames, cap, nullptr, 0, 0, max_dist, 0, max_gap, h_dmax, d_row_ptr, d_work,
                           d_col, d_len, d_gap, d_cost_units, d_cost, n_arcs, nullptr, d_len_table, stream);
}

int axt_build_arcs_vis(const int32_t *d_x, const int32_t *d_y, const int32_t *d_count, int n_frames, int cap,
                       const axt_grid *grid, int H, int W, int max_dist, int conn8, int max_gap, const int32_t *h_dmax,
                       const float *d_hist, const double *d_hist_sum, double vis_weight, double miss_rate,
                       double edge_cost_thr, int64_t *d_row_ptr, int32_t *d_work, int32_t *d_col, int16_t *d_len,
                       uint8_t *d_gap, int64_t *d_cost, int64_t *n_arcs, void *stream)
{
    AXT_REQUIRE(d_hist && d_hist_sum, "axt_build_arcs_vis: null histogram argument");
    AXT_REQUIRE(vis_weight > 0.0 && vis_weight <= 1.0, "axt_build_arcs_vis: weight %g outside (0,1]", vis_weight);
    VisParams vp;
    vp.hist = d_hist;
    vp.hsum = d_hist_sum;
    vp.w = vis_weight;
    vp.thr = edge_cost_thr;
    for (int g = 0; g < 8; ++g) vp.mp[g] = pow(miss_rate, (double)g);          // miss_rate ** (gap - 1), as Python computes it
    return build_arcs_impl(d_x, d_y, d_count, n_frames, cap, grid, H, W, max_dist, conn8, max_gap, h_dmax, d_row_ptr, d_work,
                           d_col, d_len, d_gap, nullptr, d_cost, n_arcs, &vp, nullptr, stream);
}

int axt_build_arcs_rows(const int32_t *d_x, const int32_t *d_y, const int32_t *d_count, const int32_t *d_src_count,
                        int n_frames, int cap, const axt_grid *grid, int H, int W, int max_dist, int conn8, int max_gap,
                        const int32_t *h_dmax, const float *d_hist, const double *d_hist_sum, double vis_weight,
                        double miss_rate, double edge_cost_thr, int64_t *d_row_ptr, int32_t *d_work, int32_t *d_col,
                        int16_t *d_len, uint8_t *d_gap, const int64_t *d_cost_units, int64_t *d_cost, int64_t *n_arcs,
                        const int16_t *d_len_table, void *stream)
{
    if (!d_hist)
        return build_arcs_impl(d_x, d_y, d_count, n_frames, cap, d_len_table ? nullptr : grid, H, W, max_dist, conn8, max_gap,
                               h_dmax, d_row_ptr, d_work, d_col, d_len, d_gap, d_cost_units, d_cost, n_arcs, nullptr, d_len_table,
                               stream, d_src_count);
    AXT_REQUIRE(d_hist_sum, "axt_build_arcs_rows: null histogram sums");
    AXT_REQUIRE(vis_weight > 0.0 && vis_weight <= 1.0, "axt_build_arcs_rows: weight %g outside (0,1]", vis_weight);
    VisParams vp;
    vp.hist = d_hist;
    vp.hsum = d_hist_sum;
    vp.w = vis_weight;
    vp.thr = edge_cost_thr;
    for (int g = 0; g < 8; ++g) vp.mp[g] = pow(miss_rate, (double)g);
    return build_arcs_impl(d_x, d_y, d_count, n_frames, cap, d_len_table ? nullptr : grid, H, W, max_dist, conn8, max_gap, h_dmax,
                           d_row_ptr, d_work, d_col, d_len, d_gap, nullptr, d_cost, n_arcs, &vp, d_len_table, stream, d_src_count);
}

}  // extern "C"
