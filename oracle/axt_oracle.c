/*
 * oracle/axt_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the numeric kernels on AxTrack's detect+associate hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (axtrack_amd/) never does and has no CPU fallback.
 *
 * Each function cites the reference lines (relative to /root/reference) it restates.
 * Pinning: checked against the .npz fixtures under tests/golden/, which were produced by running the
 * reference's own Python (tests/golden/make_golden.py). The two third-party pieces that
 * are absent from the reference tree -- pyastar2d (grid A*) and libmot/ortools (min-cost
 * flow) -- are restated from their published algorithms; their parity is UNPINNED
 * (see DESIGN.md), the conventions chosen are documented at orc_astar_len / orc_mcf_ssp.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Number of threads of the parallel loops below. The environment variable is only read when the OpenMP runtime starts,
 * which in a process that has imported torch happened long ago (with one thread per visible CPU). */
void orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n >= 1) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------
 * CNNBlock.forward: Conv2d(3x3, pad 1, bias) -> BatchNorm2d(eval, eps 1e-5) -> LeakyReLU(0.1)
 * model.py:5-18 (block), model.py:85-103 (padding=(1,1), kernel 3, stride from ARCHITECTURE).
 * NCHW f32. Accumulation order: cin, ky, kx (textbook direct convolution) in f32, the
 * reference's own order (oneDNN) is unspecified, so CNN parity is by tolerance.
 * BN is applied UNFOLDED, as the reference does: (y - mean) / sqrt(var + eps) * gamma + beta.
 * ------------------------------------------------------------------------------------ */
void orc_conv3x3_bn_lrelu(const float *x, int B, int Cin, int H, int W,
                          const float *w, const float *bias,
                          const float *gamma, const float *beta, const float *mean, const float *var,
                          int Cout, int stride, float slope, float *out)
{
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co) {
            float *o = out + ((size_t)b * Cout + co) * Ho * Wo;
            for (int i = 0; i < Ho * Wo; ++i) o[i] = bias[co];
            for (int ci = 0; ci < Cin; ++ci) {
                const float *xi = x + ((size_t)b * Cin + ci) * H * W;
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx) {
                        const float wv = w[(((size_t)co * Cin + ci) * 3 + ky) * 3 + kx];
                        for (int oy = 0; oy < Ho; ++oy) {
                            const int iy = oy * stride + ky - 1;
                            if (iy < 0 || iy >= H) continue;
                            int ox0 = 0, ox1 = Wo;
                            while (ox0 < Wo && ox0 * stride + kx - 1 < 0) ++ox0;
                            while (ox1 > ox0 && (ox1 - 1) * stride + kx - 1 >= W) --ox1;
                            const float *xr = xi + (size_t)iy * W + kx - 1;
                            float *orow = o + (size_t)oy * Wo;
                            for (int ox = ox0; ox < ox1; ++ox) orow[ox] += wv * xr[ox * stride];
                        }
                    }
            }
            const float inv = 1.0f / sqrtf(var[co] + 1e-5f);
            for (int i = 0; i < Ho * Wo; ++i) {
                float v = (o[i] - mean[co]) * inv * gamma[co] + beta[co];
                o[i] = v > 0.f ? v : v * slope;
            }
        }
}

/* nn.MaxPool2d(2,2): model.py:100-101 */
void orc_maxpool2(const float *x, int BC, int H, int W, float *out)
{
    const int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < BC; ++c)
        for (int y = 0; y < Ho; ++y)
            for (int xx = 0; xx < Wo; ++xx) {
                const float *p = x + ((size_t)c * H + 2 * y) * W + 2 * xx;
                float m = p[0];
                if (p[1] > m) m = p[1];
                if (p[W] > m) m = p[W];
                if (p[W + 1] > m) m = p[W + 1];
                out[((size_t)c * Ho + y) * Wo + xx] = m;
            }
}

/* nn.Linear (+ optional nn.Sigmoid): model.py:105-117. w is [out, in] row-major. */
void orc_linear(const float *x, int B, int In, const float *w, const float *bias, int Out,
                int sigmoid, float *out)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int o = 0; o < Out; ++o) {
            const float *xr = x + (size_t)b * In, *wr = w + (size_t)o * In;
            /* 8 partial sums keep the f32 error near what a blocked BLAS gives */
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int i = 0;
            for (; i + 8 <= In; i += 8)
                for (int k = 0; k < 8; ++k) acc[k] += xr[i + k] * wr[i + k];
            float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
            for (; i < In; ++i) s += xr[i] * wr[i];
            s += bias[o];
            out[(size_t)b * Out + o] = sigmoid ? 1.0f / (1.0f + expf(-s)) : s;
        }
}

/* ------------------------------------------------------------------------------------
 * pyastar2d.astar_path(weights, source, target, max_path_length) as called at
 * utils.py:379, consumed at AxonDetections.py:624,736 -- only the LENGTH of the path
 * (number of cells including both end points) is used downstream.
 *
 * pyastar2d (fork LoaloaF/pyastar2d, pin pyastar2d==1.0.2, axtr.yml:140) is NOT in the
 * reference tree: PARITY UNPINNED. Convention restated from the published upstream
 * algorithm: A* on the pixel grid, cost of a move = weight of the cell moved INTO,
 * 4-connected by default (conn8 != 0: 8-connected, diagonal moves cost the same), optimal
 * because the L1 / Chebyshev heuristic times min(weights) >= 1 is admissible. Returns the
 * number of cells on a minimum-cost path over the whole grid, or 0 ("None") when that path
 * has more than max_len cells. For weights in {1, 65536} (AxonDetections.py:598) every minimum-cost
 * path has the same cell count, so the result does not depend on tie-breaking; we
 * therefore run Dijkstra on (cost, cells) which is exact and simple.
 * ------------------------------------------------------------------------------------ */
typedef struct { double cost; int32_t cell; } heap_item;

static void heap_push(heap_item *h, int *n, heap_item it)
{
    int i = (*n)++;
    while (i > 0) {
        int p = (i - 1) / 2;
        if (h[p].cost <= it.cost) break;
        h[i] = h[p];
        i = p;
    }
    h[i] = it;
}

static heap_item heap_pop(heap_item *h, int *n)
{
    heap_item top = h[0], last = h[--(*n)];
    int i = 0;
    for (;;) {
        int c = 2 * i + 1;
        if (c >= *n) break;
        if (c + 1 < *n && h[c + 1].cost < h[c].cost) ++c;
        if (h[c].cost >= last.cost) break;
        h[i] = h[c];
        i = c;
    }
    h[i] = last;
    return top;
}

int orc_astar_len(const float *weights, int H, int W, int sy, int sx, int ty, int tx,
                  int max_len, int conn8)
{
    if (sy < 0 || sy >= H || sx < 0 || sx >= W || ty < 0 || ty >= H || tx < 0 || tx >= W) return 0;
    /* the search runs on the whole grid, like upstream pyastar2d */
    const int y0 = 0, y1 = H - 1, x0 = 0, x1 = W - 1;
    const int wh = y1 - y0 + 1, ww = x1 - x0 + 1;
    double *cost = (double *)malloc(sizeof(double) * wh * ww);
    int32_t *cells = (int32_t *)malloc(sizeof(int32_t) * wh * ww);
    heap_item *heap = (heap_item *)malloc(sizeof(heap_item) * (size_t)wh * ww * 4 + 16);
    for (int i = 0; i < wh * ww; ++i) { cost[i] = INFINITY; cells[i] = 0; }
    int hn = 0, result = 0;
    const int s = (sy - y0) * ww + (sx - x0), t = (ty - y0) * ww + (tx - x0);
    if (ty < y0 || ty > y1 || tx < x0 || tx > x1) goto done;
    cost[s] = 0; cells[s] = 1;
    heap_push(heap, &hn, (heap_item){0.0, s});
    static const int dy8[8] = {-1, 1, 0, 0, -1, -1, 1, 1}, dx8[8] = {0, 0, -1, 1, -1, 1, -1, 1};
    const int nn = conn8 ? 8 : 4;
    while (hn > 0) {
        heap_item it = heap_pop(heap, &hn);
        if (it.cost > cost[it.cell]) continue;
        if (it.cell == t) break;
        const int cy = it.cell / ww, cx = it.cell % ww;
        for (int k = 0; k < nn; ++k) {
            const int ny = cy + dy8[k], nx = cx + dx8[k];
            if (ny < 0 || ny >= wh || nx < 0 || nx >= ww) continue;
            const int nc = ny * ww + nx;
            const double c = it.cost + (double)weights[(size_t)(ny + y0) * W + (nx + x0)];
            if (c < cost[nc]) {
                cost[nc] = c;
                cells[nc] = cells[it.cell] + 1;
                heap_push(heap, &hn, (heap_item){c, nc});
            }
        }
    }
    if (cells[t] > 0 && cells[t] <= max_len) result = cells[t];
done:
    free(cost); free(cells); free(heap);
    return result;
}

/* D[i][j] for all pairs, with the euclidean gate of AxonDetections.py:617-629 and the
 * None -> max_dist conversion of :736. src = detections at t_bef (rows), dst = at t. */
void orc_path_matrix(const float *weights, int H, int W,
                     const int64_t *src_x, const int64_t *src_y, int ns,
                     const int64_t *dst_x, const int64_t *dst_y, int nd,
                     int max_dist, int conn8, int32_t *D)
{
#pragma omp parallel for schedule(dynamic, 4)
    for (int p = 0; p < ns * nd; ++p) {
        const int i = p / nd, j = p % nd;
        const double dy = (double)(src_y[i] - dst_y[j]), dx = (double)(src_x[i] - dst_x[j]);
        int len = 0;
        if (sqrt(dy * dy + dx * dx) < (double)max_dist)
            len = orc_astar_len(weights, H, W, (int)src_y[i], (int)src_x[i], (int)dst_y[j], (int)dst_x[j],
                                max_dist, conn8);
        D[p] = len ? len : max_dist;
    }
}

/* ------------------------------------------------------------------------------------
 * libmot.data_association.MinCostFlowTracker.compute_trajectories(), call site
 * AxonDetections.py:663-690. libmot (fork LoaloaF/libmot, libmot==0.0.2, ortools 9.0) is
 * NOT in the reference tree: PARITY UNPINNED. Restated from the published formulation
 * (Zhang, Li, Nevatia, CVPR 2008, as implemented by libmot): unit-capacity network with
 * integer arc costs; the number of unit flows F in [min_flow, max_flow] minimising the
 * total cost is searched (libmot: Fibonacci search over F, each probe a min-cost-flow
 * solve; the cost is convex in F, so successive shortest paths reaches the same optimum).
 *
 * This oracle is the slow, obviously-correct form: Bellman-Ford shortest augmenting paths.
 * Graph is given as arc lists; node 0 = source, node 1 = sink.
 *   returns number of unit flows pushed (0 if < min_flow is infeasible -> *feasible = 0)
 *   flow_out[a] = 1 if arc a carries flow.
 * Stop rule: push while (F < min_flow) or (F < max_flow and next path cost < 0).
 * ------------------------------------------------------------------------------------ */
int orc_mcf_ssp(int n_nodes, int n_arcs, const int32_t *tail, const int32_t *head, const int64_t *cost,
                int min_flow, int max_flow, uint8_t *flow_out, int64_t *total_cost, int *feasible)
{
    int64_t *dist = (int64_t *)malloc(sizeof(int64_t) * n_nodes);
    int32_t *pred = (int32_t *)malloc(sizeof(int32_t) * n_nodes);
    memset(flow_out, 0, n_arcs);
    const int64_t INF = INT64_MAX / 4;
    int F = 0;
    int64_t tot = 0;
    *feasible = 1;
    while (F < max_flow) {
        for (int v = 0; v < n_nodes; ++v) { dist[v] = INF; pred[v] = -1; }
        dist[0] = 0;
        for (int it = 0; it < n_nodes; ++it) {
            int changed = 0;
            for (int a = 0; a < n_arcs; ++a) {
                if (!flow_out[a]) {              /* forward residual arc */
                    if (dist[tail[a]] < INF && dist[tail[a]] + cost[a] < dist[head[a]]) {
                        dist[head[a]] = dist[tail[a]] + cost[a]; pred[head[a]] = a; changed = 1;
                    }
                } else {                          /* backward residual arc */
                    if (dist[head[a]] < INF && dist[head[a]] - cost[a] < dist[tail[a]]) {
                        dist[tail[a]] = dist[head[a]] - cost[a]; pred[tail[a]] = a; changed = 1;
                    }
                }
            }
            if (!changed) break;
        }
        if (dist[1] >= INF) break;
        if (F >= min_flow && dist[1] >= 0) break;
        /* augment */
        int v = 1;
        while (v != 0) {
            const int a = pred[v];
            if (!flow_out[a] && head[a] == v) { flow_out[a] = 1; v = tail[a]; }
            else { flow_out[a] = 0; v = head[a]; }
        }
        tot += dist[1];
        ++F;
    }
    if (F < min_flow) *feasible = 0;
    *total_cost = tot;
    free(dist); free(pred);
    return F;
}

/* ------------------------------------------------------------------------------------
 * f-3: appearance features and their distance, mincostflow_models.py:30-65,107-113.
 * The three cv2 calls the reference makes (cv2 4.5.1 is absent here: PARITY UNPINNED) are restated
 * from OpenCV's published behaviour:
 *   calcHist([crop],[0],None,[180],[0,1])       bin = floor((double)v * 180), counted iff 0 <= bin < 180
 *                                               (v == 1.0 and everything outside [0,1) is dropped), f32 counts
 *   normalize(h, h, 0, 1, NORM_MINMAX)          scale = 1/(max-min) if max-min > DBL_EPSILON else 0,
 *                                               shift = -min*scale; h = (float)scale * h + (float)shift in f32
 *   compareHist(f1, f2, HISTCMP_BHATTACHARYYA)  s12 = sum sqrt(a*b), s1 = sum a, s2 = sum b in double, in bin order;
 *                                               s = s1*s2; s = |s| > FLT_EPSILON ? 1/sqrt(s) : 1;
 *                                               d = sqrt(max(1 - s12*s, 0))
 * The crop is feature_model's: rows [x1, x1+box) and columns [x2, x2+box) with x1 = max(y - box/2, 0),
 * x2 = max(x - box/2, 0) (a box that starts outside the image is SHIFTED inside, not clipped), clipped by
 * numpy slicing at the bottom / right edge.
 * ------------------------------------------------------------------------------------ */
void orc_box_histograms(const float *image, int H, int W, const int64_t *x, const int64_t *y, int n, int box,
                        float *out /* [n,180] */)
{
    for (int k = 0; k < n; ++k) {
        float h[180];
        for (int i = 0; i < 180; ++i) h[i] = 0.f;
        int64_t r0 = y[k] - box / 2, c0 = x[k] - box / 2;
        if (r0 < 0) r0 = 0;
        if (c0 < 0) c0 = 0;
        int64_t r1 = r0 + box, c1 = c0 + box;
        if (r1 > H) r1 = H;
        if (c1 > W) c1 = W;
        for (int64_t r = r0; r < r1; ++r)
            for (int64_t c = c0; c < c1; ++c) {
                const int idx = (int)floor((double)image[r * W + c] * 180.0);
                if ((unsigned)idx < 180u) h[idx] += 1.f;
            }
        float mn = h[0], mx = h[0];
        for (int i = 1; i < 180; ++i) { if (h[i] < mn) mn = h[i]; if (h[i] > mx) mx = h[i]; }
        const double scale = ((double)mx - (double)mn > 2.220446049250313e-16) ? 1.0 / ((double)mx - (double)mn) : 0.0;
        const float fs = (float)scale, fb = (float)(-(double)mn * scale);
        for (int i = 0; i < 180; ++i) {
            volatile float t = h[i] * fs;           /* f32 multiply, then f32 add: no fused multiply-add */
            out[(size_t)k * 180 + i] = t + fb;
        }
    }
}

void orc_bhattacharyya(const float *h1, int n1, const float *h2, int n2, double *out /* [n1,n2] */)
{
    for (int i = 0; i < n1; ++i)
        for (int j = 0; j < n2; ++j) {
            double s12 = 0, s1 = 0, s2 = 0;
            for (int b = 0; b < 180; ++b) {
                const double a = h1[(size_t)i * 180 + b], c = h2[(size_t)j * 180 + b];
                s12 += sqrt(a * c);
                s1 += a;
                s2 += c;
            }
            double s = s1 * s2;
            s = fabs(s) > 1.1920928955078125e-07 ? 1.0 / sqrt(s) : 1.0;
            const double d = 1.0 - s12 * s;
            out[(size_t)i * n2 + j] = sqrt(d > 0 ? d : 0);
        }
}
