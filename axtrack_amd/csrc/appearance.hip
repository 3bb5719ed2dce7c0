// Appearance features of the detections: the reference's feature_model
// (axtrack/mincostflow_models.py:30-65) on the GPU. Per detection a 180-bin histogram on [0,1) of the 70x70
// crop of the frame the tracker is shown (the stitched centre frame, AxonDetections.py:682-685), min-max
// normalised. cv2.calcHist / cv2.normalize are absent from the reference tree; their arithmetic is restated from
// OpenCV's published behaviour (parity unpinned, DESIGN.md "Unpinned third-party semantics").
//
// HBM-bound byte work: 4900 pixels read per detection (mostly L2 hits: neighbouring boxes overlap), 720 B written.
// One wavefront per detection, histogram in LDS.
#include "axt_common.h"

// every rounding in this file is part of the contract with the CPU restatement: no fused multiply-adds
#pragma clang fp contract(off)

namespace {

__global__ __launch_bounds__(64) void box_hist_kernel(const float *__restrict__ frames, int H, int W, int t_offset,
                                                      const int *__restrict__ x, const int *__restrict__ y,
                                                      const int *__restrict__ count, int cap, int box,
                                                      float *__restrict__ hist, double *__restrict__ hsum)
{
    __shared__ int bins[180];
    __shared__ float norm[180];
    const int f = blockIdx.y, i = blockIdx.x, lane = threadIdx.x;
    if (i >= min(count[f], cap)) return;
    for (int b = lane; b < 180; b += 64) bins[b] = 0;
    __syncthreads();
    const float *img = frames + (long)(f + t_offset) * H * W;
    const long s = (long)f * cap + i;
    // feature_model's crop: a box that starts outside the image is shifted inside (max(.., 0)), numpy slicing clips
    // the far edge (mincostflow_models.py:56-60; rows from Y = y - box/2, columns from X = x - box/2)
    const int r0 = max(y[s] - box / 2, 0), c0 = max(x[s] - box / 2, 0);
    const int r1 = min(r0 + box, H), c1 = min(c0 + box, W);
    const int nr = max(r1 - r0, 0), nc = max(c1 - c0, 0);
    for (int e = lane; e < nr * nc; e += 64) {
        const int r = e / nc, c = e - r * nc;
        const int idx = (int)floor((double)img[(long)(r0 + r) * W + c0 + c] * 180.0);       // cv2.calcHist: f64 bin index
        if ((unsigned)idx < 180u) atomicAdd(&bins[idx], 1);
    }
    __syncthreads();
    int mn = 0x7fffffff, mx = 0;
    for (int b = lane; b < 180; b += 64) { mn = min(mn, bins[b]); mx = max(mx, bins[b]); }
    for (int o = 32; o > 0; o >>= 1) { mn = min(mn, __shfl_xor(mn, o)); mx = max(mx, __shfl_xor(mx, o)); }
    // cv2.normalize(NORM_MINMAX, 0..1): f32 counts, f64 scale/shift rounded to f32, one f32 multiply and one f32 add
    const double scale = ((double)mx - (double)mn > 2.220446049250313e-16) ? 1.0 / ((double)mx - (double)mn) : 0.0;
    const float fs = (float)scale, fb = (float)(-(double)mn * scale);
    for (int b = lane; b < 180; b += 64) {
        const float t = (float)bins[b] * fs;      // two roundings (fp contract is off in this file)
        const float v = t + fb;
        norm[b] = v;
        hist[s * 180 + b] = v;
    }
    __syncthreads();
    if (lane == 0) {                       // bin sum in bin order, f64: the s1 / s2 of cv2.compareHist
        double acc = 0.0;
        for (int b = 0; b < 180; ++b) acc += (double)norm[b];
        hsum[s] = acc;
    }
}

}  // namespace

extern "C" int axt_box_histograms(const float *d_frames, int T_all, int H, int W, int t_offset, const int32_t *d_x,
                                  const int32_t *d_y, const int32_t *d_count, int n_frames, int cap, int box,
                                  float *d_hist, double *d_hist_sum, void *stream)
{
    AXT_REQUIRE(d_frames && d_x && d_y && d_count && d_hist && d_hist_sum, "axt_box_histograms: null argument");
    AXT_REQUIRE(n_frames >= 1 && cap >= 1 && box >= 1 && t_offset >= 0 && t_offset + n_frames <= T_all,
                "axt_box_histograms: frames [%d,%d) outside the timelapse of %d", t_offset, t_offset + n_frames, T_all);
    hipLaunchKernelGGL(box_hist_kernel, dim3(cap, n_frames), dim3(64), 0, (hipStream_t)stream, d_frames, H, W, t_offset, d_x,
                       d_y, d_count, cap, box, d_hist, d_hist_sum);
    AXT_LAUNCH_CHECK();
    return AXT_OK;
}
