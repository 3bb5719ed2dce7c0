// Detection metrics: the reference's compute_TP_FP_FN (axtrack/AxonDetections.py:409-466) for every
// frame and every confidence threshold in one launch. Evaluation-side consumer of the detections; integer work.
//
// One wavefront per (frame, threshold). Labels are visited in order (the result depends on it): a label's
// candidates are the detections closer than min_dist (dx^2 + dy^2 < min_dist^2, exact for integer anchors) whose
// confidence exceeds the threshold; the closest wins (ties: first); if that detection was already claimed by an
// earlier label, the label is a false negative -- it does not fall back to its second choice (:447-452).
// Reference quirk kept: an empty side is replaced by ONE row (conf, x, y) = (0, 0, 0) (:434-437), i.e. a phantom
// label / detection at the origin takes part.
#include "axt_common.h"

namespace {

// wave-wide minimum of a 32-bit key: DPP inside the rows of 16 lanes, v_readlane across the four rows
#define AXT_DPP_MIN32(v, ctrl)                                                             \
    {                                                                                      \
        const unsigned o_ = __builtin_amdgcn_update_dpp(v, v, ctrl, 0xF, 0xF, false);      \
        v = o_ < v ? o_ : v;                                                               \
    }
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    AXT_DPP_MIN32(v, 0xB1);
    AXT_DPP_MIN32(v, 0x4E);
    AXT_DPP_MIN32(v, 0x141);
    AXT_DPP_MIN32(v, 0x140);
    unsigned best = 0xffffffffu;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned o = __builtin_amdgcn_readlane(v, r * 16);
        best = o < best ? o : best;
    }
    return best;
}

__global__ __launch_bounds__(64) void confusion_kernel(
    const float *__restrict__ conf, const int *__restrict__ x, const int *__restrict__ y, const int *__restrict__ count,
    int cap, const int *__restrict__ gx, const int *__restrict__ gy, const int *__restrict__ gcount, int gcap,
    const double *__restrict__ thrs, int n_thr, int min_d2, int k_mask, int *__restrict__ confusion,
    unsigned char *__restrict__ fp_mask, unsigned char *__restrict__ fn_mask)
{
    extern __shared__ unsigned taken[];                  // one bit per detection
    const int f = blockIdx.x, k = blockIdx.y, lane = threadIdx.x;
    const int nd_real = min(count[f], cap), ng_real = min(gcount[f], gcap);
    const int nd = max(nd_real, 1), ng = max(ng_real, 1);      // the phantom row of an empty side
    const double thr = thrs[k];
    for (int w = lane; w < (nd + 31) / 32; w += 64) taken[w] = 0;
    __syncthreads();
    const long d0 = (long)f * cap, g0 = (long)f * gcap;
    int n_fn = 0;
    for (int i = 0; i < ng; ++i) {
        const int lx = ng_real ? gx[g0 + i] : 0, ly = ng_real ? gy[g0 + i] : 0;
        unsigned best = 0xffffffffu;                     // (d2 << 11 | j): d2 < min_d2 <= 2^20, j < 2^11
        for (int j = lane; j < nd; j += 64) {
            const int dx = (nd_real ? x[d0 + j] : 0) - lx, dy = (nd_real ? y[d0 + j] : 0) - ly;
            const long d2 = (long)dx * dx + (long)dy * dy;
            const double c = nd_real ? (double)conf[d0 + j] : 0.0;      // f32 confidence against the f64 threshold
            if (d2 < min_d2 && c > thr) {
                const unsigned key = ((unsigned)d2 << 11) | (unsigned)j;
                best = key < best ? key : best;
            }
        }
        best = wave_min_u32(best);
        bool fn = true;
        if (best != 0xffffffffu) {
            const int j = best & 2047;
            if (!((taken[j >> 5] >> (j & 31)) & 1u)) {
                fn = false;
                __syncthreads();
                if (lane == 0) taken[j >> 5] |= 1u << (j & 31);
            }
        }
        __syncthreads();
        n_fn += fn;
        if (k == k_mask && fn_mask && lane == 0 && i < ng_real) fn_mask[g0 + i] = fn;
    }
    int tp = 0, fp = 0;
    for (int j = lane; j < nd; j += 64) {
        const bool t = (taken[j >> 5] >> (j & 31)) & 1u;
        const bool above = (nd_real ? (double)conf[d0 + j] : 0.0) > thr;
        tp += t;
        fp += !t && above;
        if (k == k_mask && fp_mask && j < nd_real) fp_mask[d0 + j] = !t && above;
    }
    for (int o = 32; o > 0; o >>= 1) { tp += __shfl_xor(tp, o); fp += __shfl_xor(fp, o); }
    if (lane == 0) {
        int *c = confusion + ((long)f * 3) * n_thr + k;
        c[0] = tp;
        c[n_thr] = fp;
        c[2 * n_thr] = n_fn;
    }
}

}  // namespace

extern "C" int axt_detection_confusion(const float *d_conf, const int32_t *d_x, const int32_t *d_y, const int32_t *d_count,
                                       int n_frames, int cap, const int32_t *d_gx, const int32_t *d_gy,
                                       const int32_t *d_gcount, int gcap, const double *d_thrs, int n_thr, int min_dist,
                                       int k_mask, int32_t *d_confusion, uint8_t *d_fp_mask, uint8_t *d_fn_mask,
                                       void *stream)
{
    AXT_REQUIRE(d_conf && d_x && d_y && d_count && d_gx && d_gy && d_gcount && d_thrs && d_confusion, "axt_detection_confusion: null argument");
    AXT_REQUIRE(n_frames >= 1 && cap >= 1 && cap <= 2048 && gcap >= 1 && n_thr >= 1, "axt_detection_confusion: bad shape");
    AXT_REQUIRE(min_dist >= 0 && min_dist <= 1024, "axt_detection_confusion: min_dist %d out of range", min_dist);
    AXT_REQUIRE(k_mask < n_thr, "axt_detection_confusion: mask threshold index %d >= %d", k_mask, n_thr);
    hipLaunchKernelGGL(confusion_kernel, dim3(n_frames, n_thr), dim3(64), sizeof(unsigned) * ((cap + 31) / 32 + 1),
                       (hipStream_t)stream, d_conf, d_x, d_y, d_count, cap, d_gx, d_gy, d_gcount, gcap, d_thrs, n_thr,
                       min_dist * min_dist, k_mask, d_confusion, d_fp_mask, d_fn_mask);
    AXT_LAUNCH_CHECK();
    return AXT_OK;
}
