// Timelapse preprocessing as ONE fused, HBM-bound pass on gfx950 (the step in front of the
// hot path): reference Timelapse._read_tiff / _clip_image_values / _log_adjust_image / _standardize
// (axtrack/Timelapse.py:205-326), which the reference runs as five numpy passes over the dense timelapse plus a
// scipy.sparse round trip.
//
//   x = u16 * (1/65535)          skimage.util.img_as_float32 (Timelapse.py:207)
//   x = mask ? x : 0             :217
//   x = max(x - offset, 0)       :219-223     (offset already divided by 2^16 by the caller)
//   x = x < clip ? 0 : x         :245-249
//   x = log2(1 + x)              skimage.exposure.adjust_log(x, gain=1) (:255-258)
//   x = x / scale                :312 (the mean is not subtracted)
//
// skimage and tifffile are absent from the reference tree and from this image: img_as_float32 and adjust_log are
// restated from their published behaviour -- PARITY UNPINNED for those two steps.
// Algorithmic traffic: 2 B in (+ 1 B of mask, cache-resident) and 4 B out per pixel; 8 pixels per lane
// (16-byte load, two 16-byte stores).
#include "axt_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float prep_one(float x, bool m, float offset, float clip, int log_correct, float scale)
{
    x = m ? x : 0.f;
    if (offset != 0.f) {
        x = __fsub_rn(x, offset);
        x = x < 0.f ? 0.f : x;
    }
    if (clip != 0.f) x = x < clip ? 0.f : x;
    if (log_correct) x = log2f(__fadd_rn(1.0f, x));
    return __fdiv_rn(x, scale);
}

__global__ __launch_bounds__(256) void preprocess_u16_kernel(const unsigned short *__restrict__ raw,
                                                             const unsigned char *__restrict__ mask, long n_px,
                                                             long frame_px, float offset, float clip, int log_correct,
                                                             float scale, float *__restrict__ out)
{
    const float inv = 1.0f / 65535.0f;
    const long stride = (long)gridDim.x * blockDim.x * 8;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n_px; i += stride) {
        if (i + 8 <= n_px && (frame_px % 8 == 0)) {
            const u16x8 v = *reinterpret_cast<const u16x8 *>(raw + i);
            const long mp = i % frame_px;
            f32x4 o0, o1;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool m = mask ? mask[mp + j] != 0 : true;
                const float r = prep_one(__fmul_rn((float)v[j], inv), m, offset, clip, log_correct, scale);
                if (j < 4) o0[j] = r; else o1[j - 4] = r;
            }
            *reinterpret_cast<f32x4 *>(out + i) = o0;
            *reinterpret_cast<f32x4 *>(out + i + 4) = o1;
        } else {
            for (long k = i; k < n_px && k < i + 8; ++k) {
                const bool m = mask ? mask[k % frame_px] != 0 : true;
                out[k] = prep_one(__fmul_rn((float)raw[k], inv), m, offset, clip, log_correct, scale);
            }
        }
    }
}

}  // namespace

extern "C" int axt_preprocess_u16(const uint16_t *d_raw, const uint8_t *d_mask, int T, int H, int W, float offset,
                                  float clip_lower, int log_correct, float scale, float *d_out, void *stream)
{
    AXT_REQUIRE(d_raw && d_out, "null argument");
    AXT_REQUIRE(T > 0 && H > 0 && W > 0 && scale > 0.f, "bad argument");
    const long n = (long)T * H * W, frame = (long)H * W;
    long blocks = (n / 8 + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(preprocess_u16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_raw, d_mask, n,
                       frame, offset, clip_lower, log_correct, scale, d_out);
    AXT_LAUNCH_CHECK();
    return AXT_OK;
}
