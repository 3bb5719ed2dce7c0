// Micro-benchmark: issue rate of the bf16 MFMA shapes on gfx950, to price an FP32-by-split-bf16 convolution
// (x = hi + mid + lo, 6 products kept) against the f32 MFMA path the CNN uses now. See DESIGN.md section 9.
// hipcc --offload-arch=gfx950 -O3 mfma_bf16_rate.hip -o mfma_bf16_rate && ./mfma_bf16_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(threadIdx.x * 0.002f - i); }
    float s = 0;
    if (SHAPE == 0) {
        f4 c[8];
        for (int i = 0; i < 8; ++i) c[i] = (f4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    } else {
        f16v c[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += c[i][j];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    float *out; hipMalloc(&out, 2048 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int shape = 0; shape < 2; ++shape) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, out, iters);
            else            hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double per_it = shape == 0 ? 8 * 16384.0 : 4 * 32768.0;
            const double fl = 2048.0 * 4 * iters * per_it;
            printf("%s bf16: %.3f ms  %.1f TFLOP/s  (/6 = %.1f f32-equivalent)\n", shape == 0 ? "16x16x32" : "32x32x16", ms, fl / ms / 1e9, fl / ms / 6e9);
        }
    }
    return 0;
}
