// Micro-benchmark: issue rate of the f32-input MFMA shapes on gfx950 (is 4x4x1 as fast per FLOP as 16x16x4?).
// hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
    f4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = (f4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (SHAPE == 0) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
            else            c[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
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
            const double flops_per = shape == 0 ? 2048.0 : 512.0;
            const double fl = 2048.0 * 4 * iters * 8 * flops_per;
            printf("%s: %.3f ms  %.1f TFLOP/s\n", shape == 0 ? "16x16x4" : "4x4x1", ms, fl / ms / 1e9);
        }
    }
    return 0;
}
