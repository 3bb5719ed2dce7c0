// Micro-benchmark (round 2): issue rate and clock of the two bf16 MFMA shapes on gfx950, on RANDOM operands, one wave per SIMD
// and two, with the accumulator counts a convolution tile uses (16x16x32: 16 tiles; 32x32x16: 4 tiles -- 64 registers both).
// Cycles per MFMA come from s_memtime around the loop, the clock from s_memtime / s_memrealtime (100 MHz).
// hipcc --offload-arch=gfx950 -O3 mfma_bf16_shapes.hip -o mfma_bf16_shapes && ./mfma_bf16_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const u4 *__restrict__ data, float *out, long long *clk, int iters)
{
    bf8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf8, data[(threadIdx.x + 256 * i) & 1023]);
        b[i] = __builtin_bit_cast(bf8, data[(threadIdx.x + 256 * i + 77) & 1023]);
    }
    float s = 0;
    long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;
    if (SHAPE == 0) {
        f4 c[16];
        for (int i = 0; i < 16; ++i) c[i] = (f4){0, 0, 0, 0};
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], c[i], 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    } else {
        f16v c[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0;
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[i], c[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + 1) & 3], b[i], c[i], 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += c[i][j];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main()
{
    const int nblk_max = 512;
    float *out; hipMalloc(&out, nblk_max * 256 * 4);
    long long *clk; hipMalloc(&clk, nblk_max * 2 * 8);
    u4 *data; hipMalloc(&data, 1024 * 16);
    unsigned short h[8192];
    srand(1);
    for (int i = 0; i < 8192; ++i) { float x = (rand() / (float)RAND_MAX) * 2 - 1; unsigned u; memcpy(&u, &x, 4); h[i] = u >> 16; }
    hipMemcpy(data, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 40000;
    for (int per_cu = 1; per_cu <= 2; ++per_cu)
        for (int shape = 0; shape < 2; ++shape) {
            const int nblk = 256 * per_cu;
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(nblk), dim3(256), 0, 0, data, out, clk, iters);
                else            hipLaunchKernelGGL(k<1>, dim3(nblk), dim3(256), 0, 0, data, out, clk, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            long long hc[2 * nblk_max];
            hipMemcpy(hc, clk, nblk * 16, hipMemcpyDeviceToHost);
            double cyc = 0, real = 0;
            for (int i = 0; i < nblk; ++i) { cyc += hc[2 * i]; real += hc[2 * i + 1]; }
            const double n_mfma = shape == 0 ? 16.0 * iters : 8.0 * iters;
            const double flop_per = shape == 0 ? 16384.0 : 32768.0;
            const double fl = (double)nblk * 4 * n_mfma * flop_per;
            printf("%s bf16, %d wave(s)/SIMD: %.3f ms  %.0f TFLOP/s (/6 = %.0f f32-equivalent)  %.1f cycles/MFMA/wave  clock %.2f GHz\n",
                   shape == 0 ? "16x16x32" : "32x32x16", per_cu, best, fl / best / 1e9, fl / best / 6e9,
                   cyc / nblk / n_mfma, cyc / real * 0.1);
        }
    return 0;
}
