// Probe: operand / result lane layout of v_mfma_f32_4x4x1_16b_f32 on gfx950.
// hipcc --offload-arch=gfx950 -O2 mfma_layout.hip -o mfma_layout && ./mfma_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float *out)
{
    const int l = threadIdx.x;
    // A value encodes (lane), B value encodes (lane): D = A*B -> recover which lanes met
    float a = (float)(l + 1), b = (float)(1000 * (l + 1));
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}
int main()
{
    float *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            // expected: D[lane l][vgpr i] = A[lane 4*(l/4)+i] * B[lane l]
            const float want = (float)(4 * (l / 4) + i + 1) * (float)(1000 * (l + 1));
            if (h[l * 4 + i] != want) { ok = 0; if (l < 8) printf("lane %d vgpr %d: got %.0f want %.0f\n", l, i, h[l * 4 + i], want); }
        }
    printf("layout D[l][i] = A[4*(l/4)+i] * B[l]: %s\n", ok ? "CONFIRMED" : "MISMATCH");
    return 0;
}
