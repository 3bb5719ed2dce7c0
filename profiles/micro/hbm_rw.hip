// Micro-benchmark: streaming write / read / read+write bandwidth of one MI355X (what bounds the two stride-2 conv
// blocks, which move 1.65 GB and 1.98 GB per 252 tile-forwards).
// hipcc --offload-arch=gfx950 -O3 hbm_rw.hip -o hbm_rw && ./hbm_rw
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void wr(f4 *p, long n) { for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = f4{1, 2, 3, 4}; }
__global__ void rd(const f4 *p, long n, f4 *o) { f4 a = {0, 0, 0, 0}; for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) a += p[i]; if (a[0] == 123.f) o[0] = a; }
__global__ void cp(const f4 *p, f4 *q, long n) { for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) q[i] = p[i]; }
int main()
{
    const long bytes = 1320l << 20, n = bytes / 16;
    f4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 3; ++which)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(wr, dim3(4096), dim3(256), 0, 0, a, n);
            if (which == 1) hipLaunchKernelGGL(rd, dim3(4096), dim3(256), 0, 0, a, n, b);
            if (which == 2) hipLaunchKernelGGL(cp, dim3(4096), dim3(256), 0, 0, a, b, n);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s %.3f ms  %.2f TB/s\n", which == 0 ? "write 1.32 GB" : which == 1 ? "read 1.32 GB " : "copy 1.32 GB (r+w 2.64)", ms, (which == 2 ? 2 : 1) * bytes / ms / 1e9);
        }
    // the same three streams over a buffer that fits the 256 MB Infinity Cache (84 MB = block-0 output of 16 tile-forwards)
    const long small = 84l << 20, ns = small / 16;
    for (int which = 0; which < 3; ++which) {
        hipEventRecord(e0);
        for (int rep = 0; rep < 16; ++rep) {
            if (which == 0) hipLaunchKernelGGL(wr, dim3(4096), dim3(256), 0, 0, a, ns);
            if (which == 1) hipLaunchKernelGGL(rd, dim3(4096), dim3(256), 0, 0, a, ns, b);
            if (which == 2) hipLaunchKernelGGL(cp, dim3(4096), dim3(256), 0, 0, a, b, ns);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("84 MB x16 %s %.3f ms  %.2f TB/s\n", which == 0 ? "write" : which == 1 ? "read " : "copy ", ms, (which == 2 ? 2 : 1) * 16.0 * small / ms / 1e9);
    }
    return 0;
}
