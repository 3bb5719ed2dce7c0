// Probe: what does an LDS-DMA buffer load (buffer_load_dword ... lds) write for a lane whose offset fails the range check?
// Build: hipcc --offload-arch=gfx950 -O2 -o lds_dma_oob lds_dma_oob.hip ; prints the LDS image after the load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const float *src, int n, float *out, int width)
{
    __shared__ __attribute__((aligned(16))) float buf[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) buf[i] = -7.0f;           // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, n * 4, 0x00020000);
    const int lane = threadIdx.x;
    // odd lanes in range (element lane), even lanes beyond the range
    if (width == 4) {
        const unsigned off = (lane & 1) ? lane * 4u : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)buf, 4, (int)off, 0, 0, 0);
    } else {
        const unsigned off = (lane & 1) ? lane * 16u : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)buf, 16, (int)off, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = buf[i];
}
int main()
{
    const int n = 1024;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = 100.f + i;
    float *d, *o;
    hipMalloc(&d, n * 4);
    hipMalloc(&o, 512 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int width : {4, 16}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, n, o, width);
        std::vector<float> r(512);
        hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
        printf("width %d:", width);
        for (int i = 0; i < (width == 4 ? 16 : 32); ++i) printf(" %g", r[i]);
        printf("\n");
    }
    return 0;
}
