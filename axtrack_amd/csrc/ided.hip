// IDed_dets_all on the GPU (AxonDetections._agg_all_IDed_dets, AxonDetections.py:825-842): the dense
// [n_rows, 3*n_frames] f64 table (anchor_x, anchor_y, conf per frame; NaN where an axon is absent) that
// inference() hands back. Its size grows with frames x identities, so it is filled where the detections
// and the trajectory ids already live and crosses PCIe once.
//
// Column of frame f: 3*slot(f), with slot(f) = f, or -- reproducing the reference's label quirk, where
// pd.concat drops the frames that have no IDed detection (:831) and labels are column_position//3 --
// the rank of f among the frames that have one.
#include "axt_common.h"

namespace {

// slot[f] as defined above; single workgroup, ballot scan over chunks of 1024 frames
__global__ __launch_bounds__(1024) void ided_slot_kernel(const int *__restrict__ track, const int *__restrict__ count,
                                                         int n_frames, int cap, int quirk, int *__restrict__ slot)
{
    __shared__ int wtot[16];
    __shared__ int base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int f0 = 0; f0 < n_frames; f0 += 1024) {
        const int f = f0 + threadIdx.x;
        bool present = false;
        if (f < n_frames) {
            const int n = min(count[f], cap);
            for (int k = 0; k < n && !present; ++k) present = track[(long)f * cap + k] >= 0;
        }
        const unsigned long long mk = __ballot(present);
        if (lane == 0) wtot[wave] = __popcll(mk);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wtot[w];
        if (f < n_frames) slot[f] = quirk ? (present ? off + __popcll(mk & ((1ull << lane) - 1ull)) : -1) : f;
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int w = 0; w < 16; ++w) tot += wtot[w];
            base += tot;
        }
        __syncthreads();
    }
}

__global__ void ided_fill_kernel(double *__restrict__ table, long n)
{
    const long stride = (long)gridDim.x * blockDim.x;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);      // numpy's float('nan') bit pattern
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) table[i] = nan;
}

__global__ void ided_scatter_kernel(const int *__restrict__ track, const float *__restrict__ conf, const int *__restrict__ x,
                                    const int *__restrict__ y, const int *__restrict__ count, int n_frames, int cap,
                                    const int *__restrict__ slot, const int *__restrict__ id_row, int n_ids, int n_rows,
                                    double *__restrict__ table)
{
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= (long)n_frames * cap) return;
    const int f = s / cap, k = s - (long)f * cap;
    if (k >= min(count[f], cap)) return;
    const int id = track[s];
    if (id < 0 || id >= n_ids) return;
    const int row = id_row ? id_row[id] : id;
    if (row < 0 || row >= n_rows) return;
    double *c = table + (long)row * 3 * n_frames + 3l * slot[f];
    c[0] = (double)x[s];
    c[1] = (double)y[s];
    c[2] = (double)conf[s];
}

}  // namespace

extern "C" int axt_ided_table(const int32_t *d_track, const float *d_conf, const int32_t *d_x, const int32_t *d_y,
                              const int32_t *d_count, int n_frames, int cap, int n_ids, const int32_t *d_id_row, int n_rows,
                              int label_quirk, int32_t *d_work, double *d_table, void *stream)
{
    AXT_REQUIRE(d_track && d_conf && d_x && d_y && d_count && d_work && n_frames >= 1 && cap >= 1, "axt_ided_table: bad argument");
    AXT_REQUIRE(n_ids >= 0 && n_rows >= 0 && (d_id_row || n_rows == n_ids), "axt_ided_table: n_rows %d != n_ids %d without a row map", n_rows, n_ids);
    if (n_rows == 0) return AXT_OK;
    AXT_REQUIRE(d_table, "axt_ided_table: null table");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ided_slot_kernel, dim3(1), dim3(1024), 0, st, d_track, d_count, n_frames, cap, label_quirk ? 1 : 0, d_work);
    AXT_LAUNCH_CHECK();
    const long cells = (long)n_rows * 3 * n_frames;
    const long want = (cells + 255) / 256;
    hipLaunchKernelGGL(ided_fill_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, st, d_table, cells);
    AXT_LAUNCH_CHECK();
    const long slots = (long)n_frames * cap;
    hipLaunchKernelGGL(ided_scatter_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, st, d_track, d_conf, d_x, d_y,
                       d_count, n_frames, cap, (const int *)d_work, d_id_row, n_ids, n_rows, d_table);
    AXT_LAUNCH_CHECK();
    return AXT_OK;
}
