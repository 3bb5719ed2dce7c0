// Shared helpers for libaxtrack_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <stdio.h>
#include <string.h>

#include "../../include/axtrack_hip.h"

void axt_set_error(const char *fmt, ...);

#define AXT_CHECK_HIP(expr)                                                                   \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            axt_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
            return AXT_EHIP;                                                                  \
        }                                                                                     \
    } while (0)

#define AXT_REQUIRE(cond, ...)                      \
    do {                                            \
        if (!(cond)) {                              \
            axt_set_error(__VA_ARGS__);             \
            return AXT_EINVAL;                      \
        }                                           \
    } while (0)

#define AXT_LAUNCH_CHECK() AXT_CHECK_HIP(hipGetLastError())

static inline int axt_cdiv(int a, int b) { return (a + b - 1) / b; }

// Launch attributes (hipFuncSetAttribute) and occupancy answers belong to the CURRENT device, not to the process: a
// second device in one process (Detector(device='cuda:1')) must get its own. One of these per launcher remembers, per
// device index, that the launcher's kernels are set up there (a bit per device; devices beyond 63 are set up on every
// call). No other state survives a call, as include/axtrack_hip.h promises.
struct AxtOncePerDevice {
    std::atomic<uint64_t> done{0};
    int dev = 0;
    // true: the kernels of this launcher still need their attributes on the current device (call mark() afterwards)
    bool pending()
    {
        if (hipGetDevice(&dev) != hipSuccess) dev = 64;
        return dev < 0 || dev >= 64 || !(done.load(std::memory_order_acquire) >> dev & 1);
    }
    void mark() { if (dev >= 0 && dev < 64) done.fetch_or(1ull << dev, std::memory_order_release); }
};
template <typename K>
static inline int axt_max_dynamic_lds(K kernel, int bytes, AxtOncePerDevice &once)
{
    if (!once.pending()) return AXT_OK;
    AXT_CHECK_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    once.mark();
    return AXT_OK;
}

// the kept tiles of a frame (tile row, tile column), passed to the kernels by value
struct TileList { int n; short yx[2 * 256]; };

// cnn_front.hip: conv blocks 0 + 1 in one launch: frames -> block 1's output [B,40,128,128]
int axt_launch_conv_fused01(const float *in, const float *w0, const float *b0, const float *w1, const float *b1, float *out,
                            int B, hipStream_t st, int Hf, int Wf, int t0, int tstep, int item0, int n_tiles, const TileList &tl);
