// Shared helpers for libaxtrack_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
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

// the kept tiles of a frame (tile row, tile column), passed to the kernels by value
struct TileList { int n; short yx[2 * 256]; };

// cnn_front.hip: conv blocks 0 + 1 in one launch: frames -> block 1's output [B,40,128,128]
int axt_launch_conv_fused01(const float *in, const float *w0, const float *b0, const float *w1, const float *b1, float *out,
                            int B, hipStream_t st, int Hf, int Wf, int t0, int tstep, int item0, int n_tiles, const TileList &tl);
