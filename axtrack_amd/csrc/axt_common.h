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
