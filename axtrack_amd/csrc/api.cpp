// Error reporting and small host-side helpers of libaxtrack_hip.so.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>

#include "../../include/axtrack_hip.h"

static thread_local char g_err[512] = "";

void axt_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *axt_last_error(void) { return g_err; }
extern "C" int axt_abi_version(void) { return 1; }

static inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

extern "C" int64_t axt_arc_cost_int(double cost, int kind, int64_t a, int64_t b)
{
    const uint64_t key = ((uint64_t)kind << 60) ^ ((uint64_t)a << 30) ^ (uint64_t)b;
    const int64_t pert = (int64_t)(splitmix64(key) & 0xFFFFull);
    return ((int64_t)nearbyint(cost * 1e6)) * 65536 + pert;
}
