// Path lengths on a masked grid (weights {1 on mask, 65536 off}, reference AxonDetections.py:598).
// Placeholder entry point: the masked search is not built yet, the call fails loudly.
#include "axt_common.h"

int axt_path_cost_masked(const int32_t *, const int32_t *, int, const int32_t *, const int32_t *, int, const uint8_t *,
                         int, int, int, int, int32_t *, hipStream_t)
{
    axt_set_error("axt_path_cost: masked grids are not implemented in this build");
    return AXT_EINVAL;
}
