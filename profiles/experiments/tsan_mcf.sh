#!/bin/bash
# ThreadSanitizer over the host flow solver with the team search forced onto small networks (CPU build only: the pool offers no GPU
# sanitizers). Dumps two networks from tests/helpers.py, builds mcf.cpp + api.cpp with -fsanitize=thread around a 30-line driver.
#   bash profiles/experiments/tsan_mcf.sh        -> no "WARNING: ThreadSanitizer" lines (round 4: clean)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
D=$(mktemp -d)
cd $D
python3 - <<PY
import sys, numpy as np
sys.path.insert(0, '$R'); sys.path.insert(0, '$R/tests')
from helpers import c3_network, moving_network
for name, net in (('c3', c3_network()[:6]), ('mv', moving_network(100, 512, 90, seed=11)[:6])):
    with open(f'{name}.bin', 'wb') as f:
        np.array([len(net[0]), len(net[4])], np.int64).tofile(f)
        for a, t in zip(net, (np.int64, np.int64, np.int64, np.int64, np.int32, np.int64)):
            np.ascontiguousarray(a, t).tofile(f)
PY
cat > drv.cpp <<CPP
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include "$R/include/axtrack_hip.h"
int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb");
    int64_t hd[2];
    if (fread(hd, 8, 2, f) != 2) return 2;
    const int n = (int)hd[0];
    const int64_t m = hd[1];
    std::vector<int64_t> obs(n), en(n), ex(n), rp(n + 1), cost(m);
    std::vector<int32_t> col(m), nxt(n), tr(n);
    if (fread(obs.data(), 8, n, f) + fread(en.data(), 8, n, f) + fread(ex.data(), 8, n, f) + fread(rp.data(), 8, n + 1, f) + fread(col.data(), 4, m, f) + fread(cost.data(), 8, m, f) == 0) return 2;
    int nt = 0;
    int64_t total = 0;
    const int rc = axt_mcf_solve(n, obs.data(), en.data(), ex.data(), rp.data(), col.data(), cost.data(), 5, 100000, nxt.data(), tr.data(), &nt, &total);
    printf("rc %d tracks %d total %lld\n", rc, nt, (long long)total);
    return rc;
}
CPP
g++ -std=c++17 -O1 -g -fsanitize=thread -pthread drv.cpp $R/axtrack_amd/csrc/mcf.cpp $R/axtrack_amd/csrc/api.cpp -o drv_tsan
export AXT_MCF_PAR_MIN_N=0 AXT_MCF_PAR_SWITCH=48
AXT_MCF_THREADS=5 ./drv_tsan c3.bin
AXT_MCF_THREADS=6 ./drv_tsan mv.bin
AXT_MCF_ONE_PHASE=1 AXT_MCF_MIN_LEAF=512 AXT_MCF_THREADS=6 ./drv_tsan c3.bin
AXT_MCF_TWO_PHASE=1 AXT_MCF_ENDS_THREADS=4 AXT_MCF_THREADS=6 ./drv_tsan mv.bin
