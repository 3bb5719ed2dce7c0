#!/bin/bash
cd $(dirname $0)/../..
for v in 0 1; do if [ $v = 1 ]; then export AXT_MCF_ONE_PHASE=1; else unset AXT_MCF_ONE_PHASE; fi
  echo "one_phase=$v c3: $(python profiles/mcf_timing.py 2>&1 | grep solve | awk '{printf "%s ", $2}')"
done
unset AXT_MCF_ONE_PHASE
AXT_MCF_DEBUG=1 python profiles/mcf_timing.py 2>&1 | grep "lsap" | tail -n 1
python profiles/tmp_mcf/c4net.py 4 2 > /tmp/c4gen.log 2>&1
cat > /tmp/c4run.py <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.')
from axtrack_amd import hotpath as hp
z = np.load('/tmp/c4net.npz'); net = [z[f'arr_{i}'] for i in range(7)]
for _ in range(3):
    t = time.perf_counter(); res = hp.mcf_solve(*net[:6], 5, 1800); print('c4 solve %.3f s' % (time.perf_counter() - t), res[2], res[3], flush=True)
PY
for v in 0 1; do if [ $v = 1 ]; then export AXT_MCF_ONE_PHASE=1; else unset AXT_MCF_ONE_PHASE; fi
  echo "one_phase=$v"; python /tmp/c4run.py 2>&1 | tail -n 3
done
