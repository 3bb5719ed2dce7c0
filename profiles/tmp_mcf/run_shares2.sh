#!/bin/bash
cd $(dirname $0)/../..
for v in 0 1; do if [ $v = 1 ]; then export AXT_MCF_ONE_PHASE=1; else unset AXT_MCF_ONE_PHASE; fi
  python bench.py --frames 132 --size 1024 --assoc mcf --steps 3 --warmup 1 --no-verify --cpu-frames 0 --no-profile > /tmp/b.json 2>/dev/null
  echo "one_phase=$v c4 share: $(python -c "import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stages'])")"
done
