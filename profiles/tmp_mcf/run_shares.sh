#!/bin/bash
cd $(dirname $0)/../..
for v in 0 1; do if [ $v = 1 ]; then export AXT_MCF_ONE_PHASE=1; else unset AXT_MCF_ONE_PHASE; fi
  echo "one_phase=$v c5 share: $(AXT_MCF_DEBUG=1 python profiles/c5_stage.py 64 2>&1 | grep 'lsap: second\|assign_ids #1' | sed 's/relax.*//' | tr '\n' ' ')"
  echo "one_phase=$v c4 share: $(AXT_MCF_DEBUG=1 python bench.py --frames 132 --size 1024 --assoc mcf --steps 2 --warmup 1 --no-verify --cpu-frames 0 --no-profile 2>&1 | grep 'lsap: second' | tail -n 1 | sed 's/relax.*//')"
done
