#!/bin/bash
cd $(dirname $0)
python c4net.py 4 2 > /tmp/c4gen.log 2>&1
nproc
for rep in 1 2; do
for tr in 0 2 4; do
  echo "c3 transfer $tr: $(REPS=5 AXT_TRANSFER=$tr python run_dbg.py 2>&1 | grep '^rc' | awk '{printf "%.1f ", $4}')"
done
for tr in 0 2 4; do
  echo "c4 transfer $tr: $(AXT_TRANSFER=$tr python run_c4.py 2>&1 | tail -1)"
done
done
