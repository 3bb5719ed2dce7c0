#!/bin/bash
cd $(dirname $0)/../..
for leaf in 256 512 1024 2048 4096; do for th in 8 16 32; do
  echo "min_leaf $leaf threads $th: $(AXT_MCF_MIN_LEAF=$leaf AXT_MCF_THREADS=$th python profiles/mcf_timing.py 2>&1 | grep solve | awk '{printf "%s ", $2}')"
done; done
