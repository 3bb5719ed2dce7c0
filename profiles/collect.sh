#!/bin/bash
# Collects the rocprofv3 summaries that profiles/summarize.py condenses (run on the GPU box from the repo root):
#   bash profiles/collect.sh <tag>      -> gpurun_out/<tag>_{stats,fetch,write}/...
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1
python3 $R/profiles/profile_meta.py ${tag} --cpu-frames 0 --no-verify --no-host-variant
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_stats -o p --output-format csv -- python3 $R/bench.py --cpu-frames 0 --no-verify --no-host-variant --steps 5 --warmup 2 > $R/gpurun_out/${tag}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${tag}_fetch -o p --output-format csv -- python3 $R/bench.py --cpu-frames 0 --no-verify --no-host-variant --steps 2 --warmup 1 > $R/gpurun_out/${tag}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${tag}_write -o p --output-format csv -- python3 $R/bench.py --cpu-frames 0 --no-verify --no-host-variant --steps 2 --warmup 1 > $R/gpurun_out/${tag}_write.log 2>&1
cd $R
ls gpurun_out/${tag}_stats gpurun_out/${tag}_fetch gpurun_out/${tag}_write
