#!/bin/bash
# Matrix-pipe / wait counters of the conv kernels (run on the GPU box from the repo root):
#   bash profiles/collect_pmc.sh <tag> [bench.py arguments]   -> gpurun_out/<tag>_pmc/..., gpurun_out/<tag>_pmc.csv
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
python3 $R/profiles/profile_meta.py ${tag}_pmc --cpu-frames 0 --no-verify --no-host-variant "$@"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT \
    -d $R/gpurun_out/${tag}_pmc -o p --output-format csv -- python3 $R/bench.py --cpu-frames 0 --no-verify --no-host-variant --steps 2 --warmup 1 "$@" > $R/gpurun_out/${tag}_pmc.log 2>&1
cd $R
python3 profiles/pmc_summary.py $(find gpurun_out/${tag}_pmc -name '*counter_collection.csv' | head -n 1) conv > gpurun_out/${tag}_pmc.csv
cat gpurun_out/${tag}_pmc.csv
