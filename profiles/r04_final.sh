#!/bin/bash
# Round 4: the bench lines and rocprofv3 summaries committed as profiles/<tag>_* (run on the GPU box from the repo root):
#   bash profiles/r04_final.sh <tag> profile   -> the rocprofv3 passes + their meta files (summarised by profiles/summarize.py / pmc_summary.py)
#   bash profiles/r04_final.sh <tag> bench     -> the bench lines (their roofline counters come from the committed summaries of the same sources)
#   bash profiles/r04_final.sh <tag> full      -> BASELINE configs 4 and 5 at full size on one GPU (profiles/full_config.py)
set -x
T=${1:-r04z}
mkdir -p gpurun_out
if [ "$2" = profile ]; then
  timeout -k 10 500 bash profiles/collect.sh $T || exit 1
  timeout -k 10 300 bash profiles/collect_pmc.sh $T
  exit $?
fi
if [ "$2" = full ]; then
  AXT_MCF_DEBUG=1 timeout -k 10 400 python profiles/full_config.py c4 > gpurun_out/${T}_c4_full.log 2>&1 || exit 1
  AXT_MCF_DEBUG=1 timeout -k 10 400 python profiles/full_config.py c5 > gpurun_out/${T}_c5_full.log 2>&1
  exit $?
fi
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err || exit 1
timeout -k 10 300 python bench.py --assoc mcf --steps 10 --warmup 2 > gpurun_out/${T}_bench_mcf.json 2>> gpurun_out/${T}_bench.err || exit 1
timeout -k 10 300 python bench.py --workload c4 --steps 5 --warmup 2 > gpurun_out/${T}_bench_c4.json 2>> gpurun_out/${T}_bench.err || exit 1
timeout -k 10 400 python bench.py --workload c5 --steps 5 --warmup 2 > gpurun_out/${T}_bench_c5.json 2>> gpurun_out/${T}_bench.err || exit 1
timeout -k 10 300 python bench.py --workload c2 --steps 20 --warmup 5 > gpurun_out/${T}_bench_c2.json 2>> gpurun_out/${T}_bench.err || exit 1
timeout -k 10 300 python bench.py --input host --steps 20 --warmup 5 --cpu-frames 0 > gpurun_out/${T}_bench_host.json 2>> gpurun_out/${T}_bench.err || exit 1
for w in assoc-c3 assoc-c4; do for a in mcf hungarian; do
  timeout -k 10 300 python bench.py --workload $w --assoc $a --steps 5 --warmup 2 > gpurun_out/${T}_bench_${w}_${a}.json 2>> gpurun_out/${T}_bench.err || exit 1
done; done
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --single-device --assoc mcf --frames 132 --steps 3 --warmup 1 --no-verify > gpurun_out/${T}_bench_2ranks_mcf.json 2>> gpurun_out/${T}_bench.err || exit 1
timeout -k 10 400 python bench.py --gpus 2 --backend gloo --single-device --workload c4 --steps 2 --warmup 1 --cpu-frames 0 > gpurun_out/${T}_bench_2ranks_c4.json 2>> gpurun_out/${T}_bench.err || exit 1
