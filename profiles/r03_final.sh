#!/bin/bash
# Round 3, final tree: the bench lines and rocprofv3 summaries committed as profiles/r03z_* (run on the GPU box from the repo root).
set -x
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03z_bench.json 2> gpurun_out/r03z_bench.err || exit 1
timeout -k 10 300 python bench.py --assoc mcf --steps 10 --warmup 2 > gpurun_out/r03z_bench_mcf.json 2>> gpurun_out/r03z_bench.err || exit 1
timeout -k 10 300 python bench.py --input host --steps 20 --warmup 5 --cpu-frames 0 > gpurun_out/r03z_bench_host.json 2>> gpurun_out/r03z_bench.err || exit 1
for w in assoc-c3 assoc-c4; do for a in mcf hungarian; do
  timeout -k 10 300 python bench.py --workload $w --assoc $a --steps 5 --warmup 2 > gpurun_out/r03z_bench_${w}_${a}.json 2>> gpurun_out/r03z_bench.err || exit 1
done; done
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --single-device --assoc mcf --frames 132 --steps 3 --warmup 1 --no-verify > gpurun_out/r03z_bench_2ranks_mcf.json 2>> gpurun_out/r03z_bench.err || exit 1
timeout -k 10 500 bash profiles/collect.sh r03z || exit 1
timeout -k 10 300 bash profiles/collect_pmc.sh r03z
