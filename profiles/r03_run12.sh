set -x
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03k_bench.json 2> gpurun_out/r03k_bench.err || exit 1
timeout -k 10 300 python bench.py --assoc mcf --steps 10 --warmup 2 > gpurun_out/r03k_bench_mcf.json 2>> gpurun_out/r03k_bench.err || exit 1
timeout -k 10 300 python bench.py --input host --steps 20 --warmup 5 --cpu-frames 0 > gpurun_out/r03k_bench_host.json 2>> gpurun_out/r03k_bench.err || exit 1
timeout -k 10 500 bash profiles/collect.sh r03k || exit 1
timeout -k 10 300 bash profiles/collect_pmc.sh r03k
