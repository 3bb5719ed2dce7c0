set -x
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "cnn or winograd or benchmarked_launch or end_to_end_detections" > gpurun_out/r03b_cnntests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r03b_cnntests.log
tail -5 gpurun_out/r03b_cnntests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 > gpurun_out/r03b_bench.json 2> gpurun_out/r03b_bench.err && \
timeout -k 10 300 bash profiles/collect_pmc.sh r03b
