set -x
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "host_resident or winograd_and_direct or benchmarked_launch" > gpurun_out/r03i_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r03i_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --input host --steps 20 --warmup 5 --cpu-frames 0 > gpurun_out/r03i_bench_host.json 2> gpurun_out/r03i_bench.err || exit 1
timeout -k 10 300 python bench.py --input host --chunk 64 --steps 20 --warmup 5 --cpu-frames 0 --no-verify > gpurun_out/r03i_bench_host64.json 2>> gpurun_out/r03i_bench.err || exit 1
timeout -k 10 600 bash profiles/ab_cnn.sh > gpurun_out/r03i_ab.log 2>&1; cat gpurun_out/r03i_ab.log
