set -x
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "host_resident" > gpurun_out/r03h_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r03h_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --input host --steps 20 --warmup 5 --cpu-frames 0 > gpurun_out/r03h_bench_host.json 2> gpurun_out/r03h_bench.err || exit 1
timeout -k 10 300 python bench.py --input host --chunk 32 --steps 20 --warmup 5 --cpu-frames 0 --no-verify > gpurun_out/r03h_bench_host32.json 2>> gpurun_out/r03h_bench.err || exit 1
timeout -k 10 300 python bench.py --input host --chunk 128 --steps 20 --warmup 5 --cpu-frames 0 --no-verify > gpurun_out/r03h_bench_host128.json 2>> gpurun_out/r03h_bench.err || exit 1
AXT_MCF_DEBUG=1 timeout -k 10 300 python profiles/full_config.py c4 2>&1 | grep -v "leaf\|separator" > gpurun_out/r03h_c4_full_onephase.log
AXT_MCF_TWO_PHASE=1 AXT_MCF_DEBUG=1 timeout -k 10 300 python profiles/full_config.py c4 2>&1 | grep -v "leaf\|separator" > gpurun_out/r03h_c4_full_twophase.log
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace -d $GRAFT_REPO_ROOT/gpurun_out/r03h_trace -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --input host --cpu-frames 0 --no-verify --steps 3 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/r03h_trace.log 2>&1
