set -x
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "hungarian or host_resident or two_rank or end_to_end_against or full_size_properties" > gpurun_out/r03g_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r03g_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03g_bench.json 2> gpurun_out/r03g_bench.err || exit 1
timeout -k 10 300 python bench.py --input host --steps 20 --warmup 5 --cpu-frames 0 > gpurun_out/r03g_bench_host.json 2>> gpurun_out/r03g_bench.err || exit 1
for w in assoc-c3 assoc-c4; do for a in mcf hungarian; do timeout -k 10 300 python bench.py --workload $w --assoc $a --steps 5 --warmup 2 > gpurun_out/r03g_${w}_${a}.json 2>> gpurun_out/r03g_bench.err || exit 1; done; done
timeout -k 10 300 python profiles/full_config.py c4 > gpurun_out/r03g_c4_full.log 2>&1
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03g_stats -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-frames 0 --no-verify --no-host-variant --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/r03g_stats.log 2>&1
