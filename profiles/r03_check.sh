set -x
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "other_workloads or host_resident" > gpurun_out/r03p_tests.log 2>&1; tail -4 gpurun_out/r03p_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r03p_smoke.log 2>&1; tail -2 gpurun_out/r03p_smoke.log
timeout -k 10 500 python profiles/fuzz_parity.py 16 31 > gpurun_out/r03p_fuzz.log 2>&1; tail -3 gpurun_out/r03p_fuzz.log
