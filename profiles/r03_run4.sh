set -x
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "winograd_and_direct or benchmarked_launch" > gpurun_out/r03d_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r03d_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python profiles/cnn_kernels.py > gpurun_out/r03d_kernels.log 2>&1 && cat gpurun_out/r03d_kernels.log && \
timeout -k 10 300 python profiles/wino_stamps.py 40 1 > gpurun_out/r03d_stamps_40p.log 2>&1; cat gpurun_out/r03d_stamps_40p.log
