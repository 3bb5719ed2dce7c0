set -x
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace -d $GRAFT_REPO_ROOT/gpurun_out/r03j_trace -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --input host --cpu-frames 0 --no-verify --steps 3 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/r03j_trace.log 2>&1
