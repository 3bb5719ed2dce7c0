set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "full_size_4 or configs_4_and_5 or substitute or bare_multi or timepoint or three_allowed or changes_over_time" > gpurun_out/r03a_newtests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03a_newtests.log
tail -5 gpurun_out/r03a_newtests.log
( nproc; AXT_MCF_DEBUG=1 python profiles/mcf_timing.py static 2>&1 | grep -v "leaf\|separator" | tail -3
  AXT_MCF_DEBUG=1 python profiles/mcf_timing.py moving c3 2>&1 | grep -v "leaf\|separator" | tail -3
  echo one-phase; REPEAT=3 AXT_MCF_DEBUG=1 python profiles/mcf_timing.py moving c4 2>&1 | grep -v "leaf\|separator" | tail -3
  echo two-phase-par; REPEAT=3 AXT_MCF_TWO_PHASE=1 AXT_MCF_DEBUG=1 python profiles/mcf_timing.py moving c4 2>&1 | grep -v "leaf\|separator" | tail -3
  echo two-phase-serial; REPEAT=2 AXT_MCF_TWO_PHASE=1 AXT_MCF_ENDS_THREADS=1 AXT_MCF_DEBUG=1 python profiles/mcf_timing.py moving c4 2>&1 | grep -v "leaf\|separator" | tail -3
  echo static16; REPEAT=3 AXT_MCF_DEBUG=1 python profiles/mcf_timing.py static 16 2>&1 | grep -v "leaf\|separator" | tail -3
  echo static16-two-par; REPEAT=3 AXT_MCF_TWO_PHASE=1 AXT_MCF_DEBUG=1 python profiles/mcf_timing.py static 16 2>&1 | grep -v "leaf\|separator" | tail -3
) > gpurun_out/r03a_mcf_host.log 2>&1
python bench.py > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err; echo bench rc=$?
python bench.py --assoc mcf --no-verify > gpurun_out/r03a_bench_mcf.json 2>> gpurun_out/r03a_bench.err; echo bench rc=$?
