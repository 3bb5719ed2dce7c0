# bench.py headline with and without the fused front kernel, interleaved on one box
for r in 1 2; do
  for f in 0 1; do
    AXT_FUSE_S2=$f timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-host-variant --cpu-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse=$f', d['value'], d['ms_per_step'], d.get('stages'))" || exit 1
  done
done
