for r in 1 2 3; do
  for v in tree profiles/variants/hung_old.so; do
    if [ "$v" = tree ]; then unset AXT_LIB_PATH; else export AXT_LIB_PATH=$PWD/$v; fi
    python bench.py --steps 20 --warmup 5 --cpu-frames 0 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['stages'], d.get('verified'))"
  done
done
