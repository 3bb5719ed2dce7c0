for v in "--assoc hungarian" "--assoc mcf" "--assoc mcf --replicated-solve"; do
  timeout -k 10 300 python bench.py --gpus 2 --backend gloo --single-device $v --frames 132 --steps 3 --warmup 1 --no-verify 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('$v', b['ms_per_step'], b['stages'], b.get('collectives_ms'))"
done
