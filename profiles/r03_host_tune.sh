for ramp in "16,32,48,64" "16,48,64" "16,32,64" "32,64" "16,32,48,64,80"; do for chunk in 96 128; do
  AXT_STREAM_RAMP=$ramp timeout -k 10 300 python bench.py --input host --chunk $chunk --steps 20 --warmup 5 --cpu-frames 0 --no-verify 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('ramp $ramp chunk $chunk', b['ms_per_step'], b['stages'])"
done; done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-verify --no-host-variant 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('resident', b['ms_per_step'], b['stages'])"
