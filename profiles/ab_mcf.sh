# A/B of flow-solver builds on ONE box's host cores: the in-tree library and every profiles/variants/*.so, interleaved.
#   bash profiles/ab_mcf.sh      (config-3 network x5, one GPU's share of config 4 x3, config 4 whole x2 per build and round)
run() {  # $1 = label, rest = command; prints min of the "solve" lines
  l=$1; shift
  "$@" 2>/dev/null | grep "^solve" | awk '{print $2}' | sort -n | awk -v l="$l" '{a[NR]=$1} END{printf "%s min %.1f median %.1f ms (n=%d)\n", l, a[1], a[int((NR+1)/2)], NR}'
}
for r in 1 2 3; do
  for v in tree profiles/variants/*.so; do
    if [ "$v" = tree ]; then unset AXT_LIB_PATH; else export AXT_LIB_PATH=$PWD/$v; fi
    echo "$(basename $v) c3       $(REPEAT=10 python profiles/mcf_timing.py static 2>/dev/null | tail -1)"
    REPEAT=4  run "$(basename $v) c4-share" python profiles/experiments/mcf_c4_offline.py 132
    REPEAT=2  run "$(basename $v) c4      " python profiles/experiments/mcf_c4_offline.py
  done
done
