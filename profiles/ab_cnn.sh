# A/B of detector builds on ONE box: the in-tree library and every profiles/variants/*.so, three rounds interleaved
for r in 1 2 3; do
  echo "tree: $(python profiles/cnn_kernels.py 2>/dev/null | tail -1)"
  for v in profiles/variants/*.so; do echo "$(basename $v): $(AXT_LIB_PATH=$PWD/$v python profiles/cnn_kernels.py 2>/dev/null | tail -1)"; done
done
