# timing-only builds of conv_s2_fused (AXT_FUSED_ABLATE bit sets) -> profiles/variants/fused_a<bits>.so
set -e
cd "$(dirname "$0")/../../axtrack_amd/csrc"
mkdir -p ../../profiles/variants
for a in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DAXT_FUSED_ABLATE=$a -c cnn_front.hip -o /tmp/cnn_front_a$a.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../profiles/variants/fused_a$a.so cnn.o /tmp/cnn_front_a$a.o detect.o assoc.o hungarian.o path_bfs.o preproc.o ided.o appearance.o metrics.o mcf.o api.o
done
