#!/bin/bash
# Diagnostic: bf16 filter-gradient kernel with pieces of its K-loop removed (FO_ABLATE_W bits, see wgrad_bf16.hip); timing only.
#   bash tools/ablate_wgrad.sh build   (here)        bash tools/ablate_wgrad.sh run [filter]   (GPU box)
set -u
cd "$(dirname "$0")/.."
CS=faceoff_amd/csrc
if [ "${1:-build}" = "build" ]; then
  for m in 1 2 3 4 5 6 7; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iinclude -DFO_ABLATE_W=$m -c $CS/wgrad_bf16.hip -o /tmp/wg_ab$m.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/wg_ab$m.o $(ls $CS/*.o | grep -v /wgrad_bf16.o) -o tools/_libfaceoff_wab$m.so || exit 1
  done
else
  python tools/bench_wgrad_bf16.py "${2:-conv3d64}" 2>&1
  for m in 1 2 3 4 5 6 7; do
    echo "FO_ABLATE_W=$m"; FACEOFF_HIP_LIB=$PWD/tools/_libfaceoff_wab$m.so python tools/bench_wgrad_bf16.py "${2:-conv3d64}" 2>&1
  done
fi
