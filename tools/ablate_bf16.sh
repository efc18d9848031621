#!/bin/bash
# Diagnostic: bf16 conv kernel with pieces of its K-loop removed (FO_ABLATE_H bits, see conv_bf16.hip); timing only.
#   bash tools/ablate_bf16.sh build   (here)        bash tools/ablate_bf16.sh run [filter]   (GPU box)
set -u
cd "$(dirname "$0")/.."
CS=faceoff_amd/csrc
if [ "${1:-build}" = "build" ]; then
  for m in 1 3 4 7; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iinclude -DFO_ABLATE_H=$m -c $CS/conv_bf16.hip -o /tmp/cbf_ab$m.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/cbf_ab$m.o $(ls $CS/*.o | grep -v /conv_bf16.o) -o tools/_libfaceoff_hab$m.so || exit 1
  done
else
  python tools/bench_bf16.py "${2:-fwd}" 2>&1 | grep TFLOP
  for m in 1 3 4 7; do
    echo "FO_ABLATE_H=$m"; FACEOFF_HIP_LIB=$PWD/tools/_libfaceoff_hab$m.so python tools/bench_bf16.py "${2:-fwd}" 2>&1 | grep TFLOP
  done
fi
