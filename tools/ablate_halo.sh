#!/bin/bash
# Diagnostic: the halo-tile bf16 conv kernel with pieces removed (FO_ABLATE_H bits 8 no DMA, 16 no MFMAs / fragment reads, 32 no stores)
set -u
cd "$(dirname "$0")/.."
CS=faceoff_amd/csrc
if [ "${1:-build}" = "build" ]; then
  for m in 8 16 32; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iinclude -DFO_ABLATE_H=$m -c $CS/conv_bf16.hip -o /tmp/cbh_ab$m.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/cbh_ab$m.o $(ls $CS/*.o | grep -v /conv_bf16.o) -ldl -o tools/_libfaceoff_hh$m.so || exit 1
  done
else
  python tools/bench_bf16.py "${2:-conv1_2 fwd}" 2>&1 | grep TFLOP
  for m in 8 16 32; do
    echo "FO_ABLATE_H=$m"; FACEOFF_HIP_LIB=$PWD/tools/_libfaceoff_hh$m.so python tools/bench_bf16.py "${2:-conv1_2 fwd}" 2>&1 | grep TFLOP
  done
fi
