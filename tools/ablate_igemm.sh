#!/bin/bash
# Diagnostic: build libfaceoff variants with pieces of the igemm K-loop removed (FO_ABLATE bits, see conv_igemm.hip)
# and time one Conv3d launch with each.  Results of the ablated builds are wrong by construction; only time matters.
#   bash tools/ablate_igemm.sh build      (here, cross-compiles)      bash tools/ablate_igemm.sh run   (on the GPU box)
set -u
cd "$(dirname "$0")/.."
CS=faceoff_amd/csrc
if [ "${1:-build}" = "build" ]; then
  for m in 1 2 3 4 8 12 15; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iinclude -DFO_ABLATE=$m -c $CS/conv_igemm.hip -o /tmp/igemm_ab$m.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/igemm_ab$m.o $(ls $CS/*.o | grep -v /conv_igemm.o) -o tools/_libfaceoff_ab$m.so || exit 1
  done
else
  python tools/bench_kernels.py "${2:-conv3d_b fwd}" 2>&1 | grep TFLOP
  for m in 1 2 3 4 8 12 15; do
    echo "FO_ABLATE=$m"; FACEOFF_HIP_LIB=$PWD/tools/_libfaceoff_ab$m.so python tools/bench_kernels.py "${2:-conv3d_b fwd}" 2>&1 | grep TFLOP
  done
fi
