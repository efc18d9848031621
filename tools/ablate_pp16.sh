#!/bin/bash
# Diagnostic: conv_bf16_pp16_kernel with pieces of its K loop removed (FO_ABLATE_PP bits: 1 no LDS-DMA, 2 one MFMA per phase instead of TM x TN,
# 4 every fragment read from block 0 -- i.e. one ds_read address per operand, the compiler keeps one read each); timing only, results are wrong.
#   bash tools/ablate_pp16.sh build   (here)        bash tools/ablate_pp16.sh run "<layer filter>"   (GPU box)
set -u
cd "$(dirname "$0")/.."
CS=faceoff_amd/csrc
if [ "${1:-build}" = "build" ]; then
  for m in 64; do
    D="-DFO_ABLATE_PP=$m"; [ $m = L0 ] && D="-DFO_ABLATE_PP_LINES=1"; [ $m = L2 ] && D="-DFO_ABLATE_PP_LINES=1 -DFO_ABLATE_PP=2"      # L*: whole-line pieces
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iinclude $D -c $CS/conv_bf16.hip -o /tmp/cpp_ab$m.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/cpp_ab$m.o $(ls $CS/*.o | grep -v /conv_bf16.o) -ldl -o tools/_libfaceoff_pp$m.so || exit 1
  done
else
  shift
  for flt in "$@"; do
    echo "== $flt: full kernel"; ROUNDS=5 python tools/ab_bf16.py "$flt" base
    for m in 64; do
      echo "== $flt: FO_ABLATE_PP=$m"; ROUNDS=5 FACEOFF_HIP_LIB=$PWD/tools/_libfaceoff_pp$m.so python tools/ab_bf16.py "$flt" base
    done
  done
fi
