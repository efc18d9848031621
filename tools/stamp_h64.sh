#!/bin/bash
# Diagnostic: the 64-input-channel halo-tile kernel with in-kernel clock stamps (FO_STAMP_H64).   bash tools/stamp_h64.sh   (here), then python tools/stamp_h64.py on the GPU box
set -u
cd "$(dirname "$0")/.."
CS=faceoff_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iinclude -DFO_STAMP_H64 ${EXTRA:-} -c $CS/conv_bf16.hip -o /tmp/cb_stamp_h.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/cb_stamp_h.o $(ls $CS/*.o | grep -v /conv_bf16.o) -ldl -o tools/_libfaceoff_stamp_h64${TAG:-}.so || exit 1
