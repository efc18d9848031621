#!/bin/bash
# Diagnostic: the extended-tile bf16 conv kernel with in-kernel clock stamps (FO_STAMP_PPH) -- segment lengths of its K loop.
#   bash tools/stamp_pph.sh build   (here)        python tools/stamp_pph.py "<layer> <kind>"   (GPU box)
set -u
cd "$(dirname "$0")/.."
CS=faceoff_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iinclude -DFO_STAMP_PPH ${EXTRA:-} -c $CS/conv_bf16.hip -o /tmp/cb_stamp.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/cb_stamp.o $(ls $CS/*.o | grep -v /conv_bf16.o) -ldl -o tools/_libfaceoff_stamp_pph${TAG:-}.so || exit 1
