#!/bin/bash
# A/B of two builds of the library on one device: config 3 as timed (bench.py --perceptual --vqvae-dtype bf16), alternating processes.
#   tools/ab_step.sh <old.so> [rounds]      (the new build is the in-tree faceoff_amd/libfaceoff_hip.so)
# <old.so> must report the same fo_version() as the tree's include/faceoff_hip.h (FO_ABI_VERSION): faceoff_amd/_lib.py refuses any other
# library at load time, so build the old kernels with the current api.cpp / header, not from a checkout older than the last ABI bump.
OLD=$1; R=${2:-3}
ARGS="--perceptual --vqvae-dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-c3 --no-x6-leg --no-direct-leg --no-c5 --no-h2d-leg --no-kernel-events"
for i in $(seq $R); do
  for v in old new; do
    if [ $v = old ]; then export FACEOFF_HIP_LIB=$OLD; else unset FACEOFF_HIP_LIB; fi
    python bench.py $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'])"
  done
done
