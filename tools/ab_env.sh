#!/bin/bash
# A/B of environment switches on config 3 as timed, alternating processes on one device:  tools/ab_env.sh rounds "VAR=1" "VAR2=0" ...   (first variant: no switch)
R=$1; shift
ARGS="--perceptual --vqvae-dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-c3 --no-x6-leg --no-direct-leg --no-c5 --no-h2d-leg --no-kernel-events"
for i in $(seq $R); do
  for v in base "$@"; do
    if [ "$v" = base ]; then python bench.py $ARGS 2>/dev/null; else env $v python bench.py $ARGS 2>/dev/null; fi | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'])"
  done
done
