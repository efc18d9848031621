#!/bin/bash
# as tools/ab_env.sh on the headline workload (config 2, fp32):  tools/ab_env_c2.sh rounds "VAR=1" ...
R=$1; shift
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-c3 --no-x6-leg --no-direct-leg --no-c5 --no-h2d-leg --no-kernel-events"
for i in $(seq $R); do
  for v in base "$@"; do
    if [ "$v" = base ]; then python bench.py $ARGS 2>/dev/null; else env $v python bench.py $ARGS 2>/dev/null; fi | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'])"
  done
done
