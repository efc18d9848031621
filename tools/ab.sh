#!/bin/bash
# A/B on ONE box:  bash tools/ab.sh <tag> "<env A>" "<env B>" [bench args]   (boxes differ by a few per cent: never compare across calls)
TAG=$1; A=$2; B=$3; shift 3
mkdir -p gpurun_out/$TAG
for rep in 1 2; do
  for v in A B; do
    if [ $v = A ]; then E="$A"; else E="$B"; fi
    env $E python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-c3 --no-c5 --no-direct-leg --no-x6-leg "$@" > gpurun_out/$TAG/$v$rep.json 2> gpurun_out/$TAG/$v$rep.err
    python - gpurun_out/$TAG/$v$rep.json "$v$rep [$E]" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d.get('kernels',{})
top=sorted(k.items(), key=lambda kv:-kv[1]['ms_per_step'])[:4]
print(sys.argv[2], 'step', d['ms_per_step'], 'serial', d.get('ms_per_step_serial'), ' '.join(f"{n}={v['ms_per_step']}" for n,v in top))
PY
  done
done
