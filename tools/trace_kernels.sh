#!/bin/bash
# per-kernel durations of one microbenchmark:  bash tools/trace_kernels.sh "<filter>"
export TMPDIR=/tmp
OUT=gpurun_out/trace_one
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 tools/bench_kernels.py "$1" > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/trace_one/**/*kernel_stats.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{float(r['AverageNs'])/1e3:10.1f} us avg  x{r['Calls']:>4s}  {r['Name'][:110]}")
PY
grep TFLOP $OUT/log.txt
