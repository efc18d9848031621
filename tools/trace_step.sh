#!/bin/bash
# rocprofv3 per-kernel totals of tools/step_breakdown.py <args>:  bash tools/trace_step.sh --bf16
export TMPDIR=/tmp
OUT=gpurun_out/trace_step
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 tools/step_breakdown.py "$@" > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv,glob,re
rows=[]
for f in glob.glob('gpurun_out/trace_step/**/*kernel_stats.csv',recursive=True):
    rows+=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f"total GPU kernel time {tot/1e6:.2f} ms over 5 steps (2 warm-up + 3 timed)")
for r in rows[:45]:
    n=re.sub(r"\(anonymous namespace\)::","",r['Name']); n=re.sub(r"^void ","",n)
    print(f"{float(r['TotalDurationNs'])/5e6:8.3f} ms/step {float(r['AverageNs'])/1e3:9.1f} us avg  x{int(r['Calls'])//5:>4d}  {n[:120]}")
PY
head -2 $OUT/log.txt
