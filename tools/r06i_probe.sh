O=gpurun_out/r06i; mkdir -p $O
for cfg in "" "--bf16 --lpips" "--gan"; do
  n=$(echo "c2$cfg" | tr -d ' -'); python tools/step_ledger.py $cfg --top 12 > $O/ledger_$n.txt 2>&1; head -16 $O/ledger_$n.txt | cut -c1-200
  python - $O/ledger_$n.txt <<'PY'
import re,sys
over=0;n=0
for l in open(sys.argv[1]):
    m=re.match(r"\s*(\d+)\s+([\d.]+)\s+([\d.]+)\s+(-?[\d.]+)",l)
    if m and float(m[3])>float(m[2])*1.02: over+=float(m[3])-float(m[2]); n+=1
print("floor above measured in the top list:", n, "launches", round(over,3), "ms")
PY
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CONFIG=c3 bash profiles/collect.sh r06_c3 all > $O/collect_c3.log 2>&1; tail -5 $O/collect_c3.log
