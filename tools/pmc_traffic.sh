#!/bin/bash
# HBM traffic per launch of any python script's kernels:  bash tools/pmc_traffic.sh <tag> <script> [args...]
# (HBM read = 2 x FETCH_SIZE KB on gfx950 for 16-B/lane streams, write = WRITE_SIZE KB: MI355X_MICROARCH.md, HBM section)
TAG=$1; shift
OUT=gpurun_out/pmct_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/f -o pmc --pmc FETCH_SIZE -- python3 "$@" > $OUT/f.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/w -o pmc --pmc WRITE_SIZE -- python3 "$@" > $OUT/w.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/l -o pmc --pmc TCC_HIT_sum TCC_MISS_sum -- python3 "$@" > $OUT/l.log 2>&1
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
vals=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(dict)
for f in glob.glob(out+'/*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:60]
        vals[k][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[k][r['Dispatch_Id']]=float(r['End_Timestamp'])-float(r['Start_Timestamp'])
for k in sorted(vals, key=lambda k:-sum(dur[k].values())):
    d=sum(dur[k].values())/len(dur[k])
    if d<50e3: continue
    a={c:sum(v)/len(v) for c,v in vals[k].items()}
    rd=2*a.get('FETCH_SIZE',0)*1024; wr=a.get('WRITE_SIZE',0)*1024
    h,m=a.get('TCC_HIT_sum',0),a.get('TCC_MISS_sum',0)
    print(f"{k} dur={d/1e3:.1f}us HBM read {rd/1e6:.0f} MB write {wr/1e6:.0f} MB -> {(rd+wr)/d:.2f} GB/s... = {(rd+wr)/d/1e3:.2f} TB/s; L2 hit {h/(h+m+1e-9):.3f} (hits {h:.3g} misses {m:.3g})")
PY
