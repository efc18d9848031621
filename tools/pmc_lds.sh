#!/bin/bash
# LDS / issue counters over any python script:  bash tools/pmc_lds.sh <tag> <script> [args...]
TAG=$1; shift
OUT=gpurun_out/pmcl_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/a -o pmc \
  --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE \
  -- python3 "$@" > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/b -o pmc \
  --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
  -- python3 "$@" > $OUT/b.log 2>&1
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
    wc=a.get('SQ_WAVE_CYCLES',1)
    print(k, f"dur={d/1e3:.1f}us")
    print("   "+" ".join(f"{c}={v:.4g} ({v/wc:.3f}/wave-cycle)" for c,v in sorted(a.items())))
PY
