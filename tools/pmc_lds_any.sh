#!/bin/bash
# rocprofv3 LDS / memory counter passes over any python script:  bash tools/pmc_lds_any.sh <tag> <script> [args...]
TAG=$1; shift
OUT=gpurun_out/pmcl_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/a -o pmc --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -- python3 "$@" > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/b -o pmc --pmc FETCH_SIZE -- python3 "$@" > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/c -o pmc --pmc TCC_HIT_sum TCC_MISS_sum -- python3 "$@" > $OUT/c.log 2>&1
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
vals=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(dict)
for f in glob.glob(out+'/*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:80]
        vals[k][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[k][r['Dispatch_Id']]=float(r['End_Timestamp'])-float(r['Start_Timestamp'])
for k in sorted(vals, key=lambda k:-sum(dur[k].values())):
    d=sum(dur[k].values())/len(dur[k])
    if d<50e3: continue
    a={c:sum(v)/len(v) for c,v in vals[k].items()}
    wc=a.get('SQ_WAVE_CYCLES',1)
    print(f"{k[:80]:80s} dur={d/1e3:8.1f}us lds_conflict/idx_active={a.get('SQ_LDS_BANK_CONFLICT',0)/max(a.get('SQ_LDS_IDX_ACTIVE',1),1):.3f} lds_active/gui={a.get('SQ_LDS_IDX_ACTIVE',0)/max(a.get('GRBM_GUI_ACTIVE',1)/8*256,1):.3f} wait_lds={a.get('SQ_WAIT_INST_LDS',0)/wc:.3f} wait_inst={a.get('SQ_WAIT_INST_ANY',0)/wc:.3f} fetchMB={2*a.get('FETCH_SIZE',0)/1024:.0f} l2hit={a.get('TCC_HIT_sum',0)/max(a.get('TCC_HIT_sum',0)+a.get('TCC_MISS_sum',0),1):.3f}")
PY
