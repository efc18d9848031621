#!/bin/bash
# rocprofv3 SQ counter pass over any python script:  bash tools/pmc_any.sh <tag> <script> [args...]
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/sq -o pmc \
  --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE \
  -- python3 "$@" > $OUT/sq.log 2>&1
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
vals=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(dict); grid={}
for f in glob.glob(out+'/sq/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:70]+' g'+r.get('Grid_Size','')
        vals[k][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[k][r['Dispatch_Id']]=float(r['End_Timestamp'])-float(r['Start_Timestamp'])
for k in sorted(vals, key=lambda k:-sum(dur[k].values())):
    d=sum(dur[k].values())/len(dur[k])
    a={c:sum(v)/len(v) for c,v in vals[k].items()}
    if 'GRBM_GUI_ACTIVE' not in a or d<50e3: continue
    clk=a['GRBM_GUI_ACTIVE']/8/d
    share=a.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(a['GRBM_GUI_ACTIVE']/8*4*256)
    wc=a.get('SQ_WAVE_CYCLES',1)
    print(f"{k[:95]:95s} n={len(dur[k])} dur={d/1e3:8.1f}us clk={clk:.3f} mfma_busy={share:.3f} wait_any={a.get('SQ_WAIT_ANY',0)/wc:.3f} valu={a.get('SQ_INSTS_VALU',0):.3g}")
PY
