#!/bin/bash
# rocprofv3 counter pass over one microbenchmark:  bash tools/pmc_kernel.sh "<filter>" <tag>
FLT="$1"; TAG=${2:-k}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/sq -o pmc \
  --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE \
  -- python3 tools/bench_kernels.py "$FLT" > $OUT/sq.log 2>&1
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
vals=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(dict)
for f in glob.glob(out+'/sq/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:90]
        vals[k][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[k][r['Dispatch_Id']]=float(r['End_Timestamp'])-float(r['Start_Timestamp'])
for k in vals:
    d=sum(dur[k].values())/len(dur[k])
    a={c:sum(v)/len(v) for c,v in vals[k].items()}
    if 'GRBM_GUI_ACTIVE' not in a: continue
    clk=a['GRBM_GUI_ACTIVE']/8/d
    share=a.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(a['GRBM_GUI_ACTIVE']/8*4*256)
    wc=a.get('SQ_WAVE_CYCLES',1)
    print(f"{k[:70]:70s} n={len(dur[k])} dur={d/1e3:9.1f}us clk={clk:.3f}GHz mfma_busy={share:.3f} wait_any={a.get('SQ_WAIT_ANY',0)/wc:.3f} wait_inst={a.get('SQ_WAIT_INST_ANY',0)/wc:.3f} active={a.get('SQ_ACTIVE_INST_ANY',0)/wc:.3f} valu_insts={a.get('SQ_INSTS_VALU',0):.3g}")
PY
