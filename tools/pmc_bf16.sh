#!/bin/bash
# rocprofv3 counter pass over the bf16 conv microbenchmark:  bash tools/pmc_bf16.sh "<filter>" <tag>
FLT="$1"; TAG=${2:-h}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/sq -o pmc \
  --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE \
  -- python3 tools/bench_bf16.py "$FLT" > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/sq2 -o pmc \
  --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE \
  -- python3 tools/bench_bf16.py "$FLT" > $OUT/sq2.log 2>&1
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
for sub in ('sq','sq2'):
    vals=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(dict)
    for f in glob.glob(out+'/'+sub+'/**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:60]
            if 'conv_bf16' not in k: continue
            vals[k][r['Counter_Name']].append(float(r['Counter_Value']))
            dur[k][r['Dispatch_Id']]=float(r['End_Timestamp'])-float(r['Start_Timestamp'])
    for k in vals:
        d=sum(dur[k].values())/len(dur[k])
        a={c:sum(v)/len(v) for c,v in vals[k].items()}
        cyc=a.get('GRBM_GUI_ACTIVE',0)/8
        print(k, f"dur={d/1e3:.1f}us clk={cyc/d:.3f}GHz", {c:(f"{v:.4g}", f"{v/(cyc*4*256):.3f}/simdcyc" ) for c,v in a.items() if c!='GRBM_GUI_ACTIVE'})
PY
