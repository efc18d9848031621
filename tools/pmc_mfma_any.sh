#!/bin/bash
# rocprofv3 occupancy / MFMA counter pass over any python script:  bash tools/pmc_mfma_any.sh <tag> <script> [args...]
TAG=$1; shift
OUT=gpurun_out/pmcm_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/a -o pmc --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- python3 "$@" > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/b -o pmc --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU -- python3 "$@" > $OUT/b.log 2>&1
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
    print(f"{k:60s} dur={d/1e3:8.1f}us", " ".join(f"{c}={v:.4g}" for c,v in sorted(a.items())))
PY
