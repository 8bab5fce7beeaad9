#!/bin/bash
# PMC counters of the decode kernels.  Run on the GPU box.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/p1 -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-adaptive-leg "$@" > $OUT/p1.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_WAVES --kernel-trace --output-format csv -d $OUT/p2 -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-adaptive-leg "$@" > $OUT/p2.log 2>&1
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: [0,0.0]); dur=collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=(r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])
        agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
for f in glob.glob("$OUT/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0][:40]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
for (kn,cn),(n,v) in sorted(agg.items()):
    if v>0 and "decode" in kn: print("%-42s %-24s calls=%d per_call=%.4g"%(kn,cn,n,v/n))
for k,v in dur.items():
    if "decode" in k or "assemble" in k: print("%-42s alone ms %s"%(k, ["%.2f"%x for x in v]))
PY
