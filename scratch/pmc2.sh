#!/bin/bash
# PMC counters of the frozen-table kernels, two passes (instruction mix / waits; LDS / flat / memory).  Run on the GPU box.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/p1 -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-decode --no-adaptive-leg "$@" > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/p2 -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-decode --no-adaptive-leg "$@" > $OUT/p2.log 2>&1
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: [0,0.0])
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=(r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])
        agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
for (kn,cn),(n,v) in sorted(agg.items()):
    if v>0 and any(x in kn for x in ("_c", "_f", "exc", "count", "hist")): print("%-42s %-24s calls=%d per_call=%.4g"%(kn,cn,n,v/n))
PY
