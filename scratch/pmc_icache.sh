#!/bin/bash
# instruction-cache counters of the chain kernels.  Run on the GPU box.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p1 -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-decode --no-adaptive-leg "$@" > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD --kernel-trace --output-format csv -d $OUT/p2 -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-decode --no-adaptive-leg "$@" > $OUT/p2.log 2>&1
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: [0,0.0])
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=(r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])
        agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
for (kn,cn),(n,v) in sorted(agg.items()):
    if v>0 and any(x in kn for x in ("_c", "_f", "exc", "count", "hist")): print("%-42s %-28s calls=%d per_call=%.4g"%(kn,cn,n,v/n))
PY
tail -3 $OUT/p1.log $OUT/p2.log
