#!/bin/bash
# SQ instruction / wait counters of one bench configuration: bash scratch/pmc_sq2.sh <tag> [bench args]   -> gpurun_out/<tag>/pmc.txt
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
LEAN="--steps 1 --warmup 1 --no-cpu-baseline --no-genome-leg --no-format6-leg --no-adaptive-leg --no-size-sweep"
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_BUSY_CYCLES"; do
  name=$(echo $grp | cut -d' ' -f1)
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $ROOT/bench.py $LEAN "$@" > $OUT/pmc_$name.json 2> $OUT/pmc_$name.log < /dev/null )
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
with open("$OUT/pmc.txt", "w") as o:
    for (kn, cn), (n, v) in sorted(agg.items()):
        if v > 0 and kn.startswith("k_"): o.write("%-34s %-22s calls=%d sum=%.4g per_call=%.4g\n" % (kn, cn, n, v, v / n))
PY
grep -c . $OUT/pmc.txt
