#!/bin/bash
# format 6 (one block, one wave per stream): SQ counters of the encode and decode kernels -> gpurun_out/<tag>/pmc.txt
#   bash scratch/pmc_f6.sh <tag> [reads]
TAG=$1; N=${2:-60000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/scratch/f6_probe.py $N > $OUT/trace.txt 2> $OUT/trace.log < /dev/null )
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  name=$(echo $grp | cut -d' ' -f1)
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $ROOT/scratch/f6_probe.py $N > $OUT/pmc_$name.txt 2> $OUT/pmc_$name.log < /dev/null )
  echo "pass $name: $?"
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
        if v > 0 and kn.startswith("k_"): o.write("%-34s %-28s calls=%d sum=%.5g per_call=%.5g\n" % (kn, cn, n, v, v / n))
PY
grep -c . $OUT/pmc.txt; cat $OUT/trace.txt
