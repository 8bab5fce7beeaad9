#!/bin/bash
# Round 5: the judged artifacts for the default bench line.  Run ON THE GPU BOX from the repo root:
#   bash profiles/r05_refresh.sh r05z     -> gpurun_out/<tag>/...   (then, in the dev container: python profiles/r05_collect.py r05z)
# 1. the default line as the driver runs it; 2. the same command under rocprofv3 --kernel-trace --stats;
# 3. PMC passes, one counter group per pass (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE cannot share a pass), --kernel-trace only.
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python3 bench.py --steps 20 --warmup 5 < /dev/null 2> $OUT/bench.err | tail -1 > $OUT/default_bench.json
cut -c1-300 $OUT/default_bench.json
LEAN="--steps 3 --warmup 1 --no-cpu-baseline --no-genome-leg --no-format6-leg --no-adaptive-leg --no-size-sweep --no-host-leg"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $ROOT/bench.py $LEAN > $OUT/prof.json 2> $OUT/prof.log < /dev/null )
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_trace.csv
head -6 $OUT/kernel_stats.csv | cut -c1-120
PMC="--steps 1 --warmup 1 --no-cpu-baseline --no-genome-leg --no-format6-leg --no-adaptive-leg --no-size-sweep --no-host-leg"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  name=$(echo $grp | cut -d' ' -f1)
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $ROOT/bench.py $PMC > $OUT/pmc_$name.json 2> $OUT/pmc_$name.log < /dev/null )
  echo "pmc pass $name done" 
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
with open("$OUT/pmc_summary.txt", "w") as o:
    for (kn, cn), (n, v) in sorted(agg.items()):
        if v > 0: o.write("%-34s %-22s calls=%d sum=%.4g per_call=%.4g\n" % (kn, cn, n, v, v / n))
print(open("$OUT/pmc_summary.txt").read().count("\n"), "counter lines")
PY
# 4. (round 5) the genome-sampled call: bench line, kernel trace, and the other workloads' lines
G="--kind 3 --steps 3 --warmup 1 --no-cpu-baseline --no-genome-leg --no-format6-leg --no-adaptive-leg --no-size-sweep --no-host-leg"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gprof -- python3 $ROOT/bench.py $G > $OUT/genome10M_bench.json 2> $OUT/gprof.log < /dev/null )
f=$(find $OUT/gprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/genome10M_kernel_stats.csv
f=$(find $OUT/gprof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/genome10M_kernel_trace.csv
rm -rf $OUT/gprof $OUT/prof
X="--steps 5 --warmup 2 --no-genome-leg --no-format6-leg --no-adaptive-leg --no-cpu-baseline --no-size-sweep --no-host-leg"
for w in "qlt --workload qlt" "binned --kind 2" "l4 --level 4" "l1 --level 1" "long --kind 1" "genome2M --kind 3 --reads 2000000" "genome10M_plain --kind 3" "reads2M --reads 2000000" "reads5M --reads 5000000"; do
  set -- $w; name=$1; shift
  python3 bench.py $X "$@" 2> $OUT/bench_$name.err < /dev/null | tail -1 > $OUT/bench_$name.json
  python3 -c "
import json; d=json.load(open('$OUT/bench_$name.json')); print('$name', d['value'], d['ms_per_step'], d.get('ratio'), 'dec', (d.get('decode') or {}).get('value'), (d.get('decode') or {}).get('round_trip_identical'))"
done
rm -rf $OUT/pmc_*/  2>/dev/null
ls $OUT | head -40
