#!/bin/bash
# scratch/all_pmc.sh [bench flags]: VALU + SALU instructions per kernel of one encode (+decode) call -> stdout, largest first
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/allpmc; mkdir -p $OUT
B="--steps 1 --warmup 1 --no-cpu-baseline --no-genome-leg --no-format6-leg --no-adaptive-leg --no-size-sweep --no-host-leg"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p -- python3 $ROOT/bench.py $B "$@" > $OUT/b.json 2> $OUT/b.log < /dev/null )
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = agg[kn][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
rows = []
for kn, c in agg.items():
    n = c["SQ_INSTS_VALU"][0]
    rows.append((c["SQ_INSTS_VALU"][1] + c["SQ_INSTS_SALU"][1], kn, n, c["SQ_INSTS_VALU"][1], c["SQ_INSTS_SALU"][1], c["SQ_WAVE_CYCLES"][1]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("total VALU+SALU over the run: %.3e" % tot)
for t, kn, n, v, s_, w in rows[:28]: print("%-40s launches %3d  VALU %.3e  SALU %.3e  sum %.3e (%.1f%%)  wave-cycles %.3e" % (kn[:40], n, v, s_, t, 100 * t / tot, w))
PY
rm -rf $OUT/p
