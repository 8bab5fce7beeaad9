#!/bin/bash
# SQ_INSTS_VALU / SALU / LDS of one kernel for the in-tree library and every scratch/libsfq_*.so
K=${1:-k_rec_tokens}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/pv
cp slimfastq_amd/libslimfastq_amd.so /tmp/lib_orig.so
one() {
  rm -rf $ROOT/gpurun_out/pv/p
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $ROOT/gpurun_out/pv/p -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-adaptive-leg --no-genome-leg --no-format6-leg --no-decode > /dev/null 2> $ROOT/gpurun_out/pv/log < /dev/null )
  python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(float); n=collections.Counter()
for f in glob.glob("$ROOT/gpurun_out/pv/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$K" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
print("   ", {k: "%.3g" % (v/max(1,n[k])) for k,v in sorted(agg.items())})
PY
}
echo "== in-tree"; one
for f in scratch/libsfq_*.so; do [ -e "$f" ] || continue; cp $f slimfastq_amd/libslimfastq_amd.so; echo "== $f"; one; done
cp /tmp/lib_orig.so slimfastq_amd/libslimfastq_amd.so
