#!/bin/bash
# scratch/tok_pmc.sh: instruction counts of k_rec_tokens under every scratch/libsfq_*.so (and the in-tree library), one rocprofv3 --pmc pass each
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/tokpmc; mkdir -p $OUT
cp slimfastq_amd/libslimfastq_amd.so /tmp/lib_orig.so
B="--steps 1 --warmup 1 --reads 2000000 --no-decode --no-cpu-baseline --no-genome-leg --no-format6-leg --no-adaptive-leg --no-size-sweep --no-host-leg"
one() {
  name=$1
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p_$name -- python3 $ROOT/bench.py $B > $OUT/$name.json 2> $OUT/$name.log < /dev/null )
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("$OUT/p_$name/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "k_rec_tok" in kn or "k_rec_code" in kn:
            agg[(kn, r["Counter_Name"])][0] += 1; agg[(kn, r["Counter_Name"])][1] += float(r["Counter_Value"])
print("== $name")
for (kn, cn), (n, v) in sorted(agg.items()): print("  %-22s %-20s per_call=%.4g" % (kn, cn, v / n))
PY
  rm -rf $OUT/p_$name
}
one intree
for f in scratch/libsfq_*.so; do cp $f slimfastq_amd/libslimfastq_amd.so; one $(basename $f .so); done
cp /tmp/lib_orig.so slimfastq_amd/libslimfastq_amd.so
