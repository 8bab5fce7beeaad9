#!/bin/bash
# Collect PMC counters for the bench workload, one counter group per pass (MI355X_MICROARCH.md "rocprofv3 PMC
# slots": FETCH_SIZE and WRITE_SIZE cannot share a pass).  Usage: collect_pmc.sh <tag> [bench args...]
# Run on the GPU box from the repo root; writes gpurun_out/pmc_<tag>_<group>/
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_${TAG}_${name} -- python3 $ROOT/bench.py "$@" > $ROOT/gpurun_out/pmc_${TAG}_${name}.log 2>&1
done
python3 - <<PY
import csv, glob, collections, os
root="$ROOT/gpurun_out"
for d in sorted(glob.glob(root+"/pmc_${TAG}_*/")):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(lambda: [0,0.0])
        for r in csv.DictReader(open(f)):
            k=(r["Kernel_Name"].split("(")[0], r["Counter_Name"])
            agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
        for (kn,cn),(n,v) in sorted(agg.items()):
            if v>0 and ("encode" in kn or "newlines" in kn or "compact" in kn): print("%-28s %-22s calls=%d sum=%.4g per_call=%.4g"%(kn,cn,n,v,v/n))
PY
