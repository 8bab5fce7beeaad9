ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_dec -- python3 $ROOT/scratch/decode_bench.py 4000000 1024 > $ROOT/gpurun_out/pmc_dec.log 2>&1
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: [0,0.0])
for f in glob.glob("$ROOT/gpurun_out/pmc_dec/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=(r["Kernel_Name"].split("(")[0], r["Counter_Name"])
        agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
for (kn,cn),(n,v) in sorted(agg.items()):
    if v>0 and ("decode" in kn): print("%-28s %-22s calls=%d per_call=%.4g"%(kn,cn,n,v/n))
PY
