ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
name=$(echo $grp | cut -d' ' -f1)
rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_mem_$name -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $ROOT/gpurun_out/pmc_mem_$name.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: [0,0.0])
for f in glob.glob("$ROOT/gpurun_out/pmc_mem_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=(r["Kernel_Name"].split("(")[0], r["Counter_Name"])
        agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
for (kn,cn),(n,v) in sorted(agg.items()):
    if v>0 and ("encode" in kn or "model" in kn or "rc_lanes" in kn): print("%-28s %-22s calls=%d per_call=%.4g"%(kn,cn,n,v/n))
PY
