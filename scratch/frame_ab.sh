#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
cp slimfastq_amd/libslimfastq_amd.so /tmp/lib_orig.so
for f in /tmp/lib_orig.so scratch/libsfq_fr_*.so; do
  cp $f slimfastq_amd/libslimfastq_amd.so 2>/dev/null
  rm -rf gpurun_out/trf
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/trf -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-size-sweep --no-adaptive-leg --no-cpu-baseline --no-format6-leg --no-genome-leg --no-decode > /dev/null 2>&1 )
  echo "== $f: $(python3 -c "
import csv,glob
f=glob.glob('gpurun_out/trf/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'k_frame' in r['Name']: print(r['Calls'], float(r['AverageNs'])/1e6, float(r['MinNs'])/1e6)
")"
done
cp /tmp/lib_orig.so slimfastq_amd/libslimfastq_amd.so
