#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
cp slimfastq_amd/libslimfastq_amd.so /tmp/lib_orig.so
for f in /tmp/lib_orig.so scratch/libsfq_cap*.so; do
  cp $f slimfastq_amd/libslimfastq_amd.so 2>/dev/null
  for r in 2000000 10000000; do
    echo "== $f reads $r: $(python3 bench.py --kind 3 --reads $r --steps 3 --warmup 1 --no-size-sweep --no-cpu-baseline --no-genome-leg --no-format6-leg --no-adaptive-leg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'ratio', d['ratio'], 'dec', d['decode']['ms'], d['decode']['round_trip_identical'])")"
  done
done
cp /tmp/lib_orig.so slimfastq_amd/libslimfastq_amd.so
