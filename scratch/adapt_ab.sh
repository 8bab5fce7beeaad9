#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
cp slimfastq_amd/libslimfastq_amd.so /tmp/lib_orig.so
for f in /tmp/lib_orig.so scratch/libsfq_*.so; do
  cp $f slimfastq_amd/libslimfastq_amd.so 2>/dev/null
  echo "== $f: $(python3 bench.py --tables 0 --steps 3 --warmup 1 --no-size-sweep --no-cpu-baseline --no-genome-leg --no-format6-leg 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'ratio', d['ratio'], d['phase_ms'], 'dec', d['decode']['ms'], d['decode']['round_trip_identical'])" 2>&1 | tail -1)"
done
cp /tmp/lib_orig.so slimfastq_amd/libslimfastq_amd.so
