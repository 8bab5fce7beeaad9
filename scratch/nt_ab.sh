#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
cp slimfastq_amd/libslimfastq_amd.so /tmp/lib_keep.so
for f in scratch/libsfq_qalone_*.so; do echo "== $f"; cp $f slimfastq_amd/libslimfastq_amd.so; timeout -k 10 200 python3 scratch/qalone.py 10000000 0 2>&1 | grep "EXP" | tr '\n' ' '; echo; done
cp /tmp/lib_keep.so slimfastq_amd/libslimfastq_amd.so
B="--steps 5 --warmup 2 --no-size-sweep --no-cpu-baseline --no-genome-leg --no-format6-leg --no-adaptive-leg --no-host-leg"
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['phase_ms'], d['decode']['ms'], d['decode']['round_trip_identical'], d['decode']['phase_ms'])"; }
for i in 1 2; do
  echo "== in-tree"; cp /tmp/lib_keep.so slimfastq_amd/libslimfastq_amd.so; python3 bench.py $B 2>/dev/null | show
  for v in "$@"; do echo "== $v"; cp scratch/libsfq_$v.so slimfastq_amd/libslimfastq_amd.so; python3 bench.py $B 2>/dev/null | show; done
done
cp /tmp/lib_keep.so slimfastq_amd/libslimfastq_amd.so
