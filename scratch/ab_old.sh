#!/bin/bash
# in-tree library against scratch/libsfq_old.so, default encode, alternating
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
cp slimfastq_amd/libslimfastq_amd.so /tmp/lib_keep.so
B="--steps 10 --warmup 3 --no-size-sweep --no-cpu-baseline --no-genome-leg --no-format6-leg --no-adaptive-leg --no-host-leg --no-decode"
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['phase_ms'], d['roofline']['coder_ms'])"; }
for i in 1 2 3; do
  echo "== new"; cp /tmp/lib_keep.so slimfastq_amd/libslimfastq_amd.so; python3 bench.py $B 2>/dev/null | show
  echo "== old"; cp scratch/libsfq_old.so slimfastq_amd/libslimfastq_amd.so; python3 bench.py $B 2>/dev/null | show
done
echo "== new qlt alone"; cp /tmp/lib_keep.so slimfastq_amd/libslimfastq_amd.so; python3 bench.py $B --workload qlt 2>/dev/null | show
echo "== old qlt alone"; cp scratch/libsfq_old.so slimfastq_amd/libslimfastq_amd.so; python3 bench.py $B --workload qlt 2>/dev/null | show
cp /tmp/lib_keep.so slimfastq_amd/libslimfastq_amd.so
