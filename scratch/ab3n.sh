#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
LEAN="--no-cpu-baseline --no-adaptive-leg --no-genome-leg --no-format6-leg --no-decode"
cp slimfastq_amd/libslimfastq_amd.so /tmp/lib_keep.so
for i in 1 2 3; do
  for v in $1 scratch/libsfq_OLD.so; do
    cp $v slimfastq_amd/libslimfastq_amd.so
    python3 bench.py --steps 20 --warmup 5 $LEAN 2>/dev/null < /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['phase_ms']['device_total'], d['ratio'], d['roofline']['coder_ms'])"
  done
done
cp /tmp/lib_keep.so slimfastq_amd/libslimfastq_amd.so
