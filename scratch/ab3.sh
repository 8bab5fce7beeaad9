#!/bin/bash
# the default bench three times alternating between the in-tree library and scratch/libsfq_OLD.so
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
LEAN="--no-cpu-baseline --no-adaptive-leg --no-genome-leg --no-format6-leg"
cp slimfastq_amd/libslimfastq_amd.so /tmp/lib_new.so
for i in 1 2 3; do
  for v in new OLD; do
    if [ $v = new ]; then cp /tmp/lib_new.so slimfastq_amd/libslimfastq_amd.so; else cp scratch/libsfq_OLD.so slimfastq_amd/libslimfastq_amd.so; fi
    python3 bench.py --steps 20 --warmup 5 $LEAN 2>/dev/null < /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['phase_ms']['device_total'], 'dec', d['decode']['ms'])"
  done
done
cp /tmp/lib_new.so slimfastq_amd/libslimfastq_amd.so
