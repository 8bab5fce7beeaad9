#!/bin/bash
# scratch/multi_ab.sh: in-tree vs scratch/libsfq_prev.so on several workloads, same box, alternating
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
cp slimfastq_amd/libslimfastq_amd.so /tmp/lib_new.so
X="--steps 10 --warmup 3 --no-decode --no-adaptive-leg --no-cpu-baseline --no-format6-leg --no-genome-leg --no-size-sweep --no-host-leg"
one() { python3 bench.py $X "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%.3f' % d['ms_per_step'], end=' ')"; }
for w in "--reads 2000000" "--reads 2700000" "--reads 5000000" "--workload qlt" "--kind 2" "--level 1"; do
  echo -n "$w : new "; cp /tmp/lib_new.so slimfastq_amd/libslimfastq_amd.so; one $w; one $w
  echo -n " prev "; cp scratch/libsfq_prev.so slimfastq_amd/libslimfastq_amd.so; one $w; one $w
  echo -n " new "; cp /tmp/lib_new.so slimfastq_amd/libslimfastq_amd.so; one $w; echo
done
cp /tmp/lib_new.so slimfastq_amd/libslimfastq_amd.so
