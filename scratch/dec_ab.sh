#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
cp slimfastq_amd/libslimfastq_amd.so /tmp/lib_orig.so
run() { for k in 1 2; do python3 bench.py --steps 6 --warmup 2 --no-adaptive-leg --no-cpu-baseline --no-format6-leg --no-genome-leg --no-size-sweep --no-host-leg "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); dec=d['decode']; print('  enc %.3f dec %.3f ms ok %s %s' % (d['ms_per_step'], dec['ms'], dec['round_trip_identical'], dec['phase_ms']))"; done; }
echo "== in-tree"; run "$@"
for f in scratch/libsfq_*.so; do [ -f $f ] || continue; cp $f slimfastq_amd/libslimfastq_amd.so; echo "== $f"; run "$@"; done
cp /tmp/lib_orig.so slimfastq_amd/libslimfastq_amd.so
echo "== in-tree again"; run "$@"
