#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
cp slimfastq_amd/libslimfastq_amd.so /tmp/lib_orig.so
one() { python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-genome-leg --no-decode 2>/dev/null < /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['adaptive_tables'], {k:v for k,v in d['format6'].items() if 'MBps' in k})"; }
echo "== in-tree"; one
for f in scratch/libsfq_*.so; do [ -e "$f" ] || continue; cp $f slimfastq_amd/libslimfastq_amd.so; echo "== $f"; one; done
cp /tmp/lib_orig.so slimfastq_amd/libslimfastq_amd.so
