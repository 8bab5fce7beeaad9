#!/bin/bash
# what do the quality kernel's row gathers cost?  normal build with lds_rows 0 / 240, then a build whose gathers are arithmetic (output garbage)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
echo "== q alone, lds 0";   python3 $ROOT/scratch/frozen_rt.py 10000000 0 1024 0 1 x 0   2>&1 | grep "tables=1 encode"
echo "== q alone, lds 240"; python3 $ROOT/scratch/frozen_rt.py 10000000 0 1024 0 1 x 240 2>&1 | grep "tables=1 encode"
echo "== all, lds 0";       python3 $ROOT/scratch/frozen_rt.py 10000000 0 1024 0 7 x 0   2>&1 | grep "tables=1 encode"
echo "== all, lds 240";     python3 $ROOT/scratch/frozen_rt.py 10000000 0 1024 0 7 x 240 2>&1 | grep "tables=1 encode"
cp $ROOT/slimfastq_amd/libslimfastq_amd.so /tmp/lib_orig.so; cp $ROOT/scratch/libsfq_NOGATHER.so $ROOT/slimfastq_amd/libslimfastq_amd.so
echo "== NOGATHER q alone"; python3 $ROOT/scratch/frozen_rt.py 10000000 0 1024 0 1 x 0 2>&1 | grep "tables=1 encode"
echo "== NOGATHER all";     python3 $ROOT/scratch/frozen_rt.py 10000000 0 1024 0 7 x 0 2>&1 | grep "tables=1 encode"
cp /tmp/lib_orig.so $ROOT/slimfastq_amd/libslimfastq_amd.so
