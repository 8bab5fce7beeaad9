#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
cp slimfastq_amd/libslimfastq_amd.so /tmp/lib_keep.so
for i in 1 2; do
  echo "== in-tree"; cp /tmp/lib_keep.so slimfastq_amd/libslimfastq_amd.so; python3 scratch/dec_loop.py 10000000 6 2>/dev/null | tail -1
  for f in scratch/libsfq_*.so; do echo "== $f"; cp $f slimfastq_amd/libslimfastq_amd.so; python3 scratch/dec_loop.py 10000000 6 2>/dev/null | tail -1; done
done
cp /tmp/lib_keep.so slimfastq_amd/libslimfastq_amd.so
