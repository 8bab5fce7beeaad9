#!/bin/bash
# one pytest selection with every scratch/libsfq_<variant>.so in place of the in-tree library
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
cp slimfastq_amd/libslimfastq_amd.so /tmp/lib_orig.so
for f in scratch/libsfq_*.so; do
  cp $f slimfastq_amd/libslimfastq_amd.so
  echo "== $f"; timeout -k 10 300 python -m pytest "$@" -q -x < /dev/null 2>&1 | tail -2
done
cp /tmp/lib_orig.so slimfastq_amd/libslimfastq_amd.so
