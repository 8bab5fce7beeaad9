#!/bin/bash
# scratch/mk_gm_variant.sh <name> <gm source file>: scratch/libsfq_<name>.so = the in-tree library with that gm.hip in place of its own
set -e
N=$1; SRC=$2
D=slimfastq_amd/build
cp $SRC /tmp/gm_variant.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -ffp-contract=off -Islimfastq_amd/csrc -Iinclude -c /tmp/gm_variant.hip -o /tmp/gm_variant.o
OBJS=$(ls $D/*.o | grep -v gm.hip.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scratch/libsfq_$N.so $OBJS /tmp/gm_variant.o -lpthread
echo built scratch/libsfq_$N.so
