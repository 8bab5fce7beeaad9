#!/bin/bash
# build scratch/libsfq_<NAME>.so: the in-tree objects with ONE source recompiled under extra defines
#   bash scratch/mk_variant.sh NAME prior.hip -DHIST_T=128
NAME=$1; SRC=$2; shift 2
cd $(dirname $0)/..
B=slimfastq_amd/build
X=""; [[ $SRC == *.cpp ]] && X="-x hip"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -ffp-contract=off "$@" $X -c slimfastq_amd/csrc/$SRC -o /tmp/var_$NAME.o || exit 1
OBJS=$(ls $B/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scratch/libsfq_$NAME.so $OBJS /tmp/var_$NAME.o -lpthread && echo built scratch/libsfq_$NAME.so
