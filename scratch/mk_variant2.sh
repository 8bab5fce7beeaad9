#!/bin/bash
# scratch/mk_variant2.sh <name> <file in csrc> [extra compiler flags]: scratch/libsfq_<name>.so = the in-tree library with that file rebuilt with the flags
set -e
N=$1; F=$2; shift 2
D=slimfastq_amd/build
X=""; case $F in *.cpp) X="-x hip";; esac
/opt/rocm/bin/hipcc $X -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -ffp-contract=off "$@" -c slimfastq_amd/csrc/$F -o /tmp/variant_$N.o
OBJS=$(ls $D/*.o | grep -v "/$F.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scratch/libsfq_$N.so $OBJS /tmp/variant_$N.o -lpthread
echo built scratch/libsfq_$N.so
