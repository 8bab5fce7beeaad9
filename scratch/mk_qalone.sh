#!/bin/bash
# scratch/libsfq_qalone_<NAME>.so: the quality decoder timed by itself (api.cpp -DSFQ_EXP_QDEC_ALONE), chains.hip under extra defines
#   bash scratch/mk_qalone.sh NAME [-D...]
NAME=$1; shift
cd $(dirname $0)/..
B=slimfastq_amd/build
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -ffp-contract=off"
[ -f /tmp/var_api_qalone.o ] || /opt/rocm/bin/hipcc $F -DSFQ_EXP_QDEC_ALONE -x hip -c slimfastq_amd/csrc/api.cpp -o /tmp/var_api_qalone.o || exit 1
/opt/rocm/bin/hipcc $F "$@" -c slimfastq_amd/csrc/chains.hip -o /tmp/var_chains_$NAME.o || exit 1
OBJS=$(ls $B/*.o | grep -v "/api.cpp.o" | grep -v "/chains.hip.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scratch/libsfq_qalone_$NAME.so $OBJS /tmp/var_api_qalone.o /tmp/var_chains_$NAME.o -lpthread && echo built scratch/libsfq_qalone_$NAME.so
