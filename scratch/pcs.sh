#!/bin/bash
# PC sampling of one bench configuration: bash scratch/pcs.sh <tag> <method: stochastic|host_trap> <unit> <interval> [bench args]
TAG=$1; METHOD=$2; UNIT=$3; IVL=$4; shift 4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $METHOD --pc-sampling-unit $UNIT --pc-sampling-interval $IVL --kernel-trace --output-format csv -d $OUT/prof -- python3 $ROOT/bench.py "$@" > $OUT/bench.json 2> $OUT/prof.log < /dev/null )
echo "rc=$?"
tail -5 $OUT/prof.log | cut -c1-300
find $OUT/prof -type f | head -20
for f in $(find $OUT/prof -name "*pc_sampling*.csv"); do wc -l $f; head -3 $f | cut -c1-400; done
