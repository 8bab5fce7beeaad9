#!/bin/bash
# the bases of the genome-sampled call alone on the chip (--models 2): what each of the match model's kernels takes by itself
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r05s
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r05s/prof -- python3 $ROOT/bench.py --kind 3 --models 2 --steps 3 --warmup 1 --no-decode --no-adaptive-leg --no-cpu-baseline --no-format6-leg --no-genome-leg --no-size-sweep --no-host-leg > $ROOT/gpurun_out/r05s/bench.json 2> $ROOT/gpurun_out/r05s/prof.log )
f=$(find gpurun_out/r05s/prof -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r05s/kernel_stats.csv
f=$(find gpurun_out/r05s/prof -name "*kernel_trace.csv" | head -1); cp $f gpurun_out/r05s/kernel_trace.csv; rm -rf gpurun_out/r05s/prof
head -12 gpurun_out/r05s/kernel_stats.csv | cut -c1-60,160-260
tail -c 700 gpurun_out/r05s/bench.json
