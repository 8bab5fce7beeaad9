#!/bin/bash
# each model's kernels alone: the default bench with one model at a time, kernel trace
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/alone
for m in 1 2 4; do
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/alone/p$m -- python3 $ROOT/bench.py --steps 3 --warmup 1 --models $m --no-cpu-baseline --no-adaptive-leg --no-genome-leg --no-format6-leg --no-decode > /dev/null 2> $ROOT/gpurun_out/alone/log$m < /dev/null )
  t=$(find $ROOT/gpurun_out/alone/p$m -name "*kernel_trace.csv" | head -1)
  echo "== models $m"; python3 scratch/tl.py $t 0.1 | grep -v "== \|k_count_new\|k_write_new\|k_validate\|copyBuffer"
  rm -rf $ROOT/gpurun_out/alone/p$m
done
