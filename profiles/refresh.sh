#!/bin/bash
# Refresh the judged artifacts for the default bench configuration.  Run ON THE GPU BOX from the repo root:
#   bash profiles/refresh.sh r01j      -> gpurun_out/<tag>_*  (copy into profiles/ afterwards: profiles/refresh_collect.py <tag>)
set -e
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python bench.py --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/${TAG}_default_bench_with_cpu_baseline.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_prof -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/${TAG}_prof.log 2>&1 )
cp gpurun_out/${TAG}_prof/*/*kernel_stats.csv gpurun_out/${TAG}_default_bench_kernel_stats.csv
bash profiles/collect_pmc.sh $TAG --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_pmc_summary.txt 2>&1
cut -c1-160 gpurun_out/${TAG}_default_bench_with_cpu_baseline.json
head -4 gpurun_out/${TAG}_default_bench_kernel_stats.csv | cut -c1-120
