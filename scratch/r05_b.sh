#!/bin/bash
# frozen-table parity tests, then the genome-sampled call traced
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
TAG=${1:-r05b}
mkdir -p gpurun_out/$TAG
timeout -k 10 400 python -m pytest tests/test_frozen_tables.py -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -15 gpurun_out/$TAG/pytest.log
[ $rc -ne 0 ] && exit 1
bash scratch/prof_bench.sh $TAG/genome --kind 3 --steps 3 --warmup 1 --no-adaptive-leg --no-cpu-baseline --no-format6-leg --no-genome-leg --no-size-sweep --no-host-leg
for t in genome; do f=$(find gpurun_out/$TAG/$t/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/$TAG/$t/kernel_trace.csv; rm -rf gpurun_out/$TAG/$t/prof; done
