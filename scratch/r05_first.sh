#!/bin/bash
# round 5, first call: GPU tests, then kernel traces of the default call and of a genome-sampled call
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r05a
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r05a/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r05a/pytest.log
bash scratch/prof_bench.sh r05a/default --steps 5 --warmup 2 --no-adaptive-leg --no-cpu-baseline --no-format6-leg --no-genome-leg --no-size-sweep --no-host-leg
bash scratch/prof_bench.sh r05a/genome --kind 3 --steps 3 --warmup 1 --no-adaptive-leg --no-cpu-baseline --no-format6-leg --no-genome-leg --no-size-sweep --no-host-leg
for t in default genome; do f=$(find gpurun_out/r05a/$t/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r05a/$t/kernel_trace.csv; rm -rf gpurun_out/r05a/$t/prof; done
ls -la gpurun_out/r05a/*
