#!/bin/bash
# the round's closing artifacts (run on the GPU box from the repo root): default bench profile set, other workloads, decode counters, CLI
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT; mkdir -p gpurun_out
bash profiles/collect_r02.sh r02z > gpurun_out/r02z_collect.log 2>&1 || exit 1
echo "collect done" > gpurun_out/r02z_progress.txt
cd $ROOT
python bench.py --steps 3 --warmup 1 --workload qlt --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02z_bench_qlt.json || exit 1
python bench.py --steps 3 --warmup 1 --kind 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02z_bench_binned.json || exit 1
python bench.py --steps 3 --warmup 1 --level 4 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02z_bench_l4.json || exit 1
python bench.py --steps 3 --warmup 1 --kind 1 --cpu-sample-reads 2000 2>/dev/null | tail -1 > gpurun_out/r02z_bench_long.json || exit 1
echo "workloads 1 done" >> gpurun_out/r02z_progress.txt
python bench.py --steps 3 --warmup 1 --kind 3 --reads 2000000 --cpu-sample-reads 2000000 2>/dev/null | tail -1 > gpurun_out/r02z_bench_genome.json || exit 1
python bench.py --steps 3 --warmup 1 --kind 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02z_bench_genome_10M.json || exit 1
echo "workloads 2 done" >> gpurun_out/r02z_progress.txt
bash scratch/pmc_dec.sh r02z_pmcdec > gpurun_out/r02z_decode_pmc.txt 2>&1 || exit 1
python scratch/cli_e2e.py 20000000 > gpurun_out/r02z_cli_end_to_end_20M_reads.txt 2>&1 || exit 1
echo "all done" >> gpurun_out/r02z_progress.txt
