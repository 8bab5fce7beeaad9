#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r05r
timeout -k 10 900 python scratch/ratio_table.py > gpurun_out/r05r/ratio_table.json 2> gpurun_out/r05r/ratio_table.err; echo "ratio table rc $?"; tail -3 gpurun_out/r05r/ratio_table.err | cut -c1-300
timeout -k 10 300 python scratch/cli_e2e.py 20000000 > gpurun_out/r05r/cli_e2e_20M.txt 2>&1; echo "cli rc $?"; cat gpurun_out/r05r/cli_e2e_20M.txt | tail -6
