#!/bin/bash
# the other workloads, one bench line each, kept whole under gpurun_out/wl4/ (round 4)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/wl4
X="--no-genome-leg --no-format6-leg --no-adaptive-leg --no-cpu-baseline --no-size-sweep --no-host-leg"
run() { name=$1; shift; python3 bench.py --steps 5 --warmup 2 $X "$@" 2>gpurun_out/wl4/$name.err < /dev/null | tail -1 > gpurun_out/wl4/$name.json
  python3 -c "
import json; d=json.load(open('gpurun_out/wl4/$name.json')); print('$name', d['value'], d['ms_per_step'], d.get('ratio'), 'dec', (d.get('decode') or {}).get('value'), (d.get('decode') or {}).get('round_trip_identical'))"; }
run qlt --workload qlt
run binned --kind 2
run l4 --level 4
run l1 --level 1
run long --kind 1
run genome2M --kind 3 --reads 2000000
run reads2M --reads 2000000
run reads5M --reads 5000000
