#!/bin/bash
# default bench with explicit chain lengths: bash scratch/cr_ab.sh 49 44 39 36
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
B="--steps 8 --warmup 3 --no-size-sweep --no-cpu-baseline --no-genome-leg --no-format6-leg --no-adaptive-leg --no-host-leg"
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['chains_per_gpu'], d['value'], d['ms_per_step'], d['ratio'], d['phase_ms'], 'dec', d['decode']['ms'], d['decode']['round_trip_identical'])"; }
for r in "$@"; do echo "== chain-reads $r"; python3 bench.py $B --chain-reads $r 2>/dev/null | show; done
