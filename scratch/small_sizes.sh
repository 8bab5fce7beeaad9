#!/bin/bash
# encode / decode ms at small call sizes (in-tree library): bash scratch/small_sizes.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
B="--steps 10 --warmup 3 --no-size-sweep --no-cpu-baseline --no-genome-leg --no-format6-leg --no-adaptive-leg --no-host-leg"
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['ratio'], d['phase_ms'], 'dec', d['decode']['ms'], d['decode']['round_trip_identical'])"; }
for n in 600000 1350000 2000000 5000000 10000000; do echo "== reads $n"; python3 bench.py $B --reads $n 2>/dev/null | show; done
