#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for i in 1 2; do python3 bench.py --kind 3 --steps 3 --warmup 1 --no-cpu-baseline --no-adaptive-leg --no-genome-leg --no-format6-leg 2>/dev/null < /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ratio'], d['phase_ms'], 'dec', d['decode']['value'], d['decode']['round_trip_identical'])"; done
