#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
X="--no-genome-leg --no-format6-leg --no-adaptive-leg --no-cpu-baseline --no-size-sweep --no-host-leg --no-decode"
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['phase_ms']['device_total'])"; }
for a in "--workload qlt --steps 5 --warmup 2" "--workload qlt --steps 20 --warmup 5" "--kind 2 --steps 5 --warmup 2" "--kind 2 --steps 20 --warmup 5" "--steps 5 --warmup 2" "--steps 20 --warmup 5"; do echo "== $a"; python3 bench.py $X $a 2>/dev/null | show; done
