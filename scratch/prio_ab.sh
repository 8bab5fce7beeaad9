#!/bin/bash
# stream priorities of the auxiliary streams (bit 0: headers, 1: exceptions, 2: bases), plain and distributed path
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
B="--steps 8 --warmup 3 --no-size-sweep --no-cpu-baseline --no-genome-leg --no-format6-leg --no-adaptive-leg --no-host-leg --no-decode"
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['phase_ms'])"; }
for m in 2 3 0 7 1; do echo "== prio mask $m plain"; SFQ_EXP_PRIO=$m python3 bench.py $B 2>/dev/null | show; echo "== prio mask $m dist"; SFQ_EXP_PRIO=$m python3 bench.py $B --force-dist 2>/dev/null | show; done
