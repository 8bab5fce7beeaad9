#!/bin/bash
# decode leg of the default bench for several --dec-lds-rows values (args), in-tree lib
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
B="--steps 3 --warmup 1 --no-size-sweep --no-cpu-baseline --no-genome-leg --no-format6-leg --no-adaptive-leg"
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['decode']['ms'], d['decode']['phase_ms'])"; }
for r in "$@"; do echo "== dec-lds-rows $r"; python3 bench.py $B --dec-lds-rows $r 2>/dev/null | show; done
