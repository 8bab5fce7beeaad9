#!/bin/bash
# default encode with different LDS images for the quality chains (0 = automatic 800 rows; 4294967295 = none: the 256-lane kernel)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
B="--steps 8 --warmup 3 --no-size-sweep --no-cpu-baseline --no-genome-leg --no-format6-leg --no-adaptive-leg --no-host-leg --no-decode"
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['chains_per_gpu'], d['value'], d['ms_per_step'], d['phase_ms'], d['roofline']['coder_ms'])"; }
for r in "$@"; do echo "== lds-rows $r"; python3 bench.py $B --lds-rows $r 2>/dev/null | show; done
