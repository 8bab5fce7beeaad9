#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
run() { for k in 1 2; do timeout -k 10 150 python3 bench.py --steps 15 --warmup 4 --no-decode --no-adaptive-leg --no-cpu-baseline --no-format6-leg --no-genome-leg --no-size-sweep --no-host-leg "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('  enc %.3f ms %s' % (d['ms_per_step'], d['phase_ms']))"; done; }
for q in 0 1 0 1; do echo "== SFQ_TOK_EARLY=$q $@"; SFQ_TOK_EARLY=$q run "$@"; done
