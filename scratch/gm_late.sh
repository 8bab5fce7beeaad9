#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
run() { for n in 10000000 2000000; do python3 bench.py --kind 3 --reads $n --steps 5 --warmup 2 --no-decode --no-adaptive-leg --no-cpu-baseline --no-format6-leg --no-genome-leg --no-size-sweep --no-host-leg 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('  $n enc %.2f ms %s' % (d['ms_per_step'], d['phase_ms']))"; done; }
echo "== default"; run
export SFQ_GM_QLT_LATE=1
echo "== quality chains behind the plan"; run
