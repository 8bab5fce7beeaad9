#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for n in 0 400 800 1200 1600 2400 0; do
  python3 bench.py --steps 10 --warmup 3 --lds-rows $n --no-decode --no-adaptive-leg --no-cpu-baseline --no-format6-leg --no-genome-leg --no-size-sweep --no-host-leg 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('lds_rows $n enc %.3f ms %s' % (d['ms_per_step'], d['phase_ms']))"
done
