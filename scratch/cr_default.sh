#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for cr in "$@"; do
  for k in 1 2; do python3 bench.py --steps 12 --warmup 3 --chain-reads $cr --no-adaptive-leg --no-cpu-baseline --no-format6-leg --no-genome-leg --no-size-sweep --no-host-leg 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); dec=d.get('decode') or {}
print('cr $cr enc %.3f ms ratio %.4f dec %.3f ms  %s' % (d['ms_per_step'], d['ratio'], dec.get('ms', 0), d['phase_ms']))"; done
done
