#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
LEAN="--no-cpu-baseline --no-adaptive-leg --no-genome-leg --no-format6-leg"
for opt in "" "--lds-rows 512" "--lds-rows 800" "--chain-reads 40" "--chain-reads 32" "--chain-reads 40 --lds-rows 512"; do
  echo "== $opt"
  python3 bench.py --steps 10 --warmup 3 $LEAN $opt 2>/dev/null < /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   enc', d['value'], d['ms_per_step'], d['ratio'], d['phase_ms']); print('   dec', d['decode']['value'], d['decode']['ms'], d['decode']['phase_ms'])"
done
