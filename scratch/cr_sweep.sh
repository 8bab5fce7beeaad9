#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
LEAN="--no-cpu-baseline --no-adaptive-leg --no-genome-leg --no-format6-leg"
for cr in 0 40 32 24; do
  python3 bench.py --steps 10 --warmup 3 $LEAN --chain-reads $cr 2>/dev/null < /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[cr $cr]', d['ms_per_step'], d['ratio'], d['config']['chains_per_gpu'], d['roofline']['coder_ms'], 'dec', d['decode']['ms'], d['decode']['phase_ms'])"
done
