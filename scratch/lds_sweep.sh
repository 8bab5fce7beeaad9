#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
LEAN="--no-cpu-baseline --no-adaptive-leg --no-genome-leg --no-format6-leg --no-decode"
for rep in 1 2; do
for opt in "" "--lds-rows 256" "--lds-rows 512" "--lds-rows 800" "--lds-rows 1024"; do
  python3 bench.py --steps 20 --warmup 5 $LEAN $opt 2>/dev/null < /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$opt]', d['ms_per_step'], d['phase_ms']['device_total'], d['roofline']['coder_ms'])"
done; done
