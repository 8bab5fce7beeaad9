#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
run() { python3 bench.py --steps 12 --warmup 3 --chain-reads $2 --no-decode --no-adaptive-leg --no-cpu-baseline --no-format6-leg --no-genome-leg --no-size-sweep --no-host-leg 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read())
print('threads $1 cr $2 enc %.3f ms ratio %.4f %s' % (d['ms_per_step'], d['ratio'], d['phase_ms']))"; }
SFQ_QLT_THREADS=1024 run 1024 40
SFQ_QLT_THREADS=768 run 768 52
SFQ_QLT_THREADS=768 run 768 40
SFQ_QLT_THREADS=768 run 768 27
SFQ_QLT_THREADS=512 run 512 40
SFQ_QLT_THREADS=512 run 512 20
SFQ_QLT_THREADS=1024 run 1024 40
