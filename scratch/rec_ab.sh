#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
run() { python3 bench.py --steps 8 --warmup 2 --no-adaptive-leg --no-cpu-baseline --no-format6-leg --no-genome-leg --no-size-sweep --no-host-leg 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); dec=d['decode']; print('  want $1 floor $2: enc %.3f dec %.3f ms ratio %.4f ok %s enc %s dec %s' % (d['ms_per_step'], dec['ms'], d['ratio'], dec['round_trip_identical'], d['phase_ms'], dec['phase_ms']))"; }
for cfg in "61440 128" "81920 96" "122880 64" "163840 48" "245760 32" "61440 128"; do set -- $cfg; SFQ_REC_WANT=$1 SFQ_REC_FLOOR=$2 run $1 $2; done
