#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for cr in 0 $((0x80000000+6700)) $((0x80000000+6500)) 0; do
python3 bench.py --kind 1 --steps 8 --warmup 2 --chain-reads $cr --no-adaptive-leg --no-cpu-baseline --no-format6-leg --no-genome-leg --no-size-sweep --no-host-leg 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); dec=d['decode']; print('  cr $cr: enc %.3f ms %.1f GB/s dec %.3f ms ratio %.4f chains %s ok %s' % (d['ms_per_step'], d['value']/1000, dec['ms'], d['ratio'], d['config'].get('chains_per_gpu'), dec['round_trip_identical']))"
done
