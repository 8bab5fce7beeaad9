#!/bin/bash
# genome-sampled call (10 M and 2 M reads) with the in-tree library and with every scratch/libsfq_<variant>.so in its place, same box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
cp slimfastq_amd/libslimfastq_amd.so /tmp/lib_orig.so
run() {
  for n in 10000000 2000000; do
  python3 bench.py --kind 3 --reads $n --steps 5 --warmup 2 --no-adaptive-leg --no-cpu-baseline --no-format6-leg --no-genome-leg --no-size-sweep --no-host-leg 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('  $n enc %.2f ms (gen %.2f) | dec %.2f ms (gen %.2f)' % (d['ms_per_step'], d['phase_ms']['gen'], d['decode']['ms'], d['decode']['phase_ms']['gen']))"
  done
}
echo "== in-tree"; run
for f in scratch/libsfq_*.so; do cp $f slimfastq_amd/libslimfastq_amd.so; echo "== $f"; run; done
cp /tmp/lib_orig.so slimfastq_amd/libslimfastq_amd.so
echo "== in-tree again"; run
