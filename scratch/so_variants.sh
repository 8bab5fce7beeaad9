#!/bin/bash
# default bench with the in-tree library and with every scratch/libsfq_<variant>.so in its place
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
cp slimfastq_amd/libslimfastq_amd.so /tmp/lib_orig.so
echo "== in-tree"; python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-adaptive-leg --no-decode 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['coder_ms'], d['phase_ms'])"
for f in scratch/libsfq_*.so; do
  cp $f slimfastq_amd/libslimfastq_amd.so
  echo "== $f"; python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-adaptive-leg --no-decode 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['coder_ms'], d['phase_ms'])"
done
cp /tmp/lib_orig.so slimfastq_amd/libslimfastq_amd.so
