#!/bin/bash
# default bench (encode + decode), lean, for the in-tree library and every scratch/libsfq_*.so; kernel trace timelines with TRACE=1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; TAG=${1:-ab}; mkdir -p gpurun_out/$TAG
LEAN="--no-cpu-baseline --no-adaptive-leg --no-genome-leg --no-format6-leg"
cp slimfastq_amd/libslimfastq_amd.so /tmp/lib_orig.so
one() {
  python3 bench.py --steps 10 --warmup 3 $LEAN 2>/dev/null < /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   enc', d['value'], d['ms_per_step'], d['phase_ms']); print('   dec', d['decode']['value'], d['decode']['ms'], d['decode']['phase_ms'])"
  if [ -n "$TRACE" ]; then
    ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/$TAG/p_$1 -- python3 $ROOT/bench.py --steps 3 --warmup 1 $LEAN > /dev/null 2> $ROOT/gpurun_out/$TAG/p_$1.log < /dev/null )
    t=$(find $ROOT/gpurun_out/$TAG/p_$1 -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && cp $t $ROOT/gpurun_out/$TAG/trace_$1.csv && python3 scratch/tl.py $ROOT/gpurun_out/$TAG/trace_$1.csv 0.3
    rm -rf $ROOT/gpurun_out/$TAG/p_$1
  fi
}
echo "== in-tree"; one intree
for f in scratch/libsfq_*.so; do [ -e "$f" ] || continue; n=$(basename $f .so); cp $f slimfastq_amd/libslimfastq_amd.so; echo "== $n"; one $n; done
cp /tmp/lib_orig.so slimfastq_amd/libslimfastq_amd.so
