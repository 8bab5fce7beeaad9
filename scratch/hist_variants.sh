#!/bin/bash
# default bench with every scratch/libsfq_<variant>.so in place of the in-tree library: ms per step + the avg duration of one kernel (kernel trace)
#   bash scratch/hist_variants.sh <tag> <kernel-name-prefix>
TAG=$1; K=${2:-k_qlt_hist}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT; mkdir -p gpurun_out/$TAG
cp slimfastq_amd/libslimfastq_amd.so /tmp/lib_orig.so
LEAN="--no-cpu-baseline --no-adaptive-leg --no-genome-leg --no-format6-leg --no-decode"
one() {
  python3 bench.py --steps 10 --warmup 3 $LEAN 2>/dev/null < /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], d['ms_per_step'], d['ratio'], d['phase_ms'])"
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/$TAG/p_$1 -- python3 $ROOT/bench.py --steps 3 --warmup 1 $LEAN > /dev/null 2> $ROOT/gpurun_out/$TAG/p_$1.log < /dev/null )
  f=$(find $ROOT/gpurun_out/$TAG/p_$1 -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && grep "$K" "$f" | cut -d, -f1-5 | cut -c1-100
  t=$(find $ROOT/gpurun_out/$TAG/p_$1 -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && cp $t $ROOT/gpurun_out/$TAG/trace_$1.csv
  rm -rf $ROOT/gpurun_out/$TAG/p_$1
}
echo "== in-tree"; one intree
for f in scratch/libsfq_*.so; do
  n=$(basename $f .so); cp $f slimfastq_amd/libslimfastq_amd.so
  echo "== $n"; one $n
done
cp /tmp/lib_orig.so slimfastq_amd/libslimfastq_amd.so
