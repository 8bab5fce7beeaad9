#!/bin/bash
# scratch/trace_dec.sh: kernel trace of the default DECODE call -> the last decode call's kernels over 0.1 ms
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trdec; mkdir -p $OUT
LEAN="--steps 2 --warmup 1 --no-cpu-baseline --no-genome-leg --no-format6-leg --no-adaptive-leg --no-size-sweep --no-host-leg"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -- python3 $ROOT/bench.py $LEAN "$@" > $OUT/prof.json 2> $OUT/prof.log < /dev/null )
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_assemble' in r['Kernel_Name']]
iend = idx[-1]
# the call starts at the first kernel after the previous assemble
i0 = idx[-2] + 1 if len(idx) > 1 else 0
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:iend + 1]:
    s = (int(r['Start_Timestamp']) - t0) / 1e6; e = (int(r['End_Timestamp']) - t0) / 1e6
    if e - s > 0.02: print('%-44s %7.2f %7.2f  %5.2f' % (r['Kernel_Name'].replace('void ', '')[:44], s, e, e - s))
PY
rm -rf $OUT/prof
