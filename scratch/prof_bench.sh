#!/bin/bash
# kernel trace of the default bench line: gpurun_out/<tag>/kernel_stats.csv (+ the bench's own JSON)
#   bash scratch/prof_bench.sh <tag> [bench args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/$TAG
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/$TAG/prof -- python3 $ROOT/bench.py "$@" > $ROOT/gpurun_out/$TAG/bench.json 2> $ROOT/gpurun_out/$TAG/prof.log )
f=$(find $ROOT/gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $ROOT/gpurun_out/$TAG/kernel_stats.csv; head -16 $ROOT/gpurun_out/$TAG/kernel_stats.csv | cut -c1-110,200-; fi
python3 - <<PY
import json
d = json.loads(open("$ROOT/gpurun_out/$TAG/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["ratio"], d["phase_ms"], d["roofline"]["coder_ms"])
PY
