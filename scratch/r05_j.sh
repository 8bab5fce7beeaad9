#!/bin/bash
# frozen tests, then genome-sampled calls at 10 M and 2 M reads (no profiler), then the 10 M one traced
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
T=${1:-r05j}
mkdir -p gpurun_out/$T
timeout -k 10 400 python -m pytest tests/test_frozen_tables.py -m gpu -x -q > gpurun_out/$T/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/$T/pytest.log
[ $rc -ne 0 ] && exit 1
for n in 10000000 2000000; do
timeout -k 10 300 python bench.py --kind 3 --reads $n --steps 5 --warmup 2 --no-adaptive-leg --no-cpu-baseline --no-format6-leg --no-genome-leg --no-size-sweep --no-host-leg > gpurun_out/$T/genome$n.json 2> gpurun_out/$T/genome$n.err
python - <<PY
import json
d=json.loads(open("gpurun_out/$T/genome$n.json").read().strip().splitlines()[-1])
print($n, "enc %.2f ms %.1f GB/s ratio %.4f %s | dec %.2f ms %.1f GB/s %s" % (d["ms_per_step"], d["value"]/1e3, d["ratio"], d["phase_ms"], d["decode"]["ms"], d["decode"]["value"]/1e3, d["decode"]["phase_ms"]))
PY
done
bash scratch/prof_bench.sh $T/genome --kind 3 --steps 3 --warmup 1 --no-adaptive-leg --no-cpu-baseline --no-format6-leg --no-genome-leg --no-size-sweep --no-host-leg > /dev/null
f=$(find gpurun_out/$T/genome/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/$T/genome/kernel_trace.csv; rm -rf gpurun_out/$T/genome/prof
