#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
T=${1:-r05q}
mkdir -p gpurun_out/$T
timeout -k 10 600 python -m pytest tests/test_frozen_tables.py -m gpu -x -q > gpurun_out/$T/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -4 gpurun_out/$T/pytest.log
[ $rc -ne 0 ] && exit 1
for a in "--kind 1" "--kind 2" "--level 1" "--level 4" "--workload qlt"; do
timeout -k 10 300 python bench.py --steps 10 --warmup 3 $a --no-adaptive-leg --no-cpu-baseline --no-format6-leg --no-genome-leg --no-size-sweep --no-host-leg > gpurun_out/$T/b.json 2> gpurun_out/$T/b.err
python - <<PY
import json
d=json.loads(open("gpurun_out/$T/b.json").read().strip().splitlines()[-1])
dec=d.get("decode") or {"ms":0,"value":0,"phase_ms":None}
print("[$a] enc %.2f ms %.1f GB/s ratio %.4f %s | dec %.2f ms %.1f GB/s" % (d["ms_per_step"], d["value"]/1e3, d["ratio"], d["phase_ms"], dec["ms"], dec["value"]/1e3))
PY
done
