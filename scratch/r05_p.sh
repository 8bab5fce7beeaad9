#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
T=${1:-r05p}
mkdir -p gpurun_out/$T
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/$T/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -4 gpurun_out/$T/pytest.log
[ $rc -ne 0 ] && exit 1
for a in "" "--reads 2000000" "--kind 1"; do
timeout -k 10 300 python bench.py --steps 10 --warmup 3 $a --no-adaptive-leg --no-cpu-baseline --no-format6-leg --no-genome-leg --no-size-sweep --no-host-leg > gpurun_out/$T/b.json 2> gpurun_out/$T/b.err
python - <<PY
import json
d=json.loads(open("gpurun_out/$T/b.json").read().strip().splitlines()[-1])
print("[$a] enc %.2f ms %.1f GB/s ratio %.4f %s | dec %.2f ms %.1f GB/s %s" % (d["ms_per_step"], d["value"]/1e3, d["ratio"], d["phase_ms"], d["decode"]["ms"], d["decode"]["value"]/1e3, d["decode"]["phase_ms"]))
PY
done
