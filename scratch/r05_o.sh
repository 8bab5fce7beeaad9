#!/bin/bash
# parity tests that exercise the framing, then the default call (framing time in roofline_framing)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
T=${1:-r05o}
mkdir -p gpurun_out/$T
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/$T/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -4 gpurun_out/$T/pytest.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-adaptive-leg --no-cpu-baseline --no-format6-leg --no-genome-leg --no-size-sweep --no-host-leg > gpurun_out/$T/default.json 2> gpurun_out/$T/default.err
python - <<PY
import json
d=json.loads(open("gpurun_out/$T/default.json").read().strip().splitlines()[-1])
print("enc %.2f ms %.1f GB/s ratio %.4f %s | dec %.2f ms %.1f GB/s" % (d["ms_per_step"], d["value"]/1e3, d["ratio"], d["phase_ms"], d["decode"]["ms"], d["decode"]["value"]/1e3))
print(d["roofline_framing"])
PY
