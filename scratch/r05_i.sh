#!/bin/bash
# the whole GPU suite, then the genome-sampled call at 2 M reads and the default call (no profiler)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
T=${1:-r05i}
mkdir -p gpurun_out/$T
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/$T/pytest.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/$T/pytest.log
timeout -k 10 300 python bench.py --kind 3 --reads 2000000 --steps 5 --warmup 2 --no-adaptive-leg --no-cpu-baseline --no-format6-leg --no-genome-leg --no-size-sweep --no-host-leg > gpurun_out/$T/genome2M.json 2> gpurun_out/$T/genome2M.err
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-adaptive-leg --no-cpu-baseline --no-format6-leg --no-genome-leg --no-size-sweep --no-host-leg > gpurun_out/$T/default.json 2> gpurun_out/$T/default.err
python - <<PY
import json
for n in ("genome2M","default"):
    d=json.loads(open("gpurun_out/$T/%s.json"%n).read().strip().splitlines()[-1])
    print(n, "enc %.2f ms %.1f GB/s ratio %.4f %s | dec %.2f ms %.1f GB/s %s" % (d["ms_per_step"], d["value"]/1e3, d["ratio"], d["phase_ms"], d["decode"]["ms"], d["decode"]["value"]/1e3, d["decode"]["phase_ms"]))
PY
