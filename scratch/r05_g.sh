#!/bin/bash
# genome-sampled call at several chain lengths: how the decode scales with the number of chains
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r05g
for cr in 12 24 49; do
  timeout -k 10 300 python bench.py --kind 3 --steps 3 --warmup 1 --chain-reads $cr --no-adaptive-leg --no-cpu-baseline --no-format6-leg --no-genome-leg --no-size-sweep --no-host-leg > gpurun_out/r05g/cr$cr.json 2> gpurun_out/r05g/cr$cr.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05g/cr$cr.json").read().strip().splitlines()[-1])
print("chain_reads $cr: enc %.1f ms %.1f GB/s ratio %.4f | dec %.1f ms %.1f GB/s %s" % (d["ms_per_step"], d["value"]/1e3, d["ratio"], d["decode"]["ms"], d["decode"]["value"]/1e3, d["decode"]["phase_ms"]))
PY
done
