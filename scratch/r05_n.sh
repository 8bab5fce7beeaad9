#!/bin/bash
# the default bench line as the driver runs it (every leg), then a one-rank rehearsal of the N > 1 path with the C4 leg, small
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r05n
( time timeout -k 10 900 python bench.py > gpurun_out/r05n/default_full.json 2> gpurun_out/r05n/default_full.err ) 2> gpurun_out/r05n/default_full.time
tail -3 gpurun_out/r05n/default_full.time
python - <<PY
import json
d=json.loads(open("gpurun_out/r05n/default_full.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "decode", d["decode"]["value"], "roofline", {k:d["roofline"][k] for k in ("kernel_name","over","achieved","frac","avg_ms","traffic")})
print("framing", d.get("roofline_framing"))
print("adaptive", d.get("adaptive_tables"), "host", d.get("host_to_host"), "f6", d.get("format6"))
print("genome", d.get("genome_sampled"))
print("cpu", d.get("cpu_baseline"), d.get("ratio_vs_reference"), d.get("ratio_vs_reference_full"))
for r in d.get("size_sweep", []): print("  sweep", r["reads"], r["encode_MBps"], r.get("decode_MBps"), r.get("ours_over_reference"))
PY
SFQ_BENCH_C4=1 timeout -k 10 300 python bench.py --force-dist --reads 500000 --steps 3 --warmup 1 --c4-reads-per-gpu 300000 > gpurun_out/r05n/c4_rehearsal.json 2> gpurun_out/r05n/c4_rehearsal.err; tail -c 600 gpurun_out/r05n/c4_rehearsal.json; tail -3 gpurun_out/r05n/c4_rehearsal.err
