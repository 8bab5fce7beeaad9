"""Copy what profiles/refresh.sh left under gpurun_out/ into profiles/ and rebuild the traffic JSON bench.py reads.
Usage (in the dev container, after the gpurun call): python profiles/refresh_collect.py r01j"""
import json
import os
import re
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
for name in ("default_bench_with_cpu_baseline.json", "default_bench_kernel_stats.csv"):
    shutil.copy(os.path.join(go, "%s_%s" % (tag, name)), os.path.join(pr, "%s_%s" % (tag, name)))
summary = open(os.path.join(go, "%s_pmc_summary.txt" % tag)).read()
head = ("# %s: rocprofv3 PMC, one counter group per pass (profiles/collect_pmc.sh %s --steps 1 --warmup 1 --no-cpu-baseline): the default bench,\n"
        "# 10 M x 150 bp, 9766 blocks, 1.5 G symbols per model.  calls=2 = warm-up + timed step.  FETCH_SIZE / WRITE_SIZE in KB as reported\n"
        "# (not doubled: the traffic is 64-byte table sectors, not wide streaming reads).\n" % (tag, tag))
open(os.path.join(pr, "%s_pmc_summary.txt" % tag), "w").write(head + summary)
kern = {"qlt_encode": "k_qlt_encode_k2", "gen_encode": "void k_gen_encode_k<2>", "rec_encode": "k_rec_encode_w_fast"}
out = {"source": "profiles/%s_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, per launch)" % tag,
       "config": {"reads": 10000000, "read_len": 150, "level": 3, "kind": 0, "block_reads": 1024, "kernel": 0},
       "note": "bytes = counter (KB) x 1024; FETCH_SIZE is not doubled: the gfx950 x2 correction is for wide coalesced streaming reads, "
               "this traffic is 64-byte sectors of the adaptive tables (random rows); the coalesced text read is < 3 % of it", "kernels": {}}
for key, name in kern.items():
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        m = re.search(r"^%s\s+%s\s+calls=\d+ sum=\S+ per_call=(\S+)" % (re.escape(name), counter), summary, re.M)
        vals[counter] = int(float(m.group(1)) * 1024)
    out["kernels"][key] = {"kernel": name.replace("void ", ""), "fetch_bytes": vals["FETCH_SIZE"], "write_bytes": vals["WRITE_SIZE"]}
json.dump(out, open(os.path.join(pr, "%s_pmc_traffic.json" % tag), "w"), indent=2)
print(json.dumps(out["kernels"], indent=1))
