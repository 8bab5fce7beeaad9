"""Copy what profiles/r05_refresh.sh left under gpurun_out/<tag>/ into profiles/ and build the traffic JSON bench.py reads.
    python profiles/r05_collect.py r05z"""
import json, os, re, shutil, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
for src, dst in (("default_bench.json", "%s_default_bench.json"), ("kernel_stats.csv", "%s_default_bench_kernel_stats.csv"), ("kernel_trace.csv", "%s_default_bench_kernel_trace.csv")):
    if os.path.exists(os.path.join(go, src)):
        shutil.copy(os.path.join(go, src), os.path.join(pr, dst % tag))
summary = open(os.path.join(go, "pmc_summary.txt")).read()
head = ("# %s: rocprofv3 --pmc, one counter group per pass, --kernel-trace only (profiles/r05_refresh.sh): the default bench call,\n"
        "# 10 M x 150 bp, -l 3, frozen tables; calls = warm-up + timed step.  FETCH_SIZE / WRITE_SIZE in KB as reported.\n" % tag)
open(os.path.join(pr, "%s_pmc_summary.txt" % tag), "w").write(head + summary)
kern = {"qlt_encode": "k_qlt_encode_c", "gen_encode": "k_gen_encode_c", "rec_encode": "k_rec_tokens", "frame": "k_frame", "qlt_decode": "k_qlt_decode_c", "gen_decode": "k_gen_decode_c"}
for extra in ("genome10M_bench.json", "genome10M_kernel_stats.csv", "genome10M_kernel_trace.csv"):
    if os.path.exists(os.path.join(go, extra)):
        shutil.copy(os.path.join(go, extra), os.path.join(pr, "%s_%s" % (tag, extra)))
import glob
for f in glob.glob(os.path.join(go, "bench_*.json")):
    shutil.copy(f, os.path.join(pr, "%s_%s" % (tag, os.path.basename(f))))
out = {"source": "profiles/%s_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, per launch)" % tag,
       "config": {"reads": 10000000, "read_len": 150, "level": 3, "kind": 0, "block_reads": 1024, "kernel": 0, "tables": 1},
       "note": "bytes = counter (KB) x 1024, as reported; kernels matched by name prefix.  The guide's gfx950 correction (FETCH_SIZE reads half of a wide "
               "coalesced stream) is calibrated for 16 B/lane coalesced loads; these kernels read 16-byte pieces per LANE from 64 different lines, for "
               "which the counter is uncalibrated: the true fetch lies between the reported figure and twice it.", "kernels": {}}
for key, name in kern.items():
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        m = re.search(r"^%s.*?\s%s\s+calls=\d+ sum=\S+ per_call=(\S+)" % (re.escape(name), counter), summary, re.M)
        vals[counter] = int(float(m.group(1)) * 1024) if m else None
    out["kernels"][key] = {"kernel": name, "fetch_bytes": vals["FETCH_SIZE"], "write_bytes": vals["WRITE_SIZE"]}
json.dump(out, open(os.path.join(pr, "%s_pmc_traffic.json" % tag), "w"), indent=2)
print(json.dumps(out["kernels"], indent=1))
