#!/bin/bash
# which chain kernels slow each other down?  the encode with one, two, three models; durations of the chain kernels from a kernel trace
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for m in 4 2 1 6 5 3 7; do
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -- python3 $ROOT/scratch/frozen_rt.py 10000000 0 1024 0 $m x > /tmp/pp.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pp/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
last={}
for r in rows:
    n=r["Kernel_Name"].split("(")[0]
    for key in ("k_qlt_encode_c","k_gen_encode_c","k_rec_encode_f","k_gen_exc_w","k_qlt_hist","k_gen_count","k_rec_count_f"):
        if key in n: last.setdefault(key,[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
print("models $m:", {k:"%.2f"%(sum(v[-len(v)//3:])/max(1,len(v[-len(v)//3:]))) for k,v in last.items()})
PY
  grep "tables=1 encode" /tmp/pp.log | cut -c1-60
done
