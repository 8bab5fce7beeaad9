#!/bin/bash
# alone-time of the header kernels (a PMC run serialises the kernels): prints the duration of every k_rec_* launch
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES --kernel-trace --output-format csv -d /tmp/rv -- python3 $ROOT/scratch/frozen_rt.py 10000000 0 1024 0 1 x > /tmp/rv.log 2>&1
python3 - <<PY
import csv,glob,collections
d=collections.defaultdict(list)
for f in glob.glob("/tmp/rv/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "k_rec_" in r["Kernel_Name"]: d[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
for k,v in sorted(d.items()): print(k, ["%.2f"%x for x in v])
PY
