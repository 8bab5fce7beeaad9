"""timeline of the last encode and decode call in a rocprofv3 kernel trace:  python scratch/tl.py gpurun_out/<tag>/kernel_trace.csv [min_ms]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
mn = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def show(s, stop):
    t0 = int(rows[s]["Start_Timestamp"])
    for r in rows[s:]:
        a = (int(r["Start_Timestamp"]) - t0) / 1e6; b = (int(r["End_Timestamp"]) - t0) / 1e6
        if a > stop: break
        if b - a > mn: print("%8.3f %8.3f %7.3f  q%s %s" % (a, b, b - a, r.get("Queue_Id"), r["Kernel_Name"][:64]))
enc = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_count_newlines")]
print("== encode"); show(enc[-1], 26)
dec = [i for i, r in enumerate(rows) if "k_usr_decode_l" in r["Kernel_Name"]]
if dec: print("== decode"); show(dec[-1], 30)
