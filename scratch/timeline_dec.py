import csv,glob,sys
f=glob.glob(sys.argv[1]+"/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if ("k_usr_fill" in r["Kernel_Name"] or r["Kernel_Name"].startswith("k_usr_decode_l"))]
s=[i for i in idx if "k_usr_fill" in rows[i]["Kernel_Name"]][-1] if any("k_usr_fill" in rows[i]["Kernel_Name"] for i in idx) else idx[-1]
t0=int(rows[s]["Start_Timestamp"])
for r in rows[s:]:
    a=(int(r["Start_Timestamp"])-t0)/1e6; b=(int(r["End_Timestamp"])-t0)/1e6
    if b-a>0.02: print("%8.3f %8.3f %7.3f  q%s %s"%(a,b,b-a,r.get("Queue_Id"),r["Kernel_Name"][:70]))
