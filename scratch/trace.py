import csv,sys,glob
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows=[r for r in rows if any(k in r['Kernel_Name'] for k in ('k_qlt','k_gen','k_rec','k_rc','k_usr'))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step: take last N kernels
last=rows[-8:]
t0=min(int(r['Start_Timestamp']) for r in last)
for r in last:
    print('%-40s start %8.2f ms  end %8.2f ms  grid %s'%(r['Kernel_Name'][:40],(int(r['Start_Timestamp'])-t0)/1e6,(int(r['End_Timestamp'])-t0)/1e6,r.get('Grid_Size_X', r.get('Grid_Size',''))))
