import csv,sys
f=sys.argv[1]; which=sys.argv[2] if len(sys.argv)>2 else 'enc'; thr=float(sys.argv[3]) if len(sys.argv)>3 else 0.15
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
key='k_qlt_encode_c' if which=='enc' else 'k_qlt_decode_c'
start='k_frame' if which=='enc' else 'k_usr_fill'
idx=[i for i,r in enumerate(rows) if key in r['Kernel_Name']]
i=idx[-1]; j=i
while j>0 and start not in rows[j]['Kernel_Name']: j-=1
t0=int(rows[j]['Start_Timestamp'])
endk='k_compact_chains' if which=='enc' else 'k_assemble'
last=None
for r in rows[j:]:
    s=(int(r['Start_Timestamp'])-t0)/1e6; e=(int(r['End_Timestamp'])-t0)/1e6
    if e-s>thr: print('%-50s %8.2f -> %8.2f  (%.2f) grid %s'%(r['Kernel_Name'][:50],s,e,e-s,r.get('Grid_Size_X', r.get('Grid_Size',''))))
    if endk in r['Kernel_Name']: last=e
    if last and s>last+1: break
