import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'], d['phase_ms'], d['roofline']['coder_ms'])
for r in d.get('size_sweep',[]):
    print('  %8d reads %6.2f GB chains %7d enc %8.1f MB/s %7.3f ms (dev %6.3f; q %5.2f g %5.2f r %5.2f) dec %8.1f MB/s %7.3f ms %s ratio %.4f ok=%s'%(r['reads'],r['raw_bytes']/1e9,r['chains'],r['encode_MBps'],r['encode_ms'],r['device_ms'],r['coder_ms']['qlt'],r['coder_ms']['gen'],r['coder_ms']['rec'],r.get('decode_MBps',0),r.get('decode_ms',0),r.get('decode_phase_ms'),r['ratio'],r.get('round_trip_identical')))
for k in ('decode','adaptive_tables','format6','genome_sampled','ratio_vs_reference','cpu_baseline'):
    if k in d: print(k, d[k])
