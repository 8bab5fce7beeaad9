import os, subprocess, sys, time
sys.path.insert(0, '/root/repo')
from slimfastq_amd import capi
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = '/tmp/sfq_e2e'; os.makedirs(d, exist_ok=True)
fq = capi.synth_fastq(4_000_000, 150, seed=2)
src, sfq, back = d + '/in.fq', d + '/out.sfq', d + '/back.fq'
open(src, 'wb').write(fq)
for rep in range(2):
    for name in ('slimfastq-amd.old', 'slimfastq-amd'):
        cli = os.path.join(root, 'slimfastq_amd', 'bin', name)
        t0 = time.time(); subprocess.check_call([cli, '-u', src, '-f', sfq, '-O']); t1 = time.time()
        subprocess.check_call([cli, '-d', '-f', sfq, '-u', back, '-O']); t2 = time.time()
        print('%-20s compress %.2f s (%.0f MB/s)  decompress %.2f s (%.0f MB/s)' % (name, t1 - t0, len(fq) / (t1 - t0) / 1e6, t2 - t1, len(fq) / (t2 - t1) / 1e6), flush=True)
for f in (src, sfq, back):
    os.remove(f)
