"""End-to-end (file to file, PCIe and page cache included) rate of the CLI on this box."""
import os, subprocess, sys, time
sys.path.insert(0, '/root/repo')
from slimfastq_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'slimfastq_amd', 'bin', 'slimfastq-amd')
d = '/tmp/sfq_e2e'; os.makedirs(d, exist_ok=True)
fq = capi.synth_fastq(n, 150, seed=2)
src, sfq, back = d + '/in.fq', d + '/out.sfq', d + '/back.fq'
open(src, 'wb').write(fq)
for label, cmd in (('compress', [cli, '-z', '-u', src, '-f', sfq, '-O']), ('compress again (tables allocated per process)', [cli, '-z', '-u', src, '-f', sfq, '-O']),
                   ('decompress', [cli, '-z', '-d', '-f', sfq, '-u', back, '-O'])):
    t0 = time.time(); subprocess.check_call(cmd); dt = time.time() - t0
    print('%-48s %6.2f s  %7.1f MB/s of FASTQ' % (label, dt, len(fq) / dt / 1e6), flush=True)
assert open(back, 'rb').read() == fq
print('sizes: fastq %d, sfq %d' % (len(fq), os.path.getsize(sfq)))
for f in (src, sfq, back):
    os.remove(f)
