import sys
sys.path.insert(0, '/root/repo')
from slimfastq_amd import capi
ctx = capi.Context(0)
for n, br, ps in ((1000000, 1024, 2), (1000000, 1024, capi.PRIOR_AUTO), (300000, 1024, 1), (1000000, 1024, 7)):
    fq = capi.synth_fastq(n, 150, seed=1)
    enc = ctx.encode_host(fq, level=3, block_reads=br, prior_step=ps)
    try:
        ok = ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq
    except Exception as e:
        ok = str(e)[:90]
    print(n, br, ps, ok, len(enc.prior), flush=True)
