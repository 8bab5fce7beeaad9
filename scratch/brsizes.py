import sys
sys.path.insert(0, '/root/repo')
from slimfastq_amd import capi
fq = capi.synth_fastq(600000, 150, seed=1)
ctx = capi.Context(0)
base = None
for br in (4096, 1024, 512, 256, 128):
    enc = ctx.encode_host(fq, level=3, block_reads=br, prior_step=capi.PRIOR_AUTO)
    sb = list(enc.res.stream_bytes)
    extra = len(enc.first_hdrs) + len(enc.prior) + 14 * len(enc.blocks)
    tot = sum(sb) + extra
    if base is None: base = (sb, extra, tot)
    print('br %5d blocks %5d rec %9d gen %9d qlt %9d other %7d hdr+idx+prior %8d total %10d (+%.2f%% vs 4096)' % (
        br, len(enc.blocks), sb[0], sb[1], sb[2], sum(sb[3:]), extra, tot, 100.0 * (tot - base[2]) / base[2]))
