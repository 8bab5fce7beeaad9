"""sizes of everything a decoder needs, by part: python scratch/parts.py <reads> [chain_reads ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slimfastq_amd import capi
n = int(sys.argv[1])
fq = capi.synth_fastq(n, 150, seed=17)
ctx = capi.Context(0)
for c in [int(x) for x in sys.argv[2:]] or [0]:
    for lvl in (3,):
        e = ctx.encode_host(fq, level=lvl, block_reads=capi.BLOCK_AUTO, prior_step=capi.PRIOR_AUTO, tables=capi.TABLES_FROZEN, chain_reads=c)
        sb = list(e.res.stream_bytes)
        print("reads %d c=%d l%d chains %d: streams %d (rec %d gen %d qlt %d) first %d qlt.pri %d chn.idx %d rec.pri %d blk %d  total %d" % (
            n, c, lvl, e.res.n_chains, e.res.total_bytes, sb[0], sb[1], sb[2], len(e.first_hdrs), len(e.prior), len(e.chains), len(e.rec_prior), 14 * len(e.blocks), e.archive_bytes))
