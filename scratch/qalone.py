"""quality decoder alone (scratch/libsfq_qalone.so): python scratch/qalone.py [reads] [dec_lds_rows...]"""
import sys, os, shutil
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np, torch
from slimfastq_amd import capi
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
fq = capi.synth_fastq(reads, 150, seed=1)
d = torch.from_numpy(np.frombuffer(fq, np.uint8).copy()).cuda(); n = len(fq); del fq
ctx = capi.Context(0)
cap = capi.lib().sfq_encode_bound(n)
out = torch.empty(cap, dtype=torch.uint8, device="cuda"); back = torch.empty(n + 4096, dtype=torch.uint8, device="cuda")
for cr in (0, 24):
    r = ctx.encode_device(d.data_ptr(), n, out.data_ptr(), cap, level=3, block_reads=1024, prior_step=capi.PRIOR_AUTO, tables=1, chain_reads=cr)
    blocks = ctx.index(r.n_blocks); first = ctx.first_headers(r.first_hdr_bytes); prior, chains, rp = ctx.prior(), ctx.chains(), ctx.rec_prior()
    for rows in [int(x) for x in sys.argv[2:]] or [0]:
        for _ in range(3):
            try:
                sys.stderr.write("chains %d lds_rows %d: " % (r.n_chains, rows)); sys.stderr.flush()
                ctx.decode_device(blocks, first, out.data_ptr(), list(r.stream_offset), back.data_ptr(), back.numel(), prior=prior, level=3, chains=chains, rec_prior=rp, lds_rows=rows)
            except capi.SfqError:
                pass
