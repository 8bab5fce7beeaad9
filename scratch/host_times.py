"""host-side marks of a decode (SFQ_HOST_TIMING): python scratch/host_times.py [reads]"""
import sys, os
os.environ["SFQ_HOST_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from slimfastq_amd import capi
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
fq = capi.synth_fastq(reads, 150, seed=1)
d = torch.from_numpy(np.frombuffer(fq, np.uint8).copy()).cuda(); n = len(fq); del fq
ctx = capi.Context(0)
cap = capi.lib().sfq_encode_bound(n)
out = torch.empty(cap, dtype=torch.uint8, device="cuda"); back = torch.empty(n + 4096, dtype=torch.uint8, device="cuda")
os.environ.pop("SFQ_HOST_TIMING")
r = ctx.encode_device(d.data_ptr(), n, out.data_ptr(), cap, level=3, block_reads=1024, prior_step=capi.PRIOR_AUTO, tables=1)
blocks = ctx.index(r.n_blocks); first = ctx.first_headers(r.first_hdr_bytes); prior, chains, rp = ctx.prior(), ctx.chains(), ctx.rec_prior()
for i in range(4):
    if i == 3: os.environ["SFQ_HOST_TIMING"] = "1"
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.decode_device(blocks, first, out.data_ptr(), list(r.stream_offset), back.data_ptr(), back.numel(), prior=prior, level=3, chains=chains, rec_prior=rp)
    torch.cuda.synchronize(); print("decode %.3f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
