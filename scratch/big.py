import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from slimfastq_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 13_500_000
fq = capi.synth_fastq(n, 150, seed=5)
nbytes = len(fq); print('bytes', nbytes, '> 2^32:', nbytes > 2**32, flush=True)
d_in = torch.from_numpy(np.frombuffer(fq, np.uint8).copy()).cuda(); del fq
ctx = capi.Context(0)
cap = capi.lib().sfq_encode_bound(nbytes)
d_out = torch.empty(cap, dtype=torch.uint8, device='cuda')
t0 = time.time()
res = ctx.encode_device(d_in.data_ptr(), nbytes, d_out.data_ptr(), cap, level=4, block_reads=1024, prior_step=capi.PRIOR_AUTO)
torch.cuda.synchronize(); print('encode %.3fs ratio %.3f blocks %d' % (time.time() - t0, nbytes / res.total_bytes, res.n_blocks), flush=True)
blocks = ctx.index(res.n_blocks); first = ctx.first_headers(res.first_hdr_bytes); prior = ctx.prior()
packed = d_out[:res.total_bytes].clone(); del d_out
d_back = torch.empty(nbytes + 4096, dtype=torch.uint8, device='cuda')
t0 = time.time()
got, _ = ctx.decode_device(blocks, first, packed.data_ptr(), list(res.stream_offset), d_back.data_ptr(), d_back.numel(), prior=prior, level=4)
torch.cuda.synchronize(); print('decode %.3fs' % (time.time() - t0), flush=True)
assert got == nbytes and torch.equal(d_back[:nbytes], d_in)
print('round trip ok at', nbytes, 'bytes')
