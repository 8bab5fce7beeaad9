"""What a rank's step costs in a multi-GPU job: rank 0's sfq_build_priors, then an encode from the given priors."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slimfastq_amd import capi
n = 10_000_000
fq = capi.synth_fastq(n, 150, seed=1)
nbytes = len(fq)
d_in = torch.from_numpy(np.frombuffer(fq, np.uint8).copy()).cuda()
ctx = capi.Context(0)
cap = capi.lib().sfq_encode_bound(nbytes)
d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pri, rp = ctx.build_priors(d_in.data_ptr(), nbytes, level=3, block_reads=1024, tables=1)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ctx.set_priors(pri, rp)
    res = ctx.encode_device(d_in.data_ptr(), nbytes, d_out.data_ptr(), cap, level=3, block_reads=1024, prior_step=capi.PRIOR_GIVEN, tables=1)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    res2 = ctx.encode_device(d_in.data_ptr(), nbytes, d_out.data_ptr(), cap, level=3, block_reads=1024, prior_step=capi.PRIOR_AUTO, tables=1)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print("build_priors %.2f ms, encode (given) %.2f ms, encode (own priors) %.2f ms; bytes %d vs %d" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, res.total_bytes, res2.total_bytes), flush=True)
