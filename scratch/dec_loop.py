"""decode of the default workload, N times in one process: ms of each call and the kernels' phases"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slimfastq_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lds = int(sys.argv[3]) if len(sys.argv) > 3 else 0
fq = capi.synth_fastq(n, 150, seed=1)
nbytes = len(fq)
d_in = torch.from_numpy(np.frombuffer(fq, np.uint8).copy()).cuda()
ctx = capi.Context(0)
cap = capi.lib().sfq_encode_bound(nbytes)
d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
res = ctx.encode_device(d_in.data_ptr(), nbytes, d_out.data_ptr(), cap, level=3, block_reads=1024, prior_step=capi.PRIOR_AUTO, tables=1)
blocks = ctx.index(res.n_blocks); first = ctx.first_headers(res.first_hdr_bytes)
prior, chains, rec_prior = ctx.prior(), ctx.chains(), ctx.rec_prior()
packed = d_out[:res.total_bytes].clone(); soff = list(res.stream_offset)
d_back = torch.empty(nbytes + 4096, dtype=torch.uint8, device="cuda")
out = []
for _ in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    got, r = ctx.decode_device(blocks, first, packed.data_ptr(), soff, d_back.data_ptr(), d_back.numel(), prior=prior, level=3, chains=chains, rec_prior=rec_prior, lds_rows=lds)
    torch.cuda.synchronize(); out.append(((time.perf_counter() - t0) * 1e3, r.kernel_ms[capi.T_QLT], r.kernel_ms[capi.T_REC], r.kernel_ms[capi.T_GEN]))
print("ok" if torch.equal(d_back[:nbytes], d_in) else "MISMATCH", " ".join("%.1f(q%.1f r%.1f g%.1f)" % o for o in out))
