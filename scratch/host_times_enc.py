"""host-side marks of an encode (SFQ_HOST_TIMING): python scratch/host_times_enc.py [reads]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from slimfastq_amd import capi
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
fq = capi.synth_fastq(reads, 150, seed=1)
d = torch.from_numpy(np.frombuffer(fq, np.uint8).copy()).cuda(); n = len(fq); del fq
ctx = capi.Context(0)
cap = capi.lib().sfq_encode_bound(n)
out = torch.empty(cap, dtype=torch.uint8, device="cuda")
for i in range(5):
    if i == 4: os.environ["SFQ_HOST_TIMING"] = "1"
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = ctx.encode_device(d.data_ptr(), n, out.data_ptr(), cap, level=3, block_reads=1024, prior_step=capi.PRIOR_AUTO, tables=1)
    torch.cuda.synchronize(); print("encode %.3f ms  device %.3f" % ((time.perf_counter() - t0) * 1e3, r.kernel_ms[6]), file=sys.stderr)
