"""format 6 (one block) encode / decode phase times: which stream bounds a single serial chain?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slimfastq_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
fq = capi.synth_fastq(n, 150, seed=1)
ctx = capi.Context(0)
t = torch.from_numpy(np.frombuffer(fq, np.uint8).copy()).cuda()
o = torch.empty(capi.lib().sfq_encode_bound(len(fq)), dtype=torch.uint8, device="cuda")
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = ctx.encode_device(t.data_ptr(), len(fq), o.data_ptr(), o.numel(), level=3, block_reads=0)
    torch.cuda.synchronize(); te = time.perf_counter() - t0
print("encode %.2f MB/s  phases ms: frame %.1f qlt %.1f gen %.1f rec %.1f usr %.1f pack %.1f total %.1f" % ((len(fq) / 1e6 / te,) + tuple(r.kernel_ms[i] for i in range(7))))
b, h = ctx.index(1), ctx.first_headers(r.first_hdr_bytes)
back = torch.empty(len(fq) + 4096, dtype=torch.uint8, device="cuda")
for kernel in (0, 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    got, rr = ctx.decode_device(b, h, o.data_ptr(), list(r.stream_offset), back.data_ptr(), back.numel(), level=3, kernel=kernel)
    torch.cuda.synchronize(); td = time.perf_counter() - t0
    ok = got == len(fq) and torch.equal(back[:len(fq)], t)
    print("decode kernel=%d %.2f MB/s ok=%s phases ms: usr %.1f qlt %.1f gen %.1f rec %.1f pack %.1f total %.1f" % (kernel, len(fq) / 1e6 / td, ok, rr.kernel_ms[4], rr.kernel_ms[1], rr.kernel_ms[2], rr.kernel_ms[3], rr.kernel_ms[5], rr.kernel_ms[6]))
