import sys, time, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from slimfastq_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
br = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
fq = capi.synth_fastq(n, 150, seed=1)
ctx = capi.Context(0)
t0 = time.time(); enc = ctx.encode_host(fq, level=3, block_reads=br, prior_step=capi.PRIOR_AUTO); t1 = time.time()
print('encode_host %.3fs (PCIe incl)  %d -> %d bytes, blocks %d' % (t1 - t0, len(fq), enc.archive_bytes, len(enc.blocks)))
L = capi.lib()
L.sfq_set_qlt_prior(ctx.handle, enc.prior, len(enc.prior))
d_streams = torch.from_numpy(enc.data).cuda()
d_out = torch.empty(len(fq) + 4096, dtype=torch.uint8, device='cuda')
soff = (C.c_uint64 * capi.NSTREAMS)(*list(enc.res.stream_offset))
p = capi.Params(3, 0, 0, 0, 0, 0, 0)
res = capi.Result(); nout = C.c_uint64()
fb = np.frombuffer(enc.first_hdrs, np.uint8)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    rc = L.sfq_decode_blocks(ctx.handle, C.byref(p), enc.blocks, len(enc.blocks), fb.ctypes.data_as(C.c_void_p), len(enc.first_hdrs),
                             C.c_void_p(d_streams.data_ptr()), soff, C.c_void_p(d_out.data_ptr()), d_out.numel(), C.byref(nout), C.byref(res))
    torch.cuda.synchronize(); dt = time.time() - t0
    assert rc == 0, L.sfq_last_error(ctx.handle)
    print('decode %.1f ms -> %.0f MB/s' % (dt * 1e3, len(fq) / dt / 1e6), [round(x, 1) for x in res.kernel_ms])
assert bytes(d_out[:nout.value].cpu().numpy()) == fq
print('roundtrip ok')
