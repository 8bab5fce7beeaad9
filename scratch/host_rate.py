"""PCIe-inclusive rate of the host-buffer entry point (sfq_encode_blocks_host): the text starts in host memory and the
streams end there.  Pageable numpy buffers, then page-locked ones (sfq_host_alloc), whole call and in slabs of 512 MiB."""
import sys, os, time, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slimfastq_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
fq = capi.synth_fastq(n, 150, seed=1)
nbytes = len(fq)
L = capi.lib()
ctx = capi.Context(0)
cap = L.sfq_encode_bound(nbytes)
p = capi.Params(3, 1024, 0, 0, 0, 0, capi.PRIOR_AUTO, capi.TABLES_FROZEN, 0, 0)
res = capi.Result()
def run(src_ptr, dst_ptr, what):
    ts = []
    for it in range(4):
        t0 = time.perf_counter()
        rc = L.sfq_encode_blocks_host(ctx._h, C.c_void_p(src_ptr), nbytes, C.byref(p), C.c_void_p(dst_ptr), cap, C.byref(res))
        assert rc == 0, rc
        ts.append(time.perf_counter() - t0)
    print("%-28s %7.1f ms  %6.2f GB/s of text (best of 3 after a warm-up; ratio %.4f)" % (what, min(ts[1:]) * 1e3, nbytes / min(ts[1:]) / 1e9, nbytes / res.total_bytes), flush=True)
src = np.frombuffer(fq, np.uint8); out = np.empty(cap, np.uint8)
run(src.ctypes.data, out.ctypes.data, "pageable in, pageable out")
L.sfq_host_alloc.restype = C.c_void_p
hp_in = L.sfq_host_alloc(ctx._h, nbytes); hp_out = L.sfq_host_alloc(ctx._h, cap)
assert hp_in and hp_out
C.memmove(hp_in, src.ctypes.data, nbytes)
run(hp_in, hp_out, "page-locked in and out")
