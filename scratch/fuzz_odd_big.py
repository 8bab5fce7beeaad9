"""one-off: the odd-header generator at a size where thousands of header chains mix fast-path and handed-over chains"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from slimfastq_amd import capi
import util
import test_frozen_tables as T
ctx = capi.Context(0)
for n, br, cr, level, seed in ((150000, 1024, 64, 3, 101), (90000, 333, 111, 2, 102)):
    t0 = time.time()
    fq = T._odd_headers_fastq(n, seed)
    enc = T.check_against_oracle(ctx, fq, level, br=br, cr=cr, step=1, what="big odd")
    ci = util.unpack_chains(enc.chains)
    want = util.reference_restoration(fq, br, level, ci["rec_chain_reads"])
    got = ctx.decode_host(enc, level=level, out_cap=2 * len(fq) + 4096)
    print(n, br, cr, level, "header chains", len(ci["rec"]), "equal:", got == want, "lossy lines:", sum(1 for a, b in zip(fq.split(b"\n"), want.split(b"\n")) if a != b), "%.0f s" % (time.time() - t0), flush=True)
    assert got == want
