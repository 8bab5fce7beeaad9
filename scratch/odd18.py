import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from slimfastq_amd import capi
from oracle import oracle as O
import util
import test_frozen_tables as T
ctx = capi.Context(0)
seed = 18
rng = np.random.default_rng(seed)
n = int(rng.integers(300, 5000)); br = int(rng.integers(30, min(900, n) + 1)); cr = int(rng.integers(1, br + 1)); level = int(rng.integers(1, 5))
fq = T._odd_headers_fastq(n, seed)
want = b"".join(O.decompress(O.compress(c, level).image) for c in util.split_records(fq, br))
enc = ctx.encode_host(fq, level=level, block_reads=br, prior_step=1, tables=capi.TABLES_FROZEN, chain_reads=cr)
got = ctx.decode_host(enc, level=level, out_cap=2 * len(fq) + 4096)
print(len(got), len(want), len(fq))
gl, wl, fl = got.split(b"\n"), want.split(b"\n"), fq.split(b"\n")
k = 0
for i in range(min(len(gl), len(wl))):
    if gl[i] != wl[i]:
        print("line", i, "record", i // 4, "in block", (i // 4) // br, "at", (i // 4) % br)
        for j in range(max(0, i - 8), i + 5, 4):
            print("  src ", fl[j][:150]); print("  want", wl[j][:150]); print("  got ", gl[j][:150])
        k += 1
        if k >= 2: break
