"""one-off: more seeds / geometries of the odd-header test"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from slimfastq_amd import capi
from oracle import oracle as O
import util
import test_frozen_tables as T
ctx = capi.Context(0)
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(300, 5000)); br = int(rng.integers(30, min(900, n) + 1)); cr = int(rng.integers(1, br + 1)); level = int(rng.integers(1, 5))
    fq = T._odd_headers_fastq(n, seed)
    try:
        enc = T.check_against_oracle(ctx, fq, level, br=br, cr=cr, step=1, what="odd seed %d" % seed)
        want = util.reference_restoration(fq, br, level, util.unpack_chains(enc.chains)["rec_chain_reads"])
        assert ctx.decode_host(enc, level=level, out_cap=2 * len(fq) + 4096) == want
        print("seed", seed, n, br, cr, level, "ok", flush=True)
    except AssertionError as e:
        bad += 1
        import traceback; tb = traceback.format_exc().splitlines()
        print("seed", seed, n, br, cr, level, "FAILED", str(e)[:300].replace("\n", " | "), " @ ", [l.strip() for l in tb if "line" in l][-2:], flush=True)
print("failures:", bad)
