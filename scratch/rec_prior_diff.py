"""which odd header makes the GPU's header prior (rec.pri) differ from the oracle's?"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from slimfastq_amd import capi
from oracle import oracle as O
import util
def gen(n, seed, kinds):
    rng = np.random.default_rng(seed); out = []; x = 1000
    for i in range(n):
        x += int(rng.integers(0, 50)); k = rng.random()
        hdr = None
        if k < 0.02 and "long" in kinds: hdr = "@long.%d %s:%d" % (i, "Z" * int(rng.integers(120, 300)), x)
        elif 0.02 <= k < 0.04 and "many" in kinds: hdr = "@many.%d " % i + ":".join(str(int(v)) for v in rng.integers(0, 99, int(rng.integers(17, 40))))
        elif 0.04 <= k < 0.06 and "big" in kinds: hdr = "@big.%d run:%d:%d" % (i, 18446744073709551000 + int(rng.integers(0, 600)), x)
        elif 0.06 <= k < 0.08 and "hex" in kinds: hdr = "@hex.%d run:%x:%d" % (i, 0xabc000 + i, x)
        elif 0.08 <= k < 0.10 and "zero" in kinds: hdr = "@zero.%d run:0%d::%d" % (i, i, x)
        if hdr is None: hdr = "@SIM.%d M7:12:FC9:%d:%d:%d:%d 1:N:0:ACGT" % (i, 1 + i // 2000, 1100 + i // 500, x, int(rng.integers(1000, 30000)))
        seq = "".join("ACGT"[int(v)] for v in rng.integers(0, 4, 60)); q = "".join(chr(33 + int(v)) for v in rng.integers(2, 41, 60))
        out += [hdr, seq, "+", q]
    return ("\n".join(out) + "\n").encode()
ctx = capi.Context(0)
fq = gen(6000, 5, ["big", "zero"])
lines = fq.split(b"\n")[:-1]
recs = [lines[i:i + 4] for i in range(0, len(lines), 4)]
pre = recs[:246 * 18]
a, b = pre[-36:-18], pre[-18:]
sub = b"\n".join(b"\n".join(r) for r in (a + b)) + b"\n"
enc = ctx.encode_host(sub, level=3, block_reads=400, prior_step=1, tables=capi.TABLES_FROZEN, chain_reads=50)
starts, lens = util.line_table(sub)
hoff, hlen = starts[0::4] + 1, lens[0::4] - 1
f = O.rec_prior_freqs(O.rec_count(sub, hoff, hlen, 18, 18, 2)); g = util.unpack_rec_prior(enc.rec_prior)
bad = np.flatnonzero(g != f)
print("two runs differ:", len(bad), [(int(x) // 256, int(x) % 256, int(g[x]), int(f[x])) for x in bad[:20]])
fq = gen(6000, 5, ["long", "many", "big", "hex", "zero"])
enc = ctx.encode_host(fq, level=3, block_reads=400, prior_step=1, tables=capi.TABLES_FROZEN, chain_reads=50)
starts, lens = util.line_table(fq)
hoff, hlen = starts[0::4] + 1, lens[0::4] - 1
f = O.rec_prior_freqs(O.rec_count(fq, hoff, hlen, 18, 18, 333)); g = util.unpack_rec_prior(enc.rec_prior)
print("all kinds differ:", int((g != f).sum()))
