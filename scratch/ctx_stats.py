"""Quality-context statistics of a FASTQ sample (level 3 contexts, qlts.hpp:62-74): how concentrated are the
symbols on the hot contexts, and how many distinct symbols does a context see?  (design input for LDS staging)"""
import sys, numpy as np
sys.path.insert(0, ".")
from slimfastq_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
L = 150
fq = capi.synth_fastq(n, L, seed=1, kind=kind)
lines = fq.split(b"\n")
q = np.frombuffer(b"".join(lines[3::4]), np.uint8).reshape(n, L).astype(np.int32) - 33
v1 = np.zeros_like(q); v1[:, 1:] = q[:, :-1]
v2 = np.zeros_like(q); v2[:, 2:] = q[:, :-2]
v3 = np.zeros_like(q); v3[:, 3:] = q[:, :-3]
drop = np.maximum(0, v1 - q)
inc = np.cumsum(drop, axis=1)
dprev = 5 + inc - drop
d3 = np.minimum(7, dprev >> 3)
ctx = (v1 | (np.maximum(v2, v3) << 6) | ((v2 == v3).astype(np.int32) << 12) | (d3 << 13)) & 0xFFFF
ctx[:, 0] = 0
pair = (ctx.astype(np.int64) << 6) | q
cnt = np.bincount(ctx.ravel(), minlength=65536)
order = np.argsort(-cnt)
tot = cnt.sum()
print("contexts touched:", (cnt > 0).sum(), "symbols:", tot)
cum = np.cumsum(cnt[order]) / tot
for k in (16, 32, 64, 128, 256, 512, 1024, 2048, 4096):
    print("top %5d contexts cover %.4f" % (k, cum[k - 1]))
pc = np.bincount(pair.ravel(), minlength=65536 * 64).reshape(65536, 64)
nd = (pc > 0).sum(axis=1)
for k in (64, 256, 1024, 4096):
    sel = order[:k]
    print("top %5d: distinct symbols per context: mean %.1f max %d; slots needed to cover 99%% of a context's hits: mean %.1f max %d" % (
        k, nd[sel].mean(), nd[sel].max(),
        np.mean([np.searchsorted(np.cumsum(np.sort(pc[c])[::-1]) / cnt[c], 0.99) + 1 for c in sel]),
        np.max([np.searchsorted(np.cumsum(np.sort(pc[c])[::-1]) / cnt[c], 0.99) + 1 for c in sel])))
# per block of 1024 reads: contexts touched, and the share of the block's symbols in the globally hottest K
for br in (128, 256, 1024):
    nb = n // br
    t = []; share = {256: [], 512: [], 1024: []}
    rank = np.empty(65536, np.int64); rank[order] = np.arange(65536)
    for b in range(min(nb, 20)):
        c = ctx[b * br:(b + 1) * br].ravel()
        t.append(len(np.unique(c)))
        for K in share: share[K].append((rank[c] < K).mean())
    print("block of %d reads: contexts touched mean %.0f; share of symbols in global top-K:" % (br, np.mean(t)), {K: round(float(np.mean(v)), 4) for K, v in share.items()})
# run statistics within 64-symbol windows (what the symbol-parallel kernel sees)
