"""Design experiment: base stream cost with generation-frozen Base2 rows (rows built from the counts of all earlier
generations, frozen inside a generation) against the reference's actual gen stream, on genome-sampled and iid reads."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from slimfastq_amd import capi
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 3
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 24
L = 150
fq = capi.synth_fastq(n, L, seed=1, kind=kind)
t0 = time.time(); ref = O.compress(fq, 3).streams; print("oracle %.1fs" % (time.time() - t0))
ref_g = len(ref["gen"])
lines = fq.split(b"\n")
g = np.frombuffer(b"".join(lines[1::4]), np.uint8).reshape(n, L)
lut = np.zeros(256, np.int64); lut[ord('C')] = 1; lut[ord('G')] = 2; lut[ord('T')] = 3
code = lut[g]
del fq, lines
ctx = np.empty((n, L), np.int64)
last = np.full(n, 0x007616c7, np.int64)
mask = (1 << bits) - 1
for i in range(L):
    ctx[:, i] = last & mask
    last = ((last << 2) | code[:, i]) & 0xFFFFFFFF
key = (ctx << 2) | code
print("reads %d bases %d reference gen %d B (%.4f bit/base)" % (n, n * L, ref_g, ref_g * 8 / (n * L)))

STEP = 1
def rows(cnt):          # Base2 rows from counts: f = 3 + STEP * n, the reference's normalize (halve, keep LSB) while any > 255
    f = 3 + STEP * cnt.reshape(-1, 4).astype(np.int64)
    while True:
        big = f.max(axis=1) > 255
        if not big.any(): break
        f[big] = (f[big] >> 1) | (f[big] & 1)
    return f

def cost(keys, f):
    c = keys >> 2; s = keys & 3
    return -np.log2(f[c, s] / f[c].sum(axis=1)).sum()

for first_div, ratio, STEP in ((64, 2.0, 2), (64, 2.0, 3), (64, 2.0, 4), (64, 2.0, 6), (256, 1.5, 3)):
    bounds = [0, max(1, n // first_div)]
    while bounds[-1] < n: bounds.append(min(n, int(bounds[-1] * ratio) + 1))
    acc = np.zeros(4 << bits, np.int64)
    tot = 0.0
    per = []
    for gi in range(len(bounds) - 1):
        lo, hi = bounds[gi], bounds[gi + 1]
        seg = key[lo:hi].ravel()
        b = cost(seg, rows(acc))
        per.append(b / seg.size)
        tot += b
        if gi + 2 < len(bounds): acc += np.bincount(seg, minlength=4 << bits)
    print("STEP %d generations first=n/%d x%.1f (%d): %.0f B = %.4f x reference; bit/base per generation: %s" % (
        STEP, first_div, ratio, len(bounds) - 1, tot / 8, tot / 8 / ref_g, " ".join("%.3f" % x for x in per)))
