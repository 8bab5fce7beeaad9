"""Round 5 design study (CPU, numpy): what the base stream of genome-sampled reads costs under
  (a) today's rule (rows f = 3 + 4 n halved to <= 255, per generation) at 20 / 22 / 24 context bits,
  (b) rows quantised to a one-byte codebook index.
Cost = sum of -log2 p (the coder's overhead is not in it).  usage: gen_codebook_study.py [reads] [kind]"""
import sys, time, numpy as np
sys.path.insert(0, ".")
from slimfastq_amd import capi
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = 150
fq = capi.synth_fastq(n, L, seed=1, kind=kind)
t0 = time.time(); ref = O.compress(fq, 3).streams; print("oracle %.1fs" % (time.time() - t0), flush=True)
ref_g = len(ref["gen"])
lines = fq.split(b"\n")
g = np.frombuffer(b"".join(lines[1::4]), np.uint8).reshape(n, L)
lut = np.zeros(256, np.uint32); lut[ord('C')] = 1; lut[ord('G')] = 2; lut[ord('T')] = 3
code = lut[g]
del fq, lines
ctx32 = np.empty((n, L), np.uint32)
last = np.full(n, 0x007616c7, np.uint32)
for i in range(L):
    ctx32[:, i] = last
    last = (last << np.uint32(2)) | code[:, i]
print("reads %d bases %d reference gen %d B (%.4f bit/base)" % (n, n * L, ref_g, ref_g * 8 / (n * L)), flush=True)

def gen_bounds(nb):
    # kernels.h gen_bounds: first generation 1/64 of the blocks, each next as long as all before
    b = [0, max(1, nb // 64)]
    while b[-1] < nb: b.append(min(nb, b[-1] * 2))
    return b

def rows_exact(cnt):
    f = 3 + 4 * cnt.reshape(-1, 4).astype(np.int64)
    while True:
        big = f.max(axis=1) > 255
        if not big.any(): break
        f[big] = (f[big] >> 1) | (f[big] & 1)
    return f

# one-byte codebook: order of the two largest counts (top 2 bits, second 2 bits) and a 4-bit level
# level tables: (f_top, f_second, f_other) out of a total of 256
def make_levels():
    lv = []
    # (p_top, p_second) pairs: flat, mild ... sharp; and two-way splits
    for pt, ps in ((0.25, 0.25), (0.40, 0.25), (0.55, 0.20), (0.70, 0.14), (0.82, 0.09), (0.90, 0.05), (0.95, 0.025), (0.975, 0.0125),
                   (0.9875, 0.006), (0.993, 0.003), (0.45, 0.45), (0.62, 0.33), (0.75, 0.22), (0.85, 0.13), (0.33, 0.33), (0.996, 0.002)):
        ft = int(round(pt * 256)); fs = max(1, int(round(ps * 256)))
        fo = (256 - ft - fs) // 2
        if fo < 1: fo = 1; ft = 256 - fs - 2
        ft = 256 - fs - 2 * fo
        lv.append((ft, fs, fo))
    return np.array(lv, np.int64)
LV = make_levels()

def rows_codebook(cnt):
    c = cnt.reshape(-1, 4).astype(np.int64)
    nctx = c.shape[0]
    order = np.argsort(-c, axis=1, kind="stable")
    top, sec = order[:, 0], order[:, 1]
    tot = c.sum(axis=1)
    ct = c[np.arange(nctx), top]; cs = c[np.arange(nctx), sec]
    # smoothed estimates as the exact rule would give: (3 + 4 n) / (12 + 4 N)
    pt = (3 + 4 * ct) / (12 + 4 * tot); ps = (3 + 4 * cs) / (12 + 4 * tot)
    po = np.maximum((1 - pt - ps) / 2, 1e-9)
    # pick the level minimising the expected cost under the smoothed estimate
    best = np.zeros(nctx, np.int64); bc = np.full(nctx, 1e30)
    for k, (ft, fs, fo) in enumerate(LV):
        e = -(pt * np.log2(ft / 256) + ps * np.log2(fs / 256) + 2 * po * np.log2(fo / 256))
        m = e < bc; best[m] = k; bc[m] = e[m]
    f = np.empty((nctx, 4), np.int64)
    f[:] = LV[best, 2][:, None]
    f[np.arange(nctx), top] = LV[best, 0]
    f[np.arange(nctx), sec] = LV[best, 1]
    return f

def cost(keys, f):
    c = keys >> 2; s = keys & 3
    return -np.log2(f[c, s] / f[c].sum(axis=1)).sum()

br = 1024
nb = (n + br - 1) // br
bounds = [min(n, b * br) for b in gen_bounds(nb)]
for bits in (24, 22, 20):
    mask = np.uint32((1 << bits) - 1)
    key = ((ctx32 & mask) << np.uint32(2)) | code
    for name, rows in (("exact", rows_exact), ("codebook", rows_codebook)):
        acc = np.zeros(4 << bits, np.int64)
        tot = 0.0; per = []
        for gi in range(len(bounds) - 1):
            seg = key[bounds[gi]:bounds[gi + 1]].ravel().astype(np.int64)
            if gi < 2: b = 2.0 * seg.size
            else: b = cost(seg, rows(acc))
            per.append(b / seg.size); tot += b
            if gi + 2 < len(bounds): acc += np.bincount(seg, minlength=4 << bits)
        print("bits %d %-8s: %.0f B = %.4f x reference; bit/base per generation: %s" % (bits, name, tot / 8, tot / 8 / ref_g, " ".join("%.3f" % x for x in per)), flush=True)
