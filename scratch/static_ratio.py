"""Design experiment: what does the quality stream cost if every chain codes with FROZEN rows built from a count
table (the transmitted prior, or counts of earlier generations) instead of adapting per symbol?  Ideal code
lengths (-log2 p) against the reference's actual qlt stream bytes on the same synthetic reads."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from slimfastq_amd import capi
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
L = 150
fq = capi.synth_fastq(n, L, seed=1, kind=kind)
t0 = time.time(); ref = O.compress(fq, 3).streams; print("oracle %.1fs" % (time.time() - t0))
ref_q = len(ref["qlt"]); ref_g = len(ref["gen"])
lines = fq.split(b"\n")
q = np.frombuffer(b"".join(lines[3::4]), np.uint8).reshape(n, L).astype(np.int32) - 33
v1 = np.zeros_like(q); v1[:, 1:] = q[:, :-1]
v2 = np.zeros_like(q); v2[:, 2:] = q[:, :-2]
v3 = np.zeros_like(q); v3[:, 3:] = q[:, :-3]
drop = np.maximum(0, v1 - q); inc = np.cumsum(drop, axis=1); d3 = np.minimum(7, (5 + inc - drop) >> 3)
ctx = (v1 | (np.maximum(v2, v3) << 6) | ((v2 == v3).astype(np.int32) << 12) | (d3 << 13)) & 0xFFFF
ctx[:, 0] = 0
pair = (ctx.astype(np.int64) << 6) | q

def rows_from_counts(cnt):          # the prior rule (prior.hip): freq = 6*count >> s, largest <= 32000
    cnt = cnt.reshape(65536, 64).astype(np.int64)
    mx = cnt.max(axis=1)
    sh = np.zeros(65536, np.int64)
    while True:
        over = ((mx * 6) >> sh) > 32000
        if not over.any(): break
        sh[over] += 1
    f = (cnt * 6) >> sh[:, None]
    return f, f.sum(axis=1)

def cost_bits(pairs, f, tot):
    c = pairs >> 6; s = pairs & 63
    p = (f[c, s] + 1) / (tot[c] + 64)
    return -np.log2(p).sum()

allp = pair.ravel()
print("reads %d  symbols %d  reference qlt stream %d B (%.4f bit/sym)" % (n, allp.size, ref_q, ref_q * 8 / allp.size))
full = np.bincount(allp, minlength=65536 * 64)
f, tot = rows_from_counts(full)
b = cost_bits(allp, f, tot)
print("frozen rows from FULL counts:            %.0f B  = %.4f x reference" % (b / 8, b / 8 / ref_q))
for step in (2, 8, 32):
    samp = np.bincount(pair[::step].ravel(), minlength=65536 * 64)
    f, tot = rows_from_counts(samp)
    b = cost_bits(allp, f, tot)
    nz = (samp > 0).sum()
    print("frozen rows from every %2d-th record:    %.0f B  = %.4f x reference   (prior: %d nonzero entries)" % (step, b / 8, b / 8 / ref_q, nz))
# generations: frozen rows from everything before the generation (gen 0 from a thin sample)
for g0 in (64, 256):
    bounds = [0]; x = max(1, n // g0)
    while bounds[-1] < n: bounds.append(min(n, max(x, bounds[-1] * 2) if bounds[-1] else x))
    tot_bits = 0.0
    samp = np.bincount(pair[::64].ravel(), minlength=65536 * 64)
    acc = np.zeros(65536 * 64, np.int64)
    for gi in range(len(bounds) - 1):
        lo, hi = bounds[gi], bounds[gi + 1]
        f, tot = rows_from_counts(acc if gi else samp)
        seg = pair[lo:hi].ravel()
        tot_bits += cost_bits(seg, f, tot)
        acc += np.bincount(seg, minlength=65536 * 64)
    print("generations (first = n/%d, doubling; %d of them), gen 0 from a 1/64 sample: %.0f B = %.4f x reference" % (g0, len(bounds) - 1, tot_bits / 8, tot_bits / 8 / ref_q))
