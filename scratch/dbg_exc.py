import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slimfastq_amd import capi
ctx = capi.Context(0)
def mk(nrec, npos):
    recs = []
    for i in range(nrec):
        seq = list("ACGT" * 10); q = list("I" * 40)
        for p in npos.get(i, []): seq[p] = "N"; q[p] = "#"
        recs.append("@r%d\n%s\n+\n%s\n" % (i, "".join(seq), "".join(q)))
    return "".join(recs).encode()
for name, fq in (("noexc", mk(50, {})), ("oneN", mk(50, {3: [5]})), ("twoN", mk(50, {3: [5], 7: [1, 30]})), ("many", mk(300, {i: [i % 40] for i in range(0, 300, 2)}))):
    for br in (capi.BLOCK_AUTO, 16):
        try:
            enc = ctx.encode_host(fq, level=3, block_reads=br, prior_step=capi.PRIOR_AUTO, tables=1)
            ok = ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq
            print(name, br, "ok" if ok else "MISMATCH", len(enc.stream("gen.Ns")), len(enc.stream("gen.Nn")), flush=True)
        except capi.SfqError as e:
            print(name, br, "ERR", str(e)[:80], flush=True)
