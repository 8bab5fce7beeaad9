import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slimfastq_amd import capi
ctx = capi.Context(0)
recs = []
for i in range(50):
    seq = list("ACGT" * 10); q = list("I" * 40)
    if i == 3: seq[5] = "N"; q[5] = "#"
    recs.append("@r%d\n%s\n+\n%s\n" % (i, "".join(seq), "".join(q)))
fq = "".join(recs).encode()
enc = ctx.encode_host(fq, level=3, block_reads=capi.BLOCK_AUTO, prior_step=capi.PRIOR_AUTO, tables=1)
print("gen.Ns", enc.stream("gen.Ns").hex(), flush=True)
try:
    print(ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq)
except capi.SfqError as e:
    print("ERR", e)
