import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slimfastq_amd import capi
ctx = capi.Context(0)
def mk(lens, exc=True):
    rnd = random.Random(5); recs = []
    for i, n in enumerate(lens):
        seq = "".join(rnd.choice("ACGT") for _ in range(n))
        qual = "".join(chr(33 + min(60, max(1, int(rnd.gauss(30, 8))))) for _ in range(n))
        if exc and n > 1000:
            seq = seq[:500] + "N" * 7 + seq[507:]; qual = qual[:500] + "!" * 7 + qual[507:]
        recs.append("@long.%d ch=%d len=%d\n%s\n+\n%s\n" % (i + 1, 100 + i, n, seq, qual))
    return "".join(recs).encode()
for lens, exc in (((150, 60000, 200), False), ((150, 70000, 200), False), ((150, 70000, 200), True), ((70000,), False), ((150, 200, 300001), False)):
    fq = mk(lens, exc)
    for tables in (1, 0):
        for br in (capi.BLOCK_AUTO, 2):
            try:
                enc = ctx.encode_host(fq, level=3, block_reads=br, tables=tables)
                ok = ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq
                print(lens, exc, tables, br, "ok" if ok else "MISMATCH", flush=True)
            except capi.SfqError as e:
                print(lens, exc, tables, br, "ERR", str(e)[:90], flush=True)
