import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slimfastq_amd import capi
ctx = capi.Context(0)
for level in (1, 2):
    fq = capi.synth_fastq(4100, 150, seed=30 + level)
    for br in (0, 512):
        for tables in (0, 1):
            for models in (0,):
                try:
                    enc = ctx.encode_host(fq, level=level, block_reads=br, tables=tables)
                    ok = ctx.decode_host(enc, level=level, out_cap=len(fq) + 4096) == fq
                    print(level, br, tables, "ok" if ok else "MISMATCH", flush=True)
                except capi.SfqError as e:
                    print(level, br, tables, "ERR", e, flush=True)
