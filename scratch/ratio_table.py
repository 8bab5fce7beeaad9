"""profiles/r05_ratio_table.json (rounds 3 and 5): compressed size of the block format (frozen tables, the default) over the reference's own,
per level and workload -- "ratio within 1 % of the reference at each -l level" (BASELINE.json north_star).
The reference side is the oracle (stream-identical to the compiled reference: tests/test_oracle.py) on the same text;
ours counts everything a decoder needs (streams, first headers, priors, chain and block index).

    python scratch/ratio_table.py > profiles/r05_ratio_table.json        (on the GPU box)
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from slimfastq_amd import capi
from oracle import oracle as O
import util

WORK = [("default: 600k x 150 bp Illumina-style reads", dict(n=600_000, kind=0)),
        ("binned: 600k x 150 bp reads, 4-level qualities", dict(n=600_000, kind=2)),
        ("genome: 2M x 150 bp reads sampled from a 10 Mbp genome (30x)", dict(n=2_000_000, kind=3)),
        ("long: 6000 reads of 10-50 kb", dict(n=6_000, kind=1))]
if "--quick" in sys.argv:
    WORK = [(w, dict(n=max(2000, k["n"] // 20), kind=k["kind"])) for w, k in WORK]
ctx = capi.Context(0)
rows = []
for what, k in WORK:
    fq = capi.synth_fastq(k["n"], 150, seed=1, kind=k["kind"])
    for level in (1, 2, 3, 4):
        t0 = time.perf_counter()
        ref = O.compress(fq, level)
        t_ref = time.perf_counter() - t0
        ref_bytes = ref.payload_bytes() - len(ref.streams["<info>"])
        enc = ctx.encode_host(fq, level=level, block_reads=capi.BLOCK_AUTO, prior_step=capi.PRIOR_AUTO, tables=capi.TABLES_FROZEN)
        assert ctx.decode_host(enc, level=level, out_cap=len(fq) + 4096) == fq
        rows.append({"workload": what, "level": level, "raw_bytes": len(fq), "reference_stream_bytes": ref_bytes, "ours_bytes": enc.archive_bytes,
                     "ours_over_reference": round(enc.archive_bytes / ref_bytes, 4), "ratio_ours": round(len(fq) / enc.archive_bytes, 4),
                     "ratio_reference": round(len(fq) / ref_bytes, 4), "reference_cpu_MBps": round(len(fq) / 1e6 / t_ref, 1), "round_trip_identical": True})
        print(rows[-1], file=sys.stderr, flush=True)
    del fq
fq = util.golden_fastq("tst7")                      # the reference's largest sample (SOLiD colour space, 21000 records)
for level in (1, 2, 3, 4):
    ref = O.compress(fq, level)
    ref_bytes = ref.payload_bytes() - len(ref.streams["<info>"])
    enc = ctx.encode_host(fq, level=level, block_reads=capi.BLOCK_AUTO, prior_step=capi.PRIOR_AUTO, tables=capi.TABLES_FROZEN)
    assert ctx.decode_host(enc, level=level, out_cap=len(fq) + 4096) == fq
    auto = ctx.encode_host(fq, level=level, block_reads=capi.BLOCK_AUTO, prior_step=capi.PRIOR_AUTO, tables=capi.TABLES_AUTO)
    assert ctx.decode_host(auto, level=level, out_cap=len(fq) + 4096) == fq
    rows.append({"workload": "samples/tst7.fq (21000 SOLiD records, 3.9 MB): frozen tables forced -- the priors' fixed cost weighs on a file this small; "
                             "the default (SFQ_TABLES_AUTO) takes adaptive tables below 64 MiB", "level": level, "raw_bytes": len(fq),
                 "reference_stream_bytes": ref_bytes, "ours_bytes": enc.archive_bytes, "ours_over_reference": round(enc.archive_bytes / ref_bytes, 4),
                 "default_tables_auto_bytes": auto.archive_bytes, "default_tables_auto_over_reference": round(auto.archive_bytes / ref_bytes, 4),
                 "ratio_ours": round(len(fq) / enc.archive_bytes, 4), "ratio_reference": round(len(fq) / ref_bytes, 4), "round_trip_identical": True})
    print(rows[-1], file=sys.stderr, flush=True)
print(json.dumps({"what": "block format (frozen tables, automatic blocks / chains / priors) over the reference, bytes a decoder needs / reference stream bytes",
                  "rows": rows}, indent=1))
