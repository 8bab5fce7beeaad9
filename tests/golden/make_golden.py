#!/usr/bin/env python3
"""Regenerate tests/golden/ -- run in the build container only (needs /root/reference and oracle/_ref).

For every fixture and level 1..4 the COMPILED REFERENCE (oracle/_ref/slimfastq_ref, built from
/root/reference by oracle/Makefile) compresses the FASTQ from stdin; the per-stream bytes, the info
page and the reference's own decode of that archive are stored.  Fixtures are data only:
  * a few of the reference's sample inputs (samples/*.fq, the data files its `make test` uses),
  * edge-case FASTQs written here (cases the samples miss: SURVEY.md section 4 / 8c).

    python tests/golden/make_golden.py [name ...]
"""
import gzip
import hashlib
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

SAMPLES = ["small", "tstb", "tstc", "tstd", "badsprintf", "badqlt", "solid", "tsta", "tst1", "tst3", "fast5.to"]
INPUT_ONLY = ["tst7"]          # the reference's largest sample: kept as an input (ratio measurements), no reference streams stored


def edge_cases():
    out = {}
    # qualities >= 63 (escape row qlts.cpp:80-86), incl. 0x7e
    out["edge_hiq"] = (b"@r1 1\nACGTACGTAC\n+\nIII``~~II}\n@r2 2\nACGTNCGTAC\n+\n~~~~#~~~`I\n"
                       b"@r3 3\nTTGTACGTAC\n+\nIIIIIIIIII\n")
    # N rules: N with '!' (implicit), N with '#', real base with '!', '.' is NOT mixed with N (would croak)
    out["edge_n"] = (b"@n.1\nACGNNACGTA\n+\nII!!#III!I\n@n.2\nNNNNNACGTA\n+\n!!!!!IIIII\n"
                     b"@n.3\nACGTAACGTN\n+\n!IIIIIIII#\n@n.4\nACGTAACGTA\n+\nIIIIIIIIII\n")
    # lowercase bases: accepted, decoded as uppercase (gens.cpp:73-77 vs 171-178) -> lossy in the reference
    out["edge_lower"] = b"@l.1\nacgtACGTac\n+\nIIIIIIIIII\n@l.2\nACGTacgtAC\n+\nIIIIIIIIII\n"
    # header fields: decimal up/down, leading zero, hex lower/upper, hex with leading zero, two zeros,
    # over-long number, shape change (rec.x), field that becomes empty (decodes as 0: SURVEY H7 i)
    hdrs = [b"a:10:00ff:AB12:x9 7", b"a:12:0100:AB13:x9 7", b"a:9:00fe:AB0F:x10 7", b"a:009:fe:ab0f:x10 08",
            b"a:0:1fe:ab10:y1 09", b"a:18446744073709551615:1ff:AB10:y1 10", b"a:18446744073709551616:200:ab:y 011",
            b"a:7:0200:ab:y 0", b"a:7:0201:ab/y 0", b"a:8:0201:ab/y 1", b"b:8:0201:ab/y 1", b"b:8::ab/y 1",
            b"b:8:5:ab/y 1", b"b:123456789012345678901234567890:5:abcdef0123456789a/y 1"]
    out["edge_hdr"] = b"".join(b"@" + h + b"\nACGTACGTAC\n+\nIIIIIIIIII\n" for h in hdrs)
    # line-length exceptions: llen change (usr.x), qlen != llen (usr.x.q), second id (usr.2id)
    out["edge_len"] = (b"@v.1 x\nACGTACGTAC\n+v.1 x\nIIIIIIIIII\n@v.2 x\nACGTACG\n+v.2 x\nIIIIIII\n"
                       b"@v.3 x\nACGTACG\n+v.3 x\nIIIII\n@v.4 x\nACGTACGTACGT\n+v.4 x\nIIIIIIIIIIIIII\n"
                       b"@v.5 x\nACGTACGTACGT\n+v.5 x\nIIIIIIIIIIII\n")
    # a single record
    out["edge_one"] = b"@only 1\nACGT\n+\nIIII\n"
    # oversize records (usrs.cpp:269-301): base lines of 70 kb and 300 kb, a header of 9000 bytes, one right at the limit
    # (65534: still on the model path) and one just over it (65535), between ordinary records of changing length; the first
    # record of the file is oversize too (determine_record's own branch, usrs.cpp:206-229).  Skewed bases and few quality
    # values keep the raw streams small.
    rng = np.random.default_rng(20261004)

    def rec(i, n, hdr=None):
        seq = "".join(rng.choice(list("ACGTN"), n, p=[0.94, 0.02, 0.02, 0.015, 0.005]))
        q = "".join(rng.choice(list("IH5#!"), n, p=[0.95, 0.02, 0.01, 0.01, 0.01]))
        return "@%s\n%s\n+\n%s\n" % (hdr or "ov.%d len=%d" % (i, n), seq, q)
    recs = [rec(0, 66000), rec(1, 150), rec(2, 70000), rec(3, 200), rec(4, 200), rec(5, 300001), rec(6, 65534), rec(7, 65535),
            rec(8, 120, hdr="long " + "h" * 9000 + " 8"), rec(9, 120), rec(10, 131072), rec(11, 90)]
    out["edge_oversize"] = "".join(recs).encode()
    return out


def main():
    assert O.ref_binary(), "build oracle/_ref first: make -C oracle ref"
    manifest = {}
    fixtures = {}
    for s in SAMPLES:
        fixtures[s] = open("/root/reference/samples/%s.fq" % s, "rb").read()
    fixtures.update(edge_cases())
    only = set(sys.argv[1:])                        # names on the command line: (re)generate those only
    if only:
        manifest = json.load(open(os.path.join(HERE, "manifest.json")))
        fixtures = {k: v for k, v in fixtures.items() if k in only}
    for name in INPUT_ONLY:
        if not only or name in only:
            with gzip.GzipFile(os.path.join(HERE, name + ".fq.gz"), "wb", mtime=0) as f:
                f.write(open("/root/reference/samples/%s.fq" % name, "rb").read())
    for name, fq in sorted(fixtures.items()):
        with gzip.GzipFile(os.path.join(HERE, name + ".fq.gz"), "wb", mtime=0) as f:
            f.write(fq)
        entry = {"bytes": len(fq), "md5": hashlib.md5(fq).hexdigest(), "levels": {}}
        arrays = {}
        for level in (1, 2, 3, 4):
            image = O.ref_compress(fq, level, quiet=True)
            a = O.parse(image)
            dec = O.ref_decompress(image)
            entry["levels"][str(level)] = {
                "streams": {k: len(v) for k, v in a.streams.items()},
                "image_md5": hashlib.md5(image).hexdigest(),
                "roundtrip_exact": dec == fq,
            }
            for k, v in a.streams.items():
                arrays["l%d/%s" % (level, k)] = np.frombuffer(v, np.uint8)
            if dec != fq:
                arrays["l%d/<decoded>" % level] = np.frombuffer(dec, np.uint8)
        np.savez_compressed(os.path.join(HERE, name + ".ref.npz"), **arrays)
        manifest[name] = entry
        print(name, len(fq), {lv: e["streams"].get("qlt") for lv, e in entry["levels"].items()},
              "lossy" if not entry["levels"]["3"]["roundtrip_exact"] else "")
    json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
