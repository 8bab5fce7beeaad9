"""Shared helpers for the tests (fixtures, FASTQ slicing).  The oracle is imported here only as a checker."""
import gzip
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
MANIFEST = json.load(open(os.path.join(GOLDEN, "manifest.json")))
MAIN_STREAMS = ["rec", "gen", "qlt"]


def golden_names(max_bytes=None):
    return sorted(n for n, e in MANIFEST.items() if max_bytes is None or e["bytes"] <= max_bytes)


def golden_fastq(name) -> bytes:
    return gzip.open(os.path.join(GOLDEN, name + ".fq.gz"), "rb").read()


def golden_streams(name, level) -> dict:
    """Reference-produced stream bytes {name: bytes} (+ '<decoded>' when the reference's own decode
    differs from the input, i.e. the reference is lossy on that fixture)."""
    z = np.load(os.path.join(GOLDEN, name + ".ref.npz"))
    pre = "l%d/" % level
    return {k[len(pre):]: z[k].tobytes() for k in z.files if k.startswith(pre)}


def split_records(fastq: bytes, n_per_block: int):
    """Cut FASTQ text into chunks of n_per_block 4-line records (the last may be short)."""
    lines = fastq.split(b"\n")
    assert lines[-1] == b""
    lines = lines[:-1]
    assert len(lines) % 4 == 0
    out = []
    for i in range(0, len(lines), 4 * n_per_block):
        out.append(b"\n".join(lines[i:i + 4 * n_per_block]) + b"\n")
    return out


def info_of(streams: dict) -> dict:
    info = {}
    for line in streams["<info>"].decode("latin1").split("\n"):
        if "=" in line:
            k, v = line.split("=", 1)
            info.setdefault(k, v)
    return info


def line_table(fastq: bytes):
    """(offsets, lengths) of every line of a 4-line FASTQ, as numpy arrays."""
    a = np.frombuffer(fastq, np.uint8)
    nl = np.flatnonzero(a == 10)
    starts = np.concatenate([[0], nl[:-1] + 1]).astype(np.uint64)
    return starts, (nl - starts).astype(np.uint32)


def unpack_prior(blob: bytes, q_rows: int):
    """"qlt.pri" -> uint32 [q_rows, 66] (slot[64], total, iend); mirrors api.cpp pack_prior."""
    rows = np.zeros((q_rows, 66), np.uint32)
    p = 0

    def vint():
        nonlocal p
        v = sh = 0
        while True:
            c = blob[p]; p += 1
            v |= (c & 0x7f) << sh; sh += 7
            if not c & 0x80:
                return v
    assert vint() == q_rows
    c = 0; first = True
    while True:
        v = vint()
        if v == 0:
            break
        c = v - 1 if first else c + v - 1
        first = False
        iend, nnz = blob[p], blob[p + 1]; p += 2
        used = set(); total = 0
        for j in range(nnz):
            sym = blob[p]; p += 1
            f = vint()
            rows[c, j] = f | (sym << 16); used.add(sym); total += f
        j = nnz
        for s in range(iend):
            if s not in used:
                rows[c, j] = s << 16; j += 1
        rows[c, 64] = total; rows[c, 65] = iend
    return rows


def _vints(blob: bytes):
    p = 0
    out = []
    while p < len(blob):
        v = sh = 0
        while True:
            c = blob[p]; p += 1
            v |= (c & 0x7f) << sh; sh += 7
            if not c & 0x80:
                break
        out.append(v)
    return out


def unpack_chains(blob: bytes, nblocks=None):
    """"chn.idx" -> dict(chain_reads, flags, qlt, gen [, seg_len, seg_blocks] [, rec_chain_reads, rec, rec_hdr_bytes]); mirrors api.cpp.
    flags bit 2: every list of sizes is stored as zigzag differences to the entry before it; bit 3: the chains are SEGMENTS of one
    record -- their length and every block's number of chains follow the chain count (nblocks must be given); bit 4: the base
    exceptions are Rice-coded gap lists (exc.hip); bit 5: the bases are coded under the generation match model (gm.hip), the bits of its index
    follow the flags."""
    v = _vints(blob)
    cr, flags = v[0], v[1]
    p = 2
    out = {"chain_reads": cr, "flags": flags}
    ng = None
    if flags & 32:                                   # bit 5: the bases are coded under the match model (gm.hip): the index's bits,
        out["gm_table_bits"] = v[p]                  # then the base chains' own geometry -- records per chain, their number
        out["gen_chain_reads"] = v[p + 1]; ng = v[p + 2]; p += 3
    n = v[p]; p += 1
    if ng is None:
        ng = n; out["gen_chain_reads"] = cr
    if flags & 8:
        assert nblocks is not None
        out["seg_len"] = v[p]; out["seg_blocks"] = v[p + 1:p + 1 + nblocks]; p += 1 + nblocks
        assert sum(out["seg_blocks"]) == n

    def sizes(raw):
        if not flags & 4:
            return np.array(raw, np.uint32)
        d = np.array([(x >> 1) ^ -(x & 1) for x in raw], np.int64)
        return np.cumsum(d).astype(np.uint32)
    out.update(qlt=sizes(v[p:p + n]), gen=sizes(v[p + n:p + n + ng]))
    p += n + ng
    if flags & 2:
        rcr, m = v[p], v[p + 1]; p += 2
        out.update(rec_chain_reads=rcr, rec=sizes(v[p:p + m]), rec_hdr_bytes=sizes(v[p + m:p + 2 * m]))
        p += 2 * m
    assert p == len(v)
    return out


def unpack_rec_prior(blob: bytes):
    """"rec.pri" -> uint32 [66 * 16 * 256] scaled frequencies; mirrors api.cpp pack_rec_prior."""
    f = np.zeros(66 * 16 * 256, np.uint32)
    p = 0

    def vint():
        nonlocal p
        v = sh = 0
        while True:
            c = blob[p]; p += 1
            v |= (c & 0x7f) << sh; sh += 7
            if not c & 0x80:
                return v
    assert vint() == 66 * 16
    r = 0; first = True
    while True:
        v = vint()
        if v == 0:
            break
        r = v - 1 if first else r + v - 1
        first = False
        for _ in range(vint()):
            sym = blob[p]; p += 1
            f[r * 256 + sym] = vint()
    assert p == len(blob)
    return f


def block_reference(chunk: bytes, level: int, gen_bits: int = 0):
    """The oracle's archive for ONE block of the block format: the reference run on a FASTQ holding only that block's
    records, with the block format's lossless rules (oracle sfqo_opts.lossless: where the reference would alter the text
    the block format departs from its bytes -- SURVEY H7; everywhere else the two are byte-identical)."""
    from oracle import oracle as O
    return O.compress(chunk, level, gen_bits=gen_bits, lossless=True)


def exc_rice_reference(chunk: bytes, solid: int = 0):
    """The oracle's Rice-coded base-exception lists of ONE block (frozen tables, "chn.idx" flag bit 4; exc.hip):
    {"gen.Ns", "gen.Nn", "gen.lc"} -> bytes."""
    from oracle import oracle as O
    starts, lens = line_table(chunk)
    ns, nn, lc, _ = O.exc_rice_block(chunk, starts[1::4] + solid, lens[1::4] - solid, starts[3::4] + solid, lens[3::4] - solid)
    return {"gen.Ns": ns, "gen.Nn": nn, "gen.lc": lc}

