"""CPU: the oracle (oracle/sfq_oracle.c) against the reference's golden vectors, and -- when the
compiled reference is present (build container) -- against the reference itself."""
import glob
import os

import numpy as np
import pytest

import util
from oracle import oracle as O
from slimfastq_amd import capi

LEVELS = (1, 2, 3, 4)


@pytest.mark.parametrize("name", util.golden_names())
def test_oracle_streams_match_reference_golden(name):
    fq = util.golden_fastq(name)
    for level in LEVELS:
        gold = util.golden_streams(name, level)
        mine = O.compress(fq, level, quiet=True)
        want = {k: v for k, v in gold.items() if k != "<decoded>"}
        assert list(mine.streams.keys()) == list(want.keys()), (name, level)
        for k in want:
            assert mine.streams[k] == want[k], (name, level, k, len(mine.streams[k]), len(want[k]))
        # decode of the reference's archive == what the reference itself decodes (== input unless lossy)
        expect = gold.get("<decoded>", fq)
        assert O.decompress(mine.image) == expect, (name, level)


def test_golden_sizes_match_survey_table():
    # rec / gen / qlt sizes measured from the reference
    t = {"tst1": {1: (5968, 61644, 85735), 3: (5968, 62221, 89043), 4: (5968, 62375, 89043)},
         "small": {2: (630, 4715, 7477), 3: (630, 4716, 7161)},
         "fast5.to": {3: (2455, 78159, 249309)},
         "tstb": {3: (45, 122, 209)}}
    for name, lv in t.items():
        for level, (r, g, q) in lv.items():
            s = util.golden_streams(name, level)
            assert (len(s["rec"]), len(s["gen"]), len(s["qlt"])) == (r, g, q)


def test_stream_level_entry_points_agree_with_whole_file():
    fq = util.golden_fastq("small")
    lines = fq.split(b"\n")[:-1]
    import numpy as np
    offs = np.cumsum([0] + [len(l) + 1 for l in lines])
    nrec = len(lines) // 4
    hoff = offs[0:4 * nrec:4] + 1
    hlen = np.array([len(l) - 1 for l in lines[0::4]])
    goff = offs[1:4 * nrec:4]; glen = np.array([len(l) for l in lines[1::4]])
    qoff = offs[3:4 * nrec:4]; qlen = np.array([len(l) for l in lines[3::4]])
    for level in LEVELS:
        a = O.compress(fq, level)
        q, _ = O.qlt_encode(fq, qoff, qlen, level)
        assert q == a.streams["qlt"]
        bits = {1: 18, 2: 22, 3: 24, 4: 26}[level]
        g, ns, nn, nb = O.gen_encode(fq, goff, glen, qoff, qlen, bits)
        assert g == a.streams["gen"]
        r, x = O.rec_encode(fq, hoff, hlen)
        assert r == a.streams["rec"]
        assert O.qlt_decode(q, qoff, qlen, len(fq), level)[qoff[0]:qoff[0] + qlen[0]] == lines[3]


def test_xfile_roundtrip_boundaries():
    # the reference's utest.cpp:187-355 boundaries for put_u/get_u (self-consistency, no golden bytes there)
    import numpy as np
    vals = list(range(300)) + [i * 77 for i in range(300)] + [0x7f, 0x80, 0x7ffd, 0x7ffe, 0x7fff, 0xffff, 0xfffe,
            (1 << 32) - 1, 1 << 32, (1 << 40) + 5, (1 << 52) + 7, (1 << 64) - 1, (1 << 64) - 122]
    s = O.xfile_encode_u(vals)
    back = O.xfile_decode_u(s, len(vals))
    assert [int(v) for v in back] == [v & ((1 << 64) - 1) for v in vals]


def test_rice_exception_lists_hold_the_references_positions():
    """Frozen tables, round 4: a block's base-exception lists are Rice-coded (oracle sfqo_exc_rice_block, exc.hip).  The
    POSITIONS they hold are the reference's (gens.cpp:91-114): N-like bases whose quality is not '!', real bases under '!',
    lowercase bases, counted from 1 over the block's base lines -- checked against a numpy statement of that, through the
    decoder; gaps over the escape threshold, empty lists and lists of one entry included."""
    import numpy as np
    from slimfastq_amd import capi
    rng = np.random.default_rng(5)
    for nrec, n_low, n_bang in ((3000, 200, 300), (40, 0, 0), (2000, 1, 1), (6000, 4000, 3000)):
        fq = bytearray(capi.synth_fastq(nrec, 150, seed=nrec))
        starts, lens = util.line_table(bytes(fq))
        goff, glen, qoff, qlen = starts[1::4], lens[1::4], starts[3::4], lens[3::4]
        for r in rng.integers(0, nrec, n_low):
            at = goff[r] + int(rng.integers(0, 150))
            if fq[at] in b"ACGTN":
                fq[at] |= 0x20
        if any(c == ord("n") for c in fq) and any(c == ord("N") for c in fq):       # (one N character per block)
            fq = bytearray(bytes(fq).replace(b"n", b"a"))
        for r in rng.integers(0, nrec, n_bang):
            fq[qoff[r] + int(rng.integers(0, 150))] = ord("!")
        fq = bytes(fq)
        ns, nn, lc, n_byte = O.exc_rice_block(fq, goff, glen, qoff, qlen)
        B = np.frombuffer(fq, np.uint8)
        bases = np.concatenate([B[goff[r]:goff[r] + glen[r]] for r in range(nrec)])
        quals = np.concatenate([B[qoff[r]:qoff[r] + qlen[r]] for r in range(nrec)])
        is_n, bang = (bases == ord("N")) | (bases == ord("n")), quals == ord("!")
        for blob, want in ((ns, is_n & ~bang), (nn, ~is_n & bang), (lc, bases >= 97)):
            want = (np.nonzero(want)[0] + 1).astype(np.uint64)
            assert np.array_equal(O.exc_rice_decode(blob), want)
            assert (len(blob) == 0) == (len(want) == 0)
        assert n_byte == (ord("N") if is_n.any() else 0)
    # the escape (a gap of 32 << k or more: 32 one bits, then the gap in 40 bits): one N, 12 000 bases into a block whose lists start at k = 8
    fq = bytearray(capi.synth_fastq(200, 150, seed=2).replace(b"N", b"A"))
    starts, lens = util.line_table(bytes(fq))
    fq[starts[1 + 4 * 80]] = ord("N")
    fq = bytes(fq)
    ns, _, _, _ = O.exc_rice_block(fq, starts[1::4], lens[1::4], starts[3::4], lens[3::4])
    assert list(O.exc_rice_decode(ns)) == [80 * 150 + 1] and len(ns) == 11        # 32 + 40 bits, then the end: a zero bit and thirteen more


@pytest.mark.skipif(O.ref_binary() is None or not os.path.isdir("/root/reference/samples"),
                    reason="compiled reference / samples only exist in the build container")
def test_oracle_vs_compiled_reference_all_samples():
    for f in sorted(glob.glob("/root/reference/samples/*.fq")):
        fq = open(f, "rb").read()
        if len(fq) > 1_000_000:
            levels = (3,)
        else:
            levels = (1, 2, 3, 4)
        for level in levels:
            ref_img = O.ref_compress(fq, level, quiet=True)
            mine = O.compress(fq, level, quiet=True)
            assert mine.image == ref_img, (f, level)          # whole container, byte for byte
            assert O.decompress(ref_img) == fq
            assert O.ref_decompress(mine.image) == fq


@pytest.mark.skipif(O.ref_binary() is None, reason="compiled reference only exists in the build container")
def test_oracle_vs_compiled_reference_synthetic_and_chunks():
    from slimfastq_amd import capi
    fq = capi.synth_fastq(6000, 150, seed=7)
    for level in (1, 3):
        assert O.compress(fq, level).image == O.ref_compress(fq, level)
    # chunk-as-standalone-file (the block format's definition) incl. a non-default context size
    for chunk in util.split_records(fq, 2500):
        assert O.compress(chunk, 3).image == O.ref_compress(chunk, 3)
    lr = capi.synth_fastq(6, 0, seed=3, kind=1)
    assert O.compress(lr, 3).image == O.ref_compress(lr, 3)


@pytest.mark.skipif(O.ref_binary() is None, reason="compiled reference only exists in the build container")
def test_pre5_header_stream_is_what_the_reference_decodes():
    """The oracle's pre-version-5 "rec" encoder (sfqo_rec_encode_pre5) is derived from RecLoad::load_pre5 (recs.cpp:463-510),
    the only statement of that layout the reference still has.  Pin it: an archive that says version=4 and holds that stream
    beside the reference's other streams must decode, with the COMPILED REFERENCE, to the text."""
    from slimfastq_amd import capi, dist_compress as dc
    import tempfile
    fq = capi.synth_fastq(1200, 100, seed=44)
    starts, lens = util.line_table(fq)
    ref = O.compress(fq, 3)
    rec4 = O.rec_encode_pre5(fq, starts[0::4] + 1, lens[0::4] - 1)
    assert rec4 != ref.streams["rec"]
    info = ref.streams["<info>"].decode("latin1").replace("version=6", "version=4")
    info = "".join(l + "\n" for l in info.split("\n") if l and not l.startswith("comp.size="))
    streams = [(k, rec4 if k == "rec" else v) for k, v in ref.streams.items() if k != "<info>"]
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "v4.sfq")
        dc.write_archive(path, info, streams)
        img = open(path, "rb").read()
    assert O.parse(img).info["version"] == "4"
    assert O.ref_decompress(img) == fq                       # the reference itself, through load_pre5
    assert O.decompress(img) == fq                           # and the oracle's restatement of it


def test_match_model_chains_decode_back_on_the_cpu():
    """The block format's base model of round 5 (oracle/sfq_oracle.c "the generation MATCH model"; the product's kernels are gm.hip) is this
    project's own rule, so the oracle pins it to ITSELF here, without a GPU: the chains it writes for reads that overlap -- 12 000 reads of
    a 60 kb genome, either strand, half a per cent of substitutions, a few N -- decode, generation by generation, with an index the decoder
    builds from what it has decoded, to the bases they were made from; the model pays (well under two bits a base); and bases that
    do not repeat leave it off, coded four to a symbol."""
    rng = np.random.default_rng(7)
    G = rng.integers(0, 4, 60_000, dtype=np.uint8)
    n, L = 12_000, 100
    comp = np.array([3, 2, 1, 0], np.uint8)
    letters = np.frombuffer(b"ACGT", np.uint8)
    recs, codes = [], []
    for i in range(n):
        p = int(rng.integers(0, len(G) - L))
        c = G[p:p + L].copy()
        if rng.integers(2):
            c = comp[c[::-1]]
        flip = rng.random(L) < 0.005
        c[flip] = (c[flip] + rng.integers(1, 4, int(flip.sum()), dtype=np.uint8)) & 3
        line = letters[c].copy()
        if i % 97 == 0:
            line[int(rng.integers(L))] = ord("N"); c = c.copy(); c[line == ord("N")] = 0       # (an N is coded as A, gens.cpp:116-136)
        codes.append(c); recs.append(b"@r%d\n" % i + line.tobytes() + b"\n+\n" + b"I" * L + b"\n")
    fq = b"".join(recs)
    starts, lens = util.line_table(fq)
    goff, glen = starts[1::4], lens[1::4]
    br, cr, tb = 64, 8, 18
    want, sizes, on = O.gm_encode_chains(fq, goff, glen, tb, br, cr)
    assert on == 1 and len(want) * 8 < 1.2 * n * L
    back = O.gm_decode_chains(want, sizes, glen, tb, br, cr)
    assert np.array_equal(back, np.concatenate(codes))
    # bases that do not repeat: the verdict says no, four bases a symbol -- two bits a base and a few bytes a chain
    iid = capi.synth_fastq(6000, 100, seed=3)
    s2, l2 = util.line_table(iid)
    flat, sizes2, on2 = O.gm_encode_chains(iid, s2[1::4], l2[1::4], 16, br, cr)
    assert on2 == 0 and 6000 * 100 // 4 <= len(flat) <= 6000 * 100 // 4 + 6 * len(sizes2)
