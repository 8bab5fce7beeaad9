"""GPU: the HIP path (through the C ABI) against the oracle and the reference's golden vectors.
Bit-exact everywhere: this is integer/byte work."""
import numpy as np
import pytest

import util
from oracle import oracle as O
from slimfastq_amd import capi

pytestmark = pytest.mark.gpu
LEVEL_BITS = {1: 18, 2: 22, 3: 24, 4: 26}
PRIOR_SYMBOLS = 4096      # the prior counts the first 4096 quality symbols of every sampled record (kernels.h)
KERNELS = (0, 1)           # 0 = default (wave-per-block) kernels, 1 = lane-per-block cross-check kernels


def assert_streams_equal(enc, want: dict, block=None, ctxmsg=""):
    for name in capi.STREAM_NAMES:
        got = enc.stream(name, block)
        exp = want.get(name, b"")
        assert got == exp, "%s stream %s: got %d bytes, want %d" % (ctxmsg, name, len(got), len(exp))


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("level", (1, 2, 3, 4))
def test_single_block_equals_reference_streams_synthetic(ctx, level, kernel):
    fq = capi.synth_fastq(3000, 150, seed=level)
    enc = ctx.encode_host(fq, level=level, block_reads=0, kernel=kernel)
    want = O.compress(fq, level).streams
    assert enc.res.n_blocks == 1 and enc.res.n_records == 3000
    assert_streams_equal(enc, want, ctxmsg="level %d" % level)
    b = enc.blocks[0]
    assert (b.llen, b.solid, b.two_id, b.gen_bits) == (150, 0, 0, LEVEL_BITS[level])
    assert enc.first_hdrs == fq[1:fq.index(b"\n")]


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("name", util.golden_names())
def test_single_block_equals_reference_golden(ctx, name, kernel):
    fq = util.golden_fastq(name)
    for level in (1, 3) if len(fq) > 300000 else (1, 2, 3, 4):
        gold = util.golden_streams(name, level)
        enc = ctx.encode_host(fq, level=level, block_reads=0, kernel=kernel)
        assert_streams_equal(enc, gold, ctxmsg="%s l%d" % (name, level))
        info = util.info_of(gold)
        b = enc.blocks[0]
        assert b.n_records == int(info["num_records"]) and b.llen == int(info["llen"])
        assert b.solid == int(info.get("usr.solid", 0)) and b.two_id == int(info["usr.2id"])
        assert b.n_byte == int(info.get("gen.N_byte", b.n_byte if b.n_byte in (0, ord("N")) else -1))
        assert enc.first_hdrs.decode("latin1") == info["rec.first"]


@pytest.mark.parametrize("kernel", KERNELS)
def test_blocks_equal_reference_run_per_chunk(ctx, kernel):
    """The block format's definition: block b's streams == the reference's streams for a FASTQ holding
    only that block's records (counters restart per block) -- with the lossless rules where the reference would alter
    the text (util.block_reference; none of them applies to this input)."""
    fq = capi.synth_fastq(5300, 150, seed=21)
    for level, br in ((3, 1000), (1, 2500), (4, 700)):
        enc = ctx.encode_host(fq, level=level, block_reads=br, kernel=kernel)
        chunks = util.split_records(fq, br)
        assert enc.res.n_blocks == len(chunks)
        for b, chunk in enumerate(chunks):
            want = util.block_reference(chunk, level, gen_bits=enc.blocks[b].gen_bits).streams
            assert_streams_equal(enc, want, block=b, ctxmsg="level %d block %d" % (level, b))
            assert enc.blocks[b].n_records == chunk.count(b"\n") // 4


@pytest.mark.parametrize("name", ["tst1", "fast5.to", "tsta", "edge_hdr", "edge_len"])
def test_blocks_on_real_samples(ctx, name):
    fq = util.golden_fastq(name)
    nrec = fq.count(b"\n") // 4
    br = max(2, nrec // 7)
    enc = ctx.encode_host(fq, level=3, block_reads=br)
    for b, chunk in enumerate(util.split_records(fq, br)):
        want = util.block_reference(chunk, 3, gen_bits=enc.blocks[b].gen_bits).streams
        assert_streams_equal(enc, want, block=b, ctxmsg="%s block %d" % (name, b))


@pytest.mark.parametrize("name", util.golden_names())
def test_decode_reference_written_streams(ctx, name):
    """North-star requirement: a reference-written archive decodes to what the reference itself decodes."""
    fq = util.golden_fastq(name)
    for level in (3,) if len(fq) > 300000 else (1, 2, 3, 4):
        gold = util.golden_streams(name, level)
        info = util.info_of(gold)
        bi = capi.BlockInfo()
        bi.first_record = 0
        bi.n_records = int(info["num_records"]); bi.llen = int(info["llen"])
        bi.solid = int(info.get("usr.solid", 0)); bi.two_id = int(info["usr.2id"])
        bi.n_byte = int(info.get("gen.N_byte", 0)); bi.gen_bits = LEVEL_BITS[level]
        first = info["rec.first"].encode("latin1")
        bi.first_hdr_off = 0; bi.first_hdr_len = len(first)
        data = b""; soff = []
        for i, s in enumerate(capi.STREAM_NAMES):
            soff.append(len(data)); data += gold.get(s, b""); bi.size[i] = len(gold.get(s, b""))
        blocks = (capi.BlockInfo * 1)(bi)
        for kernel in KERNELS:               # a wavefront per stream (decode_w.hip) / the lane-per-block cross-check kernels
            out = ctx.decode_host((blocks, first, data, soff), level=level, version=int(info["version"]), out_cap=len(fq) * 2 + 4096, kernel=kernel)
            assert out == gold.get("<decoded>", fq), (name, level, kernel)


def test_decode_pre5_header_stream(ctx):
    """RecLoad::load_pre5 (recs.cpp:463-510): archives of format versions < 5 code a numeric field as the gap to the number
    read back from the previous header's TEXT.  No such archive ships with the reference, so the stream is written by an
    encoder derived from that decoder (oracle sfqo_rec_encode_pre5); the other streams are the reference's as usual."""
    fq = capi.synth_fastq(1500, 100, seed=44)
    # no numeric header field of the synthetic reads is ever 0 or has a leading zero, and the shape is constant: what pre-5 can express
    starts, lens = util.line_table(fq)
    ref = O.compress(fq, 3).streams
    info = util.info_of(ref)
    rec4 = O.rec_encode_pre5(fq, starts[0::4] + 1, lens[0::4] - 1)
    assert rec4 != ref["rec"]
    bi = capi.BlockInfo()
    bi.first_record = 0
    bi.n_records = int(info["num_records"]); bi.llen = int(info["llen"])
    bi.two_id = int(info["usr.2id"]); bi.gen_bits = LEVEL_BITS[3]
    first = info["rec.first"].encode("latin1")
    bi.first_hdr_off = 0; bi.first_hdr_len = len(first)
    data = b""; soff = []
    for i, name in enumerate(capi.STREAM_NAMES):
        piece = rec4 if name == "rec" else ref.get(name, b"")
        soff.append(len(data)); data += piece; bi.size[i] = len(piece)
    blocks = (capi.BlockInfo * 1)(bi)
    assert ctx.decode_host((blocks, first, data, soff), level=3, version=4, out_cap=len(fq) * 2 + 4096) == fq
    # read as a version >= 5 stream the same bytes do not decode to the text
    try:
        assert ctx.decode_host((blocks, first, data, soff), level=3, version=6, out_cap=len(fq) * 2 + 4096) != fq
    except capi.SfqError:
        pass


@pytest.mark.parametrize("level", (1, 2, 3, 4))
def test_roundtrip_blocks(ctx, level):
    fq = capi.synth_fastq(4100, 150, seed=30 + level)
    for br in (0, 512):
        enc = ctx.encode_host(fq, level=level, block_reads=br)
        for kernel in KERNELS:
            assert ctx.decode_host(enc, level=level, out_cap=len(fq) + 4096, kernel=kernel) == fq, (br, kernel)


def test_roundtrip_long_reads_and_samples(ctx):
    lr = capi.synth_fastq(12, 0, seed=9, kind=1)
    for br in (0, 5):
        enc = ctx.encode_host(lr, level=3, block_reads=br)
        assert ctx.decode_host(enc, level=3, out_cap=len(lr) + 4096) == lr
    for name in ("small", "solid", "tstc", "tstd", "badqlt", "edge_len", "edge_n", "edge_hiq"):
        fq = util.golden_fastq(name)
        nrec = fq.count(b"\n") // 4
        enc = ctx.encode_host(fq, level=2, block_reads=max(1, nrec // 3))
        assert ctx.decode_host(enc, level=2, out_cap=len(fq) * 2 + 4096) == fq, name


def test_reads_beyond_the_reference_line_limit(ctx):
    """Base / quality lines over 65 534 bytes: the reference diverts such records to its raw usr.lrec / usr.lgen / usr.lqlt
    streams (usrs.cpp:269-301), and so does format 6 (-B 0) here, stream for stream; the block format codes them like any other."""
    import random
    rnd = random.Random(5)
    recs = []
    for i, n in enumerate((150, 70_000, 200, 300_001, 65_534, 65_535, 151, 131_072, 90)):
        seq = "".join(rnd.choice("ACGT") for _ in range(n))
        if n > 1000:
            seq = seq[:500] + "N" * 7 + seq[507:]
        qual = "".join(chr(33 + min(60, max(0, int(rnd.gauss(30, 8))))) for _ in range(n))
        if n > 1000:
            qual = qual[:500] + "!" * 7 + qual[507:]
        recs.append("@long.%d ch=%d len=%d\n%s\n+\n%s\n" % (i + 1, 100 + i, n, seq, qual))
    fq = "".join(recs).encode()
    for tables in (capi.TABLES_FROZEN, 0):
        for br in (capi.BLOCK_AUTO, 2, 4):
            enc = ctx.encode_host(fq, level=3, block_reads=br, tables=tables)
            assert ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq, (tables, br)
    for kernel in KERNELS:
        leg = ctx.encode_host(fq, level=3, block_reads=0, kernel=kernel)
        assert_streams_equal(leg, O.compress(fq, 3).streams, ctxmsg="oversize records, one block, kernel %d" % kernel)
        assert leg.blocks[0].n_records == 9 and leg.res.n_records == 9
        assert ctx.decode_host(leg, level=3, out_cap=len(fq) + 4096) == fq
    # what the reference's framing cannot express is refused, not mangled: a quality line over the limit beside a base line within it
    odd = b"@q 1\nACGT\n+\nIIII\n@q 2\n" + b"A" * 100 + b"\n+\n" + b"I" * 70000 + b"\n"
    with pytest.raises(capi.SfqError) as e:
        ctx.encode_host(odd, level=3, block_reads=0)
    assert e.value.code == -7          # SFQ_E_UNSUPPORTED


def test_oversize_records_through_the_cli_and_the_reference(tmp_path):
    """usr.lrec / usr.lgen / usr.lqlt both ways through the container: `slimfastq-amd -B 0` writes the archive the reference
    writes (streams identical, the oracle and -- where it travelled -- the compiled reference decode it), and decodes the
    reference's archive of the golden fixture."""
    import subprocess
    cli = _cli()
    fq = util.golden_fastq("edge_oversize")
    src = tmp_path / "ov.fq"; src.write_bytes(fq)
    leg = tmp_path / "ov.sfq"
    subprocess.check_call([cli, "-u", str(src), "-f", str(leg), "-O", "-l", "3", "-B", "0", "-q"])
    img = leg.read_bytes()
    got, want = O.parse(img), O.compress(fq, 3)
    assert {k: v for k, v in got.streams.items() if k != "<info>"} == {k: v for k, v in want.streams.items() if k != "<info>"}
    assert got.info["num_records"] == "12" and got.info["rec.first"] == want.info["rec.first"]
    assert O.decompress(img) == fq
    if O.ref_binary():
        assert O.ref_decompress(img) == fq
    refimg = tmp_path / "ov.ref.sfq"; refimg.write_bytes(want.image)
    p = subprocess.run([cli, "-d", "-f", str(refimg)], capture_output=True, check=True)
    assert p.stdout == fq


def _quirk_fastq(n, seed, mixed_n=True):
    """Everything the reference gives back altered: lowercase bases (soft-masked stretches, single ones, 'n' with
    and without quality '!'), a header field that becomes empty, numbers of 19 and 20 digits, a field of value 0 after a
    larger one, a NUL inside a header.  mixed_n: both 'n' and 'N' occur (the reference aborts on that: "switched N_byte",
    gens.cpp:107-108; the block format lists the lowercase ones in "gen.lc")."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        ln = int(rng.integers(40, 180))
        seq = rng.choice(list("ACGT"), ln)
        qual = rng.integers(2, 41, ln)
        k = rng.random()
        if k < 0.3:                                                   # a soft-masked stretch
            a = int(rng.integers(0, ln)); b = int(rng.integers(a, ln + 1))
            seq[a:b] = np.char.lower(seq[a:b])
        elif k < 0.5:
            seq = np.where(rng.random(ln) < 0.1, np.char.lower(seq), seq)
        nmask = rng.random(ln) < 0.04
        seq = np.where(nmask, np.where((rng.random(ln) < 0.5) & mixed_n, "n", "N"), seq)
        qual = np.where(nmask & (rng.random(ln) < 0.6), 0, qual)
        qual = np.where(~nmask & (rng.random(ln) < 0.02), 0, qual)       # a real (maybe lowercase) base under quality '!'
        h = rng.random()
        if h < 0.05: hdr = "q%d::%d" % (i, i * 3)                     # an empty field
        elif h < 0.10: hdr = "q%d:%d:%d" % (i, 9223372036854775800 + int(rng.integers(0, 9)), i)      # 19 digits, >= 2^63 - 8
        elif h < 0.15: hdr = "q%d:%d:%d" % (i, 18446744073709551000 + i, i)                            # 20 digits
        elif h < 0.20: hdr = "q%d:0:%d" % (i, i)
        elif h < 0.22: hdr = "q%d:a\0b:%d" % (i, i)                   # a NUL inside
        else: hdr = "q%d:%d:%d" % (i, 1000 + (i * 7) % 50, i)
        out += ["@" + hdr, "".join(seq), "+", "".join(chr(33 + int(q)) for q in qual)]
    return ("\n".join(out) + "\n").encode("latin1")


@pytest.mark.parametrize("tables", (capi.TABLES_FROZEN, capi.TABLES_ADAPTIVE))
def test_block_format_is_lossless_where_the_reference_is_not(ctx, tables):
    """SURVEY H7 / VERDICT r02 item 1: decode(encode(x)) == x in the block format for the inputs the reference mangles; every
    block's side streams (and, with adaptive tables, all of its streams) are the oracle's lossless restatement; format 6
    (-B 0) keeps the reference's bytes AND its restoration."""
    fq = _quirk_fastq(3000, 9)
    with pytest.raises(O.OracleError):                                    # 'n' and 'N' in one file: the reference gives up
        O.compress(fq, 3)
    br = 256
    for kernel in ((0,) if tables == capi.TABLES_FROZEN else (0, 1)):
        enc = ctx.encode_host(fq, level=3, block_reads=br, prior_step=capi.PRIOR_AUTO, tables=tables, chain_reads=40, kernel=kernel)
        assert enc.res.stream_bytes[capi.STREAM_NAMES.index("gen.lc")] > 0
        for b, chunk in enumerate(util.split_records(fq, br)):
            want = dict(util.block_reference(chunk, 3, gen_bits=enc.blocks[b].gen_bits).streams)
            names = ("gen.Ns", "gen.Nn", "gen.lc", "usr.x", "usr.x.q") if tables == capi.TABLES_FROZEN else [n for n in capi.STREAM_NAMES if n != "qlt"]
            if tables == capi.TABLES_FROZEN:                                # the same lists, Rice-coded (exc.hip)
                want.update(util.exc_rice_reference(chunk))
            for name in names:                                              # (the quality rows start from the prior: not the cold reference's)
                assert enc.stream(name, b) == want.get(name, b""), (kernel, b, name)
        assert ctx.decode_host(enc, level=3, out_cap=2 * len(fq) + 4096) == fq, kernel
    # format 6: the reference's streams, the reference's restoration (and its refusal of 'n' beside 'N')
    with pytest.raises(capi.SfqError) as e:
        ctx.encode_host(fq, level=3, block_reads=0)
    assert e.value.code == -8
    fq = _quirk_fastq(3000, 9, mixed_n=False)
    ref = O.compress(fq, 3)
    assert O.decompress(ref.image) != fq                                  # the reference is lossy on this input
    leg = ctx.encode_host(fq, level=3, block_reads=0)
    assert_streams_equal(leg, ref.streams, ctxmsg="quirks, one block")
    assert ctx.decode_host(leg, level=3, out_cap=2 * len(fq) + 4096) == O.decompress(ref.image)


def test_block_format_refuses_a_plus_line_it_could_not_give_back(ctx):
    """The reference keeps ONE flag for the '+' line (usrs.cpp:236-239): a second id that differs from the first comes back
    as the first, a '+' line of blanks comes back empty.  The block format refuses such a record (SFQ_E_UNSUPPORTED) instead
    of altering it; a second id that repeats the header round-trips."""
    ok = b"@a 1\nACGT\n+a 1\nIIII\n@b 2\nACGT\n+b 2\nIIII\n"
    for tables in (capi.TABLES_FROZEN, capi.TABLES_ADAPTIVE):
        enc = ctx.encode_host(ok, level=3, block_reads=capi.BLOCK_AUTO, prior_step=capi.PRIOR_AUTO, tables=tables)
        assert enc.blocks[0].two_id == 1 and ctx.decode_host(enc, level=3, out_cap=4096) == ok
        for bad in (b"@a 1\nACGT\n+a 1\nIIII\n@b 2\nACGT\n+b 3\nIIII\n", b"@a 1\nACGT\n+a 1\nIIII\n@b 2\nACGT\n+\nIIII\n",
                    b"@a 1\nACGT\n+\nIIII\n@b 2\nACGT\n+b 2\nIIII\n", b"@a 1\nACGT\n+ \nIIII\n"):
            with pytest.raises(capi.SfqError) as e:
                ctx.encode_host(bad, level=3, block_reads=capi.BLOCK_AUTO, prior_step=capi.PRIOR_AUTO, tables=tables)
            assert e.value.code == -7, bad
    # format 6 takes them, the reference's way
    bad = b"@a 1\nACGT\n+a 1\nIIII\n@b 2\nACGT\n+b 3\nIIII\n"
    leg = ctx.encode_host(bad, level=3, block_reads=0)
    assert ctx.decode_host(leg, level=3, out_cap=4096) == O.decompress(O.compress(bad, 3).image) != bad


def test_corrupt_archives_fail_cleanly_or_decode_to_something(ctx):
    """Untrusted input on the way back: flipped stream bytes, a lying block index, swapped chain sizes, damaged priors.
    Every decode must end in an SfqError or in some bytes -- never in a device fault -- and the context must decode the
    intact archive afterwards."""
    import random
    rnd = random.Random(11)
    fq = capi.synth_fastq(3000, 120, seed=21)
    for tables in (capi.TABLES_FROZEN, capi.TABLES_ADAPTIVE):
        enc = ctx.encode_host(fq, level=3, block_reads=256, prior_step=capi.PRIOR_AUTO, tables=tables, chain_reads=32)
        assert ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq
        outcomes = {"error": 0, "bytes": 0}
        for trial in range(60):
            bad = enc.clone()
            kind = trial % 6
            if kind == 0:                                   # flipped bytes anywhere in the streams
                data = bytearray(bad.data)
                for _ in range(rnd.randint(1, 8)):
                    data[rnd.randrange(len(data))] ^= 1 << rnd.randrange(8)
                bad.data = bytes(data)
            elif kind == 1:                                 # a block index that lies about lengths / flags
                b = bad.blocks[rnd.randrange(len(bad.blocks))]
                what = rnd.randrange(5)
                if what == 0: b.llen = rnd.choice((0, 1, 119, 121, 5000))
                elif what == 1: b.hdr_bytes = rnd.choice((0, 1, 7, 1 << 20))
                elif what == 2: b.two_id ^= 1
                elif what == 3: b.solid ^= 1
                else: b.n_byte = rnd.randrange(256)
            elif kind == 2 and bad.chains:                  # chain sizes moved between neighbours (the sums stay)
                ch = bytearray(bad.chains)
                i = rnd.randrange(4, max(5, len(ch) - 2))
                if 1 < ch[i] < 0x7f and 1 < ch[i + 1] < 0x7f: ch[i] -= 1; ch[i + 1] += 1
                bad.chains = bytes(ch)
            elif kind == 3 and bad.prior:                   # a damaged quality prior
                pr = bytearray(bad.prior); pr[rnd.randrange(len(pr))] ^= 0x55; bad.prior = bytes(pr)
            elif kind == 4 and bad.rec_prior:               # a damaged header prior
                pr = bytearray(bad.rec_prior); pr[rnd.randrange(len(pr))] ^= 0x33; bad.rec_prior = bytes(pr)
            else:                                           # a truncated stream buffer
                bad.data = bad.data[:rnd.randrange(len(bad.data) // 2, len(bad.data))]
            try:
                ctx.decode_host(bad, level=3, out_cap=2 * len(fq) + 4096)
                outcomes["bytes"] += 1
            except capi.SfqError:
                outcomes["error"] += 1
        assert outcomes["error"] > 0
        assert ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq


def test_a_base_exception_list_cut_short_is_refused(ctx):
    """a Rice-coded gap list ("gen.Ns" with frozen tables, exc.hip) reads zeros behind its stream, and a list's
    end is a zero gap -- a stream cut short used to end its list silently, the last N's coming back as A.  The end of a list must
    lie inside its stream."""
    fq = capi.synth_fastq(3000, 150, seed=41)                 # (one N in a thousand bases: 450 of them)
    enc = ctx.encode_host(fq, level=3, block_reads=4096, prior_step=capi.PRIOR_AUTO, tables=capi.TABLES_FROZEN, chain_reads=64)
    assert len(enc.blocks) == 1 and util.unpack_chains(enc.chains)["flags"] & 16
    s = capi.STREAM_NAMES.index("gen.Ns")
    assert enc.blocks[0].size[s] > 8
    assert ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq
    for cut in (1, 2, 5):
        bad = enc.clone()
        bad.blocks[0].size[s] -= cut
        with pytest.raises(capi.SfqError):
            ctx.decode_host(bad, level=3, out_cap=len(fq) + 4096)
    assert ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq


def test_ragged_and_error_inputs(ctx):
    one = b"@only 1\nACGT\n+\nIIII\n"
    enc = ctx.encode_host(one, level=3)
    assert ctx.decode_host(enc, level=3, out_cap=4096) == one
    for bad, code in ((b"", -4), (b"@x\nACGT\n+\nIIII", -4), (b"@x\nACGT\n+\n", -4), (b"x\nACGT\n+\nIIII\n", -4),
                      (b"@x\nACGT\n-\nIIII\n", -4), (b"@x\nACXT\n+\nIIII\n", -8), (b"@x\nAC.TN\n+\nIIIII\n", -8)):
        with pytest.raises(capi.SfqError) as e:
            ctx.encode_host(bad, level=3)
        assert e.value.code == code, bad
    # a file whose EVERY record is over the model path's line limits is refused (the reference writes an archive without a
    # first record for it: "all records were oversized", usrs.cpp:190-197)
    big = b"@big\n" + b"A" * 70000 + b"\n+\n" + b"I" * 70000 + b"\n"
    with pytest.raises(capi.SfqError) as e:
        ctx.encode_host(big, level=3)
    assert e.value.code == -7


def test_qlt_only_entry_point(ctx):
    import torch
    fq = capi.synth_fastq(2000, 150, seed=77)
    t = torch.frombuffer(bytearray(fq), dtype=torch.uint8).cuda()
    out = torch.empty(len(fq), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    res = ctx.encode_device(t.data_ptr(), len(fq), out.data_ptr(), out.numel(), level=3, block_reads=500, qlt_only=True)
    assert res.n_blocks == 4 and res.stream_bytes[capi.STREAM_NAMES.index("gen")] == 0
    got = bytes(out[:res.total_bytes].cpu().numpy())
    want = b"".join(O.compress(c, 3).streams["qlt"] for c in util.split_records(fq, 500))
    assert got == want


@pytest.mark.parametrize("kernel", KERNELS)
def test_low_complexity_bases_same_context_in_one_window(ctx, kernel):
    """Homopolymers / short tandem repeats put the same base context many times into one 64-base window:
    the wave kernel must chain those updates in order (its serial fallback)."""
    import random
    rnd = random.Random(5)
    recs = []
    for i in range(400):
        kind = i % 5
        n = rnd.choice([30, 64, 65, 150, 200])
        if kind == 0:
            seq = rnd.choice("ACGT") * n
        elif kind == 1:
            unit = "".join(rnd.choice("ACGT") for _ in range(rnd.choice([2, 3, 5, 13])))
            seq = (unit * (n // len(unit) + 1))[:n]
        elif kind == 2:
            seq = "".join(rnd.choice("ACGT") for _ in range(n))
        elif kind == 3:
            seq = "".join(rnd.choice("AAAAAAAC") for _ in range(n))
        else:
            seq = "N" * (n // 2) + "ACGT" * (n // 8 + 1)
            seq = seq[:n]
        qual = "".join(rnd.choice("!#5II") if c == "N" else rnd.choice("II5I!") if rnd.random() < 0.02 else "I" for c in seq)
        recs.append("@lc.%d\n%s\n+\n%s\n" % (i, seq, qual))
    fq = "".join(recs).encode()
    for level, br in ((1, 0), (3, 0), (4, 0), (3, 64)):
        enc = ctx.encode_host(fq, level=level, block_reads=br, kernel=kernel)
        for b, chunk in enumerate(util.split_records(fq, br) if br else [fq]):
            want = util.block_reference(chunk, level, gen_bits=enc.blocks[b].gen_bits).streams
            assert_streams_equal(enc, want, block=b, ctxmsg="lowcomplexity l%d b%d" % (level, b))
        for dk in KERNELS:                     # (homopolymers: the wave decoder forwards the rows it has just written)
            assert ctx.decode_host(enc, level=level, out_cap=len(fq) + 4096, kernel=dk) == fq, (level, br, dk)


def _cli():
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "slimfastq_amd", "bin", "slimfastq-amd")
    if not os.access(path, os.X_OK):                 # a build product (not in git): make it if this tree has not been built
        from slimfastq_amd import build
        build.build()
    return path


def test_cli_roundtrip_and_reference_compatibility(tmp_path):
    """The C++ CLI (same flags as the reference) against the reference's container format, both ways:
    it decodes archives written by the reference algorithm, and its -B 0 archives decode with the reference."""
    import subprocess
    cli = _cli()
    for name in ("small", "tstc", "solid", "edge_len"):
        fq = util.golden_fastq(name)
        src = tmp_path / (name + ".fq"); src.write_bytes(fq)
        # block format round trip
        sfq = tmp_path / (name + ".sfq"); out = tmp_path / (name + ".out")
        subprocess.check_call([cli, "-u", str(src), "-f", str(sfq), "-O", "-l", "3", "-B", "100"])
        subprocess.check_call([cli, "-d", "-f", str(sfq), "-u", str(out), "-O"])
        assert out.read_bytes() == fq, name
        # legacy (-B 0): a format-6 file; the oracle (== reference, see test_oracle.py) must decode it
        leg = tmp_path / (name + ".v6.sfq")
        subprocess.check_call([cli, "-u", str(src), "-f", str(leg), "-O", "-l", "2", "-B", "0", "-q"])
        img = leg.read_bytes()
        assert O.decompress(img) == fq, name
        got = O.parse(img)
        want = O.compress(fq, 2)
        assert {k: v for k, v in got.streams.items() if k != "<info>"} == {k: v for k, v in want.streams.items() if k != "<info>"}
        if O.ref_binary():
            assert O.ref_decompress(img) == fq, name           # the compiled reference itself, when it travelled with the tree
        # an archive written by the reference algorithm decodes with the CLI
        refimg = tmp_path / (name + ".ref.sfq"); refimg.write_bytes(want.image)
        p = subprocess.run([cli, "-d", "-f", str(refimg)], capture_output=True, check=True)
        assert p.stdout == fq, name
    # -s lists the info page and the streams; errors keep the reference's wording
    p = subprocess.run([cli, "-s", "-f", str(tmp_path / "small.sfq")], capture_output=True)
    assert b"whoami" in p.stderr and b"qlt" in p.stderr
    bad = tmp_path / "bad.fq"; bad.write_bytes(b"@x\nACXT\n+\nIIII\n")
    p = subprocess.run([cli, "-u", str(bad), "-f", str(tmp_path / "bad.sfq"), "-O"], capture_output=True)
    assert p.returncode == 1 and b"slimfastq: encoding" in p.stderr


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("level", (1, 3))
def test_warm_start_prior_matches_oracle_rule(ctx, level, kernel):
    """Format 7 warm start: the prior counted over every N-th record, the rows built from it and the blocks
    coded from it equal the oracle's restatement of the same rule; decode restores the text."""
    fq = capi.synth_fastq(6100, 150, seed=40 + level)
    step, br = 3, 500
    enc = ctx.encode_host(fq, level=level, block_reads=br, kernel=kernel, prior_step=step)
    starts, lens = util.line_table(fq)
    qoff, qlen = starts[3::4], lens[3::4]
    counts = O.qlt_histogram(fq, qoff, np.minimum(qlen, PRIOR_SYMBOLS), level, 0, step)
    rows = O.qlt_prior_rows(counts)
    got_rows = util.unpack_prior(enc.prior, 4096 if level == 1 else 65536)
    assert np.array_equal(got_rows, rows)
    want, sizes = O.qlt_encode_blocks(fq, qoff, qlen, level, br, rows)
    assert enc.stream("qlt") == want
    assert [b.size[2] for b in enc.blocks] == list(sizes)
    # the other streams are untouched by the prior: still the reference's per-chunk result
    chunks = util.split_records(fq, br)
    for b in (0, len(chunks) - 1):
        ref = util.block_reference(chunks[b], level, gen_bits=enc.blocks[b].gen_bits).streams
        assert enc.stream("gen", b) == ref["gen"] and enc.stream("rec", b) == ref["rec"]
    assert ctx.decode_host(enc, level=level, out_cap=len(fq) + 4096) == fq
    # and it pays: within a fraction of a percent of the whole-file (reference) quality stream
    whole = len(O.compress(fq, level).streams["qlt"])
    cold = ctx.encode_host(fq, level=level, block_reads=br, kernel=kernel).res.stream_bytes[2]
    assert len(want) < cold and len(want) < whole * 1.03


def test_warm_start_with_escapes_and_real_samples(ctx):
    for name in ("tst1", "fast5.to", "badqlt", "edge_hiq", "synthetic long reads"):
        fq = capi.synth_fastq(40, 150, seed=8, kind=1) if name.startswith("synthetic") else util.golden_fastq(name)
        nrec = fq.count(b"\n") // 4
        br = max(2, nrec // 9)
        enc = ctx.encode_host(fq, level=3, block_reads=br, prior_step=1)
        starts, lens = util.line_table(fq)
        solid = enc.blocks[0].solid
        qoff, qlen = starts[3::4] + solid, lens[3::4] - solid
        rows = O.qlt_prior_rows(O.qlt_histogram(fq, qoff, np.minimum(qlen, PRIOR_SYMBOLS), 3, 0, 1))
        want, _ = O.qlt_encode_blocks(fq, qoff, qlen, 3, br, rows)
        assert enc.stream("qlt") == want, name
        assert ctx.decode_host(enc, level=3, out_cap=2 * len(fq) + 4096) == fq, name


@pytest.mark.parametrize("kernel", KERNELS)
def test_headers_longer_than_one_wave_of_bytes(ctx, kernel):
    """64..126-byte headers take the two-bytes-per-lane path of the header kernel; longer ones its general path."""
    fq = capi.synth_fastq(2500, 100, seed=9, first_read=123_456_789)     # 68-byte headers
    assert max(len(l) for l in fq.split(b"\n")[0::4]) > 64
    enc = ctx.encode_host(fq, level=3, block_reads=0, kernel=kernel)
    assert_streams_equal(enc, O.compress(fq, 3).streams, ctxmsg="long headers")
    # mixed: some blocks short, some long, a shape change and a > 126-byte header in between
    lines = fq.split(b"\n")
    lines[4 * 700] = b"@" + b"x" * 140 + b":1:2"
    lines[4 * 1500] = b"@only_two fields"
    fq2 = b"\n".join(lines)
    enc = ctx.encode_host(fq2, level=3, block_reads=600, kernel=kernel)
    for b, chunk in enumerate(util.split_records(fq2, 600)):
        want = util.block_reference(chunk, 3, gen_bits=enc.blocks[b].gen_bits).streams
        assert_streams_equal(enc, want, block=b, ctxmsg="mixed headers block %d" % b)
    assert ctx.decode_host(enc, level=3, out_cap=len(fq2) + 4096) == fq2


def test_warm_start_prior_with_heavily_scaled_counts(ctx):
    """Big samples scale the prior's frequencies down to zero for rare symbols; the zero-frequency tail must
    have a canonical order (the prior is transmitted as its non-zero head only)."""
    fq = capi.synth_fastq(40000, 150, seed=77)
    enc = ctx.encode_host(fq, level=3, block_reads=2500, prior_step=1)
    starts, lens = util.line_table(fq)
    rows = O.qlt_prior_rows(O.qlt_histogram(fq, starts[3::4], np.minimum(lens[3::4], PRIOR_SYMBOLS), 3, 0, 1))
    assert int(((rows[:, :64] & 0xffff) == 0).sum()) > 0 and int(rows[:, 64].max()) > 30000
    assert np.array_equal(util.unpack_prior(enc.prior, 65536), rows)
    want, _ = O.qlt_encode_blocks(fq, starts[3::4], lens[3::4], 3, 2500, rows)
    assert enc.stream("qlt") == want
    assert ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq


def test_small_table_budget_forces_batches():
    """When the table budget holds fewer block slots than there are blocks, blocks run in batches that reuse
    the slots (epoch tags make the rows self-clearing; the Base2 tables are refilled per batch)."""
    c = capi.Context(0, table_budget=3 * (65536 * 272 + 1192 * 1040 + (4 << 18)) + (1 << 20))    # ~3 slots at level 3
    try:
        fq = capi.synth_fastq(4000, 150, seed=123)
        for kernel in KERNELS:
            enc = c.encode_host(fq, level=3, block_reads=400, kernel=kernel, prior_step=2)       # 10 blocks, 3-4 batches
            cold = c.encode_host(fq, level=3, block_reads=400, kernel=kernel)
            for b, chunk in enumerate(util.split_records(fq, 400)):
                want = util.block_reference(chunk, 3, gen_bits=cold.blocks[b].gen_bits).streams
                assert_streams_equal(cold, want, block=b, ctxmsg="batched kernel %d block %d" % (kernel, b))
            assert c.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq
            assert c.decode_host(cold, level=3, out_cap=len(fq) + 4096) == fq
    finally:
        c.close()


@pytest.mark.parametrize("kernel", KERNELS)
def test_escape_heavy_qualities(ctx, kernel):
    """Phred+64-style files: nearly every quality is an escape symbol (two coder steps each)."""
    fq = capi.synth_fastq(3000, 100, seed=5)
    lines = fq.split(b"\n")
    for i in range(3, len(lines), 4):
        lines[i] = bytes(min(c + 45, 126) for c in lines[i])
    fq = b"\n".join(lines)
    assert sum(c >= 96 for c in lines[3]) > 50
    enc = ctx.encode_host(fq, level=3, block_reads=1000, kernel=kernel)
    for b, chunk in enumerate(util.split_records(fq, 1000)):
        want = util.block_reference(chunk, 3, gen_bits=enc.blocks[b].gen_bits).streams
        assert_streams_equal(enc, want, block=b, ctxmsg="escape-heavy block %d" % b)
    assert ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq


def test_cli_slabs_make_segments_and_batch_mode(tmp_path):
    """Large inputs are compressed slab by slab (-S MiB), one archive segment per slab with its own quality prior;
    -b keeps one context for a list of jobs and reports per job."""
    import subprocess
    cli = _cli()
    fq = capi.synth_fastq(12000, 150, seed=21)                    # ~4.4 MB -> 5 slabs of 1 MiB
    src = tmp_path / "big.fq"; src.write_bytes(fq)
    sfq = tmp_path / "big.sfq"; out = tmp_path / "big.out"
    subprocess.check_call([cli, "-u", str(src), "-f", str(sfq), "-O", "-S", "1", "-B", "500"])
    p = subprocess.run([cli, "-s", "-f", str(sfq)], capture_output=True)
    assert b"seg.count" in p.stderr and b"seg.idx" in p.stderr
    subprocess.check_call([cli, "-d", "-f", str(sfq), "-u", str(out), "-O"])
    assert out.read_bytes() == fq
    # from stdin, slabs again
    p = subprocess.run([cli, "-f", str(tmp_path / "stdin.sfq"), "-O", "-S", "2"], input=fq, capture_output=True, check=True)
    p = subprocess.run([cli, "-d", "-f", str(tmp_path / "stdin.sfq")], capture_output=True, check=True)
    assert p.stdout == fq
    # one slab = no segment keys: the archive of a small file is unchanged
    one = tmp_path / "one.sfq"
    subprocess.check_call([cli, "-u", str(src), "-f", str(one), "-O", "-B", "500"])
    p = subprocess.run([cli, "-s", "-f", str(one)], capture_output=True)
    assert b"seg.count" not in p.stderr
    # batch mode: two good jobs and a bad one; the process survives the bad one
    small = tmp_path / "small.fq"; small.write_bytes(util.golden_fastq("small"))
    bad = tmp_path / "bad.fq"; bad.write_bytes(b"@x\nACXT\n+\nIIII\n")
    jobs = "%s\t%s\n%s\t%s\n%s\t%s\n" % (small, tmp_path / "b1.sfq", bad, tmp_path / "b2.sfq", src, tmp_path / "b3.sfq")
    p = subprocess.run([cli, "-b", "-O"], input=jobs.encode(), capture_output=True)
    lines = p.stdout.decode().splitlines()
    assert [l.split("\t")[0] for l in lines] == ["ok", "fail", "ok"] and p.returncode == 2
    p = subprocess.run([cli, "-b", "-d", "-O"], input=("%s\t%s\n" % (tmp_path / "b3.sfq", tmp_path / "b3.out")).encode(), capture_output=True)
    assert p.returncode == 0 and (tmp_path / "b3.out").read_bytes() == fq


def test_cli_batch_worker_survives_a_damaged_second_segment(tmp_path):
    """A -b worker decodes a multi-segment archive one segment at a time, the text of a segment written by a thread while
    the next is decoded.  A segment index that lies about its SECOND segment makes the job croak while that thread is still
    running: the worker must answer 'fail' for the job (the stack unwinds through guards that join the thread; an unjoined
    std::thread would call std::terminate) and go on to the next job."""
    import subprocess
    from slimfastq_amd import dist_compress as dc
    cli = _cli()
    fq = capi.synth_fastq(12000, 150, seed=23)
    src = tmp_path / "big.fq"; src.write_bytes(fq)
    good = tmp_path / "good.sfq"
    subprocess.check_call([cli, "-u", str(src), "-f", str(good), "-O", "-S", "1", "-B", "500", "-F"])
    a = O.parse(good.read_bytes())
    v = util._vints(a.streams["seg.idx"])
    assert v[0] >= 3 and len(v) == 1 + 5 * v[0]                 # nblocks, prior, raw, chain, rec.pri bytes per segment
    v[1 + 5 + 4] += 1 << 20                                      # the second segment claims a megabyte more of "rec.pri" than there is
    si = bytearray()
    for x in v:
        dc.put_v(si, x)
    info = "".join(l + "\n" for l in a.streams["<info>"].decode("latin1").split("\n") if l and not l.startswith("comp.size="))
    bad = tmp_path / "bad.sfq"
    dc.write_archive(str(bad), info, [(k, bytes(si) if k == "seg.idx" else d) for k, d in a.streams.items() if k != "<info>"])
    jobs = "%s\t%s\n%s\t%s\n" % (bad, tmp_path / "bad.out", good, tmp_path / "good.out")
    p = subprocess.run([cli, "-b", "-d", "-O"], input=jobs.encode(), capture_output=True)
    lines = p.stdout.decode().splitlines()
    assert [l.split("\t")[0] for l in lines] == ["fail", "ok"] and p.returncode == 2, (p.stdout, p.stderr)
    assert "segment index" in lines[0]
    assert (tmp_path / "good.out").read_bytes() == fq
    # one-shot mode: the same archive ends the process with the reference's wording and exit code, not a crash
    p = subprocess.run([cli, "-d", "-f", str(bad), "-u", str(tmp_path / "bad2.out"), "-O"], capture_output=True)
    assert p.returncode == 1 and b"slimfastq: decoding" in p.stderr and b"segment index" in p.stderr


def test_small_inputs_take_adaptive_tables_by_default(ctx, tmp_path):
    """SFQ_TABLES_AUTO (the CLI's default): frozen tables transmit their priors, which weigh too much on a small file (the
    reference's largest sample, tst7.fq, 3.9 MB: 1.15 x the reference's bytes) -- below 64 MiB of text the library codes with
    adaptive tables in blocks of 65536 records, i.e. the reference's own streams for such a file, and stays within 1 %."""
    import subprocess
    fq = util.golden_fastq("tst7")
    ref = O.compress(fq, 3)
    ref_bytes = ref.payload_bytes() - len(ref.streams["<info>"])
    enc = ctx.encode_host(fq, level=3, block_reads=capi.BLOCK_AUTO, prior_step=capi.PRIOR_AUTO, tables=capi.TABLES_AUTO)
    assert not enc.chains and not enc.prior and len(enc.blocks) == 1
    assert enc.archive_bytes <= 1.01 * ref_bytes
    assert ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq
    frozen = ctx.encode_host(fq, level=3, block_reads=capi.BLOCK_AUTO, prior_step=capi.PRIOR_AUTO, tables=capi.TABLES_FROZEN)
    assert frozen.chains and frozen.archive_bytes > 1.05 * ref_bytes          # what the default avoids
    # the CLI: the same decision by the file's size; -F forces frozen tables
    cli = _cli()
    src = tmp_path / "tst7.fq"; src.write_bytes(fq)
    a, f, out = tmp_path / "auto.sfq", tmp_path / "frozen.sfq", tmp_path / "back.fq"
    subprocess.check_call([cli, "-u", str(src), "-f", str(a), "-O", "-q"])
    subprocess.check_call([cli, "-u", str(src), "-f", str(f), "-O", "-q", "-F"])
    pa, pf = O.parse(a.read_bytes()), O.parse(f.read_bytes())
    assert "blk.tables" not in pa.info and pf.info.get("blk.tables") == "1" and "chn.idx" in pf.streams and "chn.idx" not in pa.streams
    assert sum(len(v) for k, v in pa.streams.items() if k != "<info>") <= 1.01 * ref_bytes
    for arch in (a, f):
        subprocess.check_call([cli, "-d", "-f", str(arch), "-u", str(out), "-O"])
        assert out.read_bytes() == fq


def test_multi_file_driver(tmp_path):
    """slimfastq_amd.multi (the reference's tools/slimfastq.multi): a directory of FASTQ files -> .sfq and back."""
    from slimfastq_amd import multi
    src = tmp_path / "FQ"; src.mkdir()
    files = {"a.fq": capi.synth_fastq(3000, 100, seed=1), "b.fastq": util.golden_fastq("tst1"), "c.fq": util.golden_fastq("solid"),
             "notes.txt": b"not a fastq\n"}
    for n, b in files.items():
        (src / n).write_bytes(b)
    assert multi.main(["-t", str(tmp_path / "SFQ"), "-c", "2", str(src)]) == 0
    assert sorted(p.name for p in (tmp_path / "SFQ").iterdir()) == ["a.sfq", "b.sfq", "c.sfq"]
    assert multi.main(["-d", "-t", str(tmp_path / "OUT"), str(tmp_path / "SFQ")]) == 0
    for n in ("a", "b", "c"):
        want = files[n + (".fastq" if n == "b" else ".fq")]
        assert (tmp_path / "OUT" / (n + ".fastq")).read_bytes() == want
    # existing targets are skipped unless -O; a broken file fails its job only
    (src / "bad.fq").write_bytes(b"@x\nACXT\n+\nIIII\n")
    assert multi.main(["-t", str(tmp_path / "SFQ"), str(src)]) == 2
    assert not (tmp_path / "SFQ" / "bad.sfq").exists()


def test_an_output_buffer_too_small_is_reported_and_left_untouched(ctx):
    """The streams are packed into the caller's buffer behind a check that runs on the device (frame.hip k_stream_gate: no host
    round trip between the chains and the packing): with too little room nothing is written, the call says how much it needs, and
    the context codes the next text as if nothing had happened."""
    import torch
    fq = capi.synth_fastq(40000, 150, seed=12)
    d_in = torch.from_numpy(np.frombuffer(fq, np.uint8).copy()).cuda()
    for tables in (capi.TABLES_FROZEN, capi.TABLES_ADAPTIVE):
        kw = dict(level=3, block_reads=1024, prior_step=capi.PRIOR_AUTO if tables else 0, tables=tables)
        cap = capi.lib().sfq_encode_bound(len(fq))
        d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
        res = ctx.encode_device(d_in.data_ptr(), len(fq), d_out.data_ptr(), cap, **kw)
        torch.cuda.synchronize()
        want = d_out[:res.total_bytes].clone()
        small = res.total_bytes - 1000
        d_small = torch.full((small,), 0xA5, dtype=torch.uint8, device="cuda")
        with pytest.raises(capi.SfqError) as e:
            ctx.encode_device(d_in.data_ptr(), len(fq), d_small.data_ptr(), small, **kw)
        assert e.value.code == -5 and "output needs" in str(e.value)            # SFQ_E_OVERFLOW
        torch.cuda.synchronize()
        assert bool((d_small == 0xA5).all())
        res2 = ctx.encode_device(d_in.data_ptr(), len(fq), d_out.data_ptr(), cap, **kw)
        torch.cuda.synchronize()
        assert res2.total_bytes == res.total_bytes and torch.equal(d_out[:res.total_bytes], want)


# BASELINE configs at their full per-GPU sizes: C3 (10 M x 150 bp, -l 3), C4's per-GPU share (100 M reads over 8 GPUs = 12.5 M,
# -l 4), binned qualities at -l 4, C5 long reads, the genome-sampled secondary workload; with frozen tables (the default
# of block format 7) and with adaptive tables (a wavefront per block)
@pytest.mark.parametrize("kind,n,level,tables", [(0, 10_000_000, 3, 1), (0, 12_500_000, 4, 1), (2, 4_000_000, 4, 1), (1, 20_000, 3, 1),
                                                 (3, 2_000_000, 3, 1), (0, 10_000_000, 3, 0), (0, 12_500_000, 4, 0), (1, 20_000, 3, 0)])
def test_baseline_sizes_round_trip_and_invariants(ctx, kind, n, level, tables):
    """BASELINE-size inputs, text resident in HBM.  The oracle cannot run these in seconds, so the checks are the
    size-independent ones: decode(encode(x)) == x, the block and chain indexes add up, a second encode gives the same
    bytes, the quality-only entry point gives the same stream."""
    import torch
    fq = capi.synth_fastq(n, 150, seed=3, kind=kind)
    nbytes = len(fq)
    d_in = torch.from_numpy(np.frombuffer(fq, np.uint8).copy()).cuda()
    if not (tables and kind == 0 and n == 10_000_000):
        fq = None                                 # (the BASELINE headline call keeps its text: sampled chains go to the oracle below)
    cap = capi.lib().sfq_encode_bound(nbytes)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    kw = dict(level=level, block_reads=capi.BLOCK_AUTO if kind == 1 else 1024, prior_step=capi.PRIOR_AUTO, tables=tables)
    res = ctx.encode_device(d_in.data_ptr(), nbytes, d_out.data_ptr(), cap, **kw)
    torch.cuda.synchronize()
    blocks = ctx.index(res.n_blocks)
    first = ctx.first_headers(res.first_hdr_bytes)
    prior, chains, rec_prior = ctx.prior(), ctx.chains(), ctx.rec_prior()
    assert res.n_records == n and sum(b.n_records for b in blocks) == n
    if kind != 1:
        assert res.n_blocks == (n + 1023) // 1024
    for s in range(capi.NSTREAMS):
        assert sum(b.size[s] for b in blocks) == res.stream_bytes[s]
    assert sum(res.stream_bytes) == res.total_bytes and all(b.status == 0 for b in blocks)
    if tables:
        ci = util.unpack_chains(chains, res.n_blocks)
        assert bool(ci["flags"] & 8) == (kind == 1)               # long reads: chains that are segments of one record
        assert len(ci["qlt"]) == res.n_chains and int(ci["qlt"].sum()) == res.stream_bytes[2] and int(ci["gen"].sum()) == res.stream_bytes[1]
        assert int(ci["rec"].sum()) == res.stream_bytes[0] and int(ci["rec_hdr_bytes"].sum()) == sum(b.hdr_bytes for b in blocks)
        if kind == 3:
            assert ci["flags"] & 1 and res.stream_bytes[1] * 8 < 1.7 * n * 150        # the generation tables learn the genome
        if kind == 0:
            assert not ci["flags"] & 1                                                  # iid bases: nothing to learn
    else:
        assert not chains and not rec_prior
    if kind == 0:
        assert 4.5 < nbytes / res.total_bytes < 5.5
    packed = d_out[:res.total_bytes].clone()
    soff, sbytes = list(res.stream_offset), list(res.stream_bytes)
    if fq is not None:
        _check_sampled_chains_against_oracle(fq, packed, soff, ci, prior, rec_prior, blocks, level)
        fq = None
    # same input, same bytes (atomic counting passes and table reuse under new epochs included)
    res2 = ctx.encode_device(d_in.data_ptr(), nbytes, d_out.data_ptr(), cap, **kw)
    torch.cuda.synchronize()
    assert res2.total_bytes == res.total_bytes and torch.equal(d_out[:res.total_bytes], packed)
    assert ctx.prior() == prior and ctx.chains() == chains and ctx.rec_prior() == rec_prior
    # the quality model alone (BASELINE config C2) writes the same quality stream
    res3 = ctx.encode_device(d_in.data_ptr(), nbytes, d_out.data_ptr(), cap, qlt_only=True, **kw)
    torch.cuda.synchronize()
    assert res3.stream_bytes[2] == sbytes[2]
    assert torch.equal(d_out[res3.stream_offset[2]: res3.stream_offset[2] + sbytes[2]], packed[soff[2]: soff[2] + sbytes[2]])
    # and back
    d_back = torch.empty(nbytes + 4096, dtype=torch.uint8, device="cuda")
    got, _ = ctx.decode_device(blocks, first, packed.data_ptr(), soff, d_back.data_ptr(), d_back.numel(), prior=prior, level=level,
                               chains=chains, rec_prior=rec_prior)
    torch.cuda.synchronize()
    assert got == nbytes and torch.equal(d_back[:nbytes], d_in)


def _check_sampled_chains_against_oracle(fq, packed, soff, ci, prior, rec_prior, blocks, level, per_stream=64):
    """The call is too big for the oracle, a chain is not: 64 random chains per stream of the 10 M-read call, each coded by
    the oracle from the call's own priors (a chain needs its records, the frozen rows and -- headers -- its block's first
    header, nothing else) and compared byte for byte with the GPU's."""
    rng = np.random.default_rng(2026)
    starts, lens = util.line_table(fq)
    nrec = len(starts) // 4
    br = blocks[0].n_records
    cr, rcr = ci["chain_reads"], ci["rec_chain_reads"]
    cpb, rcpb = -(-br // cr), -(-br // rcr)

    def chain_records(c, per_block, reads):
        b, j = divmod(c, per_block)
        r0 = b * br + j * reads
        return b * br, r0, min(r0 + reads, (b + 1) * br, nrec)
    qrows = O.qlt_frozen_rows(util.unpack_prior(prior, 4096 if level == 1 else 65536))
    qoffs = np.concatenate(([0], np.cumsum(ci["qlt"], dtype=np.int64)))
    goffs = np.concatenate(([0], np.cumsum(ci["gen"], dtype=np.int64)))
    for c in sorted(set(rng.integers(0, len(ci["qlt"]), per_stream).tolist()) | {0, len(ci["qlt"]) - 1}):
        _, r0, r1 = chain_records(c, cpb, cr)
        lo, hi = int(starts[4 * r0]), int(starts[4 * r1]) if r1 < nrec else len(fq)
        sub = fq[lo:hi]
        so, sl = util.line_table(sub)
        want, sizes, _ = O.qlt_encode_chains(sub, so[3::4], sl[3::4], level, r1 - r0, r1 - r0, qrows)
        got = bytes(packed[soff[2] + int(qoffs[c]): soff[2] + int(qoffs[c + 1])].cpu().numpy())
        assert got == want, ("qlt chain", c)
        assert not ci["flags"] & 1                                    # iid bases: every base chain codes with the initial row
        assert ci["flags"] & 192 == 128                               # ... two bits each, no coder (round 5b, block format 10)
        want, sizes, on = O.gm_encode_chains(sub, so[1::4], sl[1::4], 16, r1 - r0, r1 - r0)
        got = bytes(packed[soff[1] + int(goffs[c]): soff[1] + int(goffs[c + 1])].cpu().numpy())
        assert on == 0 and got == want, ("gen chain", c)
    rrows = O.rec_frozen_rows(util.unpack_rec_prior(rec_prior))
    roffs = np.concatenate(([0], np.cumsum(ci["rec"], dtype=np.int64)))
    for c in sorted(set(rng.integers(0, len(ci["rec"]), per_stream).tolist()) | {0, len(ci["rec"]) - 1}):
        b0, r0, r1 = chain_records(c, rcpb, rcr)
        lo, hi = int(starts[4 * r0]), int(starts[4 * r1]) if r1 < nrec else len(fq)
        sub = fq[lo:hi]
        if r0 != b0:                                                  # the chain starts from its block's first header
            sub = fq[int(starts[4 * b0]): int(starts[4 * b0 + 4])] + sub
        so, sl = util.line_table(sub)
        k = len(so) // 4
        want, sizes, hb = O.rec_encode_chains_frozen(sub, so[0::4] + 1, sl[0::4] - 1, k, k, rrows)
        got = bytes(packed[soff[0] + int(roffs[c]): soff[0] + int(roffs[c + 1])].cpu().numpy())
        assert got == want, ("rec chain", c)


@pytest.mark.parametrize("level", (1, 2, 3, 4))
def test_compressed_size_within_one_percent_of_the_reference_at_every_level(ctx, level):
    """BASELINE.json north_star: "compression ratio within 1 % of the reference at each -l level".  200 k synthetic reads:
    the reference's streams (the oracle, stream-identical to it) against everything a decoder of the block format needs
    (streams, first headers, priors, chain and block index); profiles/r03_ratio_table.json holds the larger table."""
    fq = capi.synth_fastq(200_000, 150, seed=17)
    ref = O.compress(fq, level)
    ref_bytes = ref.payload_bytes() - len(ref.streams["<info>"])
    enc = ctx.encode_host(fq, level=level, block_reads=capi.BLOCK_AUTO, prior_step=capi.PRIOR_AUTO, tables=capi.TABLES_FROZEN)
    assert enc.archive_bytes <= 1.01 * ref_bytes, (level, enc.archive_bytes, ref_bytes)
    assert ctx.decode_host(enc, level=level, out_cap=len(fq) + 4096) == fq


@pytest.mark.parametrize("level", (1, 2, 3, 4))
def test_bases_that_can_be_learned_come_out_no_larger_than_the_references(ctx, level):
    """reads whose bases can be learned -- every real FASTQ -- at every level: 500 k reads sampled from a 10 Mbp
    genome (7.5-fold coverage; the reference's base table has 2^18 ... 2^26 contexts by level).  Everything a decoder of the block
    format needs against the reference's streams: at most 1.01 x (round 5: the match model reads its predictions from the earlier
    reads themselves, gm.hip -- well under 1.00 x at every level), and the way back."""
    fq = capi.synth_fastq(500_000, 150, seed=23, kind=3)
    ref = O.compress(fq, level)
    ref_bytes = ref.payload_bytes() - len(ref.streams["<info>"])
    enc = ctx.encode_host(fq, level=level, block_reads=capi.BLOCK_AUTO, prior_step=capi.PRIOR_AUTO, tables=capi.TABLES_FROZEN)
    assert util.unpack_chains(enc.chains)["flags"] & 32                  # the match model is on
    assert enc.archive_bytes <= 1.01 * ref_bytes, (level, enc.archive_bytes, ref_bytes)
    assert len(enc.stream("gen")) < len(ref.streams["gen"])
    assert ctx.decode_host(enc, level=level, out_cap=len(fq) + 4096) == fq


def test_two_ranks_compress_one_file_into_one_archive(tmp_path):
    """slimfastq_amd.dist_compress: each rank codes its record-aligned share of the file, rank 0 gathers and writes one
    archive with a segment per rank; the CLI restores the file.  Two ranks share this box's one GPU, so the exchange
    runs over gloo here (RCCL needs a GPU per rank; the call pattern is the same)."""
    import subprocess, sys, os
    fq = capi.synth_fastq(9000, 150, seed=31) + util.golden_fastq("tst1")
    src = tmp_path / "in.fq"; src.write_bytes(fq)
    sfq = tmp_path / "out.sfq"; back = tmp_path / "back.fq"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29533", "-m", "slimfastq_amd.dist_compress", str(src), str(sfq), "-B", "700"],
                   check=True, cwd=root, env=env, timeout=600)
    p = subprocess.run([_cli(), "-s", "-f", str(sfq)], capture_output=True)
    assert b"seg.count" in p.stderr and b"= 2" in p.stderr
    subprocess.check_call([_cli(), "-d", "-f", str(sfq), "-u", str(back), "-O"])
    assert back.read_bytes() == fq
    # a rank's range in slabs of 1 MiB: every slab a segment, the file's one prior in the first
    import re
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29534", "-m", "slimfastq_amd.dist_compress", str(src), str(sfq), "-B", "700", "-S", "1"],
                   check=True, cwd=root, env=env, timeout=600)
    p = subprocess.run([_cli(), "-s", "-f", str(sfq)], capture_output=True)
    m = re.search(rb"seg\.count\s*=\s*(\d+)", p.stderr)
    assert m and int(m.group(1)) == 4, p.stderr[-400:]
    subprocess.check_call([_cli(), "-d", "-f", str(sfq), "-u", str(back), "-O"])
    assert back.read_bytes() == fq
    # adaptive tables (-A) on two ranks (the summed counts then hold no header sample, and none is needed)
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29535", "-m", "slimfastq_amd.dist_compress", str(src), str(sfq), "-B", "700", "-A"],
                   check=True, cwd=root, env=env, timeout=600)
    subprocess.check_call([_cli(), "-d", "-f", str(sfq), "-u", str(back), "-O"])
    assert back.read_bytes() == fq
    # one rank: a plain one-segment archive
    subprocess.run([sys.executable, "-m", "slimfastq_amd.dist_compress", str(src), str(tmp_path / "one.sfq")], check=True, cwd=root, env=env, timeout=600)
    p = subprocess.run([_cli(), "-d", "-f", str(tmp_path / "one.sfq")], capture_output=True, check=True)
    assert p.stdout == fq


def _fuzz_fastq(rng, nrec):
    """A structurally hostile little FASTQ: ragged lengths, qlen != llen, N with and without quality '!', quality '!' under
    real bases, escape qualities (>= 63), lowercase bases, header fields that grow, shrink, turn hexadecimal, get leading
    zeros, change their separators, and a second id on the '+' line for the whole file or not at all."""
    two_id = rng.random() < 0.3
    hdr_shapes = [lambda i, a, b: "r%d:%d:%d" % (i, a, b), lambda i, a, b: "run_%x/%d %05d" % (a, i, b % 9973),
                  lambda i, a, b: "X%d.%d-%X" % (i * 7, 1000 - (i % 900), b), lambda i, a, b: "id%d" % i,
                  lambda i, a, b: "m%d/%d/%d_%d" % (54006 + (i // 50), a % 97, b % 5000, i)]
    shape = hdr_shapes[rng.integers(len(hdr_shapes))]
    base_len = int(rng.integers(1, 260))
    lines = []
    a, b = int(rng.integers(1000)), int(rng.integers(100000))
    for i in range(nrec):
        if rng.random() < 0.03:
            shape = hdr_shapes[rng.integers(len(hdr_shapes))]                      # the header's shape changes
        a += int(rng.integers(-3, 40)); b = int(rng.integers(100000)) if rng.random() < 0.5 else b + 1
        hdr = shape(i, max(a, 0), b)
        n = base_len if rng.random() < 0.85 else int(rng.integers(1, 300))
        bases = rng.choice(list("ACGT"), n)
        if rng.random() < 0.2:
            bases = np.char.lower(bases) if rng.random() < 0.5 else bases
        qual = np.clip(rng.normal(30, 9, n).astype(int) + (rng.random(n) < 0.02) * 40, 0, 93)
        nmask = rng.random(n) < 0.03
        bases = np.where(nmask, "N", bases)
        qual = np.where(nmask & (rng.random(n) < 0.7), 0, qual)                    # most Ns carry quality '!' ...
        qual = np.where(~nmask & (rng.random(n) < 0.01), 0, qual)                  # ... and so do a few real bases
        qs = "".join(chr(33 + int(q)) for q in qual)
        if rng.random() < 0.04 and n > 2:
            qs = qs[: int(rng.integers(1, n))]                                     # a quality line shorter than its bases
        lines += ["@" + hdr, "".join(bases), "+" + (hdr if two_id else ""), qs]
    return ("\n".join(lines) + "\n").encode()


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_small_structurally_hostile_inputs(ctx, seed):
    """Random small inputs through the default kernels: one block (== the reference run) and ragged blocks, every
    stream against the oracle, and back."""
    rng = np.random.default_rng(1000 + seed)
    for rep in range(5):
        nrec = int(rng.integers(1, 400))
        fq = _fuzz_fastq(rng, nrec)
        level = int(rng.integers(1, 5))
        want = O.compress(fq, level)
        if want is None or not want.streams:
            continue
        enc = ctx.encode_host(fq, level=level, block_reads=0)
        assert_streams_equal(enc, want.streams, ctxmsg="fuzz seed %d rep %d one block" % (seed, rep))
        for kernel in KERNELS:
            assert ctx.decode_host(enc, level=level, out_cap=2 * len(fq) + 4096, kernel=kernel) == O.decompress(want.image), kernel
        br = int(rng.integers(1, max(2, nrec // 2 + 1)))
        enc = ctx.encode_host(fq, level=level, block_reads=br)
        for b, chunk in enumerate(util.split_records(fq, br)):
            wantb = util.block_reference(chunk, level, gen_bits=enc.blocks[b].gen_bits).streams
            assert_streams_equal(enc, wantb, block=b, ctxmsg="fuzz seed %d rep %d block %d of %d" % (seed, rep, b, br))
        for kernel in KERNELS:
            assert ctx.decode_host(enc, level=level, out_cap=2 * len(fq) + 4096, kernel=kernel) == fq, kernel     # the block format is lossless


@pytest.mark.parametrize("kernel", (1,))
def test_fuzz_other_kernel_variants(ctx, kernel):
    """The same hostile inputs through the kernel variants kept for A/B runs: they must write the default kernels' bytes."""
    rng = np.random.default_rng(77)
    for rep in range(3):
        nrec = int(rng.integers(20, 300))
        fq = _fuzz_fastq(rng, nrec)
        br = int(rng.integers(5, 60))
        for prior_step in (0, 1):
            ref = ctx.encode_host(fq, level=3, block_reads=br, prior_step=prior_step)
            enc = ctx.encode_host(fq, level=3, block_reads=br, prior_step=prior_step, kernel=kernel)
            assert bytes(enc.data) == bytes(ref.data) and enc.prior == ref.prior, "kernel %d rep %d prior %d" % (kernel, rep, prior_step)
            assert [list(b.size) for b in enc.blocks] == [list(b.size) for b in ref.blocks]


def test_automatic_block_size_follows_the_text(ctx):
    """SFQ_BLOCK_AUTO sizes blocks by bytes of text (about 376 KiB): 1024 records of 150 bp, a handful of long reads --
    a fixed record count would leave a long-read file with a few huge blocks and the chip idle."""
    fq = capi.synth_fastq(5000, 150, seed=4)
    enc = ctx.encode_host(fq, level=3, block_reads=capi.BLOCK_AUTO, prior_step=capi.PRIOR_AUTO)
    assert enc.blocks[0].n_records == 1024 and len(enc.blocks) == 5
    assert ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq
    lr = capi.synth_fastq(300, 150, seed=4, kind=1)                   # 10-50 kb reads
    enc = ctx.encode_host(lr, level=3, block_reads=capi.BLOCK_AUTO, prior_step=capi.PRIOR_AUTO)
    assert 2 <= enc.blocks[0].n_records <= 16 and len(enc.blocks) >= 300 // 16
    assert ctx.decode_host(enc, level=3, out_cap=len(lr) + 4096) == lr


def _pad_to(records, j, want_mod, modulus, line):
    """Lengthen the headers of the records before j so that line `line` (0 = header, 2 = '+') of record j starts at an offset
    that is want_mod modulo `modulus`; returns the text."""
    def start(recs):
        off = sum(len(r) for r in recs[:j])
        if line == 2:
            parts = recs[j].split(b"\n")
            off += len(parts[0]) + 1 + len(parts[1]) + 1
        return off
    need = (want_mod - start(records)) % modulus
    recs = list(records)
    k = 0
    while need:
        add = min(need, 60)
        h, rest = recs[k].split(b"\n", 1)
        recs[k] = h + b"x" * add + b"\n" + rest
        need -= add; k += 1
        assert k < j
    assert start(recs) % modulus == want_mod
    return recs


def test_line_prefixes_are_checked_wherever_a_line_starts(ctx):
    """The framing pass (frame.hip k_frame) checks '@' / '+' (usrs.cpp:311, 346) on the byte behind every line end -- from the
    window's own bytes, or, where a line starts exactly with a 64-byte window, a 16 KiB sub-tile or a 128 KiB tile, from what the
    thread, wave, sub-tile or tile before hands over.  A wrong prefix at each of those places is refused; the untouched text codes."""
    base = [b"@r%d:%d\n%s\n+\n%s\n" % (i, i * 7, b"ACGT" * 25, b"I" * 100) for i in range(2500)]
    for modulus in (64, 4096, 16384, 131072):
        for line in (0, 2):
            j = 2000 if modulus == 131072 else 700
            recs = _pad_to(base, j, 0, modulus, line)
            good = b"".join(recs)
            enc = ctx.encode_host(good, level=3, block_reads=500, prior_step=capi.PRIOR_AUTO, tables=capi.TABLES_FROZEN)
            assert ctx.decode_host(enc, level=3, out_cap=len(good) + 4096) == good
            off = sum(len(r) for r in recs[:j])
            if line == 2:
                parts = recs[j].split(b"\n")
                off += len(parts[0]) + 1 + len(parts[1]) + 1
            assert off % modulus == 0 and good[off:off + 1] == (b"+" if line else b"@")
            bad = bytearray(good); bad[off] = ord("X")
            with pytest.raises(capi.SfqError) as e:
                ctx.encode_host(bytes(bad), level=3, block_reads=500, prior_step=capi.PRIOR_AUTO, tables=capi.TABLES_FROZEN)
            assert e.value.code == -4, (modulus, line, e.value)
            # ... and one byte further on, inside the window
            recs2 = _pad_to(base, j, 1, modulus, line)
            bad = bytearray(b"".join(recs2))
            off2 = sum(len(r) for r in recs2[:j])
            if line == 2:
                parts = recs2[j].split(b"\n")
                off2 += len(parts[0]) + 1 + len(parts[1]) + 1
            assert off2 % modulus == 1 and bytes(bad[off2:off2 + 1]) == (b"+" if line else b"@")
            bad[off2] = ord("X")
            with pytest.raises(capi.SfqError) as e:
                ctx.encode_host(bytes(bad), level=3, block_reads=0)
            assert e.value.code == -4
    # the file's first byte
    bad = bytearray(b"".join(base)); bad[0] = ord("X")
    with pytest.raises(capi.SfqError) as e:
        ctx.encode_host(bytes(bad), level=3, block_reads=500, tables=capi.TABLES_ADAPTIVE)
    assert e.value.code == -4
