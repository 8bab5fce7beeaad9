"""Block format 7 with FROZEN tables (sfq_params.tables = 1): one chain per lane, rows built by counting passes.
Not reference behaviour -- the rule is restated in oracle/sfq_oracle.c and the GPU bytes must equal it chain by chain;
the context functions, alphabets, header model and the range coder's arithmetic are the reference's (pinned elsewhere)."""
import numpy as np
import pytest

from slimfastq_amd import capi
from oracle import oracle as O
import util

pytestmark = pytest.mark.gpu
PRIOR_SYMBOLS = 4096
GEN_STEP = 4


def rec_sample(nrec, max_hdr=0):
    run = 6                                       # api.cpp REC_PRIOR_RUN / REC_PRIOR_RUNS
    nruns = min(32768, max(1, nrec // run))
    if max_hdr > 127:                             # headers beyond the fast counting kernel: at most 4 MiB of header text
        nruns = min(nruns, max(1, (4 << 20) // (run * max_hdr)))
    return max(run, nrec // nruns), run, nruns


def gm_table_bits(nbytes):
    tb = 16                                       # api.cpp gm_table_bits: an entry per eight bases of the call, 2^16 .. 2^24 entries
    while tb < 24 and (1 << tb) < nbytes // 16:
        tb += 1
    return tb


def base_chains_oracle(fq, goff, glen, ci, br, cr, seg=0, other=None, quads=False):
    """The base chains as the oracle's restatement of the generation MATCH model (gm.hip; sfq_oracle.c sfqo_gm_*) writes them, and
    its verdict; "chn.idx" must say the same (flag bit 0: the model is on; bit 5 + the index's bits: it is the match model)."""
    tb = gm_table_bits(len(fq))
    if seg:
        want, sizes, on = O.gm_encode_segs(fq, goff, glen, other, tb, br, seg)
        gcr = 1
    else:
        # the base chains of a call that takes the model are cut shorter than the quality chains (api.cpp): 2 KiB of text each or more (about 819 200 a call), of equal
        # length inside a block; a call that does not take it keeps the quality chains' records
        nrec = len(goff)
        per = max(1, len(fq) // max(1, nrec))
        gcpb = -(-br // min(max(-(-2048 // per), -(-nrec // 819200)), cr))
        gcr = max(1, -(-min(br, nrec) // gcpb))
        if gcr >= cr:
            gcr = cr
        want, sizes, on = O.gm_encode_chains(fq, goff, glen, tb, br, gcr)
        if not on:
            want, sizes, on = O.gm_encode_chains(fq, goff, glen, tb, br, cr)
            gcr = cr
    assert (ci["flags"] & 1) == on and bool(ci["flags"] & 32) == bool(on)
    # no model: the bases two bits each, no coder (block format 10) -- or format 9's four a symbol, on request
    assert ci["flags"] & 192 == (0 if on else 64 if quads else 128)
    assert not on or ci["gm_table_bits"] == tb
    assert ci["gen_chain_reads"] == gcr
    return want, sizes, on


def check_against_oracle(ctx, fq, level, br, cr, step, what="", quads=False):
    enc = ctx.encode_host(fq, level=level, block_reads=br, prior_step=step, tables=capi.TABLES_FROZEN, chain_reads=cr)
    starts, lens = util.line_table(fq)
    nrec = len(starts) // 4
    solid = enc.blocks[0].solid
    ci = util.unpack_chains(enc.chains)
    got_cr, flags, qsz, gsz = ci["chain_reads"], ci["flags"], ci["qlt"], ci["gen"]
    assert got_cr == min(cr, br)
    # qualities: prior -> frozen rows -> chains
    qoff, qlen = starts[3::4] + solid, lens[3::4] - solid
    rows66 = O.qlt_prior_rows(O.qlt_histogram(fq, qoff, np.minimum(qlen, PRIOR_SYMBOLS), level, 0, step))
    assert np.array_equal(util.unpack_prior(enc.prior, 4096 if level == 1 else 65536), rows66), what
    want, sizes, extra = O.qlt_encode_chains(fq, qoff, qlen, level, br, got_cr, O.qlt_frozen_rows(rows66))
    assert list(qsz) == list(sizes), what
    assert enc.stream("qlt") == want, what
    assert sum(b.extra_hi for b in enc.blocks) == extra
    # bases: generation tables
    goff, glen = starts[1::4] + solid, lens[1::4] - solid
    want, sizes, on = base_chains_oracle(fq, goff, glen, ci, br, got_cr, quads=quads)
    assert list(gsz) == list(sizes), what
    assert enc.stream("gen") == want, what
    # headers: counted sample -> "rec.pri" -> frozen rows -> one chain per block
    hoff, hlen = starts[0::4] + 1, lens[0::4] - 1
    stride, run, nruns = rec_sample(nrec, int(hlen.max()) if len(hlen) else 0)
    f = O.rec_prior_freqs(O.rec_count(fq, hoff, hlen, stride, run, nruns))
    assert np.array_equal(util.unpack_rec_prior(enc.rec_prior), f), what
    assert flags & 2
    rcr = ci["rec_chain_reads"]
    nblocks = -(-nrec // br)
    cpb_want = max(1, 61440 // nblocks)
    rfloor = 16 if len(fq) // nrec > 4000 else 128                            # (long reads: short header chains)
    assert rcr == min(max(rfloor, -(-br // cpb_want), got_cr), br, nrec)          # api.cpp: header chains
    want, sizes, hb = O.rec_encode_chains_frozen(fq, hoff, hlen, br, rcr, O.rec_frozen_rows(f))
    assert list(ci["rec"]) == list(sizes) and list(ci["rec_hdr_bytes"]) == list(hb), what
    assert enc.stream("rec") == want, what
    # the framing's side streams are the reference's own, block by block; the base exceptions are the reference's LISTS (gaps as
    # gens.cpp:91-114 defines them) as Rice codes -- the oracle's restatement of exc.hip ("chn.idx" flag bit 4)
    assert flags & 16
    chunks = util.split_records(fq, br)
    for b in (0, len(chunks) - 1):
        ref = util.block_reference(chunks[b], level, gen_bits=enc.blocks[b].gen_bits).streams
        for name in ("usr.x", "usr.x.q", "usr.pfg", "usr.pfq"):
            assert enc.stream(name, b) == ref.get(name, b""), (what, name, b)
        rice = util.exc_rice_reference(chunks[b], solid=int(enc.blocks[b].solid))
        for name in ("gen.Ns", "gen.Nn", "gen.lc"):
            assert enc.stream(name, b) == rice[name], (what, name, b)
        assert enc.stream("rec.x", b) == b""             # a header whose shape changed is coded inside its chain
    return enc


def test_base_exceptions_in_the_references_own_coding_on_request(ctx):
    """sfq_params.kernel = 2 with frozen tables: the base exceptions go through the reference's adaptive PowerRanger rows (what
    rounds 2 and 3 wrote; "chn.idx" flag bit 4 clear) -- byte for byte the reference's gen.Ns / gen.Nn of a file holding the
    block -- and such an archive decodes; the default (Rice-coded lists, exc.hip) holds the same positions in fewer bytes."""
    fq = capi.synth_fastq(5000, 150, seed=31)
    br = 500
    old = ctx.encode_host(fq, level=3, block_reads=br, prior_step=2, tables=capi.TABLES_FROZEN, chain_reads=64, kernel=2)
    new = ctx.encode_host(fq, level=3, block_reads=br, prior_step=2, tables=capi.TABLES_FROZEN, chain_reads=64)
    assert not util.unpack_chains(old.chains)["flags"] & 16 and util.unpack_chains(new.chains)["flags"] & 16
    for b, chunk in enumerate(util.split_records(fq, br)):
        ref = util.block_reference(chunk, 3, gen_bits=old.blocks[b].gen_bits).streams
        for name in ("gen.Ns", "gen.Nn", "gen.lc"):
            assert old.stream(name, b) == ref.get(name, b""), (name, b)
            got = O.exc_rice_decode(new.stream(name, b))
            assert len(got) or not ref.get(name, b"")
    for name in ("rec", "qlt", "usr.x", "usr.x.q"):
        assert old.stream(name) == new.stream(name)
    # (the bases of a call without a model: round 4's 3 of 12 a base under kernel = 2; two bits each without a coder -- "chn.idx" flag bit 7, block format
    #  10 -- now; round 5a's four bases a symbol, flag bit 6, is still read: test_flat_bases_of_format_9_are_still_read)
    assert not util.unpack_chains(old.chains)["flags"] & 192 and util.unpack_chains(new.chains)["flags"] & 192 == 128
    assert len(new.stream("gen")) <= len(old.stream("gen"))
    assert sum(len(new.stream(n)) for n in ("gen.Ns", "gen.Nn")) < sum(len(old.stream(n)) for n in ("gen.Ns", "gen.Nn"))
    assert ctx.decode_host(old, level=3, out_cap=len(fq) + 4096) == fq
    assert ctx.decode_host(new, level=3, out_cap=len(fq) + 4096) == fq


def _folded_genome_reads(n=60000, seed=5):
    """Reads of a tiny genome (300 reads of a 10 Mbp one, rotated): enough coverage at test size for the base model to pay."""
    fq = capi.synth_fastq(n, 150, seed=seed, kind=3)
    lines = fq.split(b"\n")[:-1]
    reads = [lines[i + 1] for i in range(0, len(lines), 4)][:300]
    rng = np.random.default_rng(3)
    out = []
    for i in range(0, len(lines), 4):
        src = reads[rng.integers(len(reads))]
        k = int(rng.integers(0, 40))
        out += [lines[i], src[k:] + src[:k], lines[i + 2], lines[i + 3]]
    return b"\n".join(out) + b"\n"


def test_round_4_generation_tables_are_still_written_on_request_and_read(ctx):
    """sfq_params.kernel = 2 also keeps round 4's base model -- generation tables of Base2 rows, counted per generation (through the
    LDS bins of round 5: the same counts) -- so that what archives of rounds 2-4 hold can still be produced and is still read:
    the chains against the oracle's restatement of THAT rule, "chn.idx" without flag bit 5, and the way back."""
    fq = _folded_genome_reads()
    br, cr = 128, 32
    old = ctx.encode_host(fq, level=3, block_reads=br, prior_step=1, tables=capi.TABLES_FROZEN, chain_reads=cr, kernel=2)
    ci = util.unpack_chains(old.chains)
    assert ci["flags"] & 1 and not ci["flags"] & 32
    starts, lens = util.line_table(fq)
    want, sizes, on = O.gen_encode_chains(fq, starts[1::4], lens[1::4], old.blocks[0].gen_bits, br, cr, GEN_STEP)
    assert on == 1 and list(ci["gen"]) == list(sizes) and old.stream("gen") == want
    assert ctx.decode_host(old, level=3, out_cap=len(fq) + 4096) == fq
    new = ctx.encode_host(fq, level=3, block_reads=br, prior_step=1, tables=capi.TABLES_FROZEN, chain_reads=cr)
    assert util.unpack_chains(new.chains)["flags"] & 32
    # (which of the two is smaller here says nothing: these reads start at 12 000 distinct places, so the table knows even a read's
    #  first bases; reads that start anywhere -- tests/test_gpu_parity.py's ratio test, bench.py's genome leg -- are the comparison)


@pytest.mark.parametrize("level", (1, 2, 3, 4))
def test_frozen_streams_equal_oracle_rule_synthetic(ctx, level):
    fq = capi.synth_fastq(7000, 150, seed=70 + level)
    enc = check_against_oracle(ctx, fq, level, br=500, cr=64, step=2)
    assert ctx.decode_host(enc, level=level, out_cap=len(fq) + 4096) == fq


@pytest.mark.parametrize("level", (1, 3))
def test_frozen_histogram_of_a_few_hot_contexts(ctx, level):
    """4-level binned qualities put nearly all symbols of the sample on a handful of (context, symbol) pairs: every lane of a
    histogram workgroup queues on the same few LDS slots, whose 10-bit counts are moved on 256 at a time (prior.hip hist_add).
    The prior must be the oracle's count for count -- and the same on every run."""
    fq = capi.synth_fastq(40000, 150, seed=9, kind=2)
    enc = check_against_oracle(ctx, fq, level, br=1024, cr=64, step=1, what="binned qualities")
    for _ in range(3):
        again = ctx.encode_host(fq, level=level, block_reads=1024, prior_step=1, tables=capi.TABLES_FROZEN, chain_reads=64)
        assert again.prior == enc.prior and again.stream("qlt") == enc.stream("qlt")


def test_frozen_generation_tables_switch_on_for_genome_like_bases(ctx):
    """Reads sampled from a small genome: generation 1 is cheaper under generation 0's rows, the tables switch on, and the
    base stream gets well below 2 bits per base; iid bases leave them off."""
    fq = capi.synth_fastq(60000, 150, seed=5, kind=3)
    fq2 = _folded_genome_reads()
    enc = check_against_oracle(ctx, fq2, 3, br=128, cr=32, step=1, what="genome-like")
    ci = util.unpack_chains(enc.chains)
    assert ci["flags"] & 1
    assert int(ci["gen"].sum()) * 8 < 1.2 * 60000 * 150
    assert ctx.decode_host(enc, level=3, out_cap=len(fq2) + 4096) == fq2
    enc = ctx.encode_host(fq, level=3, block_reads=128, prior_step=1, tables=capi.TABLES_FROZEN, chain_reads=32)
    assert not util.unpack_chains(enc.chains)["flags"] & 1
    assert ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq


def test_frozen_big_generations_are_counted_through_every_nth_record(ctx):
    """A generation of more than GEN_COUNT_CAP (524 288) records is counted through every ceil(n / cap)-th of them
    (kernels.h gen_count_stride; the decoder counts the same records): 2.4 M reads sampled from a 10 Mbp genome put the
    last counted generation at ~600 k records (stride 2).  The base chains against the oracle's restatement, then back."""
    n = 2_400_000
    fq = capi.synth_fastq(n, 100, seed=11, kind=3)
    br, cr = 1024, 64
    enc = ctx.encode_host(fq, level=3, block_reads=br, prior_step=capi.PRIOR_AUTO, tables=capi.TABLES_FROZEN, chain_reads=cr)
    ci = util.unpack_chains(enc.chains)
    assert ci["flags"] & 1 and ci["chain_reads"] == cr
    nblocks = -(-n // br)
    bound = [0]; b = max(1, -(-nblocks // 64))
    while b < nblocks and len(bound) + 1 < 40:
        bound.append(b); b = max(b + 1, b * 2)
    bound.append(nblocks)
    assert max((bound[g + 1] - bound[g]) * br for g in range(len(bound) - 2)) > 524288        # a counted generation over the cap
    starts, lens = util.line_table(fq)
    want, sizes, on = base_chains_oracle(fq, starts[1::4], lens[1::4], ci, br, cr)
    assert on == 1
    assert list(ci["gen"]) == list(sizes)
    assert enc.stream("gen") == want
    assert ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq


def test_frozen_pre_verdict_leaves_the_tables_off_for_bases_that_cannot_be_learned(ctx):
    """A generation 0 of 16384 counted records or more is first looked at through every 8th record (api.cpp gen_tables_begin: over
    the bases whose context the sample has seen, does the sample's count of THIS base beat a quarter?): uniform bases end there,
    tables off, the chains the oracle's; the genome-sampled reads of the test above pass it and go on to the full verdict."""
    n = 1_200_000
    fq = capi.synth_fastq(n, 50, seed=17, kind=0)
    br, cr = 1024, 64
    assert (-(-n // br) // 64) * br >= 16384
    enc = ctx.encode_host(fq, level=3, block_reads=br, prior_step=capi.PRIOR_AUTO, tables=capi.TABLES_FROZEN, chain_reads=cr)
    ci = util.unpack_chains(enc.chains)
    starts, lens = util.line_table(fq)
    want, sizes, on = base_chains_oracle(fq, starts[1::4], lens[1::4], ci, br, cr)
    assert on == 0 and not (ci["flags"] & 1)
    assert list(ci["gen"]) == list(sizes)
    assert enc.stream("gen") == want
    assert ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq


def test_frozen_counting_passes_take_long_lines_a_stretch_per_lane(ctx):
    """Lines over 1 KiB: the base tables' counting passes give a lane a stretch of 512 bases (sixteen bases of warm-up in
    front of it) instead of a whole record; the counts -- and so every chain's bytes -- must not depend on the split."""
    fq = capi.synth_fastq(30000, 3000, seed=13, kind=3)          # 3 kb reads sampled from the 10 Mbp genome: the tables switch on
    br, cr = 64, 8
    enc = ctx.encode_host(fq, level=3, block_reads=br, prior_step=capi.PRIOR_AUTO, tables=capi.TABLES_FROZEN, chain_reads=cr)
    ci = util.unpack_chains(enc.chains)
    starts, lens = util.line_table(fq)
    assert int(lens[1::4].max()) > 1024
    want, sizes, on = base_chains_oracle(fq, starts[1::4], lens[1::4], ci, br, cr)
    assert on == 1 and (ci["flags"] & 1)
    assert list(ci["gen"]) == list(sizes)
    assert enc.stream("gen") == want
    assert ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq


def test_match_model_archives_that_are_damaged_fail_cleanly(ctx):
    """Untrusted input on the way back through the match model's decoder (its index is built from what it decodes, its pointers
    come out of that index): flipped bytes in the base stream, a chain index that lies about the base chains or the index's bits.
    Every decode ends in an SfqError or in some bytes -- never in a device fault -- and the context decodes the intact archive afterwards."""
    import random
    rnd = random.Random(5)
    fq = _folded_genome_reads(20000)
    enc = ctx.encode_host(fq, level=3, block_reads=128, prior_step=1, tables=capi.TABLES_FROZEN, chain_reads=32)
    ci = util.unpack_chains(enc.chains)
    assert ci["flags"] & 32
    s = capi.STREAM_NAMES.index("gen")
    g0, g1 = int(enc.res.stream_offset[s]), int(enc.res.stream_offset[s]) + int(enc.res.stream_bytes[s])
    outcomes = {"error": 0, "bytes": 0}
    for trial in range(40):
        bad = enc.clone()
        if trial % 2 == 0:                                     # flipped bytes in the base stream: wrong bases, wrong k-mers, wrong pointers
            data = bytearray(bad.data)
            for _ in range(rnd.randint(1, 20)):
                data[rnd.randrange(g0, g1)] ^= 1 << rnd.randrange(8)
            bad.data = bytes(data)
        else:                                                  # the chain index's head: index bits, base chains' geometry, chain counts
            ch = bytearray(bad.chains)
            ch[rnd.randrange(0, 8)] ^= 1 << rnd.randrange(7)
            bad.chains = bytes(ch)
        try:
            ctx.decode_host(bad, level=3, out_cap=2 * len(fq) + 4096)
            outcomes["bytes"] += 1
        except capi.SfqError:
            outcomes["error"] += 1
    assert outcomes["error"] > 0
    assert ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq


def test_match_model_with_chains_that_are_segments(ctx):
    """The match model (gm.hip) where chains are SEGMENTS of one record: 3 kb reads sampled from the 10 Mbp genome, cut into segments of
    700 symbols -- a segment starts as a line does (no pointer, no k-mer), the plan's lanes are the segments, the decoder's chains too.
    Every segment's bytes against the oracle's, and the way back."""
    fq = capi.synth_fastq(30000, 3000, seed=13, kind=3)
    br, seg = 64, 700
    enc = ctx.encode_host(fq, level=3, block_reads=br, prior_step=capi.PRIOR_AUTO, tables=capi.TABLES_FROZEN, chain_reads=SEG | seg)
    starts, lens = util.line_table(fq)
    nblocks = -(-(len(starts) // 4) // br)
    ci = util.unpack_chains(enc.chains, nblocks)
    assert ci["flags"] & 8 and ci["flags"] & 32 and ci["seg_len"] == seg
    want, sizes, on = base_chains_oracle(fq, starts[1::4], lens[1::4], ci, br, 1, seg, lens[3::4])
    assert on == 1 and list(ci["gen"]) == list(sizes) and enc.stream("gen") == want
    assert ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq


def test_match_model_in_colour_space(ctx):
    """SOLiD reads ("0123" behind a primer base, usrs.cpp:186-267) whose colours repeat: the staged letters are digits, their codes the
    low two bits (gm.hip gm_code), a '.' is the N.  Against the oracle, and back."""
    fq = _folded_genome_reads(40000)
    tr = bytes.maketrans(b"ACGTN", b"0123.")
    lines = fq.split(b"\n")[:-1]
    out = []
    for i in range(0, len(lines), 4):
        out += [lines[i], b"T" + lines[i + 1].translate(tr), lines[i + 2], b"!" + lines[i + 3]]
    cs = b"\n".join(out) + b"\n"
    br, cr = 128, 32
    enc = ctx.encode_host(cs, level=3, block_reads=br, prior_step=1, tables=capi.TABLES_FROZEN, chain_reads=cr)
    assert enc.blocks[0].solid == 1
    ci = util.unpack_chains(enc.chains)
    assert ci["flags"] & 32
    starts, lens = util.line_table(cs)
    want, sizes, on = base_chains_oracle(cs, starts[1::4] + 1, lens[1::4] - 1, ci, br, cr)
    assert on == 1 and list(ci["gen"]) == list(sizes) and enc.stream("gen") == want
    assert ctx.decode_host(enc, level=3, out_cap=len(cs) + 4096) == cs


@pytest.mark.parametrize("name", util.golden_names())
def test_frozen_golden_samples(ctx, name):
    fq = util.golden_fastq(name)
    nrec = fq.count(b"\n") // 4
    br = max(2, nrec // 7)
    if name == "edge_oversize":
        # the block format has no oversize streams: base / quality lines of any length are coded the usual way
        # (test_reads_beyond_the_reference_line_limit); a header over 8190 bytes (usrs.hpp:34) is refused
        with pytest.raises(capi.SfqError) as e:
            ctx.encode_host(fq, level=3, block_reads=br, prior_step=1, tables=capi.TABLES_FROZEN)
        assert e.value.code == -7
        recs = util.split_records(fq, 1)
        fq = b"".join(r for r in recs if len(r.split(b"\n")[0]) < 8000)
        enc = ctx.encode_host(fq, level=3, block_reads=br, prior_step=1, tables=capi.TABLES_FROZEN, chain_reads=1)
        assert ctx.decode_host(enc, level=3, out_cap=2 * len(fq) + 4096) == fq
        return
    # the block format is lossless, also where the reference is not (edge_lower, badsprintf, edge_hdr: SURVEY H7)
    enc = check_against_oracle(ctx, fq, 3, br=br, cr=max(1, br // 3), step=1, what=name)
    assert ctx.decode_host(enc, level=3, out_cap=2 * len(fq) + 4096) == fq, name


def test_frozen_ragged_chains_and_long_reads(ctx):
    fq = capi.synth_fastq(90, 150, seed=8, kind=1)           # 10-50 kb reads
    for br, cr in ((7, 2), (1, 1), (90, 90), (16, 5)):
        enc = check_against_oracle(ctx, fq, 3, br=br, cr=cr, step=1, what="long %d/%d" % (br, cr))
        assert ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq
    fq = capi.synth_fastq(3001, 37, seed=9)
    enc = check_against_oracle(ctx, fq, 3, br=1000, cr=33, step=1)
    assert ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq


def test_frozen_automatic_parameters_round_trip(ctx):
    fq = capi.synth_fastq(40000, 150, seed=12)
    enc = ctx.encode_host(fq, level=3, block_reads=capi.BLOCK_AUTO, prior_step=capi.PRIOR_AUTO, tables=capi.TABLES_FROZEN)
    assert enc.chains and enc.rec_prior and enc.prior
    assert ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq
    # the frozen rows code about as well as the adaptive ones
    ada = ctx.encode_host(fq, level=3, block_reads=capi.BLOCK_AUTO, prior_step=capi.PRIOR_AUTO)
    assert enc.archive_bytes < 1.02 * ada.archive_bytes


def test_given_prior_makes_block_bytes_independent_of_the_sharding(ctx):
    """Multi-GPU contract: with ONE prior for the job (built by rank 0, installed everywhere) a record block's quality and
    header bytes are the same whether one call coded the whole text or two calls coded its halves (what two ranks do)."""
    import torch
    fq = capi.synth_fastq(8000, 150, seed=77)
    halves = util.split_records(fq, 4000)
    d = torch.from_numpy(np.frombuffer(halves[0], np.uint8).copy()).cuda()
    prior, rp = ctx.build_priors(d.data_ptr(), len(halves[0]), level=3, block_reads=500, tables=capi.TABLES_FROZEN)
    assert prior and rp
    ctx.set_priors(prior, rp)
    whole = ctx.encode_host(fq, level=3, block_reads=500, prior_step=capi.PRIOR_GIVEN, tables=capi.TABLES_FROZEN, chain_reads=50)
    assert whole.prior == prior and whole.rec_prior == rp
    parts = [ctx.encode_host(h, level=3, block_reads=500, prior_step=capi.PRIOR_GIVEN, tables=capi.TABLES_FROZEN, chain_reads=50) for h in halves]
    for name in ("qlt", "rec", "gen"):               # (iid bases: the generation tables stay off in either case)
        assert whole.stream(name) == parts[0].stream(name) + parts[1].stream(name), name
    assert ctx.decode_host(whole, level=3, out_cap=len(fq) + 4096) == fq
    ctx.set_priors(b"", b"")


def test_summed_counts_give_every_rank_the_same_priors(ctx):
    """Multi-GPU priors without a serial head (sfq_count_priors / sfq_set_prior_counts, prior_step = SFQ_PRIOR_COUNTS): two
    "ranks" (two contexts on this GPU) count the sample of their halves of a file, the count tables are added up (what the
    all-reduce of dist.allreduce_prior_counts does), both install the sums and code their halves: the priors they build are
    identical, a block's quality and header bytes are what ONE call over the whole text with the same counts writes, and the
    halves decode.  The counts themselves are the oracle's histogram over every second sampled record of a half."""
    import torch
    fq = capi.synth_fastq(8000, 150, seed=78)
    halves = util.split_records(fq, 4000)
    other = capi.Context(0, table_budget=8 << 30)
    try:
        ranks = [ctx, other]
        nq, nr = capi.Context.prior_counts_words(3)
        qs, rs, dins = [], [], []
        for c, h in zip(ranks, halves):
            d = torch.from_numpy(np.frombuffer(h, np.uint8).copy()).cuda()
            dins.append(d)
            c.count_priors(d.data_ptr(), len(h), level=3, block_reads=500, prior_step=1, tables=capi.TABLES_FROZEN, sample_scale=2)
            q = torch.empty(nq, dtype=torch.int32, device="cuda"); r = torch.empty(nr, dtype=torch.int32, device="cuda")
            c.get_prior_counts(3, q.data_ptr(), r.data_ptr())
            starts, lens = util.line_table(h)
            want = O.qlt_histogram(h, starts[3::4], np.minimum(lens[3::4], PRIOR_SYMBOLS), 3, 0, 2)
            assert np.array_equal(q.cpu().numpy().view(np.uint32), want)
            qs.append(q); rs.append(r)
        qsum, rsum = qs[0] + qs[1], rs[0] + rs[1]
        torch.cuda.synchronize()
        encs = []
        for c, h, d in zip(ranks, halves, dins):
            c.set_prior_counts(3, qsum.data_ptr(), rsum.data_ptr())
            encs.append(c.encode_host(h, level=3, block_reads=500, prior_step=capi.PRIOR_COUNTS, tables=capi.TABLES_FROZEN, chain_reads=50))
        assert encs[0].prior == encs[1].prior and encs[0].rec_prior == encs[1].rec_prior and encs[0].prior
        ctx.set_prior_counts(3, qsum.data_ptr(), rsum.data_ptr())
        whole = ctx.encode_host(fq, level=3, block_reads=500, prior_step=capi.PRIOR_COUNTS, tables=capi.TABLES_FROZEN, chain_reads=50)
        assert whole.prior == encs[0].prior and whole.rec_prior == encs[0].rec_prior
        for name in ("qlt", "rec", "gen"):
            assert whole.stream(name) == encs[0].stream(name) + encs[1].stream(name), name
        for c, h, e in zip(ranks, halves, encs):
            assert c.decode_host(e, level=3, out_cap=len(h) + 4096) == h
        with pytest.raises(capi.SfqError):                    # a fresh context has no counts to code from
            fresh = capi.Context(0, table_budget=4 << 30)
            try:
                fresh.encode_host(halves[0], level=3, block_reads=500, prior_step=capi.PRIOR_COUNTS, tables=capi.TABLES_FROZEN)
            finally:
                fresh.close()
    finally:
        other.close()


@pytest.mark.parametrize("level", (1, 3))
def test_frozen_quality_rows_staged_in_lds_do_not_change_a_byte(ctx, level):
    """sfq_params.lds_rows: every workgroup of the quality chains keeps the N rows the sample saw most in LDS (picked on the
    device, chains.hip k_hot_pick) and reads the others from the table -- where a row comes from must not show in the stream."""
    fq = capi.synth_fastq(20000, 150, seed=21)
    kw = dict(level=level, block_reads=500, prior_step=capi.PRIOR_AUTO, tables=capi.TABLES_FROZEN, chain_reads=25)
    base = ctx.encode_host(fq, lds_rows=capi.LDS_ROWS_NONE, **kw)
    for rows in (0, 1, 64, 240, 5000):                  # 0: the automatic choice (800 rows)
        enc = ctx.encode_host(fq, lds_rows=rows, **kw)
        assert enc.stream("qlt") == base.stream("qlt"), rows
        assert enc.chains == base.chains and enc.prior == base.prior, rows
    assert ctx.decode_host(enc, level=level, out_cap=len(fq) + 4096) == fq
    for rows in (1, 300, 5000):                         # the decoder's own staging (rows picked by the prior's weights)
        assert ctx.decode_host(enc, level=level, out_cap=len(fq) + 4096, lds_rows=rows) == fq, rows


def _odd_headers_fastq(n, seed):
    """Headers that leave the fast header kernels' envelope now and then: longer than 127 bytes, more than 16 fields, a
    number of twenty digits (prints with a sign through "%lld": the reference is lossy there, the block format codes it as a
    string), a field that turns hexadecimal, a leading zero, an empty field (the reference gives "0" back) -- between runs
    of ordinary ones."""
    rng = np.random.default_rng(seed)
    out = []
    x = 1000
    for i in range(n):
        x += int(rng.integers(0, 50))
        k = rng.random()
        if k < 0.02:
            hdr = "@long.%d %s:%d" % (i, "Z" * int(rng.integers(120, 300)), x)
        elif k < 0.04:
            hdr = "@many.%d " % i + ":".join(str(int(v)) for v in rng.integers(0, 99, int(rng.integers(17, 40))))
        elif k < 0.06:
            hdr = "@big.%d run:%d:%d" % (i, 18446744073709551000 + int(rng.integers(0, 600)), x)
        elif k < 0.08:
            hdr = "@hex.%d run:%x:%d" % (i, 0xabc000 + i, x)
        elif k < 0.10:
            hdr = "@zero.%d run:0%d::%d" % (i, i, x)
        else:
            hdr = "@SIM.%d M7:12:FC9:%d:%d:%d:%d 1:N:0:ACGT" % (i, 1 + i // 2000, 1100 + i // 500, x, int(rng.integers(1000, 30000)))
        ln = 60
        seq = "".join("ACGT"[int(v)] for v in rng.integers(0, 4, ln))
        q = "".join(chr(33 + int(v)) for v in rng.integers(2, 41, ln))
        out += [hdr, seq, "+", q]
    return ("\n".join(out) + "\n").encode()


def test_frozen_headers_outside_the_fast_kernels_envelope(ctx):
    """The fast header kernels (k_rec_encode_f / k_rec_decode_f: 127 bytes, 16 fields, no signs) hand a chain with anything
    else to the general kernels; the bytes must be the oracle's either way, and the text must come back EXACTLY -- also the
    twenty-digit numbers and the empty fields the reference itself mangles."""
    fq = _odd_headers_fastq(6000, 5)
    br, cr = 400, 50
    enc = check_against_oracle(ctx, fq, 3, br=br, cr=cr, step=1, what="odd headers")
    assert O.decompress(O.compress(fq, 3).image) != fq                # (the reference is lossy on this input)
    assert ctx.decode_host(enc, level=3, out_cap=2 * len(fq) + 4096) == fq
    # the same through adaptive tables (a wavefront per block; the lane-per-block cross-check kernels): lossless too, and
    # every block the oracle's
    for kernel in (0, 1):
        ada = ctx.encode_host(fq, level=3, block_reads=br, tables=capi.TABLES_ADAPTIVE, kernel=kernel)
        for b, chunk in enumerate(util.split_records(fq, br)):
            want = util.block_reference(chunk, 3, gen_bits=ada.blocks[b].gen_bits).streams
            for name in capi.STREAM_NAMES:
                assert ada.stream(name, b) == want.get(name, b""), (kernel, b, name)
        assert ctx.decode_host(ada, level=3, out_cap=2 * len(fq) + 4096) == fq, kernel
    # and with the ordinary headers only: everything on the fast path, exact
    lines = fq.split(b"\n")[:-1]
    keep = [lines[i:i + 4] for i in range(0, len(lines), 4) if lines[i].startswith(b"@SIM.")]
    fq2 = b"\n".join(b"\n".join(r) for r in keep) + b"\n"
    enc = check_against_oracle(ctx, fq2, 3, br=br, cr=cr, step=1, what="plain headers")
    assert ctx.decode_host(enc, level=3, out_cap=2 * len(fq2) + 4096) == fq2


def _field_history_fastq(n, seed):
    """Headers built for the token step of the header chains (chains.hip k_rec_tokens: a record per lane, 64 at a time): what a
    changed field codes depends on whether the field has changed since the chain began or the shape last changed, on the number
    in the previous header, and on nothing else -- unless a field turns hexadecimal, which leaves the chain to the lane kernels."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        a = 1000 + i // 70                                               # still for 70 records: its first change meets a cold field type, across a round of 64 lanes
        b = ("x%dy" % i) if (i // 45) % 3 == 1 else str(50000 - 3 * i if i < 900 else i)     # a number, a string for a while, a number again; falling, then rising
        c = ("%x" % (0xabc000 + i)) if 1300 <= i < 1330 else str(i * 7)  # hexadecimal in one stretch only
        d = "007" if i % 97 == 0 else str(i % 13)                        # leading zeros now and then
        tail = (" " + "Z" * 80 + str(i)) if 1700 <= i < 1900 and i % 2 else ""       # the shape alternates over long headers: more symbols than the token buffer has room for
        e = 4_000_000_000_000 + (i % 5) * 7_000_000_000 - (i % 3) * 70_000_000     # gaps over 2^32 and over 2^24, up and down
        z = ("%X" % (0xAB00 + i)) if i % 2 else "0"                                 # upper-case hex against a bare 0
        hdr = "@H%d:%s:%s:%s:%d:%d:%s%s" % (a, b, c, d, int(rng.integers(0, 100000)), e, z, tail)
        ln = 40
        seq = "".join("ACGT"[int(v)] for v in rng.integers(0, 4, ln))
        q = "".join(chr(33 + int(v)) for v in rng.integers(2, 41, ln))
        out.append("%s\n%s\n+\n%s\n" % (hdr, seq, q))
    return "".join(out).encode()


@pytest.mark.parametrize("br,cr", ((512, 64), (2400, 300), (200, 10)))
def test_frozen_header_tokens_follow_the_field_history(ctx, br, cr):
    fq = _field_history_fastq(2400, 11)
    enc = check_against_oracle(ctx, fq, 3, br=br, cr=cr, step=1, what="field history, blocks of %d" % br)
    assert ctx.decode_host(enc, level=3, out_cap=2 * len(fq) + 4096) == fq


@pytest.mark.parametrize("seed", range(5))
def test_frozen_fuzz_structurally_hostile_inputs(ctx, seed):
    """The hostile little FASTQs of test_gpu_parity (ragged lengths, header shapes that change, hex / leading-zero / shrinking
    fields, escapes, N with and without '!', lowercase bases, short quality lines, a second id) through the frozen-table
    chains: every prior, every chain and every side stream against the oracle's restatement, and back to the input, exactly."""
    from test_gpu_parity import _fuzz_fastq
    rng = np.random.default_rng(4000 + seed)
    for rep in range(3):
        nrec = int(rng.integers(150, 2500))
        fq = _fuzz_fastq(rng, nrec)
        level = int(rng.integers(1, 5))
        br = int(rng.integers(40, min(700, nrec) + 1))
        cr = int(rng.integers(1, br + 1))
        what = "fuzz seed %d rep %d: %d records, level %d, blocks of %d, chains of %d" % (seed, rep, nrec, level, br, cr)
        enc = check_against_oracle(ctx, fq, level, br=br, cr=cr, step=1, what=what)
        assert ctx.decode_host(enc, level=level, out_cap=2 * len(fq) + 4096) == fq, what


def test_installed_counts_are_tracked_and_a_refilled_buffer_is_framed_again(ctx):
    """(a) SFQ_PRIOR_COUNTS codes only from counts that sfq_count_priors / sfq_set_prior_counts left -- what an
    ordinary encode's own sample leaves in the same buffers is not "installed", nor are counts of another level; (b) counts taken for
    ADAPTIVE tables hold no header sample: they travel as zeros, serve an adaptive encode and are refused by a frozen one;
    (c) the line index sfq_count_priors leaves for the encode that follows is reused only while the TEXT at that address is the
    one it indexed: a buffer refilled in place (same pointer, same size) is framed again."""
    import torch
    fq = capi.synth_fastq(6000, 150, seed=91)
    d = torch.from_numpy(np.frombuffer(fq, np.uint8).copy()).cuda()
    out = torch.empty(capi.lib().sfq_encode_bound(len(fq)), dtype=torch.uint8, device="cuda")
    nq, nr = capi.Context.prior_counts_words(3)
    q = torch.empty(nq, dtype=torch.int32, device="cuda"); r = torch.empty(nr, dtype=torch.int32, device="cuda")
    kw = dict(level=3, block_reads=500)
    # (a)
    ctx.count_priors(d.data_ptr(), len(fq), prior_step=1, tables=capi.TABLES_FROZEN, **kw)
    ctx.get_prior_counts(3, q.data_ptr(), r.data_ptr())
    assert int(r.sum().item()) > 0
    ctx.encode_device(d.data_ptr(), len(fq), out.data_ptr(), out.numel(), prior_step=capi.PRIOR_COUNTS, tables=capi.TABLES_FROZEN, **kw)
    ctx.encode_device(d.data_ptr(), len(fq), out.data_ptr(), out.numel(), prior_step=capi.PRIOR_AUTO, tables=capi.TABLES_FROZEN, **kw)   # an ordinary sample
    with pytest.raises(capi.SfqError):
        ctx.encode_device(d.data_ptr(), len(fq), out.data_ptr(), out.numel(), prior_step=capi.PRIOR_COUNTS, tables=capi.TABLES_FROZEN, **kw)
    with pytest.raises(capi.SfqError):
        ctx.get_prior_counts(3, q.data_ptr(), r.data_ptr())
    ctx.set_prior_counts(3, q.data_ptr(), r.data_ptr())
    with pytest.raises(capi.SfqError):                        # level 1 has 4096 quality contexts: these counts are not its
        ctx.encode_device(d.data_ptr(), len(fq), out.data_ptr(), out.numel(), level=1, block_reads=500, prior_step=capi.PRIOR_COUNTS, tables=capi.TABLES_FROZEN)
    # (b)
    ctx.count_priors(d.data_ptr(), len(fq), prior_step=1, tables=capi.TABLES_ADAPTIVE, **kw)
    ctx.get_prior_counts(3, q.data_ptr(), r.data_ptr())
    assert int(r.abs().sum().item()) == 0 and int(q.sum().item()) > 0
    ctx.set_prior_counts(3, q.data_ptr(), r.data_ptr())
    with pytest.raises(capi.SfqError):
        ctx.encode_device(d.data_ptr(), len(fq), out.data_ptr(), out.numel(), prior_step=capi.PRIOR_COUNTS, tables=capi.TABLES_FROZEN, **kw)
    res = ctx.encode_device(d.data_ptr(), len(fq), out.data_ptr(), out.numel(), prior_step=capi.PRIOR_COUNTS, tables=capi.TABLES_ADAPTIVE, **kw)
    back = torch.empty(len(fq) + 4096, dtype=torch.uint8, device="cuda")
    got, _ = ctx.decode_device(ctx.index(res.n_blocks), ctx.first_headers(res.first_hdr_bytes), out.data_ptr(), list(res.stream_offset), back.data_ptr(), back.numel(),
                               prior=ctx.prior(), level=3)
    assert got == len(fq) and bytes(back[:got].cpu().numpy()) == fq
    # (c) the same bytes in another record structure: two records of 150 bp become one of 300
    lines = fq.split(b"\n")
    other = b"\n".join(lines[0:1] + [lines[1] + lines[5]] + lines[2:3] + [lines[3] + lines[7]] + lines[8:])
    pad = len(fq) - len(other)
    assert 0 < pad < 200
    hdr = lines[0] + b"x" * pad                       # (the first header grows by what the dropped lines took: same size)
    other = hdr + other[len(lines[0]):]
    assert len(other) == len(fq)
    ctx.count_priors(d.data_ptr(), len(fq), prior_step=1, tables=capi.TABLES_FROZEN, **kw)
    ctx.get_prior_counts(3, q.data_ptr(), r.data_ptr())
    d.copy_(torch.from_numpy(np.frombuffer(other, np.uint8).copy()))
    torch.cuda.synchronize()
    ctx.set_prior_counts(3, q.data_ptr(), r.data_ptr())
    res = ctx.encode_device(d.data_ptr(), len(fq), out.data_ptr(), out.numel(), prior_step=capi.PRIOR_COUNTS, tables=capi.TABLES_FROZEN, **kw)
    assert res.n_records == 5999
    got, _ = ctx.decode_device(ctx.index(res.n_blocks), ctx.first_headers(res.first_hdr_bytes), out.data_ptr(), list(res.stream_offset), back.data_ptr(), back.numel(),
                               prior=ctx.prior(), level=3, chains=ctx.chains(), rec_prior=ctx.rec_prior())
    assert got == len(other) and bytes(back[:got].cpu().numpy()) == other


SEG = 0x80000000


def _long_reads(rng, n, lo, hi, qdiff=False):
    recs = []
    for i in range(n):
        L = int(rng.integers(lo, hi))
        seq = "".join(rng.choice(list("ACGT"), L, p=[0.3, 0.2, 0.2, 0.3]))
        if i % 7 == 3:
            seq = seq[:50] + "N" * 5 + seq[55:]
        ql = L if not (qdiff and i % 5 == 2) else max(1, L - int(rng.integers(1, 40)))
        q = np.clip(np.cumsum(rng.integers(-2, 3, ql)) + 20, 1, 60)
        qual = bytes((q + 33).astype(np.uint8)).decode()
        recs.append("@%08x-%04x-4a%02x_read%d ch=%d\n%s\n+\n%s\n" % (int(rng.integers(0, 2**32)), i * 7 % 65536, i % 256, i, i % 512, seq, qual))
    return "".join(recs).encode()


@pytest.mark.parametrize("seg,br,qdiff", ((1000, 4, False), (700, 7, True), (4096, 3, False)))
def test_long_records_are_cut_into_segments(ctx, seg, br, qdiff):
    """Chains that are SEGMENTS of one record (BASELINE config 5: reads of 10-50 kb; SFQ_CHAIN_SEGMENT, chains.hip): every
    segment's quality and base bytes against the oracle's restatement of the rule -- a record of M = max(bases, qualities) symbols
    in ceil(M / seg) equal segments, each starting as a line does --, the chain index's shape, and the way back; with quality lines
    shorter than their base lines in between (the segments follow the longer of the two)."""
    rng = np.random.default_rng(seg + br)
    fq = _long_reads(rng, 40, 2500, 9000, qdiff)
    enc = ctx.encode_host(fq, level=3, block_reads=br, prior_step=1, tables=capi.TABLES_FROZEN, chain_reads=SEG | seg)
    starts, lens = util.line_table(fq)
    nrec = len(starts) // 4
    nblocks = -(-nrec // br)
    ci = util.unpack_chains(enc.chains, nblocks)
    assert ci["flags"] & 8 and ci["seg_len"] == seg and ci["chain_reads"] == 1
    qoff, qlen, goff, glen = starts[3::4], lens[3::4], starts[1::4], lens[1::4]
    nseg = O.seg_counts(glen, qlen, seg)
    assert [int(nseg[b * br:(b + 1) * br].sum()) for b in range(nblocks)] == list(ci["seg_blocks"])
    assert enc.res.n_chains == int(nseg.sum()) and int(nseg.max()) >= 3
    rows66 = O.qlt_prior_rows(O.qlt_histogram(fq, qoff, np.minimum(qlen, PRIOR_SYMBOLS), 3, 0, 1))
    want, sizes, extra = O.qlt_encode_segs(fq, qoff, qlen, glen, 3, seg, O.qlt_frozen_rows(rows66))
    assert list(ci["qlt"]) == list(sizes) and enc.stream("qlt") == want
    want, sizes, on = base_chains_oracle(fq, goff, glen, ci, br, 1, seg, qlen)
    assert list(ci["gen"]) == list(sizes) and enc.stream("gen") == want
    assert ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq
    # the index is untrusted: a block's share that disagrees with its records' line lengths is refused
    bad = enc.clone()
    blob = bytearray(bad.chains)
    v = util._vints(bytes(blob))
    assert v[3] == seg
    # (rewrite the blob with one segment moved from block 0 to block 1)
    moved = list(v); moved[4] -= 1; moved[5] += 1
    out = bytearray()
    for x in moved:
        while x >= 0x80:
            out.append((x & 0x7f) | 0x80); x >>= 7
        out.append(x)
    bad.chains = bytes(out)
    with pytest.raises(capi.SfqError):
        ctx.decode_host(bad, level=3, out_cap=len(fq) + 4096)


def test_long_reads_take_segments_by_themselves(ctx):
    """The automatic choice: records of a chain's worth or more, fewer than 204 800 of them and a line longer than a segment --
    the call's symbols in about 204 800 segments of 4096 symbols or more; short reads stay with whole records."""
    fq = capi.synth_fastq(300, 150, seed=5, kind=1)                       # 10-50 kb reads
    enc = ctx.encode_host(fq, level=3, block_reads=capi.BLOCK_AUTO, prior_step=capi.PRIOR_AUTO, tables=capi.TABLES_FROZEN)
    ci = util.unpack_chains(enc.chains, len(enc.blocks))
    assert ci["flags"] & 8 and ci["seg_len"] == 4096 and enc.res.n_chains > 3 * 300
    assert ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq
    short = capi.synth_fastq(3000, 150, seed=5)
    enc = ctx.encode_host(short, level=3, block_reads=500, prior_step=capi.PRIOR_AUTO, tables=capi.TABLES_FROZEN)
    assert not util.unpack_chains(enc.chains)["flags"] & 8


def _headers_of_every_length(n, top, seed):
    """Headers whose lengths run through 1 .. top and whose fields end at every byte of a dword: the token step stages a header four bytes
    a time and keeps its field ends as a mask (chains.hip rt_stage) -- bit 63 / 64, the last dword's spare bytes, a header that ends on a
    separator, bytes over 0x7f, fields of ten digits and more, "00", a tab, an underscore, fields that change width."""
    rng = np.random.default_rng(seed)
    seps = [":", " ", ".", "_", "\t", "/", "-", "#", "=", "\xe9"]
    out = []
    for i in range(n):
        want = 1 + (i * 7 + i // 130) % top
        parts = ["r%d" % (i // 3)]
        k = 0
        while len("".join(parts)) < want:
            kind = (i + k) % 9
            if kind == 0: f = str(10 ** int(rng.integers(0, 12)) + i)              # one to twelve digits: the short way and the general one
            elif kind == 1: f = "00" + str(i % 7)
            elif kind == 2: f = "0" + str(1 + i % 9)
            elif kind == 3: f = "ab" if i % 2 else "AB"
            elif kind == 4: f = str(int(rng.integers(0, 100000)))
            elif kind == 5: f = ""                                                   # two separators in a row
            elif kind == 6: f = "x\xfcy"                                              # a byte over 0x7f inside a field: it ends one
            elif kind == 7: f = str(99999 - i) if i < 99999 else "7"
            else: f = "Z" * int(rng.integers(1, 9))
            parts.append(seps[(i // 11 + k) % len(seps)] + f)
            k += 1
            if k > 14: break                                                         # (sixteen fields at most stay on the token step's path)
        hdr = "".join(parts)[:want]
        ln = 30 + i % 5
        seq = "".join("ACGT"[int(v)] for v in rng.integers(0, 4, ln))
        q = "".join(chr(33 + int(v)) for v in rng.integers(2, 41, ln))
        out.append("@%s\n%s\n+\n%s\n" % (hdr, seq, q))
    return "".join(out).encode("latin-1")


@pytest.mark.parametrize("top", (61, 94, 127))
def test_frozen_header_tokens_at_every_header_length(ctx, top):
    fq = _headers_of_every_length(3000, top, 21 + top)
    enc = check_against_oracle(ctx, fq, 3, br=700, cr=90, step=1, what="headers of 1 .. %d bytes" % top)
    assert ctx.decode_host(enc, level=3, out_cap=2 * len(fq) + 4096) == fq


def _exception_heavy_fastq(n, seed):
    """Records for the exception pass (models_w.hip k_gen_exc_q: a marked record per lane, sixteen bytes a step, eight events a list): none, one, nine
    and ninety events a record; N, n and '.' under every kind of quality; '!' over real bases; lower-case runs; quality lines shorter than their bases
    (what lies behind them counts as 'I'); records over 1024 bases (the whole wave's walk) between short ones; lengths that end a sixteen-byte step early
    and late."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        ln = int(rng.integers(1, 60)) if i % 13 == 0 else (int(rng.integers(1025, 2500)) if i % 41 == 7 else int(rng.integers(90, 200)))
        b = np.array(list("ACGT"), dtype="U1")[rng.integers(0, 4, ln)]
        q = rng.integers(2, 41, ln) + 33
        kind = i % 8
        ev = 0 if kind == 0 else 1 if kind < 4 else 9 if kind < 7 else min(ln, 90)
        at = rng.choice(ln, size=min(ev, ln), replace=False)
        for j, p in enumerate(at):
            w = (i + j) % 6
            if w == 0: b[p] = "N"
            elif w == 1: b[p] = "N"; q[p] = 33
            elif w == 2: q[p] = 33
            elif w == 3: b[p] = str(b[p]).lower()
            elif w == 4: b[p] = "N"; q[p] = 34
            else: b[p] = str(b[p]).lower(); q[p] = 33
        if i % 17 == 3: b[: ln // 2] = np.char.lower(b[: ln // 2])
        ql = ln if i % 5 else max(1, ln - int(rng.integers(1, 20)))                 # a short quality line now and then
        out.append("@e%d x:%d\n%s\n+\n%s\n" % (i, i % 1000, "".join(b), "".join(chr(int(v)) for v in q[:ql])))
    return "".join(out).encode()


@pytest.mark.parametrize("br,cr", ((1024, 64), (150, 7)))
def test_exception_scan_takes_records_of_every_kind(ctx, br, cr):
    fq = _exception_heavy_fastq(4000, 33)
    enc = check_against_oracle(ctx, fq, 3, br=br, cr=cr, step=1, what="exception-heavy records, blocks of %d" % br)
    assert ctx.decode_host(enc, level=3, out_cap=2 * len(fq) + 4096) == fq
    # one N byte per block is the format's rule (gens.cpp:169): a block that holds both 'N' and '.' is refused, not mangled
    bad = fq.replace(b"N", b".", 1)
    if bad != fq and b"N" in bad:
        with pytest.raises(capi.SfqError):
            ctx.encode_host(bad, level=3, block_reads=len(fq), tables=capi.TABLES_FROZEN, chain_reads=cr)


def test_flat_bases_pack_two_bits_each(ctx):
    """A call whose bases have no model (iid bases: the match model's verdict says no) writes them without a coder -- block format 10, "chn.idx" flag bit 7:
    a chain is its bases' codes, four a byte, the first in the low bits, across its records' ends, the last byte padded with zeros; N-like bases code
    as 0 and come back through the exception lists.  Stated here in numpy, beside the oracle's C (check_against_oracle compares every chain with that)."""
    rng = np.random.default_rng(77)
    recs = []
    for i in range(3000):
        ln = int(rng.integers(1, 40)) if i % 11 == 0 else int(rng.integers(95, 160))
        b = np.array(list("ACGT"), dtype="U1")[rng.integers(0, 4, ln)]
        if i % 7 == 0: b[int(rng.integers(0, ln))] = "N"
        if i % 13 == 0: b[: ln // 3] = np.char.lower(b[: ln // 3])
        recs.append("@p%d\n%s\n+\n%s\n" % (i, "".join(b), "".join(chr(int(v)) for v in rng.integers(35, 74, ln))))
    fq = "".join(recs).encode()
    br, cr = 512, 37
    enc = check_against_oracle(ctx, fq, 3, br=br, cr=cr, step=1, what="flat bases, two bits each")
    ci = util.unpack_chains(enc.chains)
    assert ci["flags"] & 128 and not ci["flags"] & (1 | 32 | 64)
    starts, lens = util.line_table(fq)
    goff, glen = starts[1::4], lens[1::4]
    code = np.zeros(256, np.uint8)
    for ch, v in zip("ACGTacgt0123", [0, 1, 2, 3] * 3): code[ord(ch)] = v
    a = np.frombuffer(fq, np.uint8)
    want = bytearray(); sizes = []
    nrec = len(goff)
    for b0 in range(0, nrec, br):
        for r0 in range(b0, min(b0 + br, nrec), cr):
            r1 = min(r0 + cr, b0 + br, nrec)
            c = np.concatenate([code[a[int(goff[r]): int(goff[r]) + int(glen[r])]] for r in range(r0, r1)])
            c = np.concatenate([c, np.zeros(-len(c) % 4, np.uint8)]).reshape(-1, 4)
            by = (c[:, 0] | c[:, 1] << 2 | c[:, 2] << 4 | c[:, 3] << 6).astype(np.uint8)
            want += by.tobytes(); sizes.append(len(by))
    assert list(ci["gen"]) == sizes and enc.stream("gen") == bytes(want)
    assert ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq
    # the padding of a chain's last byte is zero bits, or the archive is damaged
    nb = [int(sum(int(glen[r]) for r in range(r0, min(r0 + cr, (r0 // br) * br + br, nrec)))) for b0 in range(0, nrec, br) for r0 in range(b0, min(b0 + br, nrec), cr)]
    c = next(i for i, n in enumerate(nb) if n % 4)
    bad = enc.clone()
    at = bad.res.stream_offset[capi.STREAM_NAMES.index("gen")] + int(np.sum(ci["gen"][: c + 1])) - 1
    data = bytearray(bad.data.tobytes() if isinstance(bad.data, np.ndarray) else bad.data)
    data[at] |= 0x80
    bad.data = np.frombuffer(bytes(data), np.uint8).copy() if isinstance(enc.data, np.ndarray) else bytes(data)
    with pytest.raises(capi.SfqError):
        ctx.decode_host(bad, level=3, out_cap=len(fq) + 4096)


def test_flat_bases_of_format_9_are_still_read(ctx):
    """Round 5a coded the bases of a call without a model four a symbol through the range coder ("chn.idx" flag bit 6); archives that say so decode.
    (SFQ_FLAT_QUADS=1 makes the encoder write them: a test hook, api.cpp.)"""
    import os
    fq = capi.synth_fastq(3000, 100, 5, 0)
    os.environ["SFQ_FLAT_QUADS"] = "1"
    O.lib().sfqo_set_flat_raw(0)
    try:
        enc = check_against_oracle(ctx, fq, 3, br=500, cr=50, step=1, what="format 9's quads", quads=True)
    finally:
        del os.environ["SFQ_FLAT_QUADS"]
        O.lib().sfqo_set_flat_raw(1)
    assert util.unpack_chains(enc.chains)["flags"] & 192 == 64
    assert ctx.decode_host(enc, level=3, out_cap=len(fq) + 4096) == fq
