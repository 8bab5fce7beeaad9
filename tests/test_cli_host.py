"""Host-side pieces that need no GPU: the CLI's container reader (-s), its version / usage texts, and the
multi-file driver's file pairing."""
import os
import subprocess

import pytest

from oracle import oracle as O
from slimfastq_amd import multi
from tests import util

CLI = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "slimfastq_amd", "bin", "slimfastq-amd")


@pytest.fixture(scope="module")
def cli():
    if not os.access(CLI, os.X_OK):
        from slimfastq_amd import build
        build.build()
    assert os.access(CLI, os.X_OK)
    return CLI


def test_cli_version_and_usage(cli):
    p = subprocess.run([cli, "-v"], capture_output=True)
    assert p.returncode == 0 and b"Internal format version=6" in p.stdout
    p = subprocess.run([cli, "-h"], capture_output=True)
    assert p.returncode == 0 and b"-f comp-filename" in p.stdout and b"-B reads" in p.stdout
    p = subprocess.run([cli, "-u", "x.fq"], capture_output=True)
    assert p.returncode == 1 and b"Missing essential argument: -f" in p.stderr


def test_cli_stat_reads_a_reference_written_archive(cli, tmp_path):
    """-s walks the paged container (info page, directory, node pages) on the host: no GPU involved."""
    fq = util.golden_fastq("tst1")
    img = O.compress(fq, 3).image
    f = tmp_path / "tst1.sfq"; f.write_bytes(img)
    p = subprocess.run([cli, "-s", "-f", str(f)], capture_output=True)
    assert p.returncode == 0
    text = p.stderr.decode()
    assert "whoami" in text and "num_records" in text and "2500" in text
    for name, size in (("rec", 5968), ("gen", 62221), ("qlt", 89043)):       # SURVEY.md 8c golden sizes, -l 3
        assert any(l.split(":")[1].strip() == name and l.strip().endswith(str(size)) for l in text.splitlines() if l.count(":") == 2), name


def test_multi_pairs_sources_and_targets(tmp_path):
    d = tmp_path / "in"; (d / "sub").mkdir(parents=True)
    for n in ("a.fq", "b.fastq", "c.txt", "sub/d.fq"):
        (d / n).write_bytes(b"@r\nA\n+\nI\n")

    class A:
        paths = [str(d)]; recursively = False; tgt_dir = str(tmp_path / "out")
    got = multi.find_files(A, [".fastq", ".fq"])
    assert [os.path.basename(g) for g in got] == ["a.fq", "b.fastq"]
    A.recursively = True
    assert len(multi.find_files(A, [".fastq", ".fq"])) == 3
    assert multi.target_of(str(d / "b.fastq"), A, [".fastq", ".fq"], ".sfq") == str(tmp_path / "out" / "b.sfq")
    A.tgt_dir = None
    assert multi.target_of(str(d / "a.fq"), A, [".fastq", ".fq"], ".sfq") == str(d / "a.sfq")
    assert multi.target_of(str(d / "x.sfq"), A, [".sfq"], ".fastq") == str(d / "x.fastq")


def test_dist_compress_shards_on_record_boundaries():
    """Byte ranges of the multi-rank compressor start at records even when quality lines start with '@'."""
    from slimfastq_amd import dist_compress as dc
    recs = []
    for i in range(200):
        n = 20 + (i * 7) % 30
        qual = ("@" if i % 3 == 0 else "I") + "@+I#"[i % 4] * (n - 1)      # '@' and '+' as first quality characters
        recs.append("@r%d x\n%s\n+\n%s\n" % (i, "ACGT" * 12 if False else ("ACGTN"[i % 5]) * n, qual))
    fq = "".join(recs).encode()
    starts = set()
    o = 0
    for r in recs:
        starts.add(o); o += len(r)
    for world in (1, 2, 3, 7):
        cuts = [dc.shard_bytes(fq, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == len(fq)
        for (lo, hi), (lo2, _) in zip(cuts, cuts[1:] + [(len(fq), 0)]):
            assert hi == lo2 and (lo in starts or lo == len(fq))
    for pos in range(0, len(fq), 37):
        p = dc.record_start(fq, pos)
        assert p >= pos and (p in starts or p == len(fq))


def test_dist_compress_assembles_a_segmented_archive(cli, tmp_path):
    """The writer rank's part: per-rank results -> one archive with a segment per rank (host only; the streams here
    are placeholders, the layout is what is checked -- `slimfastq-amd -s` walks it)."""
    from slimfastq_amd import capi, dist_compress as dc
    parts = []
    for r in range(3):
        blocks = (capi.BlockInfo * 2)()
        for k in range(2):
            blocks[k].n_records = 10 + k; blocks[k].first_hdr_len = 5
            for s in range(3):
                blocks[k].size[s] = 4 + s
        streams = [bytes([r]) * (2 * (4 + s)) if s < 3 else b"" for s in range(capi.NSTREAMS)]
        parts.append(dict(streams=streams, blocks=list(blocks), first=b"hdr_%d" % r * 2, prior=b"P" * (r + 1), chains=b"C" * r, rec_prior=b"",
                          raw=1000 + r, records=21))
    parts.append(dict(streams=[b""] * capi.NSTREAMS, blocks=[], first=b"", prior=b"", chains=b"", rec_prior=b"", raw=0, records=0))   # a rank with no records
    info, streams = dc.assemble(parts, 3, 1024, "x.fq")
    f = tmp_path / "seg.sfq"
    dc.write_archive(str(f), info, streams)
    p = subprocess.run([cli, "-s", "-f", str(f)], capture_output=True)
    text = p.stderr.decode()
    assert p.returncode == 0 and "seg.count" in text and "= 3" in text and "blk.count" in text and "= 6" in text
    d = dict(streams)
    assert d["qlt.pri"] == b"PPPPPP" and d["chn.idx"] == b"CCC" and len(d["rec"]) == 3 * 8 and d["seg.idx"][0] == 3
    # the flat exchange form of one rank's part (no pickle): streams, index parts, int64 trailer
    import numpy as np
    import torch

    class R:
        stream_bytes = [3, 0, 2] + [0] * (capi.NSTREAMS - 3); total_bytes = 5; n_records = 21
    t = dc.pack_part(torch.from_numpy(np.frombuffer(b"abcde-----", np.uint8).copy()), R, list(blocks), b"first", b"prior", b"chains", b"rp", 1234)
    u = dc.unpack_part(t.numpy().tobytes())
    assert u["streams"][0] == b"abc" and u["streams"][2] == b"de" and u["first"] == b"first" and u["prior"] == b"prior" and u["chains"] == b"chains"
    assert u["rec_prior"] == b"rp" and u["raw"] == 1234 and u["records"] == 21 and len(u["blocks"]) == 2 and u["blocks"][1].n_records == 11
    assert "num_records      = 63" in text
