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
