"""bench.py's argument handling (no GPU): defaults per workload kind, workload names."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def parse(mod, argv):
    old = sys.argv
    sys.argv = ["bench.py"] + argv
    try:
        return mod.parse()
    finally:
        sys.argv = old


def test_defaults_follow_the_workload_kind():
    b = load_bench()
    a = parse(b, [])
    assert (a.gpus, a.reads, a.read_len, a.level, a.block_reads, a.kind) == (1, 10_000_000, 150, 3, 1024, 0)
    assert a.tables == 1
    assert b.workload_name(a) == "synthetic 10M x 150 bp Illumina reads per GPU, full qlts+gens+recs, frozen tables (one chain per lane), -l 3"
    assert "adaptive tables" in b.workload_name(parse(b, ["--tables", "0"]))
    a = parse(b, ["--kind", "1"])
    assert a.reads == 60_000 and a.block_reads == b.capi.BLOCK_AUTO and a.cpu_sample_reads <= 6_000
    assert "long reads" in b.workload_name(a)
    a = parse(b, ["--workload", "qlt", "--reads", "2500000", "--block-reads", "512"])
    assert a.reads == 2_500_000 and a.block_reads == 512 and "qlts-only kernel" in b.workload_name(a)
    a = parse(b, ["--gpus", "8", "--steps", "5", "--warmup", "2"])
    assert (a.gpus, a.steps, a.warmup) == (8, 5, 2)
