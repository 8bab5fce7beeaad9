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
