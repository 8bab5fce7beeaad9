"""CPU: the C-ABI library loads, exports every symbol include/slimfastq_amd.h declares, refuses to
work without a GPU (no CPU fallback), and its host-only utilities are deterministic."""
import ctypes
import os
import re

import pytest

from slimfastq_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "slimfastq_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(sfq_[a-z_0-9]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(L, s), s
    assert sorted(capi.EXPORTS) == syms
    assert L.sfq_abi_version() == 3


def test_struct_layouts_match_header():
    assert ctypes.sizeof(capi.Params) == 40
    assert ctypes.sizeof(capi.BlockInfo) == 8 + 4 + 4 + 4 + 4 + 4 + 4 + 8 + 8 * 14 + 4 + 4
    assert ctypes.sizeof(capi.Result) == 8 + 4 + 4 + 112 + 112 + 8 + 8 + 4 + 4 + 64 + 32
    assert [capi.lib().sfq_stream_name(i).decode() for i in range(capi.NSTREAMS)] == capi.STREAM_NAMES


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.SfqError) as e:
        capi.Context(0)
    assert e.value.code == -2


def test_synth_is_deterministic_and_splittable():
    a = capi.synth_fastq(5000, 150, seed=11)
    assert a == capi.synth_fastq(5000, 150, seed=11)
    assert a != capi.synth_fastq(5000, 150, seed=12)
    # any range of reads can be generated independently (per-rank sharding in bench.py)
    assert a == capi.synth_fastq(2000, 150, seed=11) + capi.synth_fastq(3000, 150, seed=11, first_read=2000)
    lines = a.split(b"\n")
    assert len(lines) == 4 * 5000 + 1 and all(len(l) == 150 for l in lines[1::4]) and all(len(l) == 150 for l in lines[3::4])
    lr = capi.synth_fastq(8, 0, seed=5, kind=1)
    ll = [len(l) for l in lr.split(b"\n")[1::4]]
    assert all(10000 <= n <= 50000 for n in ll)
