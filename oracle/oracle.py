"""ctypes binding of oracle/libsfq_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libsfq_oracle.so")
    src = os.path.join(_HERE, "sfq_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libsfq_oracle.so"])
    return so


def ref_binary():
    """Path of the compiled reference (oracle/_ref/slimfastq_ref) or None."""
    p = os.path.join(_HERE, "_ref", "slimfastq_ref")
    return p if os.path.exists(p) else None


class Opts(C.Structure):
    _fields_ = [("level", C.c_int), ("quiet", C.c_int), ("gen_bits", C.c_int),
                ("orig_filename", C.c_char_p), ("orig_size", C.c_longlong), ("lossless", C.c_int)]


class GenSide(C.Structure):
    _fields_ = [("ns", C.POINTER(C.c_uint64)), ("n_ns", C.c_size_t),
                ("nn", C.POINTER(C.c_uint64)), ("n_nn", C.c_size_t), ("n_byte", C.c_int)]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, sz, u8p = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint8)
        L.sfqo_compress.restype = vp
        L.sfqo_compress.argtypes = [C.c_char_p, sz, C.POINTER(Opts)]
        L.sfqo_decompress.argtypes = [vp, C.POINTER(u8p), C.POINTER(sz)]
        L.sfqo_archive_from_image.restype = vp
        L.sfqo_archive_from_image.argtypes = [C.c_char_p, sz]
        L.sfqo_archive_image.restype = u8p
        L.sfqo_archive_image.argtypes = [vp, C.POINTER(sz)]
        L.sfqo_archive_free.argtypes = [vp]
        L.sfqo_nstreams.argtypes = [vp]
        L.sfqo_stream_name.restype = C.c_char_p
        L.sfqo_stream_name.argtypes = [vp, C.c_int]
        L.sfqo_stream_size.restype = sz
        L.sfqo_stream_size.argtypes = [vp, C.c_int]
        L.sfqo_stream_read.restype = C.c_longlong
        L.sfqo_stream_read.argtypes = [vp, C.c_int, C.c_char_p, sz]
        L.sfqo_info_get.restype = C.c_char_p
        L.sfqo_info_get.argtypes = [vp, C.c_char_p]
        L.sfqo_last_error.restype = C.c_char_p
        L.sfqo_free.argtypes = [vp]
        for f in ("sfqo_qlt_encode", "sfqo_qlt_decode", "sfqo_gen_encode", "sfqo_gen_decode_raw",
                  "sfqo_rec_encode", "sfqo_xfile_encode_u", "sfqo_xfile_decode_u"):
            getattr(L, f).restype = C.c_int
        _LIB = L
    return _LIB


class OracleError(RuntimeError):
    pass


def _err():
    return OracleError(lib().sfqo_last_error().decode("latin1"))


class Archive:
    """An in-memory .sfq container: {stream name: bytes} + info dict, directory order kept."""

    def __init__(self, handle):
        L = lib()
        self.streams = {}
        self.order = []
        n = L.sfqo_nstreams(handle)
        for i in range(n):
            name = "<info>" if i == 0 else L.sfqo_stream_name(handle, i).decode("latin1")
            size = L.sfqo_stream_size(handle, i)
            buf = C.create_string_buffer(max(size, 1))
            got = L.sfqo_stream_read(handle, i, buf, size)
            if got != size:
                raise OracleError("bad stream chain for %s" % name)
            self.streams[name] = buf.raw[:size]
            self.order.append(name)
        sz = C.c_size_t()
        p = L.sfqo_archive_image(handle, C.byref(sz))
        self.image = C.string_at(p, sz.value)
        self.info = {}
        for line in self.streams["<info>"].decode("latin1").split("\n"):
            if "=" in line:
                k, v = line.split("=", 1)
                self.info.setdefault(k, v)

    def payload_bytes(self):
        """Sum of stream bytes (the ratio metric, SURVEY.md section 5 'Metrics')."""
        return sum(len(v) for v in self.streams.values())


def compress(fastq: bytes, level=3, quiet=True, gen_bits=0, orig_filename=None, orig_size=-1, lossless=False) -> Archive:
    """lossless: the rules of this project's block format (sfq_oracle.h sfqo_opts.lossless) instead of the reference's quirks."""
    L = lib()
    o = Opts(level, int(quiet), gen_bits, orig_filename, orig_size, int(lossless))
    h = L.sfqo_compress(fastq, len(fastq), C.byref(o))
    if not h:
        raise _err()
    try:
        return Archive(h)
    finally:
        L.sfqo_archive_free(h)


def parse(image: bytes) -> Archive:
    L = lib()
    h = L.sfqo_archive_from_image(image, len(image))
    if not h:
        raise _err()
    try:
        return Archive(h)
    finally:
        L.sfqo_archive_free(h)


def decompress(image: bytes) -> bytes:
    L = lib()
    h = L.sfqo_archive_from_image(image, len(image))
    if not h:
        raise _err()
    try:
        out = C.POINTER(C.c_uint8)()
        n = C.c_size_t()
        if L.sfqo_decompress(h, C.byref(out), C.byref(n)) != 0:
            raise _err()
        data = C.string_at(out, n.value)
        L.sfqo_free(out)
        return data
    finally:
        L.sfqo_archive_free(h)


def _take(ptr, n):
    data = C.string_at(ptr, n.value) if n.value else b""
    lib().sfqo_free(ptr)
    return data


def _arr(a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    return a, a.ctypes.data_as(C.c_void_p)


def qlt_encode(buf: bytes, off, length, level=3):
    """-> (stream bytes, extra_hi count).  qlts.cpp:74-136."""
    L = lib()
    off, po = _arr(off, np.uint64)
    length, pl = _arr(length, np.uint32)
    out = C.POINTER(C.c_uint8)(); n = C.c_size_t(); hi = C.c_uint32()
    L.sfqo_qlt_encode.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                  C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]
    if L.sfqo_qlt_encode(buf, po, pl, len(off), level, C.byref(out), C.byref(n), C.byref(hi)) != 0:
        raise _err()
    return _take(out, n), hi.value


def qlt_decode(stream: bytes, off, length, total, level=3) -> bytes:
    L = lib()
    off, po = _arr(off, np.uint64)
    length, pl = _arr(length, np.uint32)
    dst = C.create_string_buffer(total + 1)
    L.sfqo_qlt_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    if L.sfqo_qlt_decode(stream, len(stream), dst, po, pl, len(off), level) != 0:
        raise _err()
    return dst.raw[:total]


def gen_encode(buf: bytes, goff, glen, qoff, qlen, gen_bits=24):
    """-> (stream, Ns positions, Nn positions, N byte).  gens.cpp:91-159."""
    L = lib()
    goff, pgo = _arr(goff, np.uint64); glen, pgl = _arr(glen, np.uint32)
    qoff, pqo = _arr(qoff, np.uint64); qlen, pql = _arr(qlen, np.uint32)
    out = C.POINTER(C.c_uint8)(); n = C.c_size_t(); side = GenSide()
    L.sfqo_gen_encode.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                  C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t), C.POINTER(GenSide)]
    if L.sfqo_gen_encode(buf, pgo, pgl, pqo, pql, len(goff), gen_bits, C.byref(out), C.byref(n), C.byref(side)) != 0:
        raise _err()
    ns = np.ctypeslib.as_array(side.ns, (side.n_ns,)).copy() if side.n_ns else np.zeros(0, np.uint64)
    nn = np.ctypeslib.as_array(side.nn, (side.n_nn,)).copy() if side.n_nn else np.zeros(0, np.uint64)
    L.sfqo_free(side.ns); L.sfqo_free(side.nn)
    return _take(out, n), ns, nn, side.n_byte


def gen_decode_raw(stream: bytes, goff, glen, total, gen_bits=24, solid=False) -> bytes:
    L = lib()
    goff, pgo = _arr(goff, np.uint64); glen, pgl = _arr(glen, np.uint32)
    dst = C.create_string_buffer(total + 1)
    L.sfqo_gen_decode_raw.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
    if L.sfqo_gen_decode_raw(stream, len(stream), dst, pgo, pgl, len(goff), gen_bits, int(solid)) != 0:
        raise _err()
    return dst.raw[:total]


def rec_encode(buf: bytes, off, length):
    """-> (stream, 1-based record numbers that went to rec.x).  recs.cpp:277-372."""
    L = lib()
    off, po = _arr(off, np.uint64); length, pl = _arr(length, np.uint32)
    out = C.POINTER(C.c_uint8)(); n = C.c_size_t()
    xr = C.POINTER(C.c_uint64)(); nx = C.c_size_t()
    L.sfqo_rec_encode.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                  C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t),
                                  C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_size_t)]
    if L.sfqo_rec_encode(buf, po, pl, len(off), C.byref(out), C.byref(n), C.byref(xr), C.byref(nx)) != 0:
        raise _err()
    x = np.ctypeslib.as_array(xr, (nx.value,)).copy() if nx.value else np.zeros(0, np.uint64)
    L.sfqo_free(xr)
    return _take(out, n), x


def xfile_encode_u(vals) -> bytes:
    L = lib()
    vals, pv = _arr(vals, np.uint64)
    out = C.POINTER(C.c_uint8)(); n = C.c_size_t()
    L.sfqo_xfile_encode_u.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    if L.sfqo_xfile_encode_u(pv, len(vals), C.byref(out), C.byref(n)) != 0:
        raise _err()
    return _take(out, n)


def xfile_decode_u(stream: bytes, count: int):
    L = lib()
    vals = np.zeros(count, np.uint64)
    L.sfqo_xfile_decode_u.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
    if L.sfqo_xfile_decode_u(stream, len(stream), vals.ctypes.data_as(C.c_void_p), count) != 0:
        raise _err()
    return vals


# ---- driving the compiled reference (container only; never on the GPU box) --------------------------
def ref_compress(fastq: bytes, level=3, quiet=True) -> bytes:
    """Run oracle/_ref/slimfastq_ref on stdin (path-independent info page) -> .sfq image."""
    import tempfile
    exe = ref_binary()
    if exe is None:
        raise OracleError("oracle/_ref/slimfastq_ref not built")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "o.sfq")
        cmd = [exe, "-f", out, "-O", "-l", str(level)] + (["-q"] if quiet else [])
        p = subprocess.run(cmd, input=fastq, capture_output=True)
        if p.returncode != 0:
            raise OracleError("reference failed: " + p.stderr.decode("latin1"))
        return open(out, "rb").read()


def ref_decompress(image: bytes) -> bytes:
    import tempfile
    exe = ref_binary()
    if exe is None:
        raise OracleError("oracle/_ref/slimfastq_ref not built")
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "i.sfq")
        open(src, "wb").write(image)
        p = subprocess.run([exe, "-d", "-f", src], capture_output=True)
        if p.returncode != 0:
            raise OracleError("reference failed: " + p.stderr.decode("latin1"))
        return p.stdout


# ---- format-7 warm start (this project's rule, restated in sfq_oracle.c) -------------------------------------
def qlt_histogram(buf: bytes, off, length, level=3, first=0, step=1):
    L = lib()
    off, po = _arr(off, np.uint64); length, pl = _arr(length, np.uint32)
    q_rows = 4096 if level == 1 else 65536
    counts = np.zeros(q_rows * 64, np.uint32)
    L.sfqo_qlt_histogram.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p]
    L.sfqo_qlt_histogram(buf, po, pl, len(off), level, first, step, counts.ctypes.data_as(C.c_void_p))
    return counts


def qlt_prior_rows(counts):
    """-> uint32 [q_rows, 66]: slot[64] (freq | sym << 16), total, iend."""
    L = lib()
    q_rows = len(counts) // 64
    rows = np.zeros(q_rows * 66, np.uint32)
    L.sfqo_qlt_prior_rows.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.sfqo_qlt_prior_rows(counts.ctypes.data_as(C.c_void_p), q_rows, rows.ctypes.data_as(C.c_void_p))
    return rows.reshape(q_rows, 66)


def qlt_encode_blocks(buf: bytes, off, length, level=3, block_reads=1024, prior_rows=None):
    """-> (concatenated per-block qlt streams, per-block sizes); prior_rows None = cold (== the reference per chunk)."""
    L = lib()
    off, po = _arr(off, np.uint64); length, pl = _arr(length, np.uint32)
    nb = (len(off) + block_reads - 1) // block_reads
    sizes = np.zeros(nb, np.uint32)
    out = C.POINTER(C.c_uint8)(); n = C.c_size_t()
    pr = None
    if prior_rows is not None:
        prior_rows = np.ascontiguousarray(prior_rows, np.uint32)
        pr = prior_rows.ctypes.data_as(C.c_void_p)
    L.sfqo_qlt_encode_blocks.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_void_p,
                                         C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t), C.c_void_p]
    if L.sfqo_qlt_encode_blocks(buf, po, pl, len(off), level, block_reads, pr, C.byref(out), C.byref(n), sizes.ctypes.data_as(C.c_void_p)) != 0:
        raise _err()
    return _take(out, n), sizes


# ---- frozen tables (block format 7, sfq_params.tables = 1): this project's own mode, restated in sfq_oracle.c ----
def _nchains(nrec, block_reads, chain_reads):
    nb = (nrec + block_reads - 1) // block_reads
    cpb = (block_reads + chain_reads - 1) // chain_reads
    last = nrec - (nb - 1) * block_reads
    return (nb - 1) * cpb + (last + chain_reads - 1) // chain_reads


def qlt_frozen_rows(prior_rows):
    """prior rows [q_rows, 66] -> frozen entries uint32 [q_rows, 64] (cum | freq << 16, total 2^16)."""
    L = lib()
    prior_rows = np.ascontiguousarray(prior_rows, np.uint32)
    q_rows = prior_rows.shape[0]
    out = np.zeros(q_rows * 64, np.uint32)
    L.sfqo_qlt_frozen_rows.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.sfqo_qlt_frozen_rows(prior_rows.ctypes.data_as(C.c_void_p), q_rows, out.ctypes.data_as(C.c_void_p))
    return out.reshape(q_rows, 64)


def qlt_encode_chains(buf: bytes, off, length, level, block_reads, chain_reads, frozen_rows):
    """-> (chain streams back to back, per-chain sizes, escapes)."""
    L = lib()
    off, po = _arr(off, np.uint64); length, pl = _arr(length, np.uint32)
    frozen_rows = np.ascontiguousarray(frozen_rows, np.uint32)
    sizes = np.zeros(_nchains(len(off), block_reads, chain_reads), np.uint32)
    out = C.POINTER(C.c_uint8)(); n = C.c_size_t(); extra = C.c_uint32()
    L.sfqo_qlt_encode_chains.restype = C.c_longlong
    L.sfqo_qlt_encode_chains.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p,
                                         C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_uint32)]
    got = L.sfqo_qlt_encode_chains(buf, po, pl, len(off), level, block_reads, chain_reads, frozen_rows.ctypes.data_as(C.c_void_p),
                                   C.byref(out), C.byref(n), sizes.ctypes.data_as(C.c_void_p), C.byref(extra))
    if got != len(sizes):
        raise _err()
    return _take(out, n), sizes, extra.value


def gen_encode_chains(buf: bytes, goff, glen, gen_bits, block_reads, chain_reads, step=4):
    """-> (chain streams back to back, per-chain sizes, gen_on)."""
    L = lib()
    goff, po = _arr(goff, np.uint64); glen, pl = _arr(glen, np.uint32)
    sizes = np.zeros(_nchains(len(goff), block_reads, chain_reads), np.uint32)
    out = C.POINTER(C.c_uint8)(); n = C.c_size_t(); on = C.c_int()
    L.sfqo_gen_encode_chains.restype = C.c_longlong
    L.sfqo_gen_encode_chains.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.c_uint32,
                                         C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_int)]
    got = L.sfqo_gen_encode_chains(buf, po, pl, len(goff), gen_bits, block_reads, chain_reads, step, C.byref(out), C.byref(n),
                                   sizes.ctypes.data_as(C.c_void_p), C.byref(on))
    if got != len(sizes):
        raise _err()
    return _take(out, n), sizes, on.value


def exc_rice_block(buf: bytes, goff, glen, qoff, qlen):
    """A block's base-exception lists as adaptive Rice codes (frozen tables, round 4) -> (gen.Ns, gen.Nn, gen.lc, n_byte)."""
    L = lib()
    goff, pg = _arr(goff, np.uint64); glen, pgl = _arr(glen, np.uint32); qoff, pq = _arr(qoff, np.uint64); qlen, pql = _arr(qlen, np.uint32)
    out = (C.POINTER(C.c_uint8) * 3)(); n = (C.c_size_t * 3)(); nb = C.c_uint32()
    L.sfqo_exc_rice_block.restype = C.c_int
    L.sfqo_exc_rice_block.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
    if L.sfqo_exc_rice_block(buf, pg, pgl, pq, pql, len(goff), out, n, C.byref(nb)) != 0:
        raise _err()
    return tuple(_take(out[i], C.c_size_t(n[i])) for i in range(3)) + (nb.value,)


def exc_rice_decode(blob: bytes):
    """The positions (from 1) a Rice-coded exception list holds."""
    L = lib()
    L.sfqo_exc_rice_decode.restype = C.c_longlong
    L.sfqo_exc_rice_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
    cnt = L.sfqo_exc_rice_decode(blob, len(blob), None, 0)
    if cnt < 0:
        raise OracleError("exception list does not end")
    pos = np.zeros(max(cnt, 1), np.uint64)
    L.sfqo_exc_rice_decode(blob, len(blob), pos.ctypes.data_as(C.c_void_p), cnt)
    return pos[:cnt]


def seg_counts(length, other, seg_len):
    """Segments per record (chains.hip): n = ceil(max(len, other) / seg_len), at least one."""
    m = np.maximum(np.asarray(length, np.uint64), np.asarray(other, np.uint64))
    return np.maximum(1, (m + seg_len - 1) // seg_len).astype(np.int64)


def qlt_encode_segs(buf: bytes, off, length, other_len, level, seg_len, frozen_rows):
    """Quality chains that are SEGMENTS of one record -> (chain streams back to back, per-chain sizes, escapes)."""
    L = lib()
    nseg = int(seg_counts(length, other_len, seg_len).sum())
    off, po = _arr(off, np.uint64); length, pl = _arr(length, np.uint32); other_len, pt = _arr(other_len, np.uint32)
    frozen_rows = np.ascontiguousarray(frozen_rows, np.uint32)
    sizes = np.zeros(nseg, np.uint32)
    out = C.POINTER(C.c_uint8)(); n = C.c_size_t(); extra = C.c_uint32()
    L.sfqo_qlt_encode_segs.restype = C.c_longlong
    L.sfqo_qlt_encode_segs.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_void_p,
                                       C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_uint32)]
    got = L.sfqo_qlt_encode_segs(buf, po, pl, pt, len(off), level, seg_len, frozen_rows.ctypes.data_as(C.c_void_p),
                                 C.byref(out), C.byref(n), sizes.ctypes.data_as(C.c_void_p), C.byref(extra))
    if got != len(sizes):
        raise _err()
    return _take(out, n), sizes, extra.value


def gen_encode_segs(buf: bytes, goff, glen, other_len, gen_bits, block_reads, seg_len, step=4):
    """Base chains that are SEGMENTS of one record -> (chain streams back to back, per-chain sizes, gen_on)."""
    L = lib()
    nseg = int(seg_counts(glen, other_len, seg_len).sum())
    goff, po = _arr(goff, np.uint64); glen, pl = _arr(glen, np.uint32); other_len, pt = _arr(other_len, np.uint32)
    sizes = np.zeros(nseg, np.uint32)
    out = C.POINTER(C.c_uint8)(); n = C.c_size_t(); on = C.c_int()
    L.sfqo_gen_encode_segs.restype = C.c_longlong
    L.sfqo_gen_encode_segs.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_uint32, C.c_uint32,
                                       C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_int)]
    got = L.sfqo_gen_encode_segs(buf, po, pl, pt, len(goff), gen_bits, block_reads, seg_len, step, C.byref(out), C.byref(n),
                                 sizes.ctypes.data_as(C.c_void_p), C.byref(on))
    if got != len(sizes):
        raise _err()
    return _take(out, n), sizes, on.value


def gm_encode_chains(buf: bytes, goff, glen, table_bits, block_reads, chain_reads):
    """Base chains under the generation MATCH model (round 5) -> (chain streams back to back, per-chain sizes, gen_on)."""
    L = lib()
    goff, po = _arr(goff, np.uint64); glen, pl = _arr(glen, np.uint32)
    sizes = np.zeros(_nchains(len(goff), block_reads, chain_reads), np.uint32)
    out = C.POINTER(C.c_uint8)(); n = C.c_size_t(); on = C.c_int()
    L.sfqo_gm_encode_chains.restype = C.c_longlong
    L.sfqo_gm_encode_chains.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t,
                                        C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_int)]
    got = L.sfqo_gm_encode_chains(buf, po, pl, len(goff), table_bits, block_reads, chain_reads, C.byref(out), C.byref(n),
                                  sizes.ctypes.data_as(C.c_void_p), C.byref(on))
    if got != len(sizes):
        raise _err()
    return _take(out, n), sizes, on.value


def gm_decode_chains(streams: bytes, sizes, glen, table_bits, block_reads, chain_reads):
    """The base chains of a call under the match model decoded on the CPU -> the bases' codes (uint8, 0..3), records back to back."""
    L = lib()
    sizes, ps = _arr(sizes, np.uint32); glen, pl = _arr(glen, np.uint32)
    out = np.zeros(int(glen.sum()), np.uint8)
    L.sfqo_gm_decode_chains.restype = C.c_int
    L.sfqo_gm_decode_chains.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p]
    if L.sfqo_gm_decode_chains(streams, ps, pl, len(glen), table_bits, block_reads, chain_reads, out.ctypes.data_as(C.c_void_p)) != 0:
        raise _err()
    return out


def gm_encode_segs(buf: bytes, goff, glen, other_len, table_bits, block_reads, seg_len):
    """The same for chains that are SEGMENTS of one record."""
    L = lib()
    nseg = int(seg_counts(glen, other_len, seg_len).sum())
    goff, po = _arr(goff, np.uint64); glen, pl = _arr(glen, np.uint32); other_len, pt = _arr(other_len, np.uint32)
    sizes = np.zeros(nseg, np.uint32)
    out = C.POINTER(C.c_uint8)(); n = C.c_size_t(); on = C.c_int()
    L.sfqo_gm_encode_segs.restype = C.c_longlong
    L.sfqo_gm_encode_segs.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_uint32,
                                      C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_int)]
    got = L.sfqo_gm_encode_segs(buf, po, pl, pt, len(goff), table_bits, block_reads, seg_len, C.byref(out), C.byref(n),
                                sizes.ctypes.data_as(C.c_void_p), C.byref(on))
    if got != len(sizes):
        raise _err()
    return _take(out, n), sizes, on.value


REC_ROWS = 66 * 16


def rec_count(buf: bytes, off, length, stride, run, nruns):
    L = lib()
    off, po = _arr(off, np.uint64); length, pl = _arr(length, np.uint32)
    counts = np.zeros(REC_ROWS * 256, np.uint32)
    L.sfqo_rec_count.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
    if L.sfqo_rec_count(buf, po, pl, len(off), stride, run, nruns, counts.ctypes.data_as(C.c_void_p)) != 0:
        raise _err()
    return counts


def rec_prior_freqs(counts):
    L = lib()
    f = np.zeros(REC_ROWS * 256, np.uint32)
    L.sfqo_rec_prior_freqs.argtypes = [C.c_void_p, C.c_void_p]
    L.sfqo_rec_prior_freqs(counts.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p))
    return f


def rec_frozen_rows(f):
    L = lib()
    f = np.ascontiguousarray(f, np.uint32)
    rows = np.zeros(REC_ROWS * 256, np.uint32)
    L.sfqo_rec_frozen_rows.argtypes = [C.c_void_p, C.c_void_p]
    L.sfqo_rec_frozen_rows(f.ctypes.data_as(C.c_void_p), rows.ctypes.data_as(C.c_void_p))
    return rows


def rec_encode_chains_frozen(buf: bytes, off, length, block_reads, chain_reads, frozen_rows):
    """-> (header chain streams back to back, per-chain sizes, per-chain header bytes)."""
    L = lib()
    off, po = _arr(off, np.uint64); length, pl = _arr(length, np.uint32)
    frozen_rows = np.ascontiguousarray(frozen_rows, np.uint32)
    nc = _nchains(len(off), block_reads, chain_reads)
    sizes = np.zeros(nc, np.uint32); hb = np.zeros(nc, np.uint32)
    out = C.POINTER(C.c_uint8)(); n = C.c_size_t()
    L.sfqo_rec_encode_chains_frozen.restype = C.c_longlong
    L.sfqo_rec_encode_chains_frozen.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p,
                                                C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t), C.c_void_p, C.c_void_p]
    got = L.sfqo_rec_encode_chains_frozen(buf, po, pl, len(off), block_reads, chain_reads, frozen_rows.ctypes.data_as(C.c_void_p),
                                          C.byref(out), C.byref(n), sizes.ctypes.data_as(C.c_void_p), hb.ctypes.data_as(C.c_void_p))
    if got != nc:
        raise _err()
    return _take(out, n), sizes, hb


def rec_encode_pre5(buf: bytes, off, length) -> bytes:
    """A "rec" stream in the pre-version-5 layout (test input for load_pre5)."""
    L = lib()
    off, po = _arr(off, np.uint64); length, pl = _arr(length, np.uint32)
    out = C.POINTER(C.c_uint8)(); n = C.c_size_t()
    L.sfqo_rec_encode_pre5.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    if L.sfqo_rec_encode_pre5(buf, po, pl, len(off), C.byref(out), C.byref(n)) != 0:
        raise _err()
    return _take(out, n)
