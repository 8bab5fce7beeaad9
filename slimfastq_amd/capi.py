"""ctypes binding of libslimfastq_amd.so (include/slimfastq_amd.h).  No fallback of any kind."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libslimfastq_amd.so")

NSTREAMS = 14
STREAM_NAMES = ["rec", "gen", "qlt", "gen.Ns", "gen.Nn", "rec.x", "usr.x", "usr.x.q", "usr.pfg", "usr.pfq",
                "gen.lc", "usr.lrec", "usr.lgen", "usr.lqlt"]
M_REC, M_GEN, M_QLT, M_USR, M_ALL = 1, 2, 4, 8, 15
T_FRAME, T_QLT, T_GEN, T_REC, T_USR, T_PACK, T_TOTAL = range(7)
PRIOR_AUTO = 0xFFFFFFFF
PRIOR_GIVEN = 0xFFFFFFFE
PRIOR_COUNTS = 0xFFFFFFFD
BLOCK_AUTO = 0xFFFFFFFF
TABLES_ADAPTIVE, TABLES_FROZEN, TABLES_AUTO = 0, 1, 2
LDS_ROWS_NONE = 0xFFFFFFFF

EXPORTS = [
    "sfq_stream_name", "sfq_ctx_create", "sfq_ctx_destroy", "sfq_last_error", "sfq_ctx_set_table_budget",
    "sfq_ctx_stream", "sfq_ctx_synchronize", "sfq_encode_bound", "sfq_encode_blocks", "sfq_encode_qlt_blocks",
    "sfq_encode_blocks_host", "sfq_get_block_index", "sfq_get_first_headers", "sfq_decode_blocks",
    "sfq_decode_blocks_host", "sfq_synth_fastq", "sfq_abi_version", "sfq_get_qlt_prior", "sfq_set_qlt_prior",
    "sfq_archive_write", "sfq_pack_block_index", "sfq_ctx_device_memory", "sfq_get_chain_index", "sfq_set_chain_index", "sfq_get_rec_prior", "sfq_set_rec_prior", "sfq_build_priors",
    "sfq_host_alloc", "sfq_host_free", "sfq_count_priors", "sfq_prior_counts_words", "sfq_get_prior_counts", "sfq_set_prior_counts",
]


class Params(C.Structure):
    _fields_ = [("level", C.c_int32), ("block_reads", C.c_uint32), ("gen_bits", C.c_int32), ("models", C.c_uint32),
                ("kernel", C.c_uint32), ("version", C.c_uint32), ("prior_step", C.c_uint32), ("tables", C.c_uint32),
                ("chain_reads", C.c_uint32), ("lds_rows", C.c_uint32)]


class BlockInfo(C.Structure):
    _fields_ = [("first_record", C.c_uint64), ("n_records", C.c_uint32), ("llen", C.c_uint32),
                ("solid", C.c_uint8), ("two_id", C.c_uint8), ("n_byte", C.c_uint8), ("gen_bits", C.c_uint8),
                ("extra_hi", C.c_uint32), ("first_hdr_len", C.c_uint32), ("first_hdr_off", C.c_uint64),
                ("size", C.c_uint64 * NSTREAMS), ("status", C.c_uint32), ("hdr_bytes", C.c_uint32)]


class Result(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("n_blocks", C.c_uint32), ("abi_version", C.c_uint32),
                ("stream_bytes", C.c_uint64 * NSTREAMS), ("stream_offset", C.c_uint64 * NSTREAMS),
                ("total_bytes", C.c_uint64), ("first_hdr_bytes", C.c_uint64), ("n_chains", C.c_uint32), ("reserved", C.c_uint32),
                ("kernel_ms", C.c_double * 8), ("coder_ms", C.c_double * 4)]


class SfqError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("slimfastq_amd error %d: %s" % (code, msg))
        self.code = code


_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process: PyTorch wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7).  If this library pulled in /opt/rocm's copy first, a later `import torch` would
    load a second runtime and see no GPU.  Pre-loading torch's copy makes the dynamic linker bind our
    DT_NEEDED libamdhip64.so.7 to it.  Set SFQ_HIP_RUNTIME=system to skip (e.g. torch-free processes
    that must use /opt/rocm)."""
    if os.environ.get("SFQ_HIP_RUNTIME") == "system":
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec and spec.origin:
        p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run `python -m slimfastq_amd.build` (there is no CPU fallback)" % LIB_PATH)
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        vp, u64, u8p = C.c_void_p, C.c_uint64, C.c_void_p
        L.sfq_stream_name.restype = C.c_char_p
        L.sfq_ctx_create.argtypes = [C.POINTER(vp), C.c_int]
        L.sfq_ctx_destroy.argtypes = [vp]
        L.sfq_ctx_destroy.restype = None
        L.sfq_last_error.argtypes = [vp]
        L.sfq_last_error.restype = C.c_char_p
        L.sfq_ctx_set_table_budget.argtypes = [vp, u64]
        L.sfq_ctx_device_memory.argtypes = [vp]
        L.sfq_ctx_device_memory.restype = u64
        L.sfq_ctx_stream.argtypes = [vp]
        L.sfq_ctx_stream.restype = vp
        L.sfq_ctx_synchronize.argtypes = [vp]
        L.sfq_host_alloc.argtypes = [vp, C.c_uint64]; L.sfq_host_alloc.restype = C.c_void_p
        L.sfq_host_free.argtypes = [vp, C.c_void_p]; L.sfq_host_free.restype = None
        L.sfq_encode_bound.argtypes = [u64]
        L.sfq_encode_bound.restype = u64
        for f in (L.sfq_encode_blocks, L.sfq_encode_qlt_blocks, L.sfq_encode_blocks_host):
            f.argtypes = [vp, u8p, u64, C.POINTER(Params), u8p, u64, C.POINTER(Result)]
        L.sfq_get_block_index.argtypes = [vp, C.POINTER(BlockInfo), C.c_uint32]
        L.sfq_get_first_headers.argtypes = [vp, u8p, u64]
        L.sfq_decode_blocks.argtypes = [vp, C.POINTER(Params), C.POINTER(BlockInfo), C.c_uint32, u8p, u64, u8p,
                                        C.POINTER(u64), u8p, u64, C.POINTER(u64), C.POINTER(Result)]
        L.sfq_decode_blocks_host.argtypes = [vp, C.POINTER(Params), C.POINTER(BlockInfo), C.c_uint32, u8p, u64, u8p, u64,
                                             C.POINTER(u64), u8p, u64, C.POINTER(u64), C.POINTER(Result)]
        L.sfq_get_qlt_prior.argtypes = [vp, u8p, u64]
        L.sfq_get_qlt_prior.restype = C.c_int64
        L.sfq_set_qlt_prior.argtypes = [vp, u8p, u64]
        L.sfq_build_priors.argtypes = [vp, u8p, u64, C.POINTER(Params)]
        L.sfq_count_priors.argtypes = [vp, u8p, u64, C.POINTER(Params), C.c_uint32]
        L.sfq_prior_counts_words.argtypes = [C.c_int, C.POINTER(u64), C.POINTER(u64)]
        L.sfq_prior_counts_words.restype = None
        L.sfq_get_prior_counts.argtypes = [vp, C.c_int, u8p, u8p]
        L.sfq_set_prior_counts.argtypes = [vp, C.c_int, u8p, u8p]
        L.sfq_get_rec_prior.argtypes = [vp, u8p, u64]
        L.sfq_get_rec_prior.restype = C.c_int64
        L.sfq_set_rec_prior.argtypes = [vp, u8p, u64]
        L.sfq_get_chain_index.argtypes = [vp, u8p, u64]
        L.sfq_get_chain_index.restype = C.c_int64
        L.sfq_set_chain_index.argtypes = [vp, u8p, u64]
        L.sfq_synth_fastq.argtypes = [u64, u64, C.c_uint32, u64, C.c_int, u8p, u64]
        L.sfq_synth_fastq.restype = C.c_int64
        L.sfq_archive_write.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(vp), C.POINTER(u64)]
        L.sfq_pack_block_index.argtypes = [C.POINTER(BlockInfo), C.c_uint32, u8p, u64]
        L.sfq_pack_block_index.restype = C.c_int64
        _lib = L
    return _lib


def synth_fastq(n_reads, read_len=150, seed=1, kind=0, first_read=0) -> bytes:
    """Deterministic synthetic FASTQ (host-side generator inside the library)."""
    L = lib()
    need = L.sfq_synth_fastq(first_read, n_reads, read_len, seed, kind, None, 0)
    if need < 0:
        raise SfqError(need, "synth")
    buf = np.empty(need, np.uint8)
    got = L.sfq_synth_fastq(first_read, n_reads, read_len, seed, kind, buf.ctypes.data_as(C.c_void_p), need)
    if got != need:
        raise SfqError(got, "synth")
    return buf.tobytes()


class Encoded:
    """Host copy of one sfq_encode_blocks result."""

    def __init__(self, res, blocks, first_hdrs, data, prior=b"", chains=b"", rec_prior=b""):
        self.res, self.blocks, self.first_hdrs, self.data, self.prior, self.chains = res, blocks, first_hdrs, data, prior, chains
        self.rec_prior = rec_prior

    def clone(self):
        """A deep copy (the tests damage copies)."""
        blocks = (BlockInfo * len(self.blocks))()
        C.memmove(blocks, self.blocks, C.sizeof(blocks))
        res = Result()
        C.memmove(C.byref(res), C.byref(self.res), C.sizeof(Result))
        data = self.data.copy() if isinstance(self.data, np.ndarray) else bytes(self.data)
        return Encoded(res, blocks, bytes(self.first_hdrs), data, bytes(self.prior), bytes(self.chains), bytes(self.rec_prior))

    def stream(self, s, block=None) -> bytes:
        """Bytes of stream s (an id or a name): the whole concatenation, or one block's part."""
        if isinstance(s, str):
            s = STREAM_NAMES.index(s)
        off = self.res.stream_offset[s]
        if block is None:
            return bytes(self.data[off:off + self.res.stream_bytes[s]])
        for b in range(block):
            off += self.blocks[b].size[s]
        return bytes(self.data[off:off + self.blocks[block].size[s]])

    @property
    def payload_bytes(self):
        return int(self.res.total_bytes)

    @property
    def archive_bytes(self):
        """Everything a decoder needs: streams + first headers + quality prior + ~the block index."""
        return int(self.res.total_bytes) + len(self.first_hdrs) + len(self.prior) + len(self.chains) + len(self.rec_prior) + 14 * len(self.blocks)


class Context:
    def __init__(self, device=0, table_budget=None):
        self._h = C.c_void_p()
        rc = lib().sfq_ctx_create(C.byref(self._h), device)
        if rc != 0:
            raise SfqError(rc, "sfq_ctx_create failed (a HIP device is required; there is no CPU path)")
        if table_budget:
            lib().sfq_ctx_set_table_budget(self._h, int(table_budget))

    def close(self):
        if self._h:
            lib().sfq_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise SfqError(rc, lib().sfq_last_error(self._h).decode("latin1"))

    @property
    def handle(self):
        return self._h

    def index(self, n_blocks):
        blocks = (BlockInfo * n_blocks)()
        got = lib().sfq_get_block_index(self._h, blocks, n_blocks)
        if got < 0:
            self._check(got)
        return blocks

    def first_headers(self, nbytes):
        buf = C.create_string_buffer(max(int(nbytes), 1))
        self._check(lib().sfq_get_first_headers(self._h, buf, nbytes))
        return buf.raw[:nbytes]

    def prior(self) -> bytes:
        n = lib().sfq_get_qlt_prior(self._h, None, 0)
        if n <= 0:
            return b""
        buf = C.create_string_buffer(n)
        lib().sfq_get_qlt_prior(self._h, buf, n)
        return buf.raw[:n]

    def rec_prior(self) -> bytes:
        n = lib().sfq_get_rec_prior(self._h, None, 0)
        if n <= 0:
            return b""
        buf = C.create_string_buffer(n)
        lib().sfq_get_rec_prior(self._h, buf, n)
        return buf.raw[:n]

    def build_priors(self, d_ptr, nbytes, level=3, block_reads=BLOCK_AUTO, prior_step=PRIOR_AUTO, tables=1):
        """Priors of a device-resident text (quality prior, header prior); install elsewhere with set_priors()."""
        p = Params(level, block_reads, 0, 0, 0, 0, prior_step, tables, 0, 0)
        self._check(lib().sfq_build_priors(self._h, C.c_void_p(d_ptr), nbytes, C.byref(p)))
        return self.prior(), self.rec_prior()

    def count_priors(self, d_ptr, nbytes, level=3, block_reads=BLOCK_AUTO, prior_step=PRIOR_AUTO, tables=1, sample_scale=1):
        """The sample COUNTS of a device-resident text (every sample_scale-th record of the automatic sample): for several
        ranks that add their counts up (dist.allreduce_prior_counts) and code with prior_step = PRIOR_COUNTS."""
        p = Params(level, block_reads, 0, 0, 0, 0, prior_step, tables, 0, 0)
        self._check(lib().sfq_count_priors(self._h, C.c_void_p(d_ptr), nbytes, C.byref(p), sample_scale))

    @staticmethod
    def prior_counts_words(level=3):
        nq, nr = C.c_uint64(), C.c_uint64()
        lib().sfq_prior_counts_words(level, C.byref(nq), C.byref(nr))
        return nq.value, nr.value

    def get_prior_counts(self, level, d_qlt, d_rec):
        self._check(lib().sfq_get_prior_counts(self._h, level, C.c_void_p(d_qlt), C.c_void_p(d_rec)))

    def set_prior_counts(self, level, d_qlt, d_rec):
        self._check(lib().sfq_set_prior_counts(self._h, level, C.c_void_p(d_qlt), C.c_void_p(d_rec)))

    def set_priors(self, prior: bytes, rec_prior: bytes = b""):
        L = lib()
        self._check(L.sfq_set_qlt_prior(self._h, prior if prior else None, len(prior)))
        self._check(L.sfq_set_rec_prior(self._h, rec_prior if rec_prior else None, len(rec_prior)))

    def chains(self) -> bytes:
        n = lib().sfq_get_chain_index(self._h, None, 0)
        if n <= 0:
            return b""
        buf = C.create_string_buffer(n)
        lib().sfq_get_chain_index(self._h, buf, n)
        return buf.raw[:n]

    def encode_host(self, fastq: bytes, level=3, block_reads=0, gen_bits=0, models=0, kernel=0, prior_step=0, tables=0,
                    chain_reads=0, lds_rows=0) -> Encoded:
        L = lib()
        p = Params(level, block_reads, gen_bits, models, kernel, 0, prior_step, tables, chain_reads, lds_rows)
        res = Result()
        cap = L.sfq_encode_bound(len(fastq))
        out = np.empty(cap, np.uint8)
        src = np.frombuffer(fastq, np.uint8)
        self._check(L.sfq_encode_blocks_host(self._h, src.ctypes.data_as(C.c_void_p), len(fastq), C.byref(p),
                                             out.ctypes.data_as(C.c_void_p), cap, C.byref(res)))
        blocks = self.index(res.n_blocks)
        return Encoded(res, blocks, self.first_headers(res.first_hdr_bytes), out[:res.total_bytes].copy(), self.prior(), self.chains(), self.rec_prior())

    def encode_device(self, d_ptr, nbytes, d_out, out_cap, level=3, block_reads=0, gen_bits=0, models=0, kernel=0, qlt_only=False,
                      prior_step=0, tables=0, chain_reads=0, lds_rows=0):
        """Device-pointer entry point (ints from torch .data_ptr()). Returns the Result struct."""
        L = lib()
        p = Params(level, block_reads, gen_bits, models, kernel, 0, prior_step, tables, chain_reads, lds_rows)
        res = Result()
        f = L.sfq_encode_qlt_blocks if qlt_only else L.sfq_encode_blocks
        self._check(f(self._h, C.c_void_p(d_ptr), nbytes, C.byref(p), C.c_void_p(d_out), out_cap, C.byref(res)))
        return res

    def decode_device(self, blocks, first_hdrs: bytes, d_streams, stream_offset, d_out, out_cap, prior=b"", level=3, version=0,
                      chains=b"", lds_rows=0, rec_prior=b"", kernel=0):
        """Device-pointer decode (ints from torch .data_ptr()): returns (bytes written, Result)."""
        L = lib()
        self._check(L.sfq_set_qlt_prior(self._h, prior if prior else None, len(prior)))
        self._check(L.sfq_set_chain_index(self._h, chains if chains else None, len(chains)))
        self._check(L.sfq_set_rec_prior(self._h, rec_prior if rec_prior else None, len(rec_prior)))
        p = Params(level, 0, 0, 0, kernel, version, 0, 0, 0, lds_rows)
        res = Result()
        n = C.c_uint64()
        fb = np.frombuffer(first_hdrs if len(first_hdrs) else b"\0", np.uint8)
        soff = (C.c_uint64 * NSTREAMS)(*list(stream_offset))
        self._check(L.sfq_decode_blocks(self._h, C.byref(p), blocks, len(blocks), fb.ctypes.data_as(C.c_void_p), len(first_hdrs),
                                        C.c_void_p(d_streams), soff, C.c_void_p(d_out), out_cap, C.byref(n), C.byref(res)))
        return n.value, res

    def decode_host(self, enc_or_parts, level=3, version=0, out_cap=None, kernel=0, lds_rows=0) -> bytes:
        """Decode an Encoded (or a (blocks, first_hdrs, data, stream_offset) tuple) back to FASTQ text."""
        L = lib()
        prior = chains = rec_prior = b""
        if isinstance(enc_or_parts, Encoded):
            blocks, first, data, prior, chains = enc_or_parts.blocks, enc_or_parts.first_hdrs, enc_or_parts.data, enc_or_parts.prior, enc_or_parts.chains
            rec_prior = enc_or_parts.rec_prior
            soff = (C.c_uint64 * NSTREAMS)(*list(enc_or_parts.res.stream_offset))
        else:
            blocks, first, data, so = enc_or_parts[:4]
            if len(enc_or_parts) > 4:
                prior = enc_or_parts[4]
            if len(enc_or_parts) > 5:
                chains = enc_or_parts[5]
            if len(enc_or_parts) > 6:
                rec_prior = enc_or_parts[6]
            soff = (C.c_uint64 * NSTREAMS)(*so)
        self._check(L.sfq_set_qlt_prior(self._h, prior if prior else None, len(prior)))
        self._check(L.sfq_set_chain_index(self._h, chains if chains else None, len(chains)))
        self._check(L.sfq_set_rec_prior(self._h, rec_prior if rec_prior else None, len(rec_prior)))
        data = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8)) if not isinstance(data, np.ndarray) else data
        p = Params(level, 0, 0, 0, kernel, version, 0, 0, 0, lds_rows)
        res = Result()
        if out_cap is None:
            out_cap = 64 * len(data) + (1 << 20)
        out = np.empty(out_cap, np.uint8)
        n = C.c_uint64()
        fb = np.frombuffer(first if len(first) else b"\0", np.uint8)
        self._check(L.sfq_decode_blocks_host(self._h, C.byref(p), blocks, len(blocks), fb.ctypes.data_as(C.c_void_p), len(first),
                                             data.ctypes.data_as(C.c_void_p), len(data), soff,
                                             out.ctypes.data_as(C.c_void_p), out_cap, C.byref(n), C.byref(res)))
        return out[:n.value].tobytes()
