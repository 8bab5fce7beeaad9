"""One FASTQ file over several GPUs -> one .sfq archive.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        -m slimfastq_amd.dist_compress in.fastq out.sfq [-l 3] [-B 1024]

Rank 0 builds ONE set of priors (quality prior, header prior) from its shard and broadcasts it (a few hundred KB); then
rank r compresses a contiguous, record-aligned byte range of the file on its GPU from that prior (the blocks are
independent: no data-path collective), and ONE exchange follows: every rank's compressed streams -- as they lie in
HBM -- plus its block index, first headers and chain index go to rank 0 (slimfastq_amd.dist.gather_bytes: RCCL over
xGMI with the "nccl" backend), which writes the archive: a rank codes its range in slabs (-S, as the CLI does) and every
slab is one SEGMENT, the priors stored once, decodable by `slimfastq-amd -d`.
"""
import argparse
import ctypes as C
import mmap
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

from . import capi
from . import dist as sdist


def record_start(buf, pos: int) -> int:
    """First record boundary at or after `pos` in a FASTQ buffer: a line that starts with '@' whose line + 2
    starts with '+' and whose lines + 1 and + 3 have equal length (a quality line may itself start with '@',
    so '@' alone is not enough), checked on two consecutive records."""
    n = len(buf)
    if pos <= 0:
        return 0
    p = buf.find(b"\n", pos - 1) + 1                        # start of the next line
    while 0 < p < n:
        q, good, checked = p, True, 0
        for _ in range(2):
            starts = [q]
            for _ in range(4):
                e = buf.find(b"\n", starts[-1])
                if e < 0:
                    break
                starts.append(e + 1)
            if len(starts) < 5:
                good = checked > 0                          # the file ends here: one whole record must have been seen
                break
            l1 = starts[2] - starts[1] - 1
            l3 = starts[4] - starts[3] - 1
            if buf[starts[0]:starts[0] + 1] != b"@" or buf[starts[2]:starts[2] + 1] != b"+" or l1 != l3:
                good = False
                break
            q = starts[4]
            checked += 1
        if good:
            return p
        p = buf.find(b"\n", p) + 1
    return n


def shard_bytes(buf, rank: int, world: int):
    """Record-aligned byte range of `rank`."""
    n = len(buf)
    lo = record_start(buf, n * rank // world)
    hi = record_start(buf, n * (rank + 1) // world) if rank + 1 < world else n
    return lo, hi


def put_v(out: bytearray, v: int):
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)


def assemble(parts, level: int, block_reads: int, orig_name: str, frozen=True, shared_prior=False):
    """parts: per rank, in rank order: dict(streams=[bytes] * NSTREAMS, blocks=BlockInfo array, first=bytes,
    prior=bytes, chains=bytes, rec_prior=bytes, raw=int, records=int).  With shared_prior only the first part carries
    the priors.  Returns (info_text, [(name, bytes)])."""
    L = capi.lib()
    parts = [p for p in parts if p["records"]]
    nblocks = sum(len(p["blocks"]) for p in parts)
    allb = (capi.BlockInfo * nblocks)()
    k, rec, hoff = 0, 0, 0
    for p in parts:
        for b in p["blocks"]:
            C.memmove(C.byref(allb[k]), C.byref(b), C.sizeof(capi.BlockInfo))
            allb[k].first_record = rec
            allb[k].first_hdr_off = hoff
            rec += b.n_records
            hoff += b.first_hdr_len
            k += 1
    need = L.sfq_pack_block_index(allb, nblocks, None, 0)
    idx = (C.c_uint8 * need)()
    L.sfq_pack_block_index(allb, nblocks, idx, need)
    info = [("whoami", "slimfastq"), ("version", "10"), ("config.level", str(level)), ("orig.filename", orig_name),
            ("orig.size", str(sum(p["raw"] for p in parts))), ("blk.reads", str(block_reads)), ("blk.count", str(nblocks)),
            ("num_records", str(rec))]
    streams = []
    for s, name in enumerate(capi.STREAM_NAMES):
        data = b"".join(p["streams"][s] for p in parts)
        if data:
            streams.append((name, data))
    streams.append(("blk.idx", bytes(idx)))
    streams.append(("blk.hdr", b"".join(p["first"] for p in parts)))
    for key, name in (("prior", "qlt.pri"), ("chains", "chn.idx"), ("rec_prior", "rec.pri")):
        data = b"".join(p[key] for p in parts)
        if data:
            streams.append((name, data))
    if frozen:
        info.append(("blk.tables", "1"))
    if len(parts) > 1:
        info.append(("seg.count", str(len(parts))))
        if shared_prior:
            info.append(("seg.shared_prior", "1"))
        si = bytearray()
        put_v(si, len(parts))
        for p in parts:
            put_v(si, len(p["blocks"])); put_v(si, len(p["prior"])); put_v(si, p["raw"])
            if frozen:
                put_v(si, len(p["chains"])); put_v(si, len(p["rec_prior"]))
        streams.append(("seg.idx", bytes(si)))
    return "".join("%s=%s\n" % kv for kv in info), streams


def write_archive(path: str, info_text: str, streams):
    L = capi.lib()
    n = len(streams)
    names = (C.c_char_p * n)(*[s[0].encode() for s in streams])
    bufs = [np.frombuffer(s[1], np.uint8) if len(s[1]) else np.zeros(1, np.uint8) for s in streams]
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    sizes = (C.c_uint64 * n)(*[len(s[1]) for s in streams])
    rc = L.sfq_archive_write(path.encode(), info_text.encode(), n, names, ptrs, sizes)
    if rc:
        raise capi.SfqError(rc, "cannot write " + path)


META_WORDS = capi.NSTREAMS + 7        # stream sizes, then: blocks, first, prior, chains, rec_prior bytes, raw bytes, records


def pack_part(streams_dev, res, blocks, first: bytes, prior: bytes, chains: bytes, rec_prior: bytes, raw: int):
    """One rank's contribution as ONE byte tensor on its device: the compressed streams as they lie in HBM, then the small
    index parts, then a trailer of int64 sizes (a flat layout: nothing is pickled)."""
    bb = b"".join(bytes(b) for b in blocks)
    tail = np.array([int(x) for x in res.stream_bytes] + [len(bb), len(first), len(prior), len(chains), len(rec_prior), raw, int(res.n_records)], np.int64)
    small = np.frombuffer(bb + first + prior + chains + rec_prior + tail.tobytes(), np.uint8)
    t = torch.from_numpy(small.copy()).to(streams_dev.device)
    # the streams are packed stream after stream from offset 0 (sfq_result.stream_offset): total_bytes of them
    return torch.cat([streams_dev[:int(res.total_bytes)], t])


def unpack_part(raw: bytes):
    tail = np.frombuffer(raw[-8 * META_WORDS:], np.int64)
    sizes = [int(x) for x in tail[:capi.NSTREAMS]]
    nbb, nfirst, nprior, nchains, nrp, rawbytes, records = (int(x) for x in tail[capi.NSTREAMS:])
    off, streams = 0, []
    for n in sizes:
        streams.append(raw[off:off + n]); off += n
    bsz = C.sizeof(capi.BlockInfo)
    blocks = [capi.BlockInfo.from_buffer_copy(raw[off + i * bsz:off + (i + 1) * bsz]) for i in range(nbb // bsz)]
    off += nbb
    first = raw[off:off + nfirst]; off += nfirst
    prior = raw[off:off + nprior]; off += nprior
    chains = raw[off:off + nchains]; off += nchains
    rec_prior = raw[off:off + nrp]
    return dict(streams=streams, blocks=blocks, first=first, prior=prior, chains=chains, rec_prior=rec_prior, raw=rawbytes, records=records)


def bcast_bytes(data: bytes, src: int, dev):
    """Broadcast a byte string from rank src (sizes first)."""
    n = torch.tensor([len(data) if dist.get_rank() == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, src)
    t = torch.from_numpy(np.frombuffer(data, np.uint8).copy()).to(dev) if dist.get_rank() == src else torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    if int(n.item()):
        dist.broadcast(t, src)
    return t.cpu().numpy().tobytes()


def main(argv=None):
    ap = argparse.ArgumentParser(prog="slimfastq_amd.dist_compress", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("fastq")
    ap.add_argument("sfq")
    ap.add_argument("-l", "--level", type=int, default=3)
    ap.add_argument("-B", "--block_reads", type=int, default=-1, help="records per block (default: automatic, about 376 KiB of text)")
    ap.add_argument("-A", "--adaptive", action="store_true", help="adaptive tables (a wavefront per block) instead of frozen tables")
    ap.add_argument("-S", "--slab", type=int, default=2048, help="MiB of text per call and archive segment (default 2048)")
    ap.add_argument("--backend", default="", help="torch.distributed backend (default: nccl = RCCL when every rank has its own GPU, else gloo)")
    args = ap.parse_args(argv)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ngpu = torch.cuda.device_count()
    dev = local % max(ngpu, 1)
    backend = args.backend or ("nccl" if ngpu >= world else "gloo")     # ranks sharing a GPU cannot use RCCL
    if world > 1:
        dist.init_process_group(backend, rank=rank, world_size=world,
                                **({"device_id": torch.device("cuda", dev)} if backend == "nccl" else {}))
    cdev = torch.device("cuda", dev)
    xdev = cdev if backend == "nccl" else torch.device("cpu")          # where the exchanged tensors live
    # the rank's share of the file, cut into slabs at record boundaries: one archive segment per slab (as the CLI's -S),
    # so that a rank's device memory is sized by the slab, not by its share of the file
    slab_bytes = max(1, args.slab) << 20
    f = open(args.fastq, "rb")
    mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) if os.fstat(f.fileno()).st_size else b""
    lo, hi = shard_bytes(mm, rank, world)
    cuts = [lo]
    while cuts[-1] < hi:
        want = cuts[-1] + slab_bytes
        nxt = hi if want >= hi else min(record_start(mm, want), hi)
        cuts.append(nxt if nxt > cuts[-1] else hi)
    spans = list(zip(cuts[:-1], cuts[1:]))
    per_gpu = (world + max(ngpu, 1) - 1) // max(ngpu, 1)            # ranks sharing one GPU share its memory
    budget = int(torch.cuda.get_device_properties(dev).total_memory * 0.6 / per_gpu) if per_gpu > 1 else None
    ctx = capi.Context(dev, table_budget=budget)
    tables = capi.TABLES_ADAPTIVE if args.adaptive else capi.TABLES_FROZEN
    br = capi.BLOCK_AUTO if args.block_reads < 0 else args.block_reads
    torch.cuda.set_device(dev)

    def to_device(text):
        return torch.from_numpy(np.frombuffer(text, np.uint8).copy()).to(cdev)

    # ONE prior for the whole file and no rank the others wait for: every rank counts a 1 / world share of the
    # sample over its first slab, the counts are summed over the ranks (dist.allreduce_prior_counts), every rank builds the
    # same priors from the sums (it takes them from its first encode call and codes its other slabs from them)
    prior = rec_prior = b""
    d_first = to_device(mm[spans[0][0]:spans[0][1]]) if spans else None
    counted = False
    if world > 1:
        if d_first is None:                       # (a rank without records still takes part in the collective)
            d_first = to_device(b"@e\nA\n+\nI\n")
            sdist.allreduce_prior_counts(ctx, d_first.data_ptr(), d_first.numel(), cdev, level=args.level, block_reads=br, tables=tables, via_cpu=backend != "nccl")
            d_first = None
        else:
            sdist.allreduce_prior_counts(ctx, d_first.data_ptr(), d_first.numel(), cdev, level=args.level, block_reads=br, tables=tables, via_cpu=backend != "nccl")
        counted = True
    shared = counted
    my_parts = []
    for k, (a, b) in enumerate(spans):
        d_in = d_first if k == 0 else to_device(mm[a:b])
        nb = b - a
        cap = capi.lib().sfq_encode_bound(nb)
        d_out = torch.empty(cap, dtype=torch.uint8, device=cdev)
        if shared and k:
            ctx.set_priors(prior, rec_prior)
        res = ctx.encode_device(d_in.data_ptr(), nb, d_out.data_ptr(), cap, level=args.level, block_reads=br,
                                prior_step=(capi.PRIOR_GIVEN if k else capi.PRIOR_COUNTS) if shared else capi.PRIOR_AUTO, tables=tables)
        if shared and k == 0:
            prior, rec_prior = ctx.prior(), ctx.rec_prior()           # the file's priors, the same on every rank
        blocks = list(ctx.index(res.n_blocks))
        # with a shared prior only the file's first segment carries it
        keep = (not shared) or (rank == 0 and k == 0)
        my_parts.append(pack_part(d_out, res, blocks, ctx.first_headers(res.first_hdr_bytes), ctx.prior() if keep else b"", ctx.chains(),
                                  ctx.rec_prior() if keep else b"", nb))
        del d_in, d_out
        d_first = None
    if len(mm):
        mm.close()
    f.close()
    # a rank's segments travel as one tensor: the parts back to back, then their lengths and their count (int64)
    trailer = np.array([int(t.numel()) for t in my_parts] + [len(my_parts)], np.int64)
    part_t = torch.cat(my_parts + [torch.from_numpy(np.frombuffer(trailer.tobytes(), np.uint8).copy()).to(cdev)])

    def split_parts(raw: bytes):
        n = int(np.frombuffer(raw[-8:], np.int64)[0])
        lens = [int(x) for x in np.frombuffer(raw[-8 * (n + 1):-8], np.int64)]
        out, off = [], 0
        for ln in lens:
            out.append(unpack_part(raw[off:off + ln])); off += ln
        return out

    if world > 1:
        # the one exchange: every rank's bytes to the writer, device to device (RCCL over xGMI) when each rank has a GPU
        got = sdist.gather_bytes(part_t.to(xdev), dst=0)
        parts = [p for g in got for p in split_parts(g.cpu().numpy().tobytes())] if rank == 0 else None
    else:
        parts = split_parts(part_t.cpu().numpy().tobytes())
    if rank == 0:
        first = [p for p in parts if p["records"]]
        info, streams = assemble(parts, args.level, int(first[0]["blocks"][0].n_records) if first else 0, args.fastq,
                                 frozen=not args.adaptive, shared_prior=shared)
        write_archive(args.sfq, info, streams)
        raw = sum(p["raw"] for p in parts)
        print("%s: %d bytes -> %s: %d bytes of streams, %d segment(s)" % (args.fastq, raw, args.sfq, sum(len(s[1]) for s in streams),
                                                                          len([p for p in parts if p["records"]])))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
