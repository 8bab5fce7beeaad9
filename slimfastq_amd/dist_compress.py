"""One FASTQ file over several GPUs -> one .sfq archive.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        -m slimfastq_amd.dist_compress in.fastq out.sfq [-l 3] [-B 1024]

Rank r compresses a contiguous, record-aligned byte range of the file on its GPU (the blocks are independent:
no data-path collective), then ONE exchange: every rank's compressed streams, block index, first headers and
quality prior go to rank 0 (slimfastq_amd.dist.gather_bytes: RCCL over xGMI with the "nccl" backend), which writes
the archive -- one SEGMENT per rank, decodable by `slimfastq-amd -d`.
"""
import argparse
import ctypes as C
import mmap
import os
import pickle
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


def assemble(parts, level: int, block_reads: int, orig_name: str):
    """parts: per rank, in rank order: dict(streams=[bytes] * NSTREAMS, blocks=BlockInfo array, first=bytes,
    prior=bytes, raw=int, records=int).  Returns (info_text, [(name, bytes)])."""
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
    info = [("whoami", "slimfastq"), ("version", "7"), ("config.level", str(level)), ("orig.filename", orig_name),
            ("orig.size", str(sum(p["raw"] for p in parts))), ("blk.reads", str(block_reads)), ("blk.count", str(nblocks)),
            ("num_records", str(rec))]
    streams = []
    for s, name in enumerate(capi.STREAM_NAMES):
        data = b"".join(p["streams"][s] for p in parts)
        if data:
            streams.append((name, data))
    streams.append(("blk.idx", bytes(idx)))
    streams.append(("blk.hdr", b"".join(p["first"] for p in parts)))
    prior = b"".join(p["prior"] for p in parts)
    if prior:
        streams.append(("qlt.pri", prior))
    if len(parts) > 1:
        info.append(("seg.count", str(len(parts))))
        si = bytearray()
        put_v(si, len(parts))
        for p in parts:
            put_v(si, len(p["blocks"])); put_v(si, len(p["prior"])); put_v(si, p["raw"])
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


def main(argv=None):
    ap = argparse.ArgumentParser(prog="slimfastq_amd.dist_compress", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("fastq")
    ap.add_argument("sfq")
    ap.add_argument("-l", "--level", type=int, default=3)
    ap.add_argument("-B", "--block_reads", type=int, default=-1, help="records per block (default: automatic, about 376 KiB of text)")
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
    with open(args.fastq, "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        lo, hi = shard_bytes(mm, rank, world)
        text = mm[lo:hi]
        mm.close()
    per_gpu = (world + max(ngpu, 1) - 1) // max(ngpu, 1)            # ranks sharing one GPU share its memory
    budget = int(torch.cuda.get_device_properties(dev).total_memory * 0.6 / per_gpu) if per_gpu > 1 else None
    ctx = capi.Context(dev, table_budget=budget)
    part = dict(streams=[b""] * capi.NSTREAMS, blocks=[], first=b"", prior=b"", raw=len(text), records=0)
    if text:
        enc = ctx.encode_host(text, level=args.level, block_reads=capi.BLOCK_AUTO if args.block_reads < 0 else args.block_reads,
                              prior_step=capi.PRIOR_AUTO)
        part.update(streams=[enc.stream(s) for s in capi.STREAM_NAMES], blocks=list(enc.blocks), first=enc.first_hdrs,
                    prior=enc.prior, records=int(enc.res.n_records))
    if world > 1:
        # the one exchange: stream bytes as one tensor per rank (RCCL / gloo), the small index as pickled bytes beside them
        meta = pickle.dumps(dict(sizes=[len(s) for s in part["streams"]], blocks=[bytes(b) for b in part["blocks"]], first=part["first"],
                                 prior=part["prior"], raw=part["raw"], records=part["records"]))
        blob = b"".join(part["streams"]) + meta + len(meta).to_bytes(8, "little")
        t = torch.from_numpy(np.frombuffer(blob, np.uint8).copy())
        if backend == "nccl":
            t = t.cuda(dev)
        got = sdist.gather_bytes(t, dst=0)
        if rank == 0:
            parts = []
            for g in got:
                raw = g.cpu().numpy().tobytes()
                mlen = int.from_bytes(raw[-8:], "little")
                m = pickle.loads(raw[-8 - mlen:-8])
                off, streams = 0, []
                for n in m["sizes"]:
                    streams.append(raw[off:off + n]); off += n
                blocks = [capi.BlockInfo.from_buffer_copy(b) for b in m["blocks"]]
                parts.append(dict(streams=streams, blocks=blocks, first=m["first"], prior=m["prior"], raw=m["raw"], records=m["records"]))
    else:
        parts = [part]
    if rank == 0:
        first = [p for p in parts if p["records"]]
        info, streams = assemble(parts, args.level, int(first[0]["blocks"][0].n_records) if first else 0, args.fastq)
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
