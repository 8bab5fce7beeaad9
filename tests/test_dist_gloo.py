"""CPU, world_size 2, gloo: the N>1 path's plumbing -- record sharding, the variable-size gather of the
per-rank compressed streams to the writer rank, and the merge of the block indexes.  The encoder inside
each rank is a stand-in here (the oracle, run per block, exactly what the GPU kernels are proven to equal
in test_gpu_parity.py); what this test checks is that shards + gather + merge reproduce the single-process
result byte for byte."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOCK = 300
NREC = 2000


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _encode_blocks(fq: bytes, level: int):
    """stand-in for sfq_encode_blocks: per block, the reference's streams for that block alone"""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import util
    from oracle import oracle as O
    blocks, payload, firsts = [], {}, []
    names = ["rec", "gen", "qlt", "gen.Ns", "gen.Nn", "rec.x", "usr.x", "usr.x.q", "usr.pfg", "usr.pfq"]
    per_stream = {n: [] for n in names}
    for chunk in util.split_records(fq, BLOCK):
        a = O.compress(chunk, level, gen_bits=18)
        first = a.info["rec.first"].encode("latin1")
        firsts.append(first)
        blocks.append({"n_records": chunk.count(b"\n") // 4, "first_hdr_len": len(first),
                       "size": [len(a.streams.get(n, b"")) for n in names]})
        for n in names:
            per_stream[n].append(a.streams.get(n, b""))
    return blocks, b"".join(firsts), b"".join(b"".join(per_stream[n]) for n in names), [sum(map(len, per_stream[n])) for n in names]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from slimfastq_amd import capi, dist as sdist
    lo, hi = sdist.shard_records(NREC, rank, world, BLOCK)
    fq = capi.synth_fastq(hi - lo, 100, seed=3, first_read=lo)        # every rank generates only its shard
    blocks, firsts, payload, totals = _encode_blocks(fq, 3)
    bufs = sdist.gather_bytes(torch.frombuffer(bytearray(payload), dtype=torch.uint8), dst=0)
    meta = [None] * world
    dist.gather_object((blocks, firsts, totals), meta if rank == 0 else None, dst=0)
    if rank == 0:
        index, first_blob = sdist.merge_indexes([m[0] for m in meta], [m[1] for m in meta])
        q.put((index, first_blob, [bytes(b.numpy()) for b in bufs], [m[2] for m in meta], (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_gather_merge_equals_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    index, first_blob, payloads, totals, _ = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from slimfastq_amd import capi, dist as sdist
    whole = capi.synth_fastq(NREC, 100, seed=3)
    blocks1, firsts1, payload1, totals1 = _encode_blocks(whole, 3)
    # shards are block aligned and cover the records exactly once
    assert sdist.shard_records(NREC, 0, 2, BLOCK)[1] == sdist.shard_records(NREC, 1, 2, BLOCK)[0]
    assert [b["n_records"] for b in index] == [b["n_records"] for b in blocks1]
    assert [b["size"] for b in index] == [b["size"] for b in blocks1]
    assert first_blob == firsts1
    assert index[0]["first_record"] == 0 and index[-1]["first_record"] == NREC - index[-1]["n_records"]
    # per stream: rank 0's part followed by rank 1's part == the single-process stream
    off = [0, 0]; single_off = 0
    for s in range(10):
        merged = b""
        for r in range(world):
            merged += payloads[r][off[r]:off[r] + totals[r][s]]
            off[r] += totals[r][s]
        assert merged == payload1[single_off:single_off + totals1[s]], s
        single_off += totals1[s]


def _worker_pipelined(rank, world, port, q):
    """bench.py's N > 1 loop: the payload of batch k travels (on a group of its own) while batch k + 1 is prepared and a
    small collective of the next batch runs on the default group; two buffers take turns."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    payload_group = dist.new_group(list(range(world)))
    sys.path.insert(0, ROOT)
    from slimfastq_amd import dist as sdist
    bufs = [torch.zeros(70000, dtype=torch.uint8) for _ in range(2)]
    got, flight = [], None
    for k in range(5):
        buf = bufs[k % 2]
        n = 1000 * (rank + 1) + 13000 * k                      # ragged: every rank and every batch its own size
        note = torch.tensor([k if rank == 0 else -1])
        dist.broadcast(note, 0)                                # (the shared prior's broadcast: default group, payload k - 1 in flight)
        assert int(note) == k
        buf[:n] = torch.arange(n, dtype=torch.int64).add(7 * k + rank).remainder(251).to(torch.uint8)
        if flight is not None:
            r = sdist.gather_bytes_finish(flight)
            if rank == 0:
                got.append([bytes(t.numpy()) for t in r])
        flight = sdist.gather_bytes_start(buf[:n], dst=0, p2p_group=payload_group)
    r = sdist.gather_bytes_finish(flight)
    if rank == 0:
        got.append([bytes(t.numpy()) for t in r])
        q.put(got)
    dist.barrier()
    dist.destroy_process_group()


def test_payload_of_one_batch_travels_while_the_next_is_prepared():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipelined, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(got) == 5
    for k, parts in enumerate(got):
        for rank, b in enumerate(parts):
            n = 1000 * (rank + 1) + 13000 * k
            want = bytes((torch.arange(n, dtype=torch.int64).add(7 * k + rank).remainder(251)).to(torch.uint8).numpy())
            assert b == want, (k, rank)


class _CountsOnlyCtx:
    """Stand-in for capi.Context in dist.allreduce_prior_counts: the counting kernels' part is played by the oracle's
    histogram over every `sample_scale`-th record of the rank's shard (the rule of sfq_count_priors), the count tables live in
    host memory.  What the test checks is the collective around them."""

    def __init__(self, fq, level):
        import ctypes
        self.fq, self.level, self.ct = fq, level, ctypes
        self.q = self.r = None

    def count_priors(self, d_ptr, nbytes, level=3, block_reads=None, tables=1, sample_scale=1):
        import numpy as np
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import util
        from oracle import oracle as O
        starts, lens = util.line_table(self.fq)
        self.q = O.qlt_histogram(self.fq, starts[3::4], np.minimum(lens[3::4], 4096), level, 0, sample_scale).astype(np.uint32)
        nrec = len(starts) // 4
        self.r = O.rec_count(self.fq, starts[0::4] + 1, lens[0::4] - 1, max(6, nrec // 8), 6, 8).astype(np.uint32)

    def get_prior_counts(self, level, d_qlt, d_rec):
        self.ct.memmove(d_qlt, self.q.ctypes.data, self.q.nbytes)
        self.ct.memmove(d_rec, self.r.ctypes.data, self.r.nbytes)

    def set_prior_counts(self, level, d_qlt, d_rec):
        self.ct.memmove(self.q.ctypes.data, d_qlt, self.q.nbytes)
        self.ct.memmove(self.r.ctypes.data, d_rec, self.r.nbytes)


def _worker_counts(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from slimfastq_amd import capi, dist as sdist
    lo, hi = sdist.shard_records(NREC, rank, world, BLOCK)
    fq = capi.synth_fastq(hi - lo, 100, seed=3, first_read=lo)
    ctx = _CountsOnlyCtx(fq, 3)
    ctx.count_priors(0, len(fq), level=3, sample_scale=world)
    mine = (ctx.q.copy(), ctx.r.copy())
    import unittest.mock as mock
    with mock.patch.object(torch.cuda, "current_stream", lambda dev=None: mock.Mock(synchronize=lambda: None)):
        sdist.allreduce_prior_counts(ctx, 0, len(fq), torch.device("cpu"), level=3, tables=1, via_cpu=True)
    q.put((rank, mine[0].tobytes(), mine[1].tobytes(), ctx.q.tobytes(), ctx.r.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_prior_counts_are_summed_over_the_ranks_and_identical_everywhere():
    """Multi-GPU priors without a serial head: every rank counts a share of the sample over its own
    shard, dist.allreduce_prior_counts adds the tables up -- every rank must end with the same sums, and they must be the
    sums of what the ranks counted."""
    import numpy as np
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_counts, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want_q = sum(np.frombuffer(g[1], np.uint32).astype(np.uint64) for g in got)
    want_r = sum(np.frombuffer(g[2], np.uint32).astype(np.uint64) for g in got)
    assert want_q.sum() > 0 and want_r.sum() > 0
    for g in got:
        assert np.array_equal(np.frombuffer(g[3], np.uint32), want_q.astype(np.uint32)), g[0]
        assert np.array_equal(np.frombuffer(g[4], np.uint32), want_r.astype(np.uint32)), g[0]
