"""Multi-GPU plumbing for the one exchange step of the path: record blocks are sharded across ranks
(contiguous record ranges, so concatenation order = file order) and each rank's compressed streams go to
the writer rank.  Works with any torch.distributed backend: "nccl" (= RCCL over xGMI, CUDA tensors) in
bench.py / production, "gloo" (CPU tensors) in the CPU tests.  No data-path collective besides this gather.
"""
import torch
import torch.distributed as dist


def shard_records(n_records: int, rank: int, world: int, block_reads: int):
    """Contiguous, block-aligned record range [lo, hi) of `rank` (rank r gets blocks
    [r*B/G, (r+1)*B/G))."""
    n_blocks = (n_records + block_reads - 1) // block_reads
    b0 = rank * n_blocks // world
    b1 = (rank + 1) * n_blocks // world
    return min(b0 * block_reads, n_records), min(b1 * block_reads, n_records)


def gather_bytes_start(payload: torch.Tensor, dst: int = 0, group=None, p2p_group=None):
    """Variable-size gather of one uint8 tensor per rank to `dst` (there is no gatherv in RCCL: sizes via
    all_gather, then point-to-point sends -- one hop on the fully connected xGMI mesh).  The payload is on its
    way when this returns: gather_bytes_finish(handle) waits for it, so a caller can code its next batch of
    blocks in between (`payload` and the returned buffers must stay untouched until then).  `p2p_group`: a second
    group over the same ranks for the payload (its own communicator and stream), so that the small collectives of the
    next batch (the shared prior's broadcast, the sizes) do not queue behind a payload still in flight."""
    pg = p2p_group if p2p_group is not None else group
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = payload.device
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, torch.tensor([payload.numel()], dtype=torch.int64, device=dev), group=group) \
        if dev.type == "cuda" else dist.all_gather(list(sizes.split(1)), torch.tensor([payload.numel()], dtype=torch.int64), group=group)
    hs = [int(x) for x in sizes.cpu().tolist()]
    if rank == dst:
        bufs = [payload if r == rank else torch.empty(hs[r], dtype=torch.uint8, device=dev) for r in range(world)]
        ops = [dist.P2POp(dist.irecv, bufs[r], r, pg) for r in range(world) if r != rank and hs[r]]
    else:
        bufs = None
        ops = [dist.P2POp(dist.isend, payload, dst, pg)] if payload.numel() else []
    works = dist.batch_isend_irecv(ops) if ops else []
    return works, bufs, dev


def gather_bytes_finish(handle):
    """Waits for a gather started by gather_bytes_start; returns the list of tensors (rank order) on dst, None elsewhere."""
    works, bufs, dev = handle
    for w in works:
        w.wait()
    if dev.type == "cuda":
        torch.cuda.current_stream(dev).synchronize()       # (a "nccl" wait only orders the stream: the host must know too)
    return bufs


def gather_bytes(payload: torch.Tensor, dst: int = 0, group=None):
    return gather_bytes_finish(gather_bytes_start(payload, dst, group))


def merge_indexes(per_rank_blocks, per_rank_first_hdrs):
    """Concatenate per-rank block indexes (lists of dicts with 'n_records', 'first_hdr_len', 'size') into one
    file-order index: first_record / first_hdr_off are re-based."""
    out, rec, hoff = [], 0, 0
    for blocks in per_rank_blocks:
        for b in blocks:
            nb = dict(b)
            nb["first_record"] = rec
            nb["first_hdr_off"] = hoff
            rec += b["n_records"]
            hoff += b["first_hdr_len"]
            out.append(nb)
    return out, b"".join(per_rank_first_hdrs)
