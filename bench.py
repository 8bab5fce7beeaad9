#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: MB/s of raw FASTQ compressed (+ ratio vs reference) on synthetic
150 bp Illumina reads at -l 3, full qlts+gens+recs path, one process per GPU.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (framing + quality/base/header models + range coders + packing) over
the rank's FASTQ text, which is already resident in HBM when the timed region starts.  For N > 1 each
rank codes its own shard of records (weak scaling: per-GPU work fixed) and the step ends with the RCCL
gather of the compressed streams to rank 0.  Prints ONE JSON line on rank 0.

Beside the contract's fields the line carries, at N = 1: `roofline` (the base kernel: algorithmic bytes over its
launch time, the PMC-measured HBM traffic from profiles/, the practical random-sector peak), `cpu_baseline` (the
compiled reference, or the oracle port, on a bounded sample on one host core), `ratio_vs_reference` (same sample),
and `decode` (the same blocks decoded in HBM after the timed steps and compared with the input; never in `value`).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from slimfastq_amd import capi  # noqa: E402
from slimfastq_amd import dist as sdist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md "Chip-level parameters")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=0, help="records per GPU (default: 10M x 150 bp, the BASELINE config; 60k for --kind 1 long reads)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--level", type=int, default=3)
    ap.add_argument("--block-reads", type=int, default=0,
                    help="records per block (default: 1024, which is also what the library's automatic choice gives for 150 bp reads; "
                         "automatic for --kind 1 long reads)")
    ap.add_argument("--workload", choices=["full", "qlt"], default="full")
    ap.add_argument("--kind", type=int, default=0, help="0 = 150 bp-style Illumina reads, 1 = 10-50 kb long reads (BASELINE config 5), 2 = 4-level binned qualities, "
                    "3 = bases sampled from a 10 Mbp genome (coverage = reads x length / 1e7)")
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--tables", type=int, default=1, help="1 = frozen tables, one chain per lane (the default of block format 7); "
                    "0 = adaptive tables, a wavefront per block (the reference's per-symbol updates)")
    ap.add_argument("--chain-reads", type=int, default=0)
    ap.add_argument("--lds-rows", type=int, default=0)
    ap.add_argument("--dec-lds-rows", type=int, default=0, help="decode leg: quality rows the decoder stages in LDS (sfq_params.lds_rows)")
    ap.add_argument("--no-adaptive-leg", action="store_true", help="skip the secondary measurement with adaptive tables")
    ap.add_argument("--prior-step", type=int, default=-1, help="-1 = auto warm start (default), 0 = cold blocks, N = every N-th record")
    ap.add_argument("--models", type=int, default=0, help="debug: SFQ_M_* mask (1 rec, 2 gen, 4 qlt, 8 usr)")
    ap.add_argument("--cpu-sample-reads", type=int, default=600_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the decode-and-compare leg after the timed steps")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal: take the N > 1 path (process group, shared prior, gather) with one rank")
    ap.add_argument("--no-format6-leg", action="store_true", help="skip the format-6 (one block, the reference's own streams) encode / decode timing")
    ap.add_argument("--format6-reads", type=int, default=60_000, help="records of the format-6 leg (a single serial chain per stream: MB/s, not GB/s)")
    ap.add_argument("--no-c4-leg", action="store_true", help="--gpus 8 only: skip BASELINE configuration C4 (100 M reads at -l 4 over the 8 GPUs) behind the timed steps")
    ap.add_argument("--c4-reads-per-gpu", type=int, default=12_500_000)
    ap.add_argument("--no-genome-leg", action="store_true", help="skip the genome-sampled secondary workload (the honest stress of the base model, SURVEY 8d)")
    ap.add_argument("--genome-reads", type=int, default=10_000_000)
    ap.add_argument("--genome-ratio-reads", type=int, default=2_000_000, help="records the reference itself codes for the genome leg's ratio (30x coverage of the 10 Mbp genome)")
    ap.add_argument("--no-size-sweep", action="store_true", help="skip the size sweep (prefixes of the same text: 0.25 / 0.5 / 1 / 2 GB and all of it, encode + decode)")
    ap.add_argument("--sweep-gb", type=str, default="0.25,0.5,1,2", help="size sweep: prefix sizes in GB (the whole text is always the last point)")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the PCIe-inclusive host-to-host encode")
    ap.add_argument("--sweep-only", action="store_true", help="debug: the size sweep alone (no decode / adaptive / cpu / format-6 / genome legs)")
    args = ap.parse_args()
    if args.sweep_only:
        args.no_decode = args.no_adaptive_leg = args.no_cpu_baseline = args.no_format6_leg = args.no_genome_leg = args.no_host_leg = True
    if args.reads <= 0:
        args.reads = 60_000 if args.kind == 1 else 10_000_000
    if args.kind == 1:
        args.cpu_sample_reads = min(args.cpu_sample_reads, 6_000)      # long reads: ~60 kB per record
    if args.block_reads <= 0:
        args.block_reads = capi.BLOCK_AUTO if args.kind == 1 else 1024
    return args


def workload_name(args):
    what = "full qlts+gens+recs" if args.workload == "full" else "qlts-only kernel"
    what += ", frozen tables (one chain per lane)" if args.tables else ", adaptive tables (a wavefront per block)"
    if args.kind == 1:
        return "synthetic %d long reads (10-50 kb, log-uniform) per GPU, %s, -l %d" % (args.reads, what, args.level)
    flavour = {0: "Illumina reads", 2: "Illumina reads with 4-level binned qualities", 3: "reads sampled from a 10 Mbp genome"}.get(args.kind, "reads")
    n = "%dM" % (args.reads // 1_000_000) if args.reads >= 1_000_000 and args.reads % 1_000_000 == 0 else str(args.reads)
    return "synthetic %s x %d bp %s per GPU, %s, -l %d" % (n, args.read_len, flavour, what, args.level)


def cpu_baseline(args, seed):
    """The reference's single-thread CPU path on a bounded sample of the same workload (rank 0, N=1).
    kind "reference" = oracle/_ref/slimfastq_ref (the compiled reference) when it travelled with the
    tree; else kind "port" = oracle/sfq_oracle.c.  Both are test infrastructure, used here only as the
    reported baseline."""
    from oracle import oracle as O
    n = args.cpu_sample_reads
    fq = capi.synth_fastq(n, args.read_len, seed=seed, kind=args.kind)
    out = {"cores": 1, "sample": "%d x %d bp reads (%.1f MB), -l %d, first records of the same synthetic stream"
           % (n, args.read_len, len(fq) / 1e6, args.level), "unit": "MB/s"}
    t0 = time.perf_counter()
    a = O.compress(fq, args.level)
    t_port = time.perf_counter() - t0
    ref_payload = a.payload_bytes() - len(a.streams["<info>"])
    out.update(kind="port", value=round(len(fq) / 1e6 / t_port, 2), port_MBps=round(len(fq) / 1e6 / t_port, 2))
    t0 = time.perf_counter()
    back = O.decompress(a.image)
    out["decode_value"] = round(len(fq) / 1e6 / (time.perf_counter() - t0), 2)      # the way back, same sample (UsrLoad::decode, usrs.cpp:539-574)
    assert back == fq
    if O.ref_binary():
        try:
            t0 = time.perf_counter()
            img = O.ref_compress(fq, args.level)
            t_ref = time.perf_counter() - t0
            assert O.parse(img).streams == a.streams
            out.update(kind="reference", value=round(len(fq) / 1e6 / t_ref, 2))
            t0 = time.perf_counter()
            back = O.ref_decompress(img)
            out["decode_value"] = round(len(fq) / 1e6 / (time.perf_counter() - t0), 2)
            assert back == fq
        except Exception as e:  # the binary may not run on this host; keep the port number
            out["reference_error"] = str(e)[:100]
    return out, fq, ref_payload


def sweep_points(d_in, nbytes, args):
    """(records, bytes) of the size sweep's prefixes: the CPU baseline's sample, the sizes of --sweep-gb, the whole text."""
    CH = 1 << 30                                              # (torch.nonzero takes at most 2^31 elements a call)
    chunk_nl = [int((d_in[o:o + CH] == 10).sum().item()) for o in range(0, nbytes, CH)]

    def end_of_line(k):                                       # offset behind the k-th newline (1-based)
        o = 0
        for c in chunk_nl:
            if k <= c:
                return o + int(torch.nonzero(d_in[o:o + CH] == 10).flatten()[k - 1].item()) + 1
            k -= c; o += CH
        return nbytes
    per_rec = nbytes / args.reads
    points = [min(args.cpu_sample_reads, args.reads)] if not args.no_cpu_baseline else []
    for gb in [float(x) for x in args.sweep_gb.split(",") if x]:
        n = int(gb * 1e9 / per_rec)
        if 0 < n < args.reads and all(abs(n - q) > 0.2 * n for q in points):
            points.append(n)
    points = sorted(set(points + [args.reads]))
    return [(n, nbytes if n == args.reads else end_of_line(4 * n)) for n in points]


class ReferenceRuns:
    """The compiled reference (oracle/_ref, test infrastructure: the reported baseline only) over prefixes of the host text, one
    process per prefix, fed through stdin by a thread each, on the upper half of the host's cores (this process keeps to the lower half while
    they run) -- collected at the end of the run: the sizes of the reference's streams at every point of the size sweep, without a serial
    hour of CPU time.  Their MB/s is NOT the one-core baseline (cpu_baseline is): several of them share the memory system."""

    def __init__(self, fq, cuts, level):
        import subprocess, tempfile, threading
        from oracle import oracle as O
        self.O, self.jobs, self.dir = O, [], tempfile.mkdtemp(prefix="sfq_ref_")
        exe = O.ref_binary()
        if not exe:
            return
        view = memoryview(fq)
        for cut in cuts:
            out = os.path.join(self.dir, "%d.sfq" % cut)
            p = subprocess.Popen([exe, "-f", out, "-O", "-l", str(level), "-q"], stdin=subprocess.PIPE, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                 preexec_fn=self._away)

            def feed(p=p, cut=cut):
                try:
                    p.stdin.write(view[:cut]); p.stdin.close()
                except OSError:
                    pass
            t = threading.Thread(target=feed, daemon=True); t.start()
            self.jobs.append((cut, p, t, out, time.perf_counter()))

    @staticmethod
    def _away():
        # the legs timed while these run (format 6, the genome-sampled call) are launch-bound on this process's thread: the reference's
        # processes keep to the upper half of the cores this process may use, and this process to the lower half (ADVICE, round 4)
        try:
            cores = sorted(os.sched_getaffinity(0))
            if len(cores) >= 4:
                os.sched_setaffinity(0, set(cores[len(cores) // 2:]))
        except (AttributeError, OSError):
            pass

    def collect(self):
        """{bytes of text: (bytes of the reference's streams, seconds)}"""
        res = {}
        for cut, p, t, out, t0 in self.jobs:
            p.wait(); t.join()
            dt = time.perf_counter() - t0
            try:
                a = self.O.parse(open(out, "rb").read())
                res[cut] = (a.payload_bytes() - len(a.streams["<info>"]), dt)
            except Exception:
                pass
            try:
                os.remove(out)
            except OSError:
                pass
        try:
            os.rmdir(self.dir)
        except OSError:
            pass
        return res


def size_sweep(ctx, d_in, nbytes, d_out, cap, args, prior_step, models, points):
    """The same call over prefixes of the text (whole records): encode and decode at every size, device-resident, each
    timed on its own (the reference's loop, usrs.cpp:392-407, is size-agnostic: so should this be)."""
    d_back = torch.empty(nbytes + 4096, dtype=torch.uint8, device="cuda")
    rows = []
    for n, cut in points:
        kw = dict(level=args.level, block_reads=args.block_reads, models=models, kernel=args.kernel, prior_step=prior_step,
                  tables=args.tables, chain_reads=args.chain_reads, lds_rows=args.lds_rows)
        ctx.encode_device(d_in.data_ptr(), cut, d_out.data_ptr(), cap, **kw)
        te = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = ctx.encode_device(d_in.data_ptr(), cut, d_out.data_ptr(), cap, **kw)
            torch.cuda.synchronize(); te.append(time.perf_counter() - t0)
        row = {"reads": n, "raw_bytes": cut, "encode_MBps": round(cut / min(te) / 1e6, 1), "encode_ms": round(min(te) * 1e3, 3),
               "ratio": round(cut / r.total_bytes, 4), "chains": int(r.n_chains),
               "coder_ms": {"qlt": round(r.coder_ms[0], 3), "gen": round(r.coder_ms[1], 3), "rec": round(r.coder_ms[2], 3)},
               "device_ms": round(r.kernel_ms[capi.T_TOTAL], 3)}
        if not models:
            blocks = ctx.index(r.n_blocks)
            first = ctx.first_headers(r.first_hdr_bytes)
            prior, chains, rec_prior = ctx.prior(), ctx.chains(), ctx.rec_prior()
            row["archive_bytes"] = int(r.total_bytes) + len(first) + len(prior) + len(chains) + len(rec_prior) + 14 * len(blocks)
            soff = list(r.stream_offset)
            td = []
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                got, rd = ctx.decode_device(blocks, first, d_out.data_ptr(), soff, d_back.data_ptr(), d_back.numel(), prior=prior, level=args.level,
                                            chains=chains, rec_prior=rec_prior, lds_rows=args.dec_lds_rows)
                torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
            row.update(decode_MBps=round(cut / min(td[1:]) / 1e6, 1), decode_ms=round(min(td[1:]) * 1e3, 3),
                       round_trip_identical=bool(got == cut and torch.equal(d_back[:cut], d_in[:cut])),
                       decode_phase_ms={"qlt": round(rd.kernel_ms[capi.T_QLT], 3), "gen": round(rd.kernel_ms[capi.T_GEN], 3), "rec": round(rd.kernel_ms[capi.T_REC], 3),
                                        "assemble": round(rd.kernel_ms[capi.T_PACK], 3)})
        rows.append(row)
    del d_back
    return rows


def _varints(blob: bytes, n: int):
    """the first n varints of a blob ("chn.idx": records per chain, flags -- bit 0: the base tables are in use)"""
    out, p = [], 0
    while len(out) < n and p < len(blob):
        v = sh = 0
        while True:
            c = blob[p]; p += 1
            v |= (c & 0x7f) << sh; sh += 7
            if not c & 0x80:
                break
        out.append(v)
    return out + [0] * (n - len(out))


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path to measure)")
    torch.cuda.set_device(local_rank)
    dist = None
    multi = world > 1 or args.force_dist
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        payload_group = dist.new_group(list(range(world)))          # the streams travel on a communicator of their own
    seed = 1
    prior_step = capi.PRIOR_AUTO if args.prior_step < 0 else args.prior_step
    models = 0 if args.workload == "full" else capi.M_QLT
    if args.models:
        models = args.models

    # ---- synthetic input, resident in HBM before anything is timed ----
    t0 = time.perf_counter()
    # the synthetic text is built in host memory: bound it (long reads average ~60 kB per record)
    est = args.reads * (60_000 if args.kind == 1 else 2 * args.read_len + 70)
    if est > 24 << 30:
        raise SystemExit("bench.py: %d records of kind %d would need ~%d GB of host memory; lower --reads" % (args.reads, args.kind, est >> 30))
    fq = capi.synth_fastq(args.reads, args.read_len, seed=seed, first_read=rank * args.reads, kind=args.kind)
    t_gen = time.perf_counter() - t0
    nbytes = len(fq)
    # the symbols the models see (the roofline's algorithmic bytes): counted where records differ in length, else reads x length
    if args.kind == 1:
        a8 = np.frombuffer(fq, np.uint8)
        nl = np.flatnonzero(a8 == 10)
        starts = np.concatenate(([0], nl[:-1] + 1)); lens = nl - starts
        n_bases, n_quals, hdr_text = int(lens[1::4].sum()), int(lens[3::4].sum()), int(lens[0::4].sum() + lens[2::4].sum()) + 2 * args.reads
        del a8, nl, starts, lens
    else:
        n_bases = n_quals = args.reads * args.read_len
        hdr_text = nbytes - args.reads * (2 * args.read_len + 2)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                       # read-only source: it is only copied to the device
        d_in = torch.from_numpy(np.frombuffer(fq, np.uint8)).cuda(non_blocking=False)
    fq_host = fq if (not multi and not args.no_cpu_baseline and not args.no_size_sweep) else None       # (kept for the reference's runs over its prefixes)
    del fq
    ctx = capi.Context(local_rank)
    cap = capi.lib().sfq_encode_bound(nbytes)
    # N > 1: the streams of step k travel to the writer rank while step k + 1 is coded -- two output buffers take turns
    d_outs = [torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(2 if multi else 1)]
    d_out = d_outs[0]
    torch.cuda.synchronize()
    flight = {"n": 0, "gather": None}

    def land():
        if flight["gather"] is not None:
            sdist.gather_bytes_finish(flight["gather"])
            flight["gather"] = None

    def step(tables=args.tables, src=None, level=None):
        # (src: another text resident in HBM -- (tensor, bytes, output capacity, two output tensors) -- and level: the C4 leg below)
        nonlocal_in, nonlocal_n, nonlocal_cap, nonlocal_outs = (d_in, nbytes, cap, d_outs) if src is None else src
        lvl = args.level if level is None else level
        return _step(tables, nonlocal_in, nonlocal_n, nonlocal_cap, nonlocal_outs, lvl)

    def _step(tables, d_in, nbytes, cap, d_outs, level):
        ps = prior_step
        if multi and prior_step == capi.PRIOR_AUTO:
            # one prior for the whole job and no rank the others wait for: every rank counts a 1 / world share of the
            # sample over its own shard, the count tables are summed over the ranks (all_reduce: 17 MiB), every rank builds the
            # same priors from the sums -- so a record block's bytes do not depend on how many GPUs shared the file
            sdist.allreduce_prior_counts(ctx, d_in.data_ptr(), nbytes, d_in.device, level=level, block_reads=args.block_reads, tables=tables)
            ps = capi.PRIOR_COUNTS
        buf = d_outs[flight["n"] % len(d_outs)]
        flight["n"] += 1
        res = ctx.encode_device(d_in.data_ptr(), nbytes, buf.data_ptr(), cap, level=level,
                                block_reads=args.block_reads, models=models, kernel=args.kernel, prior_step=ps,
                                tables=tables, chain_reads=args.chain_reads, lds_rows=args.lds_rows)
        if multi:
            # the path's one exchange step: compressed streams to the writer rank, over RCCL/xGMI.  The previous step's
            # streams have had this step's coding to arrive; this step's leave now (the last ones land inside the timed region)
            land()
            flight["gather"] = sdist.gather_bytes_start(buf[:res.total_bytes], dst=0, p2p_group=payload_group)
        return res

    for _ in range(args.warmup):
        res = step()
    land()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    phase = np.zeros(8)
    coder = np.zeros(4)
    for _ in range(args.steps):
        res = step()
        phase += np.array(list(res.kernel_ms))
        coder += np.array(list(res.coder_ms))
    land()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    phase /= max(args.steps, 1)
    coder /= max(args.steps, 1)
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot = torch.tensor([nbytes, res.total_bytes], dtype=torch.float64, device="cuda")
        dist.all_reduce(tot)
        all_in, all_out = float(tot[0].item()), float(tot[1].item())
    else:
        all_in, all_out = float(nbytes), float(res.total_bytes)

    # ---- N = 8: BASELINE.json's configuration C4 beside the headline -- 100 M x 150 bp reads at -l 4, 12.5 M a rank, the same step
    #      (shared prior by all-reduce, the streams to the writer rank over RCCL), timed the same way; never part of `value`
    c4 = None
    if dist and (world == 8 or os.environ.get("SFQ_BENCH_C4") == "1") and not args.no_c4_leg and args.kind == 0 and args.workload == "full" and not args.models:
        try:
            del d_in
            d_outs.clear(); d_out = None
            torch.cuda.empty_cache()
            n4 = args.c4_reads_per_gpu
            f4 = capi.synth_fastq(n4, args.read_len, seed=seed, first_read=rank * n4, kind=0)
            nb4 = len(f4)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                t4 = torch.from_numpy(np.frombuffer(f4, np.uint8)).cuda()
            del f4
            cap4 = capi.lib().sfq_encode_bound(nb4)
            outs4 = [torch.empty(cap4, dtype=torch.uint8, device="cuda") for _ in range(2)]
            src4 = (t4, nb4, cap4, outs4)
            step(src=src4, level=4); land()
            dist.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                r4 = step(src=src4, level=4)
            land(); dist.barrier(); torch.cuda.synchronize()
            d4 = time.perf_counter() - t0
            t = torch.tensor([d4], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tot = torch.tensor([nb4, r4.total_bytes], dtype=torch.float64, device="cuda"); dist.all_reduce(tot)
            c4 = {"workload": "BASELINE C4: synthetic %d x %d bp reads (%d a GPU), -l 4, record blocks sharded over 8 GPUs, RCCL gather" % (n4 * world, args.read_len, n4),
                  "value": round(float(tot[0].item()) * 3 / float(t.item()) / 1e6, 2), "unit": "MB/s", "ms_per_step": round(float(t.item()) / 3 * 1e3, 3),
                  "ratio": round(float(tot[0].item()) / float(tot[1].item()), 4), "steps": 3}
            del t4, outs4
        except Exception as e:                                # (a leg, not the measurement: the headline line must come out whatever happens here)
            c4 = {"error": str(e)[:200]}
    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return
    ms_per_step = dt / args.steps * 1e3
    value = all_in * args.steps / dt / 1e6

    # ---- roofline for the dominant kernel (device time measured with HIP events on the context's stream) ----
    names = {capi.T_QLT: "qlt", capi.T_GEN: "gen", capi.T_REC: "rec"}
    sb = list(res.stream_bytes)
    hdr_bytes = hdr_text - 4 * args.reads                    # header and '+' lines without their newlines
    alg = {capi.T_QLT: n_quals + sb[2], capi.T_GEN: n_bases + sb[1] + sb[3] + sb[4], capi.T_REC: hdr_bytes + sb[0] + sb[5]}
    # The three coding kernels overlap on the chip.  "Dominant" = the one whose launches last longest: sfq_result.coder_ms,
    # HIP events recorded around the kernel's launch on the stream it is launched on (a kernel trace shows the same
    # duration for it: profiles/); a model's PHASE (phase_ms) also holds its counting passes and row building.
    cslot = {capi.T_QLT: 0, capi.T_GEN: 1, capi.T_REC: 2}
    kname = ({capi.T_QLT: "k_qlt_encode_c", capi.T_GEN: "k_gen_encode_c", capi.T_REC: "k_rec_tokens"} if args.tables and args.block_reads else
             {capi.T_QLT: "k_qlt_encode_k2", capi.T_GEN: "k_gen_encode_k", capi.T_REC: "k_rec_encode_w_fast"})
    # (round 5: the longest PHASE decides -- a model whose time is its counting / indexing passes, the bases of reads that overlap,
    #  used to be reported through a chain kernel that was a fraction of it.  Where the phase is one coding kernel and little else
    #  -- its launch at least four fifths of the phase -- the entry is that kernel's, as before; else the phase's, all its kernels named)
    base_on = bool(args.tables and args.block_reads and (_varints(ctx.chains(), 2)[1] & 1))
    match_on = bool(args.tables and args.block_reads and (_varints(ctx.chains(), 2)[1] & 32))
    if match_on:
        kname[capi.T_GEN] = "k_gm_code"
    # (the longest single piece decides: a model's coding kernel, or what its phase holds besides -- a phase that is long because its
    #  kernels WAIT, the header coder confined to the CUs the quality chains leave, is not where the call's time goes)
    dom = max(names, key=lambda k: max(float(coder[cslot[k]]), float(phase[k]) - float(coder[cslot[k]])))
    dom_ms = float(coder[cslot[dom]])
    over = "kernel"
    if dom_ms < 0.5 * float(phase[dom]):
        over = "phase"
        dom_ms = float(phase[dom])
        if dom == capi.T_GEN and match_on:
            kname[dom] = "k_gm_stage + k_gm_insert + k_gm_plan + k_gm_code"
        elif dom == capi.T_GEN and base_on:
            kname[dom] = "k_gen_bin + k_gen_bin_count + k_gen_encode_c"
    achieved = alg[dom] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": names[dom] + "_encode", "kernel_name": kname[dom], "over": over, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                "alg_bytes_per_launch": int(alg[dom]), "avg_ms": round(dom_ms, 3),
                "coder_ms": {"qlt": round(float(coder[0]), 3), "gen": round(float(coder[1]), 3), "rec": round(float(coder[2]), 3)},
                "whole_path_GBps": round((nbytes + res.total_bytes) / (phase[capi.T_TOTAL] * 1e-3) / 1e9, 3)}

    # HBM traffic of that kernel per launch: PMC counters cannot be collected from inside this process, so the
    # figure comes from the committed rocprofv3 --pmc passes over this very configuration (profiles/), else null
    try:
        import glob
        latest = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_traffic.json")))[-1]
        pmc = json.load(open(latest))
        c = pmc["config"]
        if (args.reads, args.read_len, args.level, args.kind, args.block_reads, args.kernel, args.workload, args.tables) == \
           (c["reads"], c["read_len"], c["level"], c["kind"], c["block_reads"], c["kernel"], "full", c.get("tables", 0)) and prior_step == capi.PRIOR_AUTO:
            k = pmc["kernels"][names[dom] + "_encode"]
            if over == "kernel":
                roofline["traffic"] = k["fetch_bytes"] + k["write_bytes"]
            roofline["traffic_source"] = pmc["source"]
    except (OSError, KeyError, ValueError, IndexError):
        pass

    out = {"metric": "MB/s FASTQ compressed", "value": round(value, 2), "unit": "MB/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32 integer", "data": "synthetic",
           "config": {"workload": workload_name(args),
                      "level": args.level, "block_reads": int(ctx.index(res.n_blocks)[0].n_records) if args.block_reads == capi.BLOCK_AUTO else args.block_reads,
                      "prior_step": args.prior_step, "blocks_per_gpu": int(res.n_blocks),
                      "tables": "frozen" if args.tables else "adaptive", "chains_per_gpu": int(res.n_chains),
                      "raw_bytes_per_gpu": nbytes, "parallelism": "blocks sharded x%d, RCCL gather" % world if world > 1 else "1 GPU"},
           "ratio": round(all_in / all_out, 4),
           "phase_ms": {"frame": round(phase[capi.T_FRAME], 3), "qlt": round(phase[capi.T_QLT], 3), "gen": round(phase[capi.T_GEN], 3),
                        "rec": round(phase[capi.T_REC], 3), "usr": round(phase[capi.T_USR], 3), "pack": round(phase[capi.T_PACK], 3),
                        "device_total": round(phase[capi.T_TOTAL], 3)},
           "roofline": roofline, "synth_s": round(t_gen, 2)}
    if c4 is not None:
        out["c4"] = c4
    if len(coder) > 3 and coder[3] > 0:
        # the one part of the path that IS a stream: the framing kernel reads the text once and writes the line index
        fr_bytes = nbytes + 8 * 4 * int(res.n_records)
        out["roofline_framing"] = {"bound": "hbm", "kernel_name": "k_frame", "achieved": round(fr_bytes / (float(coder[3]) * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(fr_bytes / (float(coder[3]) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "alg_bytes_per_launch": int(fr_bytes), "avg_ms": round(float(coder[3]), 3)}
    points = None
    if not multi and not args.no_size_sweep and args.kind != 1 and args.tables:
        points = sweep_points(d_in, nbytes, args)
        out["size_sweep"] = size_sweep(ctx, d_in, nbytes, d_out, cap, args, prior_step, models, points)
    if not multi and args.workload == "full" and not args.models and not args.no_decode:
        # the way back ("decode MB/s secondarily"): the same blocks decoded in HBM and compared with the input;
        # outside the timed region, never part of `value`
        blocks = ctx.index(res.n_blocks)
        first = ctx.first_headers(res.first_hdr_bytes)
        prior, chains, rec_prior = ctx.prior(), ctx.chains(), ctx.rec_prior()
        packed = d_out[:res.total_bytes].clone()
        soff = list(res.stream_offset)
        d_back = torch.empty(nbytes + 4096, dtype=torch.uint8, device="cuda")
        times = []
        for _ in range(5):                                # (the first call sizes the context's decode buffers; the fastest of the rest is reported)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            got, _r = ctx.decode_device(blocks, first, packed.data_ptr(), soff, d_back.data_ptr(), d_back.numel(), prior=prior, level=args.level,
                                        chains=chains, rec_prior=rec_prior, lds_rows=args.dec_lds_rows)
            torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
        same = bool(got == nbytes and torch.equal(d_back[:nbytes], d_in))
        out["decode"] = {"value": round(nbytes / min(times) / 1e6, 2), "unit": "MB/s FASTQ restored", "ms": round(min(times) * 1e3, 3),
                         "round_trip_identical": same, "phase_ms": {"qlt": round(_r.kernel_ms[capi.T_QLT], 3), "gen": round(_r.kernel_ms[capi.T_GEN], 3),
                                                                    "rec": round(_r.kernel_ms[capi.T_REC], 3), "assemble": round(_r.kernel_ms[capi.T_PACK], 3),
                                                                    "device_total": round(_r.kernel_ms[capi.T_TOTAL], 3)}}
        del d_back, packed
    if not multi and args.tables and args.workload == "full" and not args.models and not args.no_adaptive_leg:
        # secondary: the same call with ADAPTIVE tables (every block runs the reference's per-symbol row updates, a wavefront per
        # block) -- round 1's headline mode, kept beside the default so the two can be compared; never part of `value`
        step(0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            ra = step(0)
        torch.cuda.synchronize(); ta = (time.perf_counter() - t0) / 3
        out["adaptive_tables"] = {"value": round(nbytes / ta / 1e6, 2), "unit": "MB/s", "ms_per_step": round(ta * 1e3, 3),
                                  "ratio": round(nbytes / ra.total_bytes, 4)}
    if not multi and args.workload == "full" and not args.models and not args.no_host_leg and fq_host is not None:
        # PCIe-inclusive (SURVEY 8d "wall clock including H2D/D2H"; never `value`): sfq_encode_blocks_host over page-locked buffers --
        # the text crosses the link, the streams come back
        L = capi.lib()
        import ctypes as C
        hin = L.sfq_host_alloc(ctx.handle, nbytes); hout = L.sfq_host_alloc(ctx.handle, cap)
        if hin and hout:
            C.memmove(hin, fq_host, nbytes)
            pp = capi.Params(args.level, args.block_reads, 0, models, args.kernel, 0, prior_step, args.tables, args.chain_reads, args.lds_rows)
            rr = capi.Result(); th = []
            for _ in range(3):
                t0 = time.perf_counter()
                rc_ = L.sfq_encode_blocks_host(ctx.handle, C.c_void_p(hin), nbytes, C.byref(pp), C.c_void_p(hout), cap, C.byref(rr))
                th.append(time.perf_counter() - t0)
            if rc_ == 0:
                out["host_to_host"] = {"value": round(nbytes / min(th[1:]) / 1e6, 2), "unit": "MB/s", "ms": round(min(th[1:]) * 1e3, 3),
                                       "what": "sfq_encode_blocks_host: page-locked host text in, host streams out (H2D + the call + D2H)"}
        if hin: L.sfq_host_free(ctx.handle, hin)
        if hout: L.sfq_host_free(ctx.handle, hout)
    if not multi and not args.no_cpu_baseline:
        cb, sample, ref_payload = cpu_baseline(args, seed)
        out["cpu_baseline"] = cb
        refs = None
        if points and fq_host is not None and args.workload == "full" and not args.models:
            # the reference over every other point of the sweep, a process each, while the legs below run
            try:
                cores = sorted(os.sched_getaffinity(0))
                if len(cores) >= 4:
                    os.sched_setaffinity(0, set(cores[:len(cores) // 2]))
            except (AttributeError, OSError):
                pass
            refs = ReferenceRuns(fq_host, [cut for n, cut in points if n != args.cpu_sample_reads], args.level)
        # ratio vs the reference on the same sample: ours in blocks vs the reference's single adaptive stream
        enc = ctx.encode_host(sample, level=args.level, block_reads=args.block_reads, models=models, kernel=args.kernel, prior_step=prior_step,
                              tables=args.tables, chain_reads=args.chain_reads)
        ours = enc.archive_bytes                                   # streams + first headers + prior + block index
        if args.workload == "full":
            out["ratio_vs_reference"] = {"sample_raw": len(sample), "reference_stream_bytes": int(ref_payload),
                                         "ours_stream_bytes": int(ours), "ours_over_reference": round(ours / ref_payload, 4)}
    if not multi and args.workload == "full" and not args.models and not args.no_format6_leg and args.kind != 1:
        # format 6 (block_reads = 0): ONE adaptive chain per stream, byte-identical to the reference's -- the drop-in mode.
        # A serial chain does not parallelise: this is MB/s next to the reference's own MB/s on one core, not GB/s
        f6 = capi.synth_fastq(args.format6_reads, args.read_len, seed=seed, kind=args.kind)
        t6 = torch.from_numpy(np.frombuffer(f6, np.uint8).copy()).cuda()
        o6 = torch.empty(capi.lib().sfq_encode_bound(len(f6)), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r6 = ctx.encode_device(t6.data_ptr(), len(f6), o6.data_ptr(), o6.numel(), level=args.level, block_reads=0)
        torch.cuda.synchronize(); te = time.perf_counter() - t0
        b6 = ctx.index(1); h6 = ctx.first_headers(r6.first_hdr_bytes)
        back6 = torch.empty(len(f6) + 4096, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        got6, _r = ctx.decode_device(b6, h6, o6.data_ptr(), list(r6.stream_offset), back6.data_ptr(), back6.numel(), level=args.level)
        torch.cuda.synchronize(); td = time.perf_counter() - t0
        out["format6"] = {"sample": "%d x %d bp reads (%.1f MB), one block = the reference's own streams" % (args.format6_reads, args.read_len, len(f6) / 1e6),
                          "encode_MBps": round(len(f6) / 1e6 / te, 2), "decode_MBps": round(len(f6) / 1e6 / td, 2),
                          "round_trip_identical": bool(got6 == len(f6) and torch.equal(back6[:len(f6)], t6)),
                          "reference_encode_MBps": out.get("cpu_baseline", {}).get("value"), "reference_decode_MBps": out.get("cpu_baseline", {}).get("decode_value")}
        del t6, o6, back6
    if not multi and args.workload == "full" and not args.models and not args.no_genome_leg and args.kind == 0 and args.tables:
        # the genome-sampled workload ("the honest stress of the base model"; iid bases switch the base tables off):
        # the same call over reads sampled from a 10 Mbp genome -- encode, decode, and the ratio against the reference itself on
        # the first --genome-ratio-reads records (30x coverage at 2 M reads of 150 bp)
        del d_in
        torch.cuda.empty_cache()
        gq = capi.synth_fastq(args.genome_reads, args.read_len, seed=seed, kind=3)
        gn = len(gq)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            g_in = torch.from_numpy(np.frombuffer(gq, np.uint8)).cuda()
        gcap = capi.lib().sfq_encode_bound(gn)
        g_out = d_out if d_out.numel() >= gcap else torch.empty(gcap, dtype=torch.uint8, device="cuda")
        kw = dict(level=args.level, block_reads=args.block_reads, prior_step=prior_step, tables=args.tables, chain_reads=args.chain_reads)
        ctx.encode_device(g_in.data_ptr(), gn, g_out.data_ptr(), gcap, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            rg = ctx.encode_device(g_in.data_ptr(), gn, g_out.data_ptr(), gcap, **kw)
        torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 3
        leg = {"workload": "synthetic %d x %d bp reads sampled from a 10 Mbp genome, -l %d" % (args.genome_reads, args.read_len, args.level),
               "value": round(gn / tg / 1e6, 2), "unit": "MB/s", "ms_per_step": round(tg * 1e3, 3), "ratio": round(gn / rg.total_bytes, 4),
               "base_tables_on": bool(_varints(ctx.chains(), 2)[1] & 1), "match_model": bool(_varints(ctx.chains(), 2)[1] & 32)}
        if not args.no_decode:
            gb, gh = ctx.index(rg.n_blocks), ctx.first_headers(rg.first_hdr_bytes)
            gp, gc, grp = ctx.prior(), ctx.chains(), ctx.rec_prior()
            g_back = torch.empty(gn + 4096, dtype=torch.uint8, device="cuda")
            times = []
            for _ in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                got, _r = ctx.decode_device(gb, gh, g_out.data_ptr(), list(rg.stream_offset), g_back.data_ptr(), g_back.numel(), prior=gp, level=args.level, chains=gc, rec_prior=grp)
                torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
            leg["decode"] = {"value": round(gn / min(times) / 1e6, 2), "unit": "MB/s FASTQ restored", "ms": round(min(times) * 1e3, 3),
                             "round_trip_identical": bool(got == gn and torch.equal(g_back[:gn], g_in))}
            del g_back
        if not args.no_cpu_baseline and args.genome_ratio_reads:
            from oracle import oracle as O
            lines = 4 * min(args.genome_ratio_reads, args.genome_reads)
            cut = 0
            for _ in range(lines):
                cut = gq.index(b"\n", cut) + 1
            sample = gq[:cut]
            t0 = time.perf_counter()
            ra = O.compress(sample, args.level)
            t_ref = time.perf_counter() - t0
            ref_payload = ra.payload_bytes() - len(ra.streams["<info>"])
            enc = ctx.encode_host(sample, **kw)
            leg["ratio_vs_reference"] = {"sample_reads": lines // 4, "sample_raw": len(sample), "reference_stream_bytes": int(ref_payload), "ours_stream_bytes": int(enc.archive_bytes),
                                         "ours_over_reference": round(enc.archive_bytes / ref_payload, 4), "cpu_port_MBps": round(len(sample) / 1e6 / t_ref, 2)}
        out["genome_sampled"] = leg
    if not multi and not args.no_cpu_baseline and points and "size_sweep" in out and args.workload == "full" and not args.models:
        got = refs.collect() if refs is not None else {}
        if "ratio_vs_reference" in out:
            got[out["ratio_vs_reference"]["sample_raw"]] = (out["ratio_vs_reference"]["reference_stream_bytes"], None)
        for row in out["size_sweep"]:
            r = got.get(row["raw_bytes"])
            if r and "archive_bytes" in row:
                row["reference_stream_bytes"] = int(r[0]); row["ours_over_reference"] = round(row["archive_bytes"] / r[0], 4)
                if r[1]:
                    row["reference_MBps"] = round(row["raw_bytes"] / 1e6 / r[1], 2)       # (beside the other prefixes' processes: not the 1-core baseline)
        full = [row for row in out["size_sweep"] if row["reads"] == args.reads and "ours_over_reference" in row]
        if full:
            out["ratio_vs_reference_full"] = {"sample_raw": full[0]["raw_bytes"], "reference_stream_bytes": full[0]["reference_stream_bytes"],
                                              "ours_stream_bytes": full[0]["archive_bytes"], "ours_over_reference": full[0]["ours_over_reference"]}
    print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
