"""frozen-table mode: device-resident encode -> decode round trip with timings (python scratch/frozen_rt.py <reads> [kind] [block_reads] [chain_reads])"""
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slimfastq_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
br = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
cr = int(sys.argv[4]) if len(sys.argv) > 4 else 0
models = int(sys.argv[5]) if len(sys.argv) > 5 else 0
lds = int(sys.argv[7]) if len(sys.argv) > 7 else 0
fq = capi.synth_fastq(n, 150, seed=1, kind=kind)
nbytes = len(fq)
d_in = torch.from_numpy(np.frombuffer(fq, np.uint8).copy()).cuda()
ctx = capi.Context(0)
cap = capi.lib().sfq_encode_bound(nbytes)
d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
for tables in ((1, 0) if len(sys.argv) <= 6 else (1,)):
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = ctx.encode_device(d_in.data_ptr(), nbytes, d_out.data_ptr(), cap, level=3, block_reads=br, prior_step=capi.PRIOR_AUTO,
                                tables=tables, chain_reads=cr, models=models, lds_rows=lds)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ms = list(res.kernel_ms)
    print("tables=%d encode %.1f ms (%.1f GB/s) ratio %.4f chains %d  phases frame %.1f qlt %.1f gen %.1f rec %.1f usr %.1f pack %.1f total %.1f" % (
        tables, dt * 1e3, nbytes / dt / 1e9, nbytes / res.total_bytes, res.n_chains, ms[0], ms[1], ms[2], ms[3], ms[4], ms[5], ms[6]), flush=True)
    print("   stream bytes", dict(zip(capi.STREAM_NAMES, list(res.stream_bytes))), "prior", len(ctx.prior()), "chains idx", len(ctx.chains()), "rec prior", len(ctx.rec_prior()), "first hdrs", res.first_hdr_bytes,
          "archive ~", res.total_bytes + len(ctx.prior()) + len(ctx.chains()) + len(ctx.rec_prior()) + res.first_hdr_bytes + 14 * res.n_blocks,
          "ratio(all) %.4f" % (nbytes / (res.total_bytes + len(ctx.prior()) + len(ctx.chains()) + len(ctx.rec_prior()) + res.first_hdr_bytes + 14 * res.n_blocks)))
    if models:
        continue
    blocks = ctx.index(res.n_blocks); first = ctx.first_headers(res.first_hdr_bytes); prior = ctx.prior(); chains = ctx.chains(); rpri = ctx.rec_prior()
    packed = d_out[:res.total_bytes].clone()
    d_back = torch.empty(nbytes + 4096, dtype=torch.uint8, device="cuda")
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        got, r2 = ctx.decode_device(blocks, first, packed.data_ptr(), list(res.stream_offset), d_back.data_ptr(), d_back.numel(), prior=prior, level=3, chains=chains, rec_prior=rpri)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    same = bool(got == nbytes and torch.equal(d_back[:nbytes], d_in))
    ms = list(r2.kernel_ms)
    print("tables=%d decode %.1f ms (%.1f GB/s) identical %s  phases usr %.1f qlt %.1f gen %.1f rec %.1f pack %.1f" % (tables, dt * 1e3, nbytes / dt / 1e9, same, ms[4], ms[1], ms[2], ms[3], ms[5]), flush=True)
    if not same:
        a = d_back[:nbytes].cpu().numpy(); b = np.frombuffer(fq, np.uint8)
        bad = np.nonzero(a[:min(len(a), len(b))] != b[:min(len(a), len(b))])[0]
        print("   got", got, "want", nbytes, "first mismatch at", bad[:5], fq[max(0, int(bad[0]) - 60):int(bad[0]) + 20] if len(bad) else None, bytes(a[max(0, int(bad[0]) - 60):int(bad[0]) + 20]) if len(bad) else None)
