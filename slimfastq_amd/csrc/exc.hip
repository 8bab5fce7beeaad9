// exc.hip -- the base exceptions of the frozen-table mode as adaptive Rice codes (round 4).
//
// What is coded is the reference's: per block three lists of gaps -- "gen.Ns" (N-like bases whose quality is not '!'), "gen.Nn"
// (real bases under quality '!'), "gen.lc" (lowercase bases; the block format's own, dev_common.h) -- with the positions and gaps
// of GenSave::bad_q_or_bad_n (gens.cpp:91-114: a base's position counted from 1 over the block's base lines, each gap against the
// list's previous entry).  HOW it is coded is not: the reference sends every gap through adaptive PowerRanger rows
// (XFileSave::put -> PowerRangerU::put_u, xfile.cpp:66-69, power_ranger.hpp:138-163), a 256-slot row search and update per byte.
// models_w.hip k_gen_exc_w does exactly that with a wavefront per block -- 900 wave instructions per gap, 1.36e9 per default call,
// 18 % of everything the call issues, for the N's of a seventh of its records.  A gap list needs no model: gaps between rare
// events are geometric, and for those a Rice code whose parameter follows the running mean is within a few per cent of the
// entropy (measured: 10 % SMALLER than the XFile streams on the default workload's blocks).  It needs no tables either, so a
// decoder's block is a LANE (k_gen_exc_decode_r below); the encoder keeps a wavefront per block for the SCAN of the marked records
// (models_w.hip k_gen_exc_w<true>: a lane per block walking them alone took 7-10 ms per call, one memory round trip after the
// other) and appends a gap's bits with a few scalar instructions.
//   k = the smallest k <= 24 with (N << k) >= A;  q = v >> k;
//   q < 32: q one bits, a zero bit, the low k bits of v;   else: 32 one bits, then v in 40 bits
//   A += v, N += 1; when N reaches 32 both are halved.   Start: A = 256, N = 1.
// Bits fill bytes from the low end; a list ends with v = 0 (no gap is 0) and zero bits up to a byte; an empty list is an empty
// stream.  oracle/sfq_oracle.c sfqo_exc_rice_block restates it; "chn.idx" flag bit 4 says a call's lists are coded this way
// (archives without it -- rounds 2 and 3 -- are read through k_gen_exc_decode_w).
#include "kernels.h"
#include "dev_rice.h"

// A lane per block: "gen.Ns" -> the N byte, "gen.Nn" -> bit 7 ("keep this base": decode_l.hip merge_n), "gen.lc" -> bit 5, in
// the block's staged bases (gens.cpp:187-188)
__global__ __launch_bounds__(64) void k_gen_exc_decode_r(DecodeArgs a, u32 nblocks) {
    const u32 b = blockIdx.x * 64 + threadIdx.x;
    if (b >= nblocks) return;
    BlockDesc* d = &a.m.blocks[b];
    RiceR r_ns, r_nn, r_lc;
    r_ns.init(a.streams + a.blk_stream_off[(u64)b * SFQ_NSTREAMS + SFQ_S_GEN_NS], (u32)d->size[SFQ_S_GEN_NS]);
    r_nn.init(a.streams + a.blk_stream_off[(u64)b * SFQ_NSTREAMS + SFQ_S_GEN_NN], (u32)d->size[SFQ_S_GEN_NN]);
    r_lc.init(a.streams + a.blk_stream_off[(u64)b * SFQ_NSTREAMS + SFQ_S_GEN_LC], (u32)d->size[SFQ_S_GEN_LC]);
    const u32 n_byte = d->n_byte ? d->n_byte : 'N';                                         // gens.cpp:169
    u8* const g = a.seq_stage + a.soff[d->rec0];
    // (the match model's stage -- gm.hip -- keeps a '\n' behind every line, and a call without it starts its lines on 32-byte sectors
    //  (api.cpp, dev_chain.h LaneOut32): a list's positions count BASES, so where the lines are not back to back (a.boff: the bases before
    //  every record) the record a position lies in is looked up and the position placed in that record's line)
    const u64* const bo = a.boff ? a.boff + d->rec0 : nullptr;
    const u64* const so = a.soff + d->rec0;
    const u64 b0 = bo ? bo[0] : 0;
    const u64 nb = bo ? bo[d->nrec] - b0 : a.soff[d->rec0 + d->nrec] - a.soff[d->rec0];
    u32 bad = 0;
    // the record of base position `at` (1-based over the block's bases): guessed from the block's mean line length -- reads of one
    // length: exact --, then stepped to (a walk from the list's previous entry, record by record, was 3000 dependent loads a lane: 1.27 ms
    // per 10 M reads against 0.43 without the sentinels)
    const u32 nr = d->nrec;
    auto rec_of = [&](u64 at) -> u32 {
        const u64 P = b0 + at - 1;
        u32 k = nb ? (u32)(((at - 1) * nr) / nb) : 0u;
        if (k >= nr) k = nr - 1;
        while (k > 0 && bo[k] > P) k--;
        while (k + 1 < nr && bo[k + 1] <= P) k++;
        return k;
    };
    u64 at = 0;
    auto place = [&](u64 at) -> u8* {                    // where base `at` of the block lies in the stage
        if (!bo) return g + at - 1;
        const u32 k = rec_of(at);
        return a.seq_stage + so[k] + (b0 + at - 1 - bo[k]);
    };
    for (u64 gap = r_ns.get(); gap; gap = r_ns.get()) { at += gap; if (at > nb) { bad = 1; break; } *place(at) = (u8)n_byte; }
    at = 0;
    for (u64 gap = r_nn.get(); gap; gap = r_nn.get()) { at += gap; if (at > nb) { bad = 1; break; } *place(at) |= 0x80u; }
    at = 0;
    for (u64 gap = r_lc.get(); gap; gap = r_lc.get()) { at += gap; if (at > nb) { bad = 1; break; } *place(at) |= 0x20u; }
    if (bad | r_ns.err | r_nn.err | r_lc.err) atomicMax(&d->status, (u32)(-SFQ_E_CORRUPT));
}
void launch_gen_exc_decode_r(const DecodeArgs& a, u32 nblocks, hipStream_t st) {
    if (nblocks) hipLaunchKernelGGL(k_gen_exc_decode_r, dim3((nblocks + 63) / 64), dim3(64), 0, st, a, nblocks);
}
