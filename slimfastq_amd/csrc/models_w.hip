// models_w.hip -- wave-per-block kernels over ADAPTIVE tables (format 6, and format 7 with sfq_params.tables = 0), plus the
// two wave-per-block kernels of the frozen-table mode whose rows stay adaptive: the N / quality-0 exception pass and its
// decoder (k_gen_exc_w, k_gen_exc_decode_w).  The default coding kernels of block format 7 are in chains.hip.
//
// A 64-lane wavefront owns one record block (two, in the quality kernel): its adaptive tables, its range-coder state and
// its output cursor.  The work of a symbol is split in three stages so that only what is inherently serial runs serially:
//   (1) lane-parallel : 64 symbols are loaded coalesced, and their model contexts are computed across
//                       lanes (the quality model's running `delta` is a wave prefix sum)
//   (2) rows          : the adaptive row of each symbol is searched and updated:
//                       k_qlt_encode_k2  symbol-parallel -- rows of different contexts in different lanes, one round per
//                                        run of one symbol in one context; two blocks per wave
//                       k_rec_encode_w_* PowerRanger rows, four slots per lane (WavePw: find by ballot, cumulative
//                                        frequency by a DPP prefix sum)
//   (3) coder         : the range coder consumes the (cum, freq, tot) triples; its one divide per symbol
//                       (coder.hpp:68) is a multiply-high by a reciprocal computed for all 64 triples at once plus an
//                       exact fix-up.  Per lane group, two blocks (MultiCoder).
// The adaptive base kernel is models_k.hip.  The bytes produced are identical to models_l.hip (and so to the
// reference); tests compare all of them.
#include "kernels.h"
#include "dev_models.h"
#include "dev_wave.h"

#define LAST_QLT 63u


// =========================================================================================================
// header encode: RecSave::save (recs.cpp:277-372) for every record of the block
//
// The header model is control-heavy and light on symbols (~10 coded bytes per record): tokenise at
// non-alphanumerics, diff against the previous header, type each changed field.  All of that runs
// wave-uniformly (every lane computes the same scalars; its loads are single broadcast requests).  What
// the wave buys is the PowerRanger row: 256 slots = 4 per lane, searched with one ballot and summed
// with one DPP scan instead of a 256-step dependent walk through HBM.
// =========================================================================================================
#include "dev_rec.h"

#include "dev_wavepw.h"
#include "dev_rice.h"
#include <type_traits>

// =========================================================================================================
// N / quality-0 exceptions alone (frozen-table mode: the bases themselves are coded by chains.hip): a wave per block
// scans 64 bases at a time and codes gen.Ns / gen.Nn as models_k.hip gen_window does (bad_q_or_bad_n, gens.cpp:91-114),
// each gap through the wave-cooperative PowerRanger rows (a gap is 1-2 symbols of a 256-slot row: on one lane the row
// search is a walk of dependent HBM reads, ~100 us per gap)
// =========================================================================================================
// flags (from the quality and base chains, which have read every byte anyway): the records that may hold an N or a '!';
// the others only move the base offset on, 64 records a step.  Null: every record is looked at.
// RICE (round 4, the default with frozen tables): the gap lists as adaptive Rice codes instead (dev_rice.h, exc.hip) -- the same
// scan, the same gaps, a few scalar instructions per gap and no tables.
struct XfRiceW {               // the XfEncW interface over a RiceWU
    RiceWU w; Sink0 sink; struct { u32 err; } rc;
    __device__ __forceinline__ void init(u8* p, u32 cap, u32) { w.init(p, cap); sink.p = p; sink.pos = 0; sink.cap = cap; rc.err = 0; }
    __device__ __forceinline__ void put(WavePw&, u64 gap, u32 lane) { w.put(gap, lane); }
    __device__ __forceinline__ u32 finish(WavePw&, u32 lane) { const u32 n = w.finish(lane); sink.pos = w.ovf ? sink.cap + 1 : n; return n; }
};
template <bool RICE>
__global__ __launch_bounds__(64) void k_gen_exc_w(ModelArgs a, const u8* __restrict__ flags, u32* ticket) {
    __shared__ __attribute__((aligned(16))) u32 hot_slots[RICE ? 4 : 2 * PW_NSYM];
    __shared__ __attribute__((aligned(16))) RowHdr hot_hdr[2];
    const u32 lane = threadIdx.x, t = blockIdx.x;
    for (u32 b = next_block(ticket); b < a.nblocks; b = next_block(ticket)) {
        BlockDesc* d = &a.blocks[b];
        WavePw pw; pw.slots = a.p_slots + (size_t)t * PR_ROWS * PW_NSYM; pw.hdr = a.p_hdr + (size_t)t * PR_ROWS; pw.epoch = EPOCH_L(a.epoch_base + b + 1);
        pw.hslots = hot_slots; pw.hhdr = hot_hdr; pw.hrow0 = PR_XF_BASE + XF_GEN_NS * PR_XF_ROWS; pw.hn = 2;
        if (!RICE && lane < 2) pw.fresh_hot(lane);         // (no block's epoch: the rows start fresh, power_ranger.hpp:36-47)
        typename std::conditional<RICE, XfRiceW, XfEncW>::type x_ns, x_nn, x_lc;          // XfEncW: the whole wave codes a gap, lane = four slots of the PowerRanger row
        x_ns.init(a.arena + d->out_off[SFQ_S_GEN_NS], d->out_cap[SFQ_S_GEN_NS], XF_GEN_NS);
        x_nn.init(a.arena + d->out_off[SFQ_S_GEN_NN], d->out_cap[SFQ_S_GEN_NN], XF_GEN_NN);
        x_lc.init(a.arena + d->out_off[SFQ_S_GEN_LC], d->out_cap[SFQ_S_GEN_LC], XF_GEN_LC);
        const u32 solid = d->solid;
        const u64 rec0 = d->rec0; const u32 nrec = d->nrec;
        u64 genofs = 0, ns_index = 0, nn_index = 0, lc_index = 0;
        u32 n_byte = 0; int bad = 0;
        // 64 records a step: lane = record; the base offsets of the marked ones come from a wave scan over the line lengths.
        // A marked record is taken 256 bases at a time (four loads per lane in flight).
        for (u32 k0 = 0; k0 < nrec; k0 += 64) {
          const u32 kk = k0 + lane;
          const bool have = kk < nrec;
          const u64 rr = rec0 + (have ? kk : 0);
          // every lane its own record's line bounds (the marked ones are handed round with readlane: no round trip of their own)
          const u64 lg0 = a.line_off[4 * rr + 1] + solid, lg1 = a.line_off[4 * rr + 2] - 1;
          const u64 lq0 = a.line_off[4 * rr + 3] + solid, lq1 = a.line_off[4 * rr + 4] - 1;
          const u32 my_llen = have && lg1 > lg0 ? (u32)(lg1 - lg0) : 0u;
          const u32 my_qlen = have && lq1 > lq0 ? (u32)(lq1 - lq0) : 0u;
          const u32 incl = wave_incl_scan(my_llen);
          const u32 excl = incl - my_llen;
          u64 todo = __ballot(have && (flags ? flags[rr] != 0 : true));
          const u64 genofs0 = genofs;
          // the first 256 bases / qualities of a marked record are fetched while the one before it is coded
          auto first = [&](u32 pk, u32 (&gch)[4], u32 (&qch)[4]) {
              const u8* gp = a.fq + (((u64)rl((u32)(lg0 >> 32), pk) << 32) | rl((u32)lg0, pk));
              const u8* qp = a.fq + (((u64)rl((u32)(lq0 >> 32), pk) << 32) | rl((u32)lq0, pk));
              const u32 llen = rl(my_llen, pk), qlen = rl(my_qlen, pk);
#pragma unroll
              for (u32 u = 0; u < 4; u++) {
                  const u32 idx = 64 * u + lane;
                  gch[u] = idx < llen ? gp[idx] : 'A';
                  qch[u] = (idx < llen && idx < qlen) ? qp[idx] : 40u;                  // gens.cpp:153
              }
          };
          u32 ng[4], nq[4];
          if (todo) first((u32)__ffsll((long long)todo) - 1u, ng, nq);
          while (todo) {
            const u32 pick = (u32)__ffsll((long long)todo) - 1u;
            todo &= todo - 1;
            genofs = genofs0 + rl(excl, pick);
            const u8* gp = a.fq + (((u64)rl((u32)(lg0 >> 32), pick) << 32) | rl((u32)lg0, pick));
            const u8* qp = a.fq + (((u64)rl((u32)(lq0 >> 32), pick) << 32) | rl((u32)lq0, pick));
            const u32 llen = rl(my_llen, pick), qlen = rl(my_qlen, pick);
            u32 gch[4], qch[4];
#pragma unroll
            for (u32 u = 0; u < 4; u++) { gch[u] = ng[u]; qch[u] = nq[u]; }
            if (todo) first((u32)__ffsll((long long)todo) - 1u, ng, nq);
            for (u32 base0 = 0; base0 < llen; base0 += 256) {
                if (base0) {
#pragma unroll
                    for (u32 u = 0; u < 4; u++) {
                        const u32 idx = base0 + 64 * u + lane;
                        gch[u] = idx < llen ? gp[idx] : 'A';
                        qch[u] = (idx < llen && idx < qlen) ? qp[idx] : 40u;                  // gens.cpp:153
                    }
                }
#pragma unroll
                for (u32 u = 0; u < 4; u++) {
                    const u32 base = base0 + 64 * u;
                    if (base >= llen) break;
                    const u32 m = llen - base < 64 ? llen - base : 64;
                    const bool in = lane < m;
                    const u32 n = gencode_w(gch[u]);
                    if (__ballot(in && n > 4)) bad = SFQ_E_GENCHAR;
                    const u64 mN = __ballot(in && n == 4), mQ = __ballot(in && qch[u] == '!');
                    u64 mx = mN | mQ;
                    while (mx) {
                        const u32 bit = (u32)__ffsll((long long)mx) - 1u;
                        mx &= mx - 1;
                        const u64 pos = genofs + bit + 1;
                        const bool is_n = (mN >> bit) & 1, is_q = (mQ >> bit) & 1;
                        if (!is_n) {
                            x_nn.put(pw, pos - nn_index, lane);
                            nn_index = pos;
                        } else {
                            u32 ch = rl(gch[u], bit);
                            if (a.lossless && ch == 'n') ch = 'N';                    // its case travels in "gen.lc"
                            if (!n_byte) n_byte = ch;
                            if (ch != n_byte) bad = SFQ_E_GENCHAR;
                            if (!is_q) { x_ns.put(pw, pos - ns_index, lane); ns_index = pos; }
                        }
                    }
                    if (a.lossless) {                                                 // lowercase bases: "gen.lc" (dev_common.h)
                        u64 ml = __ballot(in && is_lower_base(gch[u]));
                        while (ml) {
                            const u32 bit = (u32)__ffsll((long long)ml) - 1u;
                            ml &= ml - 1;
                            const u64 pos = genofs + bit + 1;
                            x_lc.put(pw, pos - lc_index, lane);
                            lc_index = pos;
                        }
                    }
                    genofs += m;
                }
            }
          }
          genofs = genofs0 + rl(incl, 63);
        }
        const u32 sz_ns = x_ns.finish(pw, lane), sz_nn = x_nn.finish(pw, lane), sz_lc = x_lc.finish(pw, lane);
        if (lane == 0) {
            d->n_byte = n_byte;
            d->size[SFQ_S_GEN_NS] = sz_ns;
            d->size[SFQ_S_GEN_NN] = sz_nn;
            d->size[SFQ_S_GEN_LC] = sz_lc;
            if (x_ns.sink.pos > x_ns.sink.cap || x_nn.sink.pos > x_nn.sink.cap || x_lc.sink.pos > x_lc.sink.cap) atomicMax(&d->status, (u32)(-SFQ_E_OVERFLOW));
            if (x_ns.rc.err | x_nn.rc.err | x_lc.rc.err) atomicMax(&d->status, (u32)(-SFQ_E_CORRUPT));
            if (bad) atomicMax(&d->status, (u32)(-bad));
        }
    }
}
// ---------------------------------------------------------------------------------------------------------------------
// The same lists (RICE only), the SCAN a marked record per lane (round 5).  k_gen_exc_w hands every marked record to the whole wave -- 64 lanes
// looking at 150 bases, ballots, a loop per event: 400 wave instructions a record, 0.6 G a default call for the N's of a seventh of its records.
// Here the marked records of a block queue up in LDS and are taken 64 at a time, a lane each, sixteen bytes a step: what is no plain base
// (either case of ACGT -- one v_perm lookup --, or a colour-space digit), what sits under a '!', what is lower case is found four bytes at
// a time and leaves a bit; only those bytes are looked at singly and become EVENTS (position, kind, character) in the lane's list.  The lists
// are then emitted in the records' order by the wave -- the gaps and their Rice codes are sequential by nature -- with no text read again.
// A record of more than EXQ_MAXLEN bases, or with more events than its list holds, is walked by the whole wave as before.
// ---------------------------------------------------------------------------------------------------------------------
#define EXQ_QCAP 128u
#define EXQ_EVCAP 8u
#define EXQ_MAXLEN 1024u
struct ExqLds {
    u64 g[EXQ_QCAP], q[EXQ_QCAP], ofs[EXQ_QCAP];
    u32 llen[EXQ_QCAP], qlen[EXQ_QCAP];
    u32 ev[64][EXQ_EVCAP + 1];
};
typedef u32 __attribute__((aligned(1))) u32_anyw;
// four bytes of text at p of which `avail` exist (0xFF where none does): nothing is read behind them
__device__ __forceinline__ u32 exq_ld4(const u8* p, u32 avail) {
    if (avail >= 4u) return *reinterpret_cast<const u32_anyw*>(p);
    u32 v = ~0u;
    for (u32 j = 0; j < avail; j++) v = (v & ~(0xffu << (8u * j))) | ((u32)p[j] << (8u * j));
    return v;
}
__device__ __forceinline__ u32 exq_zero(u32 x) { return ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u; }   // bit 7 of every byte of x that is 0
__device__ __forceinline__ u32 exq_nib(u32 t) { return ((t >> 7) * 0x01020408u) >> 24; }                               // those four bits as a nibble
// one marked record by the whole wave (k_gen_exc_w's walk, 64 bases a step)
struct ExqState { u64 ns_index, nn_index, lc_index; u32 n_byte; int bad; };
__device__ __forceinline__ void exq_wave_record(const ModelArgs& a, XfRiceW& x_ns, XfRiceW& x_nn, XfRiceW& x_lc, WavePw& pw, ExqState& st,
                                                const u8* gp, const u8* qp, u32 llen, u32 qlen, u64 genofs, u32 lane) {
    for (u32 base = 0; base < llen; base += 64) {
        const u32 idx = base + lane;
        const u32 gch = idx < llen ? gp[idx] : 'A';
        const u32 qch = (idx < llen && idx < qlen) ? qp[idx] : 40u;                      // gens.cpp:153
        const bool in = idx < llen;
        const u32 n = gencode_w(gch);
        if (__ballot(in && n > 4)) st.bad = SFQ_E_GENCHAR;
        const u64 mN = __ballot(in && n == 4), mQ = __ballot(in && qch == '!');
        u64 mx = mN | mQ;
        while (mx) {
            const u32 bit = (u32)__ffsll((long long)mx) - 1u;
            mx &= mx - 1;
            const u64 pos = genofs + base + bit + 1;
            const bool is_n = (mN >> bit) & 1, is_q = (mQ >> bit) & 1;
            if (!is_n) { x_nn.put(pw, pos - st.nn_index, lane); st.nn_index = pos; }
            else {
                u32 ch = rl(gch, bit);
                if (a.lossless && ch == 'n') ch = 'N';                                    // its case travels in "gen.lc"
                if (!st.n_byte) st.n_byte = ch;
                if (ch != st.n_byte) st.bad = SFQ_E_GENCHAR;
                if (!is_q) { x_ns.put(pw, pos - st.ns_index, lane); st.ns_index = pos; }
            }
        }
        if (a.lossless) {
            u64 ml = __ballot(in && is_lower_base(gch));
            while (ml) {
                const u32 bit = (u32)__ffsll((long long)ml) - 1u;
                ml &= ml - 1;
                const u64 pos = genofs + base + bit + 1;
                x_lc.put(pw, pos - st.lc_index, lane); st.lc_index = pos;
            }
        }
    }
}
__global__ __launch_bounds__(64) void k_gen_exc_q(ModelArgs a, const u8* __restrict__ flags, u32* ticket) {
    __shared__ ExqLds L;
    const u32 lane = threadIdx.x;
    WavePw pw; pw.slots = nullptr; pw.hdr = nullptr; pw.epoch = 0; pw.hslots = nullptr; pw.hhdr = nullptr; pw.hrow0 = 0; pw.hn = 0;      // (XfRiceW's interface; the Rice lists have no rows)
    for (u32 b = next_block(ticket); b < a.nblocks; b = next_block(ticket)) {
        BlockDesc* d = &a.blocks[b];
        XfRiceW x_ns, x_nn, x_lc;
        x_ns.init(a.arena + d->out_off[SFQ_S_GEN_NS], d->out_cap[SFQ_S_GEN_NS], XF_GEN_NS);
        x_nn.init(a.arena + d->out_off[SFQ_S_GEN_NN], d->out_cap[SFQ_S_GEN_NN], XF_GEN_NN);
        x_lc.init(a.arena + d->out_off[SFQ_S_GEN_LC], d->out_cap[SFQ_S_GEN_LC], XF_GEN_LC);
        const u32 solid = d->solid;
        const u64 rec0 = d->rec0; const u32 nrec = d->nrec;
        ExqState st; st.ns_index = st.nn_index = st.lc_index = 0; st.n_byte = 0; st.bad = 0;
        u64 genofs = 0;
        u32 qn = 0;                                                            // records waiting in the queue
        // the first m records of the queue: a lane each finds its events, the wave emits them in order
        auto take = [&](u32 m) {
            const bool mine = lane < m;
            const u8* gp = a.fq + (mine ? L.g[lane] : 0);
            const u8* qp = a.fq + (mine ? L.q[lane] : 0);
            const u32 llen = mine ? L.llen[lane] : 0u, qlen = mine ? L.qlen[lane] : 0u;
            u32 cnt = 0;
            bool whole = mine && llen > EXQ_MAXLEN;                            // -> the whole wave's walk
            u32 illegal = 0;
            for (u32 off = 0; __any(!whole && off < llen); off += 16) {
                u32 fl = 0;
                if (!whole && off < llen) {
                    const u32 nb = llen - off < 16u ? llen - off : 16u;
                    const u32 nq = qlen > off ? (qlen - off < 16u ? qlen - off : 16u) : 0u;
                    u32 fb = 0, fq = 0;
#pragma unroll
                    for (u32 dw = 0; dw < 4; dw++) {
                        const u32 w = exq_ld4(gp + off + 4 * dw, nb > 4 * dw ? nb - 4 * dw : 0u);
                        const u32 qw = exq_ld4(qp + off + 4 * dw, nq > 4 * dw ? nq - 4 * dw : 0u);
                        const u32 want = __builtin_amdgcn_perm(0u, 0x47544341u, (w >> 1) & 0x03030303u);       // "ACTG"[(c >> 1) & 3]: the letter c should be, if it is one
                        const u32 letter = exq_zero(want ^ (w & 0xDFDFDFDFu));
                        const u32 digit = exq_zero((w & 0xFCFCFCFCu) ^ 0x30303030u);
                        u32 t = ~(letter | digit) & 0x80808080u;                                                 // no plain base
                        if (a.lossless) t |= letter & ((w << 2) & 0x80808080u);                                  // a lower-case one
                        fb |= exq_nib(t) << (4 * dw);
                        fq |= exq_nib(exq_zero(qw ^ 0x21212121u)) << (4 * dw);
                    }
                    fl = (fb & ((1u << nb) - 1u)) | (fq & ((1u << nq) - 1u) & ((1u << nb) - 1u));
                }
                while (__any(fl != 0)) {
                    if (fl) {
                        const u32 j = (u32)__ffs((int)fl) - 1u;
                        fl &= fl - 1;
                        const u32 idx = off + j;
                        const u32 c = gp[idx];
                        const u32 qc = idx < qlen ? qp[idx] : 40u;                      // gens.cpp:153
                        const u32 n = gencode_w(c);
                        if (n > 4) illegal = 1;
                        const u32 kind = (n == 4 ? 1u : 0u) | (qc == '!' ? 2u : 0u) | ((a.lossless && is_lower_base(c)) ? 4u : 0u);
                        if (kind) {
                            if (cnt < EXQ_EVCAP) L.ev[lane][cnt] = idx | (kind << 16) | (c << 24);
                            else whole = true;
                            cnt++;
                        }
                    }
                }
            }
            if (__any(illegal != 0)) st.bad = SFQ_E_GENCHAR;
            for (u32 i = 0; i < m; i++) {
                const u64 ofs = L.ofs[i];
                if (rl(whole ? 1u : 0u, i)) {
                    exq_wave_record(a, x_ns, x_nn, x_lc, pw, st, a.fq + L.g[i], a.fq + L.q[i], L.llen[i], L.qlen[i], ofs, lane);
                    continue;
                }
                const u32 ne = rl(cnt, i);
                for (u32 e = 0; e < ne; e++) {
                    const u32 ev = rl(L.ev[i][e], 0);
                    const u64 pos = ofs + (ev & 0xffffu) + 1;
                    const bool is_n = (ev >> 16) & 1, is_q = (ev >> 17) & 1;
                    if (is_n | is_q) {
                        if (!is_n) { x_nn.put(pw, pos - st.nn_index, lane); st.nn_index = pos; }
                        else {
                            u32 ch = ev >> 24;
                            if (a.lossless && ch == 'n') ch = 'N';                          // its case travels in "gen.lc"
                            if (!st.n_byte) st.n_byte = ch;
                            if (ch != st.n_byte) st.bad = SFQ_E_GENCHAR;
                            if (!is_q) { x_ns.put(pw, pos - st.ns_index, lane); st.ns_index = pos; }
                        }
                    }
                    if ((ev >> 18) & 1) { x_lc.put(pw, pos - st.lc_index, lane); st.lc_index = pos; }
                }
            }
        };
        for (u32 k0 = 0; k0 < nrec; k0 += 64) {
            const u32 kk = k0 + lane;
            const bool have = kk < nrec;
            const u64 rr = rec0 + (have ? kk : 0);
            const u64 lg0 = a.line_off[4 * rr + 1] + solid, lg1 = a.line_off[4 * rr + 2] - 1;
            const u64 lq0 = a.line_off[4 * rr + 3] + solid, lq1 = a.line_off[4 * rr + 4] - 1;
            const u32 my_llen = have && lg1 > lg0 ? (u32)(lg1 - lg0) : 0u;
            const u32 my_qlen = have && lq1 > lq0 ? (u32)(lq1 - lq0) : 0u;
            const u32 incl = wave_incl_scan(my_llen);
            const bool marked = have && (flags ? flags[rr] != 0 : true);
            const u64 todo = __ballot(marked);
            if (marked) {
                const u32 at = qn + (u32)__popcll(todo & ((1ull << lane) - 1ull));
                L.g[at] = lg0; L.q[at] = lq0; L.ofs[at] = genofs + (incl - my_llen); L.llen[at] = my_llen; L.qlen[at] = my_qlen;
            }
            qn += (u32)__popcll(todo);
            genofs += rl(incl, 63);
            if (qn >= 64) {
                take(64);
                qn -= 64;
                const u64 tg = L.g[64 + lane], tq = L.q[64 + lane], to = L.ofs[64 + lane]; const u32 tl = L.llen[64 + lane], tql = L.qlen[64 + lane];
                if (lane < qn) { L.g[lane] = tg; L.q[lane] = tq; L.ofs[lane] = to; L.llen[lane] = tl; L.qlen[lane] = tql; }
            }
        }
        if (qn) take(qn);
        const u32 sz_ns = x_ns.finish(pw, lane), sz_nn = x_nn.finish(pw, lane), sz_lc = x_lc.finish(pw, lane);
        if (lane == 0) {
            d->n_byte = st.n_byte;
            d->size[SFQ_S_GEN_NS] = sz_ns;
            d->size[SFQ_S_GEN_NN] = sz_nn;
            d->size[SFQ_S_GEN_LC] = sz_lc;
            if (x_ns.sink.pos > x_ns.sink.cap || x_nn.sink.pos > x_nn.sink.cap || x_lc.sink.pos > x_lc.sink.cap) atomicMax(&d->status, (u32)(-SFQ_E_OVERFLOW));
            if (st.bad) atomicMax(&d->status, (u32)(-st.bad));
        }
    }
}
// The way back (decode_l.hip k_gen_exc_decode_l on one lane: every gap a walk of dependent reads through its row): a wave
// per block, the row search across the lanes (WavePw::get).  gen.Ns lists the N positions whose quality is not '!' (-> the
// N byte), gen.Nn the real bases under quality '!' (-> bit 7, which k_assemble reads as "keep this base"); gens.cpp:187-188.
__global__ __launch_bounds__(64) void k_gen_exc_decode_w(DecodeArgs a) {
    const u32 lane = threadIdx.x, t = blockIdx.x;
    const u32 b = a.m.batch0 + t;
    BlockDesc* d = &a.m.blocks[b];
    WavePw pw; pw.slots = a.m.p_slots + (size_t)t * PR_ROWS * PW_NSYM; pw.hdr = a.m.p_hdr + (size_t)t * PR_ROWS; pw.epoch = EPOCH_L(a.m.epoch_base + b + 1);
    XfDecW x_ns, x_nn, x_lc;
    x_ns.init(a.streams + a.blk_stream_off[(u64)b * SFQ_NSTREAMS + SFQ_S_GEN_NS], d->size[SFQ_S_GEN_NS], XF_GEN_NS);
    x_nn.init(a.streams + a.blk_stream_off[(u64)b * SFQ_NSTREAMS + SFQ_S_GEN_NN], d->size[SFQ_S_GEN_NN], XF_GEN_NN);
    x_lc.init(a.streams + a.blk_stream_off[(u64)b * SFQ_NSTREAMS + SFQ_S_GEN_LC], d->size[SFQ_S_GEN_LC], XF_GEN_LC);
    const u32 n_byte = d->n_byte ? d->n_byte : 'N';                                         // gens.cpp:169
    u8* const g = a.seq_stage + a.soff[d->rec0];
    const u64 nb = a.soff[d->rec0 + d->nrec] - a.soff[d->rec0];
    u32 bad = 0;
    for (u64 at = x_ns.get(pw, lane); at; ) {                                               // gens.cpp:187
        if (at > nb) { bad = 1; break; }
        if (lane == 0) g[at - 1] = (u8)n_byte;
        const u64 gap = x_ns.get(pw, lane);
        if (!gap) break;
        at += gap;
    }
    for (u64 at = x_nn.get(pw, lane); at; ) {                                               // gens.cpp:188
        if (at > nb) { bad = 1; break; }
        if (lane == 0) g[at - 1] |= 0x80u;
        const u64 gap = x_nn.get(pw, lane);
        if (!gap) break;
        at += gap;
    }
    // "gen.lc": the lowercase bases (bit 5; k_assemble carries it over to an N that a quality '!' makes, decode_l.hip merge_n)
    for (u64 at = x_lc.get(pw, lane); at; ) {
        if (at > nb) { bad = 1; break; }
        if (lane == 0) g[at - 1] |= 0x20u;
        const u64 gap = x_lc.get(pw, lane);
        if (!gap) break;
        at += gap;
    }
    if (lane == 0 && (bad | x_ns.rc.err | x_nn.rc.err | x_lc.rc.err)) atomicMax(&d->status, (u32)(-SFQ_E_CORRUPT));
}
void launch_gen_exc_decode_w(const DecodeArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_gen_exc_decode_w, dim3(a.m.nbatch), dim3(64), 0, st, a);
}

// =========================================================================================================
// format 6: the reference's oversize records (usrs.cpp:269-301, 471-510; frame.hip picks them).  Three XFile streams, each
// its own coder and rows, a wave each: "usr.lrec" = for every such record the gap to the one before (in file numbers),
// the header line behind its '@' and the whole '+' line; "usr.lgen" / "usr.lqlt" = its base / quality line; every line
// with its newline, character by character through the stream's string row (XFileSave::put_chr, xfile.cpp:71-74) -- which
// lives in the wave's LDS: every character reads, updates and writes that one row.
// =========================================================================================================
struct OverArgs {
    const u8* fq; const u64* line_off;      // the ORIGINAL text and its line index
    const u32* over_list; u32 n_over;       // the oversize records (0-based file numbers), ascending
    u64 out_off[3]; u32 out_cap[3];         // regions in the arena: lrec, lgen, lqlt
};
__global__ __launch_bounds__(64) void k_over_encode_w(ModelArgs a, OverArgs o) {
    __shared__ __attribute__((aligned(16))) u32 hot_slots[PW_NSYM];
    __shared__ __attribute__((aligned(16))) RowHdr hot_hdr[1];
    const u32 lane = threadIdx.x, which = blockIdx.x;                    // 0 lrec, 1 lgen, 2 lqlt
    BlockDesc* d = &a.blocks[0];
    WavePw pw; pw.slots = a.p_slots; pw.hdr = a.p_hdr; pw.epoch = EPOCH_L(a.epoch_base + 1);      // block 0's rows (every XFile has rows of its own)
    XfEncW x; x.init(a.arena + o.out_off[which], o.out_cap[which], XF_USR_LREC + which);
    pw.hslots = hot_slots; pw.hhdr = hot_hdr; pw.hrow0 = x.row0 + 14; pw.hn = 1;
    if (lane == 0) pw.fresh_hot(0);
    u64 i_long = 0;                                                     // m_last.i_long usrs.cpp:271-272
    for (u32 i = 0; i < o.n_over; i++) {
        const u64 r = o.over_list[i];
        if (which == 0) { x.put(pw, r + 1 - i_long, lane); i_long = r + 1; }
        for (u32 part = 0; part < (which == 0 ? 2u : 1u); part++) {
            // lrec: the header line behind its '@', then the '+' line whole; lgen / lqlt: the line; each with its '\n'
            const u32 ln = which == 0 ? (part ? 2u : 0u) : which == 1 ? 1u : 3u;
            const u64 b0 = o.line_off[4 * r + ln] + ((which == 0 && part == 0) ? 1u : 0u), b1 = o.line_off[4 * r + ln + 1];
            for (u64 at = b0; at < b1; at += 64) {
                const u32 m = (u32)(b1 - at < 64 ? b1 - at : 64);
                const u32 ch = lane < m ? o.fq[at + lane] : 0u;
                for (u32 j = 0; j < m; j++) pw.put(x.row0 + 14, x.rc, x.sink, rl(ch, j), lane);
            }
            x.opened = 1;
        }
    }
    const u32 sz = x.finish(pw, lane);
    if (lane == 0) {
        const int sid = SFQ_S_USR_LREC + (int)which;
        d->size[sid] = sz; d->out_off[sid] = o.out_off[which]; d->out_cap[sid] = o.out_cap[which];
        if (x.sink.pos > x.sink.cap) atomicMax(&d->status, (u32)(-SFQ_E_OVERFLOW));
        if (x.rc.err) atomicMax(&d->status, (u32)(-SFQ_E_CORRUPT));
    }
}
void launch_over_encode_w(const ModelArgs& a, const u8* fq, const u64* line_off, const u32* over_list, u32 n_over, const u64 out_off[3], const u32 out_cap[3], hipStream_t st) {
    OverArgs o; o.fq = fq; o.line_off = line_off; o.over_list = over_list; o.n_over = n_over;
    for (int k = 0; k < 3; k++) { o.out_off[k] = out_off[k]; o.out_cap[k] = out_cap[k]; }
    hipLaunchKernelGGL(k_over_encode_w, dim3(3), dim3(64), 0, st, a, o);
}
// The way back (UsrLoad::update, usrs.cpp:473-485).  which = 0 ("usr.lrec"): with store == 0 only counts -- cnt[0] = records,
// cnt[1] = bytes of their header and '+' lines -- so that the host can size the buffers; with store != 0 fills no[i] (1-based file
// number) and piece[i][0..1] = {offset, length} of the header line (behind '@') and the '+' line in txt.  which = 1 / 2
// ("usr.lgen" / "usr.lqlt"): n_over lines into txt, piece[i][2] / piece[i][3].  Lengths include the newline.
struct OverDecArgs {
    const u8* stream; u32 size; u32 which, store, n_over;
    u64* cnt; u64* no; u64* piece;           // piece: [n_over][4][2]
    u8* txt; u64 cap;
};
__global__ __launch_bounds__(64) void k_over_decode_w(ModelArgs a, OverDecArgs o) {
    __shared__ __attribute__((aligned(16))) u32 hot_slots[PW_NSYM];
    __shared__ __attribute__((aligned(16))) RowHdr hot_hdr[1];
    const u32 lane = threadIdx.x, which = o.which;
    BlockDesc* d = &a.blocks[0];
    // (rows of the call's first table slot under an epoch of their own per pass: the counting pass and the storing pass both start fresh)
    WavePw pw; pw.slots = a.p_slots; pw.hdr = a.p_hdr; pw.epoch = EPOCH_L(a.epoch_base + 1 + o.store);
    XfDecW x; x.init(o.stream, o.size, XF_USR_LREC + which);
    pw.hslots = hot_slots; pw.hhdr = hot_hdr; pw.hrow0 = x.row0 + 14; pw.hn = 1;
    if (lane == 0) pw.fresh_hot(0);
    u64 pos = 0, nrecs = 0, number = 0; u32 bad = 0;
    // one line: characters up to and including the newline (a line that does not end within the buffer is a damaged stream)
    auto line = [&](u32 slot, u64 i) {
        const u64 p0 = pos;
        for (;;) {
            const u32 c = x.valid ? pw.get(x.row0 + 14, x.rc, x.src, lane) : (u32)'\n';
            if (o.store && pos < o.cap && lane == 0) o.txt[pos] = (u8)c;
            pos++;
            if (c == '\n') break;
            if (pos >= o.cap || x.rc.err) { bad = 1; break; }              // (cap: the caller's room for the text, in the counting pass too)
        }
        if (o.store && lane == 0) { o.piece[(i * 4 + slot) * 2] = p0; o.piece[(i * 4 + slot) * 2 + 1] = pos - p0; }
    };
    if (which == 0) {
        for (u64 gap = x.get(pw, lane); gap && !bad; gap = x.get(pw, lane)) {           // usrs.cpp:449, 482
            number += gap;
            if (o.store) { if (nrecs >= o.n_over) { bad = 1; break; } if (lane == 0) o.no[nrecs] = number; }
            line(0, nrecs); if (bad) break;
            line(1, nrecs);
            nrecs++;
            if (nrecs > (1ull << 32)) bad = 1;
        }
        if (lane == 0 && o.cnt) { o.cnt[0] = nrecs; o.cnt[1] = pos; }
    } else {
        for (u64 i = 0; i < o.n_over && !bad; i++) line(which + 1, i);
    }
    if (lane == 0 && (bad | x.rc.err)) atomicMax(&d->status, (u32)(-SFQ_E_CORRUPT));
}
void launch_over_decode_w(const ModelArgs& a, const u8* stream, u32 size, u32 which, u32 store, u32 n_over, u64* cnt, u64* no, u64* piece, u8* txt, u64 cap, hipStream_t st) {
    OverDecArgs o; o.stream = stream; o.size = size; o.which = which; o.store = store; o.n_over = n_over; o.cnt = cnt; o.no = no; o.piece = piece; o.txt = txt; o.cap = cap;
    hipLaunchKernelGGL(k_over_decode_w, dim3(1), dim3(64), 0, st, a, o);
}
// a wave per oversize record: '@', the header line, the base line, the '+' line, the quality line (usrs.cpp:474-481) to its place
__global__ __launch_bounds__(64) void k_over_place(const u64* __restrict__ no, const u64* __restrict__ piece, const u8* lrec_txt, const u8* lgen_txt, const u8* lqlt_txt,
                                                   const u64* __restrict__ roff_all, u8* __restrict__ out) {
    const u32 i = blockIdx.x, lane = threadIdx.x;
    u8* dst = out + roff_all[no[i] - 1];
    if (lane == 0) dst[0] = '@';
    dst++;
    const u8* src[4] = { lrec_txt, lgen_txt, lrec_txt, lqlt_txt };
    const u32 slot[4] = { 0, 2, 1, 3 };
    for (int k = 0; k < 4; k++) {
        const u64 p0 = piece[((u64)i * 4 + slot[k]) * 2], n = piece[((u64)i * 4 + slot[k]) * 2 + 1];
        for (u64 j = lane; j < n; j += 64) dst[j] = src[k][p0 + j];
        dst += n;
    }
}
void launch_over_place(u32 n_over, const u64* no, const u64* piece, const u8* lrec_txt, const u8* lgen_txt, const u8* lqlt_txt, const u64* roff_all, u8* out, hipStream_t st) {
    if (n_over) hipLaunchKernelGGL(k_over_place, dim3(n_over), dim3(64), 0, st, no, piece, lrec_txt, lgen_txt, lqlt_txt, roff_all, out);
}

void launch_gen_exc_w(const ModelArgs& a, const u8* flags, u32* ticket, hipStream_t st) {
    // (persistent waves that take blocks off the ticket: not more of them than three quarters of the chip's wave slots --
    //  the packing kernels run beside this pass and would otherwise wait for a slot until it is through)
    const u32 grid = a.nbatch < 6144u ? a.nbatch : 6144u;
    hipLaunchKernelGGL(k_gen_exc_w<false>, dim3(grid), dim3(64), 0, st, a, flags, ticket);
}
void launch_gen_exc_r(const ModelArgs& a, const u8* flags, u32* ticket, hipStream_t st) {
    const u32 grid = a.nbatch < 6144u ? a.nbatch : 6144u;
    hipLaunchKernelGGL(k_gen_exc_q, dim3(grid), dim3(64), 0, st, a, flags, ticket);
}


// general path: any header length (tokenising and field state in per-lane scratch)
__device__ __forceinline__ void rec_encode_block_slow(const ModelArgs& a, const u32 t, const u32 b, BlockDesc* d, const u32 lane) {
    WavePw pw; pw.slots = a.p_slots + (size_t)t * PR_ROWS * PW_NSYM; pw.hdr = a.p_hdr + (size_t)t * PR_ROWS; pw.epoch = EPOCH_L(a.epoch_base + b + 1);
    Sink0 snk = { a.arena + d->out_off[SFQ_S_REC], 0, d->out_cap[SFQ_S_REC] };
    RcEncU rc; rc.init();
    XfEncW x_rec; x_rec.init(a.arena + d->out_off[SFQ_S_REC_X], d->out_cap[SFQ_S_REC_X], XF_REC_X);
    SpaceMap sm[2];
    u8  fkind[2][66];
    u64 fvalue[2][66];
    u32 cur = 0; int bad = 0;
    u64 last_index = 0;
    u32 hdr_bytes = 0;
    const u8* prev = nullptr;
    const u64 rec0 = d->rec0; const u32 nrec = d->nrec;
    for (u32 k = 0; k < nrec; k++) {
        const u64 r = rec0 + k;
        const u64 record_count = rec_count_of(a, r, rec0);
        const u64 h0 = a.line_off[4 * r] + 1, h1 = a.line_off[4 * r + 1] - 1;
        const u8* buf = a.fq + h0;
        const u32 n = h1 > h0 ? (u32)(h1 - h0) : 0;
        hdr_bytes += n;
        if (k == 0) {                                                         // recs.cpp:279-287
            cur = 0;
            if (!map_space(buf, n, sm[0])) bad = SFQ_E_FORMAT;
            for (int i = 0; i < 66; i++) { fkind[0][i] = 0; fkind[1][i] = 0; }
            prev = buf;
            continue;
        }
        const u32 prv = cur;
        cur ^= 1;
        if (!map_space(buf, n, sm[cur])) { bad = SFQ_E_FORMAT; break; }
        SpaceMap& mi = sm[cur]; SpaceMap& mp = sm[prv];
        bool shape = mi.len != mp.len;
        if (!shape) for (u32 i = 0; i < mi.len; i++) if (mi.str[i] != mp.str[i]) { shape = true; break; }
        if (a.lossless && mi.str[mi.len - 1] == 0) shape = true;              // a NUL inside (dev_common.h)
        if (shape) {                                                          // recs.cpp:292-305
            x_rec.put(pw, record_count - last_index, lane);
            last_index = record_count;
            x_rec.put_str(pw, buf, n, lane);
            for (int i = 0; i < 66; i++) fkind[cur][i] = 0;
            prev = buf;
            continue;
        }
        u64 map = 0;
        for (u32 i = 0; i < mi.len; i++)
            if (mi.wln[i] != mp.wln[i] || bytes_differ(buf + mi.off[i], prev + mp.off[i], mi.wln[i])) map |= 1ULL << i;
        pw.put_u(0 * 16 + 2, rc, snk, map, lane);                             // put_num(0, map) recs.cpp:313
        for (u32 i = 0; i < mi.len; i++) {
            if (map & (1ULL << i)) {
                const u8* bp = buf + mi.off[i];
                u64 fnum;
                u32 type = numberwang(bp, mi.wln[i], fnum, fkind[prv][i]);
                if (a.lossless && type != ST_STR && !rec_number_prints_back(type, mi.wln[i], bp[0])) type = ST_STR;
                const u32 rr = (i + 1) * 16;
                if (type == ST_STR) {                                         // recs.cpp:324-331
                    pw.put(rr + 0, rc, snk, type, lane);
                    pw.put_u(rr + 2, rc, snk, mi.wln[i], lane);
                    for (u32 j = 0; j < mi.wln[i]; j++) pw.put(rr + 1, rc, snk, bp[j], lane);
                    fkind[cur][i] = 0;
                    continue;
                }
                u64 was = fkind[prv][i] ? fvalue[prv][i] : 0;                // recs.cpp:333-348
                u64 gap;
                fkind[cur][i] = (type < ST_STR || type >= ST_DGT_Z) ? 1 : 2;
                fvalue[cur][i] = fnum;
                if (fnum < was) { gap = was - fnum; type++; }
                else gap = fnum - was;
                pw.put(rr + 0, rc, snk, type, lane);
                pw.put_u(rr + 2, rc, snk, gap, lane);
            } else {
                fkind[cur][i] = fkind[prv][i];
                fvalue[cur][i] = fvalue[prv][i];
            }
        }
        prev = buf;
    }
    rc.done(snk);
    const u32 xsz = x_rec.finish(pw, lane);
    if (lane == 0) {
        d->hdr_bytes = hdr_bytes;
        d->size[SFQ_S_REC] = snk.pos;
        d->size[SFQ_S_REC_X] = xsz;
        if (snk.pos > snk.cap || x_rec.sink.pos > x_rec.sink.cap) atomicMax(&d->status, (u32)(-SFQ_E_OVERFLOW));
        if (rc.err | x_rec.rc.err) atomicMax(&d->status, (u32)(-SFQ_E_CORRUPT));
        if (bad) atomicMax(&d->status, (u32)(-bad));
    }
}

// Header bytes held two per lane: register c0 = bytes 0..63, c1 = bytes 64..127 (lane j <-> byte 64*k + j).
struct HdrRegs {
    u32 c0, c1;
    __device__ __forceinline__ u32 at(u32 pos) const { return pos < 64 ? rl(c0, pos) : rl(c1, pos - 64); }   // pos uniform
};
// a field's type and value (dev_rec.h field_type) over a field [off, off+len) of such a header
__device__ __forceinline__ u32 nw_lanes(const HdrRegs& h, u32 off, int len, u64& num, u32 pctype) {
    return field_type([&](u32 j) -> u32 { return h.at(off + j); }, (u32)len, num, pctype);
}
// bits [a, b) of a 128-bit mask, as two 64-bit halves (per lane)
__device__ __forceinline__ void mask128(u32 a, u32 b, u64& m0, u64& m1) {
    const u64 lo_b = b >= 64 ? ~0ull : ((1ull << b) - 1), lo_a = a >= 64 ? ~0ull : ((1ull << a) - 1);
    m0 = lo_b & ~lo_a;
    const u32 a1 = a > 64 ? a - 64 : 0, b1 = b > 64 ? b - 64 : 0;
    const u64 hi_b = b1 >= 64 ? ~0ull : ((1ull << b1) - 1), hi_a = a1 >= 64 ? ~0ull : ((1ull << a1) - 1);
    m1 = hi_b & ~hi_a;
}

// fast path: every header of the block fits two bytes per lane (<= 126 bytes + terminator, <= 64 fields).
// Tokenising is two ballots, field tables are lanes (lane k = field k), the field diff against the previous
// header is a cross-lane gather; only the few coded symbols per record run serially.
#define REC_FAST_MAX 126u
__device__ __forceinline__ void rec_encode_block_fast(const ModelArgs& a, const u32 t, const u32 b, BlockDesc* d, const u32 lane) {
    WavePw pw; pw.slots = a.p_slots + (size_t)t * PR_ROWS * PW_NSYM; pw.hdr = a.p_hdr + (size_t)t * PR_ROWS; pw.epoch = EPOCH_L(a.epoch_base + b + 1);
    Sink0 snk = { a.arena + d->out_off[SFQ_S_REC], 0, d->out_cap[SFQ_S_REC] };
    RcEncU rc; rc.init();
    XfEncW x_rec; x_rec.init(a.arena + d->out_off[SFQ_S_REC_X], d->out_cap[SFQ_S_REC_X], XF_REC_X);
    const u64 rec0 = d->rec0; const u32 nrec = d->nrec;
    const u64 lt = (1ull << lane) - 1;                    // lanes below this one
    HdrRegs pb = {0, 0};                                  // previous header
    u32 psep = 0, pspos = 0, pnf = 0;                     // its k-th separator char / position (lane k), field count
    u32 ct = 0, cn_lo = 0, cn_hi = 0;                     // per field (lane k): fkind, fvalue (recs.hpp:75-76)
    u64 last_index = 0;
    u32 hdr_bytes = 0;
    int bad = 0;
    for (u32 k = 0; k < nrec; k++) {
        const u64 r = rec0 + k;
        const u64 record_count = rec_count_of(a, r, rec0);
        const u64 h0 = a.line_off[4 * r] + 1, h1 = a.line_off[4 * r + 1] - 1;
        const u32 n = h1 > h0 ? (u32)(h1 - h0) : 0;
        hdr_bytes += n;
        HdrRegs cb;
        cb.c0 = lane < n ? (u32)a.fq[h0 + lane] : 0u;
        cb.c1 = lane + 64 < n ? (u32)a.fq[h0 + 64 + lane] : 0u;
        if (lane == n) cb.c0 = '\n';                     // the terminator is a separator too (recs.cpp:148-150)
        if (lane + 64 == n) cb.c1 = '\n';
        u64 cm0 = __ballot(lane <= n && !isword(cb.c0));  // map_space: separators
        u64 cm1 = __ballot(lane + 64 <= n && !isword(cb.c1));
        // a NUL inside the text ends the reference's scan early (recs.cpp:148): cut there
        const u64 nul0 = __ballot(lane < n && cb.c0 == 0), nul1 = __ballot(lane + 64 < n && cb.c1 == 0);
        if (nul0) { const u32 z = (u32)__ffsll((long long)nul0) - 1u; cm0 &= z >= 63 ? ~0ull : ((2ull << z) - 1); cm1 = 0; }
        else if (nul1) { const u32 z = (u32)__ffsll((long long)nul1) - 1u; cm1 &= z >= 63 ? ~0ull : ((2ull << z) - 1); }
        const u32 nf0 = (u32)__popcll(cm0), nf = nf0 + (u32)__popcll(cm1);   // fields = separators (the last is the terminator)
        if (nf > 64) { bad = SFQ_E_FORMAT; break; }       // recs.cpp:153-154
        // field index of each byte, and the k-th separator's char / position gathered into lane k
        const bool sep0 = (cm0 >> lane) & 1, sep1 = (cm1 >> lane) & 1;
        const u32 rank0 = (u32)__popcll(cm0 & lt), rank1 = nf0 + (u32)__popcll(cm1 & lt);
        // (pushes from non-separator lanes go to lane 63 with value 0; a real 64th field only exists when
        //  every position is a separator, so nothing else lands there)
        u32 sepc = (u32)__builtin_amdgcn_ds_permute((int)((sep0 ? rank0 : 63u) * 4), (int)(sep0 ? cb.c0 : 0u));
        u32 spos = (u32)__builtin_amdgcn_ds_permute((int)((sep0 ? rank0 : 63u) * 4), (int)(sep0 ? lane : 0u));
        sepc |= (u32)__builtin_amdgcn_ds_permute((int)((sep1 ? (rank1 & 63) : 63u) * 4), (int)(sep1 ? cb.c1 : 0u));
        spos |= (u32)__builtin_amdgcn_ds_permute((int)((sep1 ? (rank1 & 63) : 63u) * 4), (int)(sep1 ? lane + 64 : 0u));
        if (nf == 64) {                                   // lane 63 is then a real field: its entry is the last separator
            const u32 tpos = cm1 ? 127u - (u32)__clzll((long long)cm1) : 63u - (u32)__clzll((long long)cm0);
            const u32 tch = cb.at(tpos);
            if (lane == 63) { sepc = tch; spos = tpos; }
        }
        if (k == 0) {                                     // first header -> "rec.first" (recs.cpp:279-287)
            ct = 0;
            pb = cb; psep = sepc; pspos = spos; pnf = nf;
            continue;
        }
        const bool shape = nf != pnf || __ballot(lane < nf && sepc != psep) != 0 || (a.lossless && (nul0 | nul1) != 0);
        if (shape) {                                      // recs.cpp:292-305
            x_rec.put(pw, record_count - last_index, lane);
            last_index = record_count;
            x_rec.put(pw, n, lane);                       // put_str: length, then the characters
            for (u32 j = 0; j < n; j++) pw.put(x_rec.row0 + 14, x_rec.rc, x_rec.sink, cb.at(j), lane);
            ct = 0;
            pb = cb; psep = sepc; pspos = spos; pnf = nf;
            continue;
        }
        // field tables: lane f = field f.  (Cross-lane ops stay out of lane-dependent conditionals:
        // `c ? dpp : x` would run the DPP under a partial EXEC.)
        const u32 offv = wave_shr1(spos, 0xFFFFFFFFu) + 1;          // lane 0: 0xFFFFFFFF + 1 = 0
        const u32 wlnv = spos - offv;
        const u32 poffv = wave_shr1(pspos, 0xFFFFFFFFu) + 1;
        const u32 pwlnv = pspos - poffv;
        // byte diff: byte at position p of field f against the previous header's byte at p + (poff_f - off_f)
        const u32 dv = poffv - offv;
        const u32 d0 = (u32)__builtin_amdgcn_ds_bpermute((int)((rank0 & 63) * 4), (int)dv);
        const u32 d1 = (u32)__builtin_amdgcn_ds_bpermute((int)((rank1 & 63) * 4), (int)dv);
        const u32 pp0 = lane + d0, pp1 = lane + 64 + d1;            // positions in the previous header
        const u32 g00 = (u32)__builtin_amdgcn_ds_bpermute((int)((pp0 & 63) * 4), (int)pb.c0), g01 = (u32)__builtin_amdgcn_ds_bpermute((int)((pp0 & 63) * 4), (int)pb.c1);
        const u32 g10 = (u32)__builtin_amdgcn_ds_bpermute((int)((pp1 & 63) * 4), (int)pb.c0), g11 = (u32)__builtin_amdgcn_ds_bpermute((int)((pp1 & 63) * 4), (int)pb.c1);
        const u32 pby0 = (pp0 & 64) ? g01 : g00, pby1 = (pp1 & 64) ? g11 : g10;
        const u64 dm0 = __ballot(lane < n && !sep0 && cb.c0 != pby0);
        const u64 dm1 = __ballot(lane + 64 < n && !sep1 && cb.c1 != pby1);
        u64 f0, f1;
        mask128(offv, offv + wlnv, f0, f1);
        const u64 map = __ballot(lane < nf && (wlnv != pwlnv || ((dm0 & f0) | (dm1 & f1)) != 0));
        pw.put_u(0 * 16 + 2, rc, snk, map, lane);         // put_num(0, map) recs.cpp:313
        u64 todo = map;
        while (todo) {
            const u32 i = (u32)__ffsll((long long)todo) - 1u;
            todo &= todo - 1;
            const u32 off = rl(offv, i), wln = rl(wlnv, i), pct = rl(ct, i);
            u64 fnum;
            u32 type = nw_lanes(cb, off, (int)wln, fnum, pct);
            if (a.lossless && type != ST_STR && !rec_number_prints_back(type, wln, cb.at(off))) type = ST_STR;
            const u32 rr = (i + 1) * 16;
            if (type == ST_STR) {                         // recs.cpp:324-331
                pw.put(rr + 0, rc, snk, type, lane);
                pw.put_u(rr + 2, rc, snk, wln, lane);
                for (u32 j = 0; j < wln; j++) pw.put(rr + 1, rc, snk, cb.at(off + j), lane);
                if (lane == i) ct = 0;
                continue;
            }
            const u64 was = pct ? ((u64)rl(cn_hi, i) << 32 | rl(cn_lo, i)) : 0;       // recs.cpp:333-348
            u64 gap;
            const u32 nct = (type < ST_STR || type >= ST_DGT_Z) ? 1u : 2u;
            if (lane == i) { ct = nct; cn_lo = (u32)fnum; cn_hi = (u32)(fnum >> 32); }
            if (fnum < was) { gap = was - fnum; type++; }
            else gap = fnum - was;
            pw.put(rr + 0, rc, snk, type, lane);
            pw.put_u(rr + 2, rc, snk, gap, lane);
        }
        pb = cb; psep = sepc; pspos = spos; pnf = nf;
    }
    rc.done(snk);
    const u32 xsz = x_rec.finish(pw, lane);
    if (lane == 0) {
        d->hdr_bytes = hdr_bytes;
        d->size[SFQ_S_REC] = snk.pos;
        d->size[SFQ_S_REC_X] = xsz;
        if (snk.pos > snk.cap || x_rec.sink.pos > x_rec.sink.cap) atomicMax(&d->status, (u32)(-SFQ_E_OVERFLOW));
        if (rc.err | x_rec.rc.err) atomicMax(&d->status, (u32)(-SFQ_E_CORRUPT));
        if (bad) atomicMax(&d->status, (u32)(-bad));
    }
}

// longest header of the block decides the path; two kernels (so each keeps its own register budget),
// each of which returns at once for the blocks that belong to the other
__device__ __forceinline__ bool rec_block_is_short(const ModelArgs& a, const BlockDesc* d, u32 lane) {
    u32 longest = 0;
    for (u32 k = lane; k < d->nrec; k += 64) {
        const u64 r = d->rec0 + k;
        const u32 n = (u32)(a.line_off[4 * r + 1] - a.line_off[4 * r] - 2);
        longest = n > longest ? n : longest;
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) { const u32 o = (u32)__shfl_xor((int)longest, s, 64); longest = o > longest ? o : longest; }
    return rl(longest, 0) <= REC_FAST_MAX;
}
__global__ __launch_bounds__(64, 4) void k_rec_encode_w_fast(ModelArgs a, u32* ticket) {
    const u32 lane = threadIdx.x, t = blockIdx.x;
    for (u32 b = next_block(ticket); b < a.nblocks; b = next_block(ticket)) {
        BlockDesc* d = &a.blocks[b];
        if (rec_block_is_short(a, d, lane)) rec_encode_block_fast(a, t, b, d, lane);
    }
}
__global__ __launch_bounds__(64) void k_rec_encode_w_slow(ModelArgs a, u32* ticket) {
    const u32 lane = threadIdx.x, t = blockIdx.x;
    for (u32 b = next_block(ticket); b < a.nblocks; b = next_block(ticket)) {
        BlockDesc* d = &a.blocks[b];
        if (!rec_block_is_short(a, d, lane)) rec_encode_block_slow(a, t, b, d, lane);
    }
}
void launch_rec_encode_w(const ModelArgs& a, u32* ticket_fast, u32* ticket_slow, hipStream_t st) {
    hipLaunchKernelGGL(k_rec_encode_w_fast, dim3(a.nbatch), dim3(64), 0, st, a, ticket_fast);
    hipLaunchKernelGGL(k_rec_encode_w_slow, dim3(a.nbatch), dim3(64), 0, st, a, ticket_slow);
}

// =========================================================================================================
// quality encode, symbol-parallel rows  (default quality kernel)
//
// k_qlt_encode_w above spends ~90 instructions per symbol walking one row at a time with the whole wave.
// The rows of different contexts are independent state machines, so the 64 symbols of a window can be
// modelled side by side, one per lane, as long as symbols that share a context are applied in their
// original order.  A bitonic sort of (context, position) keys gives every symbol its rank inside its
// context; round t applies the rank-t symbol of every context at once (each lane runs the Log64Ranger
// search/update on its own row in HBM).  A window needs as many rounds as its most repeated context
// (~10 on Illumina-like data) instead of 64 serial steps.  The triples are un-sorted back into position
// order and the scalar range coder (stage 3) consumes them as before.  Rows use the plain layout
// (slots[64] + RowHdr), shared with the lane-per-block kernels and decoders.
// =========================================================================================================
// Log64Ranger::put minus Encode on one row, by one lane (log64_ranger.hpp:69-112); 16-byte accesses.
// The row is in the WAVE layout (WaveRow above): 64 dwords {total, iend | count << 16, epoch tag, pad, slots 0..59}
// plus a 16-byte overflow row for slots 60..63.  Header and slots 0..11 share the row's first 64-byte sector, and
// the hot symbols live in the front slots, so a typical put reads one sector and writes one -- the table traffic
// is what bounds these kernels.  A row whose tag is stale starts from the shared prior row
// (format 7) or from zeros.
__device__ __forceinline__ u32* l64w_at(u32* row, u32* ovf, u32 i) { return i < 60 ? row + 4 + i : ovf + (i - 60); }
__device__ __forceinline__ Triple l64_model_lane(u32* row, u32* ovf, u32 epoch, const u32* prow, const u32* povf, u32 sym, u32& err) {
    const uint4 hq = *reinterpret_cast<const uint4*>(row);          // {total, iend | count << 16, epoch, pad}
    u32 total, iend, count;
    if (hq.z == epoch) { total = hq.x; iend = hq.y & 0xffffu; count = (hq.y >> 16) & 0xffu; }
    else if (prow) {
        const uint4 ph = *reinterpret_cast<const uint4*>(prow);
        total = ph.x; iend = ph.y & 0xffffu; count = 0;
        const u32 n60 = iend < 60 ? iend : 60;
        for (u32 k = 0; k < n60; k += 4) *reinterpret_cast<uint4*>(row + 4 + k) = *reinterpret_cast<const uint4*>(prow + 4 + k);
        if (iend > 60) *reinterpret_cast<uint4*>(ovf) = *reinterpret_cast<const uint4*>(povf);
    } else { total = 0; iend = 0; count = 0; }
    if (iend <= sym) { for (u32 k = iend; k <= sym; k++) *l64w_at(row, ovf, k) = k << 16; iend = sym + 1; }   // :103-105
    u32 i = 0, sumf = 0, s = 0;
    for (;;) {                                                                                        // :107
        const uint4 q = *reinterpret_cast<const uint4*>(i < 60 ? row + 4 + i : ovf);
        if ((q.x >> 16) == sym) { s = q.x; break; }
        sumf += q.x & 0xffffu;
        if ((q.y >> 16) == sym && i + 1 < iend) { s = q.y; i += 1; break; }
        sumf += q.y & 0xffffu;
        if ((q.z >> 16) == sym && i + 2 < iend) { s = q.z; i += 2; break; }
        sumf += q.z & 0xffffu;
        if ((q.w >> 16) == sym && i + 3 < iend) { s = q.w; i += 3; break; }
        sumf += q.w & 0xffffu;
        i += 4;
        if (i >= 64) { err = 1; i = 63; s = ovf[3]; break; }
    }
    Triple t; t.cum = sumf + i; t.freq = (s & 0xffffu) + 1; t.tot = total + 64;                        // :109
    // update_freq (log64_ranger.hpp:69-87), as dev_models.h Ranger::update on this layout
    u32 f = s & 0xffffu;
    bool upd = true;
    if (f > (u32)((1 << 16) - 64 - 6)) {
        if (i == 0 && f + 20u > total) upd = false;
        else {
            u32 tt = 0;
            for (u32 k = 0; k < iend; k++) { u32* p = l64w_at(row, ovf, k); const u32 nf = (*p & 0xffffu) >> 1; *p = (*p & 0xffff0000u) | nf; tt += nf; }
            total = tt;
            f >>= 1;
        }
    }
    if (upd) {
        f += 6; total += 6;
        const u32 ns = (s & 0xffff0000u) | f;
        bool placed = false;
        if (i != 0) {
            count = (count + 1) & 0xffu;
            if ((count & 0xfu) == 0) {
                u32* pp = l64w_at(row, ovf, i - 1);
                const u32 pv = *pp;
                if (f > (pv & 0xffffu)) { *pp = ns; *l64w_at(row, ovf, i) = pv; placed = true; }          // down_level :56-67
            }
        }
        if (!placed) *l64w_at(row, ovf, i) = ns;
    }
    *reinterpret_cast<uint4*>(row) = make_uint4(total, iend | (count << 16), epoch, 0u);
    return t;
}

// A RUN of L consecutive hits of one symbol in one row, by one lane: triples go to out[0..L).  When the symbol sits
// in the front slot (the usual case for long runs: the dominant symbol bubbles to the front) the repeats are
// the closed form of update_freq(0) -- freq += 6, total += 6, no count / swap (log64_ranger.hpp:69-87) -- with
// the row state in registers; otherwise every repeat is a full put.
__device__ __forceinline__ void l64_model_run_lane(u32* row, u32* ovf, u32 epoch, const u32* prow, const u32* povf,
                                                   u32 sym, u32 L, uint4* out, u32& err) {
    const Triple t0 = l64_model_lane(row, ovf, epoch, prow, povf, sym, err);
    out[0] = make_uint4(t0.cum, t0.freq, t0.tot, 0u);
    u32 r = 1;
    if (L > 1 && (row[4] >> 16) == sym) {
        u32 total = row[0], s0 = row[4], f = s0 & 0xffffu;
        for (; r < L; r++) {
            if (f > (u32)((1 << 16) - 64 - 6)) break;                     // saturation / normalize: leave to the full path
            out[r] = make_uint4(0u, f + 1, total + 64, 0u);
            f += 6; total += 6;
        }
        row[4] = (s0 & 0xffff0000u) | f;
        row[0] = total;
    }
    for (; r < L; r++) {
        const Triple t = l64_model_lane(row, ovf, epoch, prow, povf, sym, err);
        out[r] = make_uint4(t.cum, t.freq, t.tot, 0u);
    }
}


// =========================================================================================================
// quality encode, two blocks per wave  (default quality kernel)
//
// k_qlt_encode_s spends 19 scalar instructions per symbol in its range coder, every lane idle.  Here a wave owns
// TWO blocks (two table slots): stages 1-2 run for a window of each block in turn, full width, and stage 3 walks
// both windows in one instruction stream -- lanes 0..31 carry block A's coder state, lanes 32..63 block B's
// (dev_multicoder.h), the steps come through LDS.  Same bytes, fewer instructions per symbol.
// =========================================================================================================
#include "dev_multicoder.h"

struct QChain {                    // one block's progress; all uniform
    u32 act, b, k, nrec, base, n, solid, extra_hi, p1, p2, p3, carry_d;
    u64 rec0;
    const u8* p;                   // quality line of the current record
};
// put the chain on its next record that has quality symbols (or past the end)
__device__ __forceinline__ void qchain_seek(const ModelArgs& a, QChain& c) {
    while (c.k < c.nrec) {
        const u64 r = c.rec0 + c.k;
        const u64 q0 = a.line_off[4 * r + 3] + c.solid, q1e = a.line_off[4 * r + 4] - 1;
        c.n = q1e > q0 ? (u32)(q1e - q0) : 0;
        c.p = a.fq + q0;
        c.base = 0; c.p1 = c.p2 = c.p3 = 0; c.carry_d = 0;
        if (c.n) return;
        c.k++;
    }
}
// stages 1-2 for the chain's next window: steps[0..) = (cum, freq, tot, reciprocal) in coding order, neutral steps
// behind them.  Returns the number of real steps and moves the chain on.
__device__ __forceinline__ u32 qlt_window_k(const ModelArgs& a, QChain& c, u32* qs, u32* qo, PwTab& pw, u32 epoch, u32 epoch_w,
                                            uint4* steps, uint4* strip, u32& perr, const u32 lane) {
    const int level = a.level;
    const u32 m = c.n - c.base < 64 ? c.n - c.base : 64;
    // ---- stage 1 (as k_qlt_encode_s) ----
    const u32 bv = lane < m ? (u32)(u8)(c.p[c.base + lane] - '!') : 0u;
    const u32 v1 = wave_shr1(bv, c.p1);
    const u32 v2 = wave_shr1(v1, c.p2);
    const u32 v3 = wave_shr1(v2, c.p3);
    u32 ctxv, inc = 0;
    if (level == 1)      ctxv = (v1 | ((v2 & 63u) << 6)) & 0xFFFu;
    else if (level == 2) ctxv = (v1 | (((v2 | ((v3 & 15u) << 6)) & 0x3FFu) << 6)) & 0xFFFFu;
    else {
        const u32 drop = (lane < m && v1 > bv) ? v1 - bv : 0u;
        inc = wave_incl_scan(drop);
        const u32 dprev = 5u + c.carry_d + inc - drop;
        const u32 d3 = dprev >> 3;
        ctxv = (v1 | ((v2 < v3 ? v3 : v2) << 6) | ((u32)(v2 == v3) << 12) | ((d3 < 7 ? d3 : 7) << 13)) & 0xFFFFu;
        if (c.base == 0 && lane == 0) ctxv = 0;
    }
    u32 cons, nsteps;
    const u64 esc = __ballot(lane < m && bv >= LAST_QLT);
    if (!esc) {
        // ---- stage 2: runs of one symbol in one context, one round per run index (as k_qlt_encode_s) ----
        const u32 key = lane < m ? ((ctxv << 6) | lane) : (0x80000000u | (lane << 6) | lane);
        const u32 sk = bitonic_sort64(key, lane);
        const u32 sctx = sk >> 6, spos = sk & 63u;
        const bool valid = !(sk >> 31);
        const u32 ssym = (u32)__builtin_amdgcn_ds_bpermute((int)(spos * 4), (int)bv);
        const u32 prevctx = wave_shr1(sctx, 0xFFFFFFFFu), prevsym = wave_shr1(ssym, 0xFFFFFFFFu);
        const bool ctx_head = sctx != prevctx;
        const bool run_head = ctx_head || ssym != prevsym;
        const u32 cstart = wave_incl_scan_max(ctx_head ? lane : 0u);
        const u32 nheads = wave_incl_scan(run_head ? 1u : 0u);
        const u32 runidx = nheads - (u32)__builtin_amdgcn_ds_bpermute((int)(cstart * 4), (int)nheads);
        const u64 heads = __ballot(run_head);
        const u64 above = lane >= 63 ? 0ull : (heads >> (lane + 1));
        const u32 runlen = above ? (u32)__ffsll((long long)above) : 64u - lane;
        u32* const row = qs + (size_t)(sctx & 0xFFFFu) * L64_NSYM;
        u32* const ovf = qo + (size_t)(sctx & 0xFFFFu) * 4;
        const u32* const prow = a.prior_w ? a.prior_w + (size_t)(sctx & 0xFFFFu) * L64_NSYM : nullptr;
        const u32* const povf = a.prior_wovf + (size_t)(sctx & 0xFFFFu) * 4;
        __syncthreads();
        for (u32 round = 0; ; round++) {
            const bool mine = valid && run_head && runidx == round;
            if (!__ballot(mine)) break;
            if (mine) l64_model_run_lane(row, ovf, epoch_w, prow, povf, ssym, runlen, &strip[lane], perr);
        }
        __syncthreads();
        const uint4 tr = strip[lane];
        steps[spos] = valid ? make_uint4(tr.x, tr.y, tr.z, recip_exact(tr.z)) : NEUTRAL_TRIPLE;   // invalid keys sit at their own lane >= m
        cons = m; nsteps = m;
    } else {
        // escape symbols (quality >= 63: qlts.cpp:80-86) code two steps each: walk at most 32 symbols in order on lane 0
        cons = m < 32 ? m : 32;
        u32 t = 0;
        __syncthreads();
        for (u32 j = 0; j < cons; j++) {
            const u32 ctx = rl(ctxv, j), sym = rl(bv, j);
            if (lane == 0) {
                const Triple t1 = l64_model_lane(qs + (size_t)ctx * L64_NSYM, qo + (size_t)ctx * 4, epoch_w,
                                                 a.prior_w ? a.prior_w + (size_t)ctx * L64_NSYM : nullptr, a.prior_wovf + (size_t)ctx * 4,
                                                 sym < LAST_QLT ? sym : LAST_QLT, perr);
                steps[t] = make_uint4(t1.cum, t1.freq, t1.tot, recip_exact(t1.tot));
                if (sym >= LAST_QLT) {
                    const Triple t2 = Power::model(pw.slots + (size_t)PR_EXQ_ROW * PW_NSYM, pw.hdr + PR_EXQ_ROW, epoch, sym, perr);
                    steps[t + 1] = make_uint4(t2.cum, t2.freq, t2.tot, recip_exact(t2.tot));
                }
            }
            t += sym >= LAST_QLT ? 2u : 1u;
            if (sym >= LAST_QLT) c.extra_hi++;
        }
        if (lane >= t) steps[lane] = NEUTRAL_TRIPLE;
        nsteps = t;
    }
    // ---- the chain moves on by `cons` symbols ----
    const u32 np3 = cons >= 3 ? rl(bv, cons - 3) : (cons == 2 ? c.p1 : c.p2);
    const u32 np2 = cons >= 2 ? rl(bv, cons - 2) : c.p1;
    c.p3 = np3; c.p2 = np2; c.p1 = rl(bv, cons - 1);
    if (level >= 3) c.carry_d += rl(inc, cons - 1);
    c.base += cons;
    if (c.base >= c.n) { c.k++; qchain_seek(a, c); }
    return nsteps;
}

__global__ __launch_bounds__(64, 5) void k_qlt_encode_k2(ModelArgs a, u32* ticket) {
    constexpr u32 K = 2, LPC = 64 / K;
    __shared__ uint4 steps[K][64];
    __shared__ uint4 strip[64];
    const u32 lane = threadIdx.x, h = lane / LPC;
    const bool lead = (lane % LPC) == 0;
    MultiCoder dc; dc.lo = 0; dc.vr = 0xFFFFFFFFu; dc.acc = 0; dc.pos = 0; dc.cap = 0; dc.outp = nullptr; dc.err = 0;
    QChain c[K];
    u32 perr[K];
#pragma unroll
    for (u32 j = 0; j < K; j++) { c[j].act = 0; c[j].k = c[j].nrec = 0; perr[j] = 0; }
    bool drained = false;
    for (;;) {
#pragma unroll
        for (u32 j = 0; j < K; j++) {
            if (c[j].act || drained) continue;
            const u32 b = next_block(ticket);
            if (b >= a.nblocks) { drained = true; continue; }
            const BlockDesc* d = &a.blocks[b];
            c[j].act = 1; c[j].b = b; c[j].k = 0; c[j].nrec = d->nrec; c[j].rec0 = d->rec0; c[j].solid = d->solid; c[j].extra_hi = 0;
            c[j].n = 0; c[j].base = 0; c[j].p = a.fq;
            qchain_seek(a, c[j]);
            dc.reset(h == j, a.arena + d->out_off[SFQ_S_QLT], d->out_cap[SFQ_S_QLT]);
            perr[j] = 0;
        }
        if (!(c[0].act | c[1].act)) break;
        __syncthreads();                                   // one wave per workgroup: orders the LDS traffic across lanes
        u32 nmax = 0;
#pragma unroll
        for (u32 j = 0; j < K; j++) {
            if (c[j].act && c[j].k < c[j].nrec) {
                const size_t slot = (size_t)blockIdx.x * K + j;
                const u32 epoch = EPOCH_L(a.epoch_base + c[j].b + 1);
                PwTab pw; pw.slots = a.p_slots + slot * PR_ROWS * PW_NSYM; pw.hdr = a.p_hdr + slot * PR_ROWS; pw.epoch = epoch;
                const u32 n = qlt_window_k(a, c[j], a.q_slots + slot * a.q_rows * L64_NSYM, reinterpret_cast<u32*>(a.q_hdr + slot * a.q_rows), pw,
                                           epoch, EPOCH_W(a.epoch_base + c[j].b + 1), steps[j], strip, perr[j], lane);
                nmax = n > nmax ? n : nmax;
            } else steps[j][lane] = NEUTRAL_TRIPLE;
        }
        __syncthreads();
        dc.run(steps, nmax, h, lead);                      // ---- stage 3, both chains ----
#pragma unroll
        for (u32 j = 0; j < K; j++) {
            if (!c[j].act || c[j].k < c[j].nrec) continue;
            dc.done(h == j, lead);
            const u32 size = rl(dc.pos, j * LPC), cap = rl(dc.cap, j * LPC), cerr = rl(dc.err, j * LPC);
            const u64 anyerr = __ballot(perr[j] != 0);
            if (lane == 0) {
                BlockDesc* d = &a.blocks[c[j].b];
                d->extra_hi = c[j].extra_hi;
                d->size[SFQ_S_QLT] = size;
                if (size > cap) atomicMax(&d->status, (u32)(-SFQ_E_OVERFLOW));
                if (cerr || anyerr) atomicMax(&d->status, (u32)(-SFQ_E_CORRUPT));
            }
            if (h == j) dc.err = 0;
            c[j].act = 0;
        }
    }
}
void launch_qlt_encode_k(const ModelArgs& a, u32* ticket, hipStream_t st) {
    hipLaunchKernelGGL(k_qlt_encode_k2, dim3((a.nbatch + 1) / 2), dim3(64), 0, st, a, ticket);
}
