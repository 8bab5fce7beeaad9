// decode_w.hip -- wave-per-block decoders over ADAPTIVE tables: format 6 (one block: a reference-written archive) and
// block format archives written with adaptive tables.  decode_l.hip runs the same chains on ONE LANE each: every symbol
// there is a walk of dependent memory reads through its row (1.9 MB/s for a format-6 archive).  Here a wavefront owns
// the block's chain and only what is inherently serial stays serial:
//
//   k_qlt_decode_w   Log64Ranger::get (log64_ranger.hpp:114-138): a row's 64 slots lie one per lane (the wave layout of
//                    models_w.hip: one coalesced 256-byte read fetches header and slots); the cumulative search is a DPP
//                    scan and a ballot; the update touches the one or two lanes that hold the slot(s)
//   k_gen_decode_w   Base2Ranger::get (base2_ranger.hpp:86-104): the row of base t+3 is one of 64 consecutive table
//                    entries once base t-1 is known -- the wave fetches all 64 (one coalesced read per base) three bases
//                    ahead, so the table's latency hides behind three coder steps; rows updated meanwhile are forwarded
//   k_rec_decode_w   RecLoad::load (recs.cpp:374-461) run uniformly by all lanes over wave-cooperative PowerRanger rows
//                    (dev_wavepw.h: 256 slots = four per lane)
//
// Stream bytes come through WaveSrc (dev_wave.h): 256 bytes per fetch, a fetch ahead.  Decoded text leaves 64 bytes a store.
// The bytes decoded are those of decode_l.hip (sfq_params.kernel = 1 keeps that path as the cross-check).
#include "kernels.h"
#include "dev_wavepw.h"
#include "dev_rec_lane.h"

#define LAST_QLT 63u

__device__ __forceinline__ void wset_status(BlockDesc* d, int code) { if (threadIdx.x == 0) atomicMax(&d->status, (u32)(-code)); }

// 64 decoded bytes at a time: lane (i & 63) keeps byte i; a full row leaves as one coalesced store
struct WaveOut {
    u8* p; u32 n, lane, v;
    __device__ __forceinline__ void begin(u8* dst) { p = dst; n = 0; lane = threadIdx.x & 63u; v = 0; }
    __device__ __forceinline__ void put(u32 byte) {
        if (lane == (n & 63u)) v = byte;
        n++;
        if ((n & 63u) == 0) p[n - 64u + lane] = (u8)v;
    }
    __device__ __forceinline__ void end() { if (lane < (n & 63u)) p[(n & ~63u) + lane] = (u8)v; }
};

// ---- Log64Ranger::get on a row in the wave layout --------------------------------------------------------------------
// row: 64 dwords {total, iend | count << 16, epoch tag, pad, slots 0..59}; ovf: slots 60..63 (models_w.hip l64_model_lane).
// Lane L >= 4 holds slot L - 4, lanes 0..3 hold the header -- and slots 60..63 of a row that has them.
template <typename SRC>
__device__ __forceinline__ u32 l64_get_wave(u32* row, u32* ovf, u32 epoch, const u32* prow, const u32* povf, RcDec& rc, SRC& src, u32 lane) {
    u32 v = row[lane];
    u32 total, iend, count;
    bool fresh = false;                                                // the row starts from the prior row: all of it is written back
    if (rl(v, 2) == epoch) { total = rl(v, 0); const u32 ic = rl(v, 1); iend = ic & 0xffffu; count = (ic >> 16) & 0xffu; }
    else if (prow) { v = prow[lane]; total = rl(v, 0); iend = rl(v, 1) & 0xffffu; count = 0; fresh = true; }
    else { total = 0; iend = 0; count = 0; }
    u32 ov = 0;
    if (iend > 60u && lane < 4u) ov = (fresh ? povf : ovf)[lane];
    const u32 si = lane >= 4u ? lane - 4u : 60u + lane;                // this lane's slot
    u32 sv = lane >= 4u ? v : ov;
    if (si >= iend) sv = si << 16;                                     // not in the row yet: its own symbol, frequency 0 (:124-125)
    const u32 f1 = (sv & 0xffffu) + 1u;
    const u32 prob = rc.get_freq(total + 64u);
    // cumulative frequencies in slot order: slots 0..59 by a scan over lanes 4..63, then 60..63 (lanes 0..3) one by one
    const u32 lo = lane >= 4u ? f1 : 0u;
    const u32 incl = wave_incl_scan(lo);
    const u64 past = __ballot(lane >= 4u && incl > prob);
    u32 i, cum, cur;
    if (past) {
        const u32 hl = (u32)__ffsll((long long)past) - 1u;
        i = hl - 4u; cum = rl(incl - lo, hl); cur = rl(sv, hl);
    } else {
        cum = rl(incl, 63); i = 60u; cur = rl(sv, 0);
        for (u32 k = 0; k < 4u; k++) {
            const u32 f = rl(f1, k);
            i = 60u + k; cur = rl(sv, k);
            if (cum + f > prob) break;
            if (k == 3u) { rc.err = 1; break; }                         // the value lies beyond the row's total: a damaged stream
            cum += f;
        }
    }
    const u32 old_iend = iend;
    if (i >= iend) iend = i + 1u;
    rc.decode(src, cum, (cur & 0xffffu) + 1u);
    const u32 sym = (cur >> 16) & 0xffu;
    // update_freq (log64_ranger.hpp:69-87)
    u32 f = cur & 0xffffu;
    bool upd = true, halved = false;
    if (f > (u32)((1 << 16) - 64 - 6)) {
        if (i == 0 && f + 20u > total) upd = false;
        else {
            if (si < iend) sv = (sv & 0xffff0000u) | ((sv & 0xffffu) >> 1);          // normalize :51-54
            const u32 part = si < iend ? (sv & 0xffffu) : 0u;
            total = rl(wave_incl_scan(part), 63);
            f >>= 1; halved = true;
        }
    }
    u32 at = i; bool swapped = false;
    if (upd) {
        f += 6u; total += 6u;
        const u32 ns = (cur & 0xffff0000u) | f;
        if (i != 0) {
            count = (count + 1u) & 0xffu;
            if ((count & 0xfu) == 0) {
                const u32 pl = (i - 1u) < 60u ? i - 1u + 4u : i - 1u - 60u;          // the lane of slot i - 1
                const u32 pv = rl(sv, pl);
                if (f > (pv & 0xffffu)) { if (si == i) sv = pv; at = i - 1u; swapped = true; }     // down_level :56-67
            }
        }
        if (si == at) sv = ns;
    }
    // back to the table: the lanes whose slot changed (all of them after a halving or a fresh start), and the header
    const bool mine = fresh ? si < iend : halved ? si < iend : (si == at || (swapped && si == i) || (si >= old_iend && si <= i));
    if (mine) { if (lane >= 4u) row[lane] = sv; else ovf[lane] = sv; }
    if (lane < 4u) row[lane] = lane == 0 ? total : lane == 1 ? (iend | (count << 16)) : lane == 2 ? epoch : 0u;
    return sym;
}

__global__ __launch_bounds__(64) void k_qlt_decode_w(DecodeArgs a) {
    const u32 lane = threadIdx.x, t = blockIdx.x;
    const u32 b = a.m.batch0 + t;
    BlockDesc* d = &a.m.blocks[b];
    u32* const qs = a.m.q_slots + (size_t)t * a.m.q_rows * L64_NSYM;
    u32* const qo = reinterpret_cast<u32*>(a.m.q_hdr + (size_t)t * a.m.q_rows);     // overflow rows of the wave layout
    const u32 epoch_w = EPOCH_W(a.m.epoch_base + b + 1);
    WavePw pw; pw.slots = a.m.p_slots + (size_t)t * PR_ROWS * PW_NSYM; pw.hdr = a.m.p_hdr + (size_t)t * PR_ROWS; pw.epoch = EPOCH_L(a.m.epoch_base + b + 1);
    WaveSrc src; src.init(a.streams + a.blk_stream_off[(u64)b * SFQ_NSTREAMS + SFQ_S_QLT], d->size[SFQ_S_QLT]);
    RcDec rc; rc.init(src);
    const int level = a.m.level;
    const u64 rec0 = d->rec0; const u32 nrec = d->nrec;
    for (u32 k = 0; k < nrec; k++) {
        const u64 r = rec0 + k;
        const u32 n = a.qlen[r];
        WaveOut out; out.begin(a.qual_stage + a.qoff[r]);
        u32 last = 0, delta = 5, q1 = 0, q2 = 0, di = 0;
        for (u32 i = 0; i < n; i++) {
            u32 bsym = l64_get_wave(qs + (size_t)last * L64_NSYM, qo + (size_t)last * 4, epoch_w,
                                    a.m.prior_w ? a.m.prior_w + (size_t)last * L64_NSYM : nullptr, a.m.prior_wovf + (size_t)last * 4, rc, src, lane);
            if (bsym == LAST_QLT) bsym = pw.get(PR_EXQ_ROW, rc, src, lane);                 // qlts.cpp:168-171
            out.put(('!' + bsym) & 0xffu);
            if (level == 1)      last = (bsym | (last << 6)) & 0xFFFu;                     // qlts.hpp:52-57
            else if (level == 2) last = (bsym | (last << 6)) & 0xFFFFu;
            else {                                                                         // qlts.hpp:62-74, qlts.cpp:127-134
                const u32 p1 = (++di & 1u) ? q1 : q2, p2 = (di & 1u) ? q2 : q1;
                if (p1 > bsym) delta += p1 - bsym;
                const u32 d3 = delta >> 3;
                last = (bsym | ((p1 < p2 ? p2 : p1) << 6) | ((u32)(p1 == p2) << 12) | ((d3 < 7 ? d3 : 7) << 13)) & 0xFFFFu;
                if (di & 1u) q2 = bsym; else q1 = bsym;
            }
        }
        out.end();
    }
    if (rc.err) wset_status(d, SFQ_E_CORRUPT);
}
void launch_qlt_decode_w(const DecodeArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_qlt_decode_w, dim3(a.m.nbatch), dim3(64), 0, st, a);
}

// ---- GenLoad::load_x (gens.cpp:215-249) -------------------------------------------------------------------------------------
// The context of base t is the last gen_bits / 2 bases; the table entry of base t + 3 therefore lies among the 64 consecutive
// entries at ((last_t << 6) & mask) -- known before base t is decoded.  Every step issues that fetch (a dword per lane) and
// takes the row of the NEXT base out of the fetch issued three steps ago by the three bases decoded since.  A row that was
// updated after its fetch was issued (the same context again within four bases: homopolymers, short repeats) comes from the
// last rows written instead.  The N rules are applied as in k_gen_decode_l (decode_l.hip).
__global__ __launch_bounds__(64) void k_gen_decode_w(DecodeArgs a) {
    const u32 lane = threadIdx.x, t = blockIdx.x;
    const u32 b = a.m.batch0 + t;
    BlockDesc* d = &a.m.blocks[b];
    u32* const tab = a.m.g_tab + ((size_t)t << a.m.g_bits);
    WavePw pw; pw.slots = a.m.p_slots + (size_t)t * PR_ROWS * PW_NSYM; pw.hdr = a.m.p_hdr + (size_t)t * PR_ROWS; pw.epoch = EPOCH_L(a.m.epoch_base + b + 1);
    WaveSrc src; src.init(a.streams + a.blk_stream_off[(u64)b * SFQ_NSTREAMS + SFQ_S_GEN], d->size[SFQ_S_GEN]);
    RcDec rc; rc.init(src);
    XfDecW x_ns, x_nn, x_lc;
    x_ns.init(a.streams + a.blk_stream_off[(u64)b * SFQ_NSTREAMS + SFQ_S_GEN_NS], d->size[SFQ_S_GEN_NS], XF_GEN_NS);
    x_nn.init(a.streams + a.blk_stream_off[(u64)b * SFQ_NSTREAMS + SFQ_S_GEN_NN], d->size[SFQ_S_GEN_NN], XF_GEN_NN);
    x_lc.init(a.streams + a.blk_stream_off[(u64)b * SFQ_NSTREAMS + SFQ_S_GEN_LC], d->size[SFQ_S_GEN_LC], XF_GEN_LC);
    u64 ns_index = x_ns.get(pw, lane), nn_index = x_nn.get(pw, lane), lc_index = x_lc.get(pw, lane);      // gens.cpp:187-188
    const u32 n_byte = d->n_byte ? d->n_byte : 'N';                                         // gens.cpp:169
    const u32 code = d->solid ? 0x33323130u /* "0123" */ : 0x54474341u /* "ACGT" */;        // gens.cpp:173-178
    const u32 mask = (1u << d->gen_bits) - 1u;                                              // (gen_bits >= 6: sfq_decode_blocks sends smaller tables to the lane kernel)
    const u32 INIT = 0x007616c7u;                                                           // gens.cpp:139
    u64 genofs = 0;
    const u64 rec0 = d->rec0; const u32 nrec = d->nrec;
    // the rows written in the last four steps (context, row): a fetch issued before them holds the old row
    u32 uc0 = ~0u, ur0 = 0, uc1 = ~0u, ur1 = 0, uc2 = ~0u, ur2 = 0, uc3 = ~0u, ur3 = 0;
    auto newest = [&](u32 ctx, u32 fetched) {                                               // (the newest write wins)
        u32 v = fetched;
        v = ctx == uc3 ? ur3 : v; v = ctx == uc2 ? ur2 : v; v = ctx == uc1 ? ur1 : v; v = ctx == uc0 ? ur0 : v;
        return v;
    };
    for (u32 k = 0; k < nrec; k++) {
        const u64 r = rec0 + k;
        const u32 llen = a.slen[r];
        WaveOut out; out.begin(a.seq_stage + a.soff[r]);
        u32 last = INIT;
        // the fetches in flight: c1 holds the candidates of the next base, c2 those of the base after it; a fetch is indexed by the
        // bases that were unknown when it was issued (one, two or three of them: n1 / n2), gathered in h1 / h2 as they are decoded
        u32 row = newest(last & mask, tab[last & mask]);                                    // base 0: known
        u32 c1 = tab[((last << 2) & mask) + (lane & 3u)];                                   // base 1: one base unknown
        u32 c2 = tab[((last << 4) & mask) + (lane & 15u)];                                  // base 2: two
        u32 h1 = 0, h2 = 0, n1 = 1, n2 = 2;
        for (u32 i = 0; i < llen; i++) {
            const u32 ctx = last & mask;
            const u32 c3 = tab[((last << 6) & mask) + lane];                                // base i + 3: three bases unknown (this one included)
            u32 bsym;
            const u32 nrow = b2_get(row, rc, src, bsym);
            if (lane == 0) tab[ctx] = nrow;
            uc3 = uc2; ur3 = ur2; uc2 = uc1; ur2 = ur1; uc1 = uc0; ur1 = ur0; uc0 = ctx; ur0 = nrow;
            u32 ch = (code >> (8 * bsym)) & 0xffu;
            last = (last << 2) | bsym;
            // the next base's row: out of c1, by the bases decoded since it was issued
            h1 = (h1 << 2) | bsym; h2 = (h2 << 2) | bsym;
            row = newest(last & mask, rl(c1, h1 & ((1u << (2 * n1)) - 1u)));
            c1 = c2; h1 = h2; n1 = n2;
            c2 = c3; h2 = bsym; n2 = 3;
            // normalize_gen gens.cpp:200-213 (decode_l.hip k_gen_decode_l): the rule that needs the qualities is applied by k_assemble
            genofs++;
            if (nn_index == genofs) { nn_index += x_nn.get(pw, lane); ch |= 0x80u; }
            else if (ns_index == genofs) { ch = n_byte; ns_index += x_ns.get(pw, lane); }
            if (lc_index == genofs) { ch |= 0x20u; lc_index += x_lc.get(pw, lane); }
            out.put(ch);
        }
        out.end();
    }
    if (rc.err | x_ns.rc.err | x_nn.rc.err | x_lc.rc.err) wset_status(d, SFQ_E_CORRUPT);
}
void launch_gen_decode_w(const DecodeArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_gen_decode_w, dim3(a.m.nbatch), dim3(64), 0, st, a);
}

// ---- RecLoad::load (recs.cpp:374-461): every lane runs the same header model over wave-cooperative rows ----------------------
struct RecWaveDec {
    static constexpr bool inband = false;
    WavePw pw; RcDec rc; WaveSrc src; u32 lane;
    __device__ __forceinline__ u32 get(u32 row) { return pw.get(row, rc, src, lane); }
    __device__ __forceinline__ u64 get_u(u32 row0) { return pw.get_u(row0, rc, src, lane); }
    __device__ __forceinline__ u32 err() const { return rc.err; }
};
// the block's "rec.x" XFile with the interface rec_decode_lane expects of XfDec (dev_models.h), over the same wave rows
struct XfDecWL {
    XfDecW x; WavePw* pw; u32 lane;
    struct Err { u32 err; } rc;
    __device__ __forceinline__ u64 get(const PwTab&) { const u64 v = x.get(*pw, lane); rc.err = x.rc.err; return v; }
    __device__ __forceinline__ u32 get_chr(const PwTab&) { const u32 v = x.valid ? pw->get(x.row0 + 14, x.rc, x.src, lane) : 0u; rc.err = x.rc.err; return v; }
};
__global__ __launch_bounds__(64) void k_rec_decode_w(DecodeArgs a) {
    const u32 lane = threadIdx.x, t = blockIdx.x;
    const u32 b = a.m.batch0 + t;
    BlockDesc* d = &a.m.blocks[b];
    RecWaveDec cd; cd.lane = lane;
    cd.pw.slots = a.m.p_slots + (size_t)t * PR_ROWS * PW_NSYM; cd.pw.hdr = a.m.p_hdr + (size_t)t * PR_ROWS; cd.pw.epoch = EPOCH_L(a.m.epoch_base + b + 1);
    cd.src.init(a.streams + a.blk_stream_off[(u64)b * SFQ_NSTREAMS + SFQ_S_REC], d->size[SFQ_S_REC]);
    cd.rc.init(cd.src);
    XfDecWL x_rec; x_rec.pw = &cd.pw; x_rec.lane = lane; x_rec.rc.err = 0;
    x_rec.x.init(a.streams + a.blk_stream_off[(u64)b * SFQ_NSTREAMS + SFQ_S_REC_X], d->size[SFQ_S_REC_X], XF_REC_X);
    PwTab none; none.slots = nullptr; none.hdr = nullptr; none.epoch = 0;
    rec_decode_lane(a, d, d->rec0, d->nrec, b, cd, x_rec, none);
}
void launch_rec_decode_w(const DecodeArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_rec_decode_w, dim3(a.m.nbatch), dim3(64), 0, st, a);
}

// ---- UsrLoad::update (usrs.cpp:471-510): per-record line lengths and SOLiD prefixes, a wave per block -------------------
// decode_l.hip k_usr_decode_l walks a block's framing exceptions on ONE lane: every value a row search through dependent
// reads of its adaptive PowerRanger row, ~60 us each -- 1.2 ms at the head of a long-read decode, where every record is an
// exception, with nothing else running.  Here the row search is the wave's (WavePw::get) and the 64 records of a step get
// their lengths a lane each: an exception at record p changes the running values of lanes >= p.
__global__ __launch_bounds__(64) void k_usr_decode_w(DecodeArgs a, u32 prefilled) {
    const u32 lane = threadIdx.x, t = blockIdx.x;
    const u32 b = a.m.batch0 + t;
    BlockDesc* d = &a.m.blocks[b];
    if (prefilled && (d->size[SFQ_S_USR_X] | d->size[SFQ_S_USR_XQ] | d->size[SFQ_S_USR_PFG] | d->size[SFQ_S_USR_PFQ]) == 0) return;   // (decode_l.hip k_usr_fill has written this block)
    WavePw pw; pw.slots = a.m.p_slots + (size_t)t * PR_ROWS * PW_NSYM; pw.hdr = a.m.p_hdr + (size_t)t * PR_ROWS; pw.epoch = EPOCH_L(a.m.epoch_base + b + 1);
    XfDecW x_llen, x_qlen, x_sgen, x_sqlt;
    x_llen.init(a.streams + a.blk_stream_off[(u64)b * SFQ_NSTREAMS + SFQ_S_USR_X],   d->size[SFQ_S_USR_X],   XF_USR_X);
    x_qlen.init(a.streams + a.blk_stream_off[(u64)b * SFQ_NSTREAMS + SFQ_S_USR_XQ],  d->size[SFQ_S_USR_XQ],  XF_USR_XQ);
    x_sgen.init(a.streams + a.blk_stream_off[(u64)b * SFQ_NSTREAMS + SFQ_S_USR_PFG], d->size[SFQ_S_USR_PFG], XF_USR_PFG);
    x_sqlt.init(a.streams + a.blk_stream_off[(u64)b * SFQ_NSTREAMS + SFQ_S_USR_PFQ], d->size[SFQ_S_USR_PFQ], XF_USR_PFQ);
    u64 i_llen = x_llen.get(pw, lane), i_qlen = x_qlen.get(pw, lane), i_sgen = x_sgen.get(pw, lane), i_sqlt = x_sqlt.get(pw, lane);
    const u32 solid = d->solid, nrec = d->nrec;
    const u64 rec0 = d->rec0;
    u32 c_llen = d->llen, c_pfg = 0, c_pfq = 0;              // the running values behind the step at hand (uniform)
    u32 bad = 0;
    for (u32 k0 = 0; k0 < nrec; k0 += 64) {
        const u32 k = k0 + lane;
        const bool have = k < nrec;
        const u64 r = rec0 + (have ? k : 0);
        const u64 first_cnt = rec_count_of(a.m, rec0 + k0, rec0);                // the step's first record's number (the records of a block count up by one)
        const u64 last_cnt = first_cnt + 63;
        u32 llen = c_llen, qx = 0xFFFFFFFFu, pfg = c_pfg, pfq = c_pfq;          // qx: this record's own quality length, if it has one
        // every exception that falls into this step, in the order the serial loop meets them
        while (i_llen && i_llen >= first_cnt && i_llen <= last_cnt) {
            const u32 p = (u32)(i_llen - first_cnt);
            const u32 v = (u32)x_llen.get(pw, lane);
            if (lane >= p) llen = v;
            c_llen = v;
            const u64 gap = x_llen.get(pw, lane);
            if (!gap) { i_llen = 0; break; }
            i_llen += gap;
        }
        while (i_qlen && i_qlen >= first_cnt && i_qlen <= last_cnt) {
            const u32 p = (u32)(i_qlen - first_cnt);
            const u32 v = (u32)x_qlen.get(pw, lane);
            if (lane == p) qx = v;
            const u64 gap = x_qlen.get(pw, lane);
            if (!gap) { i_qlen = 0; break; }
            i_qlen += gap;
        }
        if (solid) {
            while (i_sgen && i_sgen >= first_cnt && i_sgen <= last_cnt) {
                const u32 p = (u32)(i_sgen - first_cnt);
                const u32 v = x_sgen.get_chr(pw, lane);
                if (lane >= p) pfg = v;
                c_pfg = v;
                const u64 gap = x_sgen.get(pw, lane);
                if (!gap) { i_sgen = 0; break; }
                i_sgen += gap;
            }
            while (i_sqlt && i_sqlt >= first_cnt && i_sqlt <= last_cnt) {
                const u32 p = (u32)(i_sqlt - first_cnt);
                const u32 v = x_sqlt.get_chr(pw, lane);
                if (lane >= p) pfq = v;
                c_pfq = v;
                const u64 gap = x_sqlt.get(pw, lane);
                if (!gap) { i_sqlt = 0; break; }
                i_sqlt += gap;
            }
        }
        u32 qlen = qx != 0xFFFFFFFFu ? qx : llen;
        if (llen > a.max_line || qlen > a.max_line) { bad = 1; llen = qlen = 0; }
        if (have) { a.slen[r] = llen; a.qlen[r] = qlen; a.pfg[r] = (u8)pfg; a.pfq[r] = (u8)pfq; }
    }
    if (__any(bad != 0) | x_llen.rc.err | x_qlen.rc.err | x_sgen.rc.err | x_sqlt.rc.err) { if (lane == 0) atomicMax(&d->status, (u32)(-SFQ_E_CORRUPT)); }
}
void launch_usr_decode_w(const DecodeArgs& a, hipStream_t st, u32 prefilled) {
    hipLaunchKernelGGL(k_usr_decode_w, dim3(a.m.nbatch), dim3(64), 0, st, a, prefilled);
}
