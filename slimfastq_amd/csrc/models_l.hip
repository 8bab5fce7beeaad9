// models_l.hip -- lane-per-block kernels: every lane owns one record block and runs that block's
// serial model + range-coder chains exactly as the reference does on a CPU thread, with the model
// tables in HBM.  These are the bit-exactness anchor on the device (selected with
// sfq_params.kernel = 1); the throughput kernels (models_w.hip, models_k.hip) must reproduce their bytes.
//
//   quality  : QltSave::save_1/2/3, QltLoad::load_1/2/3      qlts.cpp:74-136, 163-234
//   bases    : GenSave::save_x / normalize_gen, GenLoad      gens.cpp:91-159, 200-249
//   headers  : RecSave::save, RecLoad::load                  recs.cpp:141-461
//   framing exceptions : UsrSave::get_record / update, UsrLoad::update   usrs.cpp:126-160, 303-390, 471-510
#include "kernels.h"
#include "dev_models.h"

#define LAST_QLT 63u     // log64_ranger.hpp:34

struct Slot {
    u32 b;          // block index
    u32 epoch;
    u32* q_slots; RowHdr* q_hdr;
    PwTab pw;
    u32* g_tab;
};
__device__ __forceinline__ bool slot_init(const ModelArgs& a, Slot& s) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.nbatch) return false;
    s.b = a.batch0 + t;
    s.epoch = EPOCH_L(a.epoch_base + s.b + 1);
    s.q_slots = a.q_slots ? a.q_slots + (size_t)t * a.q_rows * L64_NSYM : nullptr;
    s.q_hdr   = a.q_hdr ? a.q_hdr + (size_t)t * a.q_rows : nullptr;
    s.pw.slots = a.p_slots + (size_t)t * PR_ROWS * PW_NSYM;
    s.pw.hdr   = a.p_hdr + (size_t)t * PR_ROWS;
    s.pw.epoch = s.epoch;
    s.g_tab = a.g_tab ? a.g_tab + ((size_t)t << a.g_bits) : nullptr;
    return true;
}
__device__ __forceinline__ void set_status(BlockDesc* d, int code) { atomicMax(&d->status, (u32)(-code)); }
__device__ __forceinline__ void finish_stream(BlockDesc* d, int s, const ByteSink& k, u32 err) {
    d->size[s] = k.pos;
    if (k.pos > k.cap) set_status(d, SFQ_E_OVERFLOW);
    if (err) set_status(d, SFQ_E_CORRUPT);
}

// qlts.hpp:62-74
__device__ __forceinline__ u32 calc_last_delta(u32& delta, u32 q, u32 q1, u32 q2) {
    if (q1 > q) delta += q1 - q;
    u32 d3 = delta >> 3;
    return (q | ((q1 < q2 ? q2 : q1) << 6) | ((u32)(q1 == q2) << 12) | ((d3 < 7 ? d3 : 7) << 13)) & 0xFFFFu;
}

// ===================================================================================================
// quality encode
// ===================================================================================================
__global__ __launch_bounds__(64) void k_qlt_encode_l(ModelArgs a) {
    Slot sl;
    if (!slot_init(a, sl)) return;
    BlockDesc* d = &a.blocks[sl.b];
    ByteSink snk = { a.arena + d->out_off[SFQ_S_QLT], 0, d->out_cap[SFQ_S_QLT] };
    RcEnc rc; rc.init();
    u32 extra_hi = 0;
    const u32 solid = d->solid;
    const int level = a.level;
    for (u64 r = d->rec0; r < d->rec0 + d->nrec; r++) {
        const u64 q0 = a.line_off[4 * r + 3] + solid;
        const u64 q1e = a.line_off[4 * r + 4] - 1;
        const u32 n = q1e > q0 ? (u32)(q1e - q0) : 0;
        const u8* p = a.fq + q0;
        u32 last = 0, delta = 5, q1 = 0, q2 = 0, di = 0;
        for (u32 i = 0; i < n; i++) {
            const u32 bsym = (u32)(u8)(p[i] - '!');
            u32* row = sl.q_slots + (size_t)last * L64_NSYM;
            RowHdr* hp = sl.q_hdr + last;
            l64_touch(row, hp, sl.epoch, a.prior_ls ? a.prior_ls + (size_t)last * L64_NSYM : nullptr, a.prior_lh + last);
            if (bsym < LAST_QLT) Log64::put(row, hp, sl.epoch, rc, snk, bsym);        // qlts.cpp:79-86
            else {
                Log64::put(row, hp, sl.epoch, rc, snk, LAST_QLT);
                sl.pw.put(PR_EXQ_ROW, rc, snk, bsym);
                extra_hi++;
            }
            if (level == 1)      last = (bsym | (last << 6)) & 0xFFFu;                  // qlts.hpp:52-54
            else if (level == 2) last = (bsym | (last << 6)) & 0xFFFFu;                 // qlts.hpp:55-57
            else if (++di & 1) { last = calc_last_delta(delta, bsym, q1, q2); q2 = bsym; }   // qlts.cpp:127-134
            else               { last = calc_last_delta(delta, bsym, q2, q1); q1 = bsym; }
        }
    }
    rc.done(snk);
    d->extra_hi = extra_hi;
    finish_stream(d, SFQ_S_QLT, snk, rc.err);
}
void launch_qlt_encode_l(const ModelArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_qlt_encode_l, dim3((a.nbatch + 63) / 64), dim3(64), 0, st, a);
}

// ===================================================================================================
// base encode
// ===================================================================================================
// gens.cpp:72-77 : 0..3 bases, 4 = N-like, 0x10 = illegal
__device__ __forceinline__ u32 gencode(u32 c) {
    switch (c) {
    case '0': case 'A': case 'a': return 0;
    case '1': case 'C': case 'c': return 1;
    case '2': case 'G': case 'g': return 2;
    case '3': case 'T': case 't': return 3;
    case '.': case 'N': case 'n': return 4;
    default: return 0x10;
    }
}
__global__ __launch_bounds__(64) void k_gen_encode_l(ModelArgs a) {
    Slot sl;
    if (!slot_init(a, sl)) return;
    BlockDesc* d = &a.blocks[sl.b];
    ByteSink snk = { a.arena + d->out_off[SFQ_S_GEN], 0, d->out_cap[SFQ_S_GEN] };
    RcEnc rc; rc.init();
    XfEnc x_ns, x_nn, x_lc;
    x_ns.init(a.arena + d->out_off[SFQ_S_GEN_NS], d->out_cap[SFQ_S_GEN_NS], XF_GEN_NS);
    x_nn.init(a.arena + d->out_off[SFQ_S_GEN_NN], d->out_cap[SFQ_S_GEN_NN], XF_GEN_NN);
    x_lc.init(a.arena + d->out_off[SFQ_S_GEN_LC], d->out_cap[SFQ_S_GEN_LC], XF_GEN_LC);
    const u32 solid = d->solid;
    const u32 mask = (1u << d->gen_bits) - 1u;
    u64 genofs = 0, ns_index = 0, nn_index = 0, lc_index = 0;       // g_genofs_count, m_last.{Ns,Nn}_index (block-relative); the last "gen.lc" entry
    u32 n_byte = 0; int bad = 0;
    for (u64 r = d->rec0; r < d->rec0 + d->nrec; r++) {
        const u64 g0 = a.line_off[4 * r + 1] + solid, g1 = a.line_off[4 * r + 2] - 1;
        const u64 q0 = a.line_off[4 * r + 3] + solid, q1 = a.line_off[4 * r + 4] - 1;
        const u32 llen = g1 > g0 ? (u32)(g1 - g0) : 0, qlen = q1 > q0 ? (u32)(q1 - q0) : 0;
        const u8* gp = a.fq + g0; const u8* qp = a.fq + q0;
        u32 last = 0x007616c7u;                                                   // gens.cpp:139
        for (u32 i = 0; i < llen; i++) {
            u32 gch = gp[i];
            const u32 qch = (i < qlen) ? qp[i] : 40u;                             // gens.cpp:153
            if (a.lossless && is_lower_base(gch)) {                               // block format: the case goes to "gen.lc" (dev_common.h)
                x_lc.put(sl.pw, genofs + 1 - lc_index); lc_index = genofs + 1;
                if (gch == 'n') gch = 'N';
            }
            u32 n = gencode(gch);                                                 // normalize_gen gens.cpp:116-136
            const bool bad_q = qch == '!';
            bool bad_n = false;
            if (n > 3) { if (n > 4) bad = SFQ_E_GENCHAR; bad_n = true; n = 0; }
            genofs++;
            if (bad_n || bad_q) {                                                 // bad_q_or_bad_n gens.cpp:91-114
                if (!bad_n) { x_nn.put(sl.pw, genofs - nn_index); nn_index = genofs; }
                else {
                    if (!n_byte) n_byte = gch;
                    if (gch != n_byte) bad = SFQ_E_GENCHAR;
                    if (!bad_q) { x_ns.put(sl.pw, genofs - ns_index); ns_index = genofs; }
                }
            }
            last &= mask;
            sl.g_tab[last] = b2_put(sl.g_tab[last], rc, snk, n);
            last = (last << 2) | n;
        }
    }
    rc.done(snk);
    d->n_byte = n_byte;
    finish_stream(d, SFQ_S_GEN, snk, rc.err);
    d->size[SFQ_S_GEN_NS] = x_ns.finish(sl.pw);
    d->size[SFQ_S_GEN_NN] = x_nn.finish(sl.pw);
    d->size[SFQ_S_GEN_LC] = x_lc.finish(sl.pw);
    if (x_ns.sink.pos > x_ns.sink.cap || x_nn.sink.pos > x_nn.sink.cap || x_lc.sink.pos > x_lc.sink.cap) set_status(d, SFQ_E_OVERFLOW);
    if (x_ns.rc.err | x_nn.rc.err | x_lc.rc.err) set_status(d, SFQ_E_CORRUPT);
    if (bad) set_status(d, bad);
}
void launch_gen_encode_l(const ModelArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_gen_encode_l, dim3((a.nbatch + 63) / 64), dim3(64), 0, st, a);
}

// ===================================================================================================
// header encode
// ===================================================================================================
#include "dev_rec_lane.h"

__global__ __launch_bounds__(64) void k_rec_encode_l(ModelArgs a) {
    Slot sl;
    if (!slot_init(a, sl)) return;
    BlockDesc* d = &a.blocks[sl.b];
    RecAdaptiveEnc cd;
    cd.pw = sl.pw; cd.snk.p = a.arena + d->out_off[SFQ_S_REC]; cd.snk.pos = 0; cd.snk.cap = d->out_cap[SFQ_S_REC]; cd.rc.init();
    XfEnc x_rec; x_rec.init(a.arena + d->out_off[SFQ_S_REC_X], d->out_cap[SFQ_S_REC_X], XF_REC_X);
    u32 hdr_bytes = 0; int bad = 0;
    rec_encode_lane(a, d->rec0, d->rec0, d->nrec, cd, x_rec, sl.pw, hdr_bytes, bad);
    cd.rc.done(cd.snk);
    d->hdr_bytes = hdr_bytes;
    finish_stream(d, SFQ_S_REC, cd.snk, cd.rc.err);
    d->size[SFQ_S_REC_X] = x_rec.finish(sl.pw);
    if (x_rec.sink.pos > x_rec.sink.cap) set_status(d, SFQ_E_OVERFLOW);
    if (x_rec.rc.err) set_status(d, SFQ_E_CORRUPT);
    if (bad) set_status(d, bad);
}
void launch_rec_encode_l(const ModelArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_rec_encode_l, dim3((a.nbatch + 63) / 64), dim3(64), 0, st, a);
}

// ===================================================================================================
// framing exceptions: the bookkeeping half of UsrSave::get_record (usrs.cpp:322-375) + update (126-160)
// ===================================================================================================
__global__ __launch_bounds__(64) void k_usr_encode_l(ModelArgs a) {
    Slot sl;
    if (!slot_init(a, sl)) return;
    BlockDesc* d = &a.blocks[sl.b];
    XfEnc x_llen, x_qlen, x_sgen, x_sqlt;
    x_llen.init(a.arena + d->out_off[SFQ_S_USR_X],   d->out_cap[SFQ_S_USR_X],   XF_USR_X);
    x_qlen.init(a.arena + d->out_off[SFQ_S_USR_XQ],  d->out_cap[SFQ_S_USR_XQ],  XF_USR_XQ);
    x_sgen.init(a.arena + d->out_off[SFQ_S_USR_PFG], d->out_cap[SFQ_S_USR_PFG], XF_USR_PFG);
    x_sqlt.init(a.arena + d->out_off[SFQ_S_USR_PFQ], d->out_cap[SFQ_S_USR_PFQ], XF_USR_PFQ);
    const u32 solid = d->solid;
    u32 llen = d->llen;
    u64 i_llen = 0, i_qlen = 0, i_sgen = 0, i_sqlt = 0;
    u32 pf_gen = 0, pf_qlt = 0;
    for (u32 k = 0; k < d->nrec; k++) {
        const u64 r = d->rec0 + k;
        const u64 rcnt = rec_count_of(a, r, d->rec0);
        const u64 g0 = a.line_off[4 * r + 1], g1 = a.line_off[4 * r + 2] - 1;
        const u64 q0 = a.line_off[4 * r + 3], q1 = a.line_off[4 * r + 4] - 1;
        if (solid) {
            const u32 c = a.fq[g0];                                            // usrs.cpp:323-327, 339-340
            if (c != pf_gen) { x_sgen.put(sl.pw, rcnt - i_sgen); x_sgen.put_chr(sl.pw, c); i_sgen = rcnt; pf_gen = c; }
        }
        const u32 sl_len = (u32)(g1 - g0) - solid;
        if (sl_len != llen) {                                                  // usrs.cpp:342-343
            x_llen.put(sl.pw, rcnt - i_llen); x_llen.put(sl.pw, sl_len); i_llen = rcnt; llen = sl_len;
        }
        if (solid) {
            const u32 c = a.fq[q0];                                            // usrs.cpp:356-360
            if (c != pf_qlt) { x_sqlt.put(sl.pw, rcnt - i_sqlt); x_sqlt.put_chr(sl.pw, c); i_sqlt = rcnt; pf_qlt = c; }
        }
        const u32 ql = (q1 - q0) >= solid ? (u32)(q1 - q0) - solid : 0;
        if (ql != llen) { x_qlen.put(sl.pw, rcnt - i_qlen); x_qlen.put(sl.pw, ql); i_qlen = rcnt; }   // usrs.cpp:371-372
        if (a.lossless && !plus_line_is_regular(a.fq, a.line_off, r, d->two_id)) set_status(d, SFQ_E_UNSUPPORTED);   // dev_common.h
    }
    d->size[SFQ_S_USR_X]   = x_llen.finish(sl.pw);
    d->size[SFQ_S_USR_XQ]  = x_qlen.finish(sl.pw);
    d->size[SFQ_S_USR_PFG] = x_sgen.finish(sl.pw);
    d->size[SFQ_S_USR_PFQ] = x_sqlt.finish(sl.pw);
    if (x_llen.sink.pos > x_llen.sink.cap || x_qlen.sink.pos > x_qlen.sink.cap ||
        x_sgen.sink.pos > x_sgen.sink.cap || x_sqlt.sink.pos > x_sqlt.sink.cap) set_status(d, SFQ_E_OVERFLOW);
    if (x_llen.rc.err | x_qlen.rc.err | x_sgen.rc.err | x_sqlt.rc.err) set_status(d, SFQ_E_CORRUPT);
}
void launch_usr_encode_l(const ModelArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_usr_encode_l, dim3((a.nbatch + 63) / 64), dim3(64), 0, st, a);
}
