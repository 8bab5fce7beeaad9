// models_k.hip -- multi-chain kernels: one wavefront codes K record blocks at once.
//
// Measured on MI355X: these kernels are bound by instruction ISSUE -- about 4.4 cycles per
// wave-instruction per SIMD whatever its type -- and the serial range-coder chain (stage 3 of models_w.hip)
// is the largest share: ~19 instructions per symbol executed by a whole wave for ONE chain, every lane
// computing the same value.  Here a wave owns K blocks (K table slots).  Stages 1-2 (contexts, rows,
// triples) run for a 64-symbol window of each block in turn, full width; stage 3 then walks all K windows in
// ONE instruction stream, lane group h = lane / (64/K) carrying block h's coder state.  The triples go through
// LDS (one ds_read_b128 per step, broadcast within a group) instead of four v_readlane per symbol.
// The bytes are the same as models_w.hip / models_l.hip / the reference: only the schedule differs.
#include "kernels.h"
#include "dev_wave.h"
#include "dev_multicoder.h"

// =========================================================================================================
// base encode, K blocks per wave: GenSave::save_x + normalize_gen (gens.cpp:91-159)
// =========================================================================================================
struct GenChains {                    // per-chain state, lane j = chain j
    u32 act, b, k, nrec, rec0l, rec0h, base, carry, mask, solid;
    u32 gl, gh, nsl, nsh, nnl, nnh;   // g_genofs_count, m_last.{Ns,Nn}_index (block-relative), 64 bit each
    u32 lcl, lch;                     // the last lowercase base listed in "gen.lc" (block format only)
    u32 nbyte, bad;
};

// one 64-base window of chain j -> trip[j][0..63]; returns the number of real steps, advances the chain
__device__ __forceinline__ u32 gen_window(const ModelArgs& a, GenChains& g, u32 j, u32* tab, uint4 (*trip)[64], XfEnc* xf, const PwTab& pw, u32 lane) {
    const u64 r = (((u64)CGET(g.rec0h, j) << 32) | CGET(g.rec0l, j)) + CGET(g.k, j);
    const u32 solid = CGET(g.solid, j), mask = CGET(g.mask, j), base = CGET(g.base, j);
    const u64 g0 = a.line_off[4 * r + 1] + solid, g1 = a.line_off[4 * r + 2] - 1;
    const u64 q0 = a.line_off[4 * r + 3] + solid, q1 = a.line_off[4 * r + 4] - 1;
    const u32 llen = g1 > g0 ? (u32)(g1 - g0) : 0, qlen = q1 > q0 ? (u32)(q1 - q0) : 0;
    const u8* gp = a.fq + g0; const u8* qp = a.fq + q0;
    const u32 carry = base ? CGET(g.carry, j) : 0x007616c7u;                          // gens.cpp:139
    u64 genofs = ((u64)CGET(g.gh, j) << 32) | CGET(g.gl, j);
    u64 ns_index = ((u64)CGET(g.nsh, j) << 32) | CGET(g.nsl, j), nn_index = ((u64)CGET(g.nnh, j) << 32) | CGET(g.nnl, j);
    u64 lc_index = ((u64)CGET(g.lch, j) << 32) | CGET(g.lcl, j);
    u32 n_byte = CGET(g.nbyte, j), bad = CGET(g.bad, j);

    const u32 m = llen - base < 64 ? llen - base : 64;
    const u32 idx = base + lane;
    const bool in = lane < m;
    const u32 gch = in ? gp[idx] : 'A';
    const u32 qch = (in && idx < qlen) ? qp[idx] : 40u;                               // gens.cpp:153
    const u32 n = gencode_w(gch);                                                     // normalize_gen gens.cpp:116-136
    const bool bad_n = in && n == 4, bad_q = in && qch == '!';
    if (__ballot(in && n > 4)) bad = (u32)(-SFQ_E_GENCHAR);
    const u32 code = n & 3u;                                                          // N is coded as 0 (A)
    const u64 mN = __ballot(bad_n), mQ = __ballot(bad_q);
    u64 mx = mN | mQ;
    while (mx) {                                                                      // bad_q_or_bad_n gens.cpp:91-114, in order
        const u32 bit = (u32)__ffsll((long long)mx) - 1u;
        mx &= mx - 1;
        const u64 pos = genofs + bit + 1;
        const bool is_n = (mN >> bit) & 1, is_q = (mQ >> bit) & 1;
        if (!is_n) {
            if (lane == 0) xf[1].put(pw, pos - nn_index);
            nn_index = pos;
        } else {
            u32 ch = rl(gch, bit);
            if (a.lossless && ch == 'n') ch = 'N';                                    // its case travels in "gen.lc"
            if (!n_byte) n_byte = ch;
            if (ch != n_byte) bad = (u32)(-SFQ_E_GENCHAR);
            if (!is_q) { if (lane == 0) xf[0].put(pw, pos - ns_index); ns_index = pos; }
        }
    }
    if (a.lossless) {                                                                 // lowercase bases: "gen.lc" (dev_common.h)
        u64 ml = __ballot(in && is_lower_base(gch));
        while (ml) {
            const u32 bit = (u32)__ffsll((long long)ml) - 1u;
            ml &= ml - 1;
            const u64 pos = genofs + bit + 1;
            if (lane == 0) xf[2].put(pw, pos - lc_index);
            lc_index = pos;
        }
    }
    genofs += m;
    // contexts: a 32-bit shift register of 2-bit codes; lane k sees the codes of lanes < k, then the carry
    u32 w = wave_shr1(code, 0u);
    w |= wave_shr1(w, 0u) << 2;
    w |= shfl_up0(w, 2, lane) << 4;
    w |= shfl_up0(w, 4, lane) << 8;
    w |= shfl_up0(w, 8, lane) << 16;
    const u32 ctx = ((lane < 16 ? carry << (2 * lane) : 0u) | w) & mask;
    const u32 ncarry = (rl(w, 63) << 2) | rl(code, 63);
    // rows: gather, update, scatter; a context repeated inside the window chains in order
    u32 row = in ? tab[ctx] : 0u;
    const u32 key = in ? ((ctx << 6) | lane) : (0x80000000u | (lane << 6) | lane);
    const u32 sk = bitonic_sort64(key, lane);
    const u32 skp = (u32)__shfl_up((int)sk, 1, 64);
    const bool dup = lane > 0 && (sk >> 6) == (skp >> 6);
    u32 cum = 0, freq = 1, tot = 1;
    if (!__ballot(dup)) {
        const u32 nrow = b2_model(row, code, cum, freq, tot);
        if (in) tab[ctx] = nrow;
    } else {
        for (u32 i = 0; i < m; i++) {
            const u32 cc = rl(ctx, i), s = rl(code, i);
            u32 cj, fj, tj;
            const u32 nrow = b2_model(tab[cc], s, cj, fj, tj);
            if (lane == 0) tab[cc] = nrow;
            if (lane == i) { cum = cj; freq = fj; tot = tj; }
        }
    }
    trip[j][lane] = in ? make_uint4(cum, freq, tot, recip_exact(tot)) : NEUTRAL_TRIPLE;
    // advance the chain
    const bool rec_done = base + 64 >= llen;
    CSET(g.base, j, rec_done ? 0u : base + 64);
    CSET(g.k, j, CGET(g.k, j) + (rec_done ? 1u : 0u));
    CSET(g.carry, j, ncarry);
    CSET(g.gl, j, (u32)genofs); CSET(g.gh, j, (u32)(genofs >> 32));
    CSET(g.nsl, j, (u32)ns_index); CSET(g.nsh, j, (u32)(ns_index >> 32));
    CSET(g.nnl, j, (u32)nn_index); CSET(g.nnh, j, (u32)(nn_index >> 32));
    CSET(g.lcl, j, (u32)lc_index); CSET(g.lch, j, (u32)(lc_index >> 32));
    CSET(g.nbyte, j, n_byte); CSET(g.bad, j, bad);
    return m;
}

template <int K>
__global__ __launch_bounds__(64) void k_gen_encode_k(ModelArgs a, u32* ticket) {
    constexpr u32 LPC = 64 / K;                        // lanes per chain
    __shared__ uint4 trip[K][64];
    __shared__ XfEnc xfs[K][3];                        // [chain][0 = gen.Ns, 1 = gen.Nn, 2 = gen.lc]: side-stream coders, used by lane 0
    const u32 lane = threadIdx.x, h = lane / LPC;
    const bool lead = (lane % LPC) == 0;
    MultiCoder dc; dc.lo = 0; dc.vr = 0xFFFFFFFFu; dc.acc = 0; dc.pos = 0; dc.cap = 0; dc.outp = nullptr; dc.err = 0;
    GenChains g; g.act = 0; g.b = g.k = g.nrec = g.rec0l = g.rec0h = g.base = g.carry = g.mask = g.solid = 0;
    g.gl = g.gh = g.nsl = g.nsh = g.nnl = g.nnh = g.lcl = g.lch = g.nbyte = g.bad = 0;
    bool drained = false;                              // the ticket counter ran out
    for (;;) {
        // (re)fill idle chains with new blocks
#pragma nounroll
        for (u32 j = 0; j < (u32)K; j++) {
            if (CGET(g.act, j) || drained) continue;
            const u32 b = next_block(ticket);
            if (b >= a.nblocks) { drained = true; continue; }
            const BlockDesc* d = &a.blocks[b];
            const u64 rec0 = d->rec0;
            CSET(g.act, j, 1u); CSET(g.b, j, b); CSET(g.k, j, 0u); CSET(g.nrec, j, d->nrec);
            CSET(g.rec0l, j, (u32)rec0); CSET(g.rec0h, j, (u32)(rec0 >> 32));
            CSET(g.base, j, 0u); CSET(g.solid, j, (u32)d->solid); CSET(g.mask, j, (1u << d->gen_bits) - 1u);
            CSET(g.gl, j, 0u); CSET(g.gh, j, 0u); CSET(g.nsl, j, 0u); CSET(g.nsh, j, 0u); CSET(g.nnl, j, 0u); CSET(g.nnh, j, 0u);
            CSET(g.lcl, j, 0u); CSET(g.lch, j, 0u);
            CSET(g.nbyte, j, 0u); CSET(g.bad, j, 0u);
            // Base2Ranger rows start at 3,3,3,3 (base2_ranger.hpp:68-71)
            u32* tab = a.g_tab + (((size_t)blockIdx.x * K + j) << a.g_bits);
            const u32 n4 = 1u << (a.g_bits - 2);
#pragma unroll 4
            for (u32 i = lane; i < n4; i += 64) reinterpret_cast<uint4*>(tab)[i] = make_uint4(B2_INIT, B2_INIT, B2_INIT, B2_INIT);
            if (lane == 0) {
                xfs[j][0].init(a.arena + d->out_off[SFQ_S_GEN_NS], d->out_cap[SFQ_S_GEN_NS], XF_GEN_NS);
                xfs[j][1].init(a.arena + d->out_off[SFQ_S_GEN_NN], d->out_cap[SFQ_S_GEN_NN], XF_GEN_NN);
                xfs[j][2].init(a.arena + d->out_off[SFQ_S_GEN_LC], d->out_cap[SFQ_S_GEN_LC], XF_GEN_LC);
            }
            dc.reset(h == j, a.arena + d->out_off[SFQ_S_GEN], d->out_cap[SFQ_S_GEN]);
        }
        if (!__ballot(lane < (u32)K && g.act)) break;
        __syncthreads();                               // one wave per workgroup: orders the LDS traffic across lanes
        // stages 1-2: one window of each active block
        u32 nmax = 0;
#pragma nounroll
        for (u32 j = 0; j < (u32)K; j++) {
            if (CGET(g.act, j)) {
                const size_t slot = (size_t)blockIdx.x * K + j;
                PwTab pw; pw.slots = a.p_slots + slot * PR_ROWS * PW_NSYM; pw.hdr = a.p_hdr + slot * PR_ROWS;
                pw.epoch = EPOCH_L(a.epoch_base + CGET(g.b, j) + 1);
                const u32 n = gen_window(a, g, j, a.g_tab + (slot << a.g_bits), trip, xfs[j], pw, lane);
                nmax = n > nmax ? n : nmax;
            } else trip[j][lane] = NEUTRAL_TRIPLE;
        }
        __syncthreads();
        // stage 3: all chains
        dc.run(trip, nmax, h, lead);
        // blocks that ran out of records: flush, publish sizes
#pragma nounroll
        for (u32 j = 0; j < (u32)K; j++) {
            if (!CGET(g.act, j) || CGET(g.k, j) < CGET(g.nrec, j)) continue;
            dc.done(h == j, lead);
            const u32 size = rl(dc.pos, j * LPC), cap = rl(dc.cap, j * LPC), cerr = rl(dc.err, j * LPC);
            const u32 bad = CGET(g.bad, j), nb = CGET(g.nbyte, j), b = CGET(g.b, j);
            if (lane == 0) {
                BlockDesc* d = &a.blocks[b];
                const size_t slot = (size_t)blockIdx.x * K + j;
                PwTab pw; pw.slots = a.p_slots + slot * PR_ROWS * PW_NSYM; pw.hdr = a.p_hdr + slot * PR_ROWS; pw.epoch = EPOCH_L(a.epoch_base + b + 1);
                d->n_byte = nb;
                d->size[SFQ_S_GEN] = size;
                d->size[SFQ_S_GEN_NS] = xfs[j][0].finish(pw);
                d->size[SFQ_S_GEN_NN] = xfs[j][1].finish(pw);
                d->size[SFQ_S_GEN_LC] = xfs[j][2].finish(pw);
                if (size > cap || xfs[j][0].sink.pos > xfs[j][0].sink.cap || xfs[j][1].sink.pos > xfs[j][1].sink.cap || xfs[j][2].sink.pos > xfs[j][2].sink.cap)
                    atomicMax(&d->status, (u32)(-SFQ_E_OVERFLOW));
                if (cerr | xfs[j][0].rc.err | xfs[j][1].rc.err | xfs[j][2].rc.err) atomicMax(&d->status, (u32)(-SFQ_E_CORRUPT));
                if (bad) atomicMax(&d->status, bad);
            }
            if (h == j) dc.err = 0;
            CSET(g.act, j, 0u);
        }
    }
}

#ifndef GEN_K
#define GEN_K 2
#endif
void launch_gen_encode_k(const ModelArgs& a, u32* ticket, hipStream_t st) {
    // grid: one workgroup per GEN_K table slots (a.nbatch is even; slots a last partial group would take stay unused)
    const u32 groups = a.nbatch / GEN_K ? a.nbatch / GEN_K : 1u;
    hipLaunchKernelGGL(k_gen_encode_k<GEN_K>, dim3(groups), dim3(64), 0, st, a, ticket);
}

// =========================================================================================================
// framing exceptions, a wave per block: UsrSave::get_record's bookkeeping (usrs.cpp:322-375) + update (126-160).
// A record is an exception when its line length differs from the previous record's (usr.x), its quality length from
// its own base length (usr.x.q), or its SOLiD prefix characters from the previous record's (usr.pfg / usr.pfq): all
// neighbour comparisons, so 64 records are tested at once and only the (rare) hits are coded, on lane 0.
// (k_usr_encode_l walks the records one by one on a single lane: 1024 dependent memory round trips per block.)
// =========================================================================================================
__global__ __launch_bounds__(64) void k_usr_encode_w(ModelArgs a) {
    const u32 lane = threadIdx.x, t = blockIdx.x;
    for (u32 b = a.batch0 + t; b < a.batch0 + a.nbatch; b += gridDim.x) {
        BlockDesc* d = &a.blocks[b];
        PwTab pw; pw.slots = a.p_slots + (size_t)t * PR_ROWS * PW_NSYM; pw.hdr = a.p_hdr + (size_t)t * PR_ROWS; pw.epoch = EPOCH_L(a.epoch_base + b + 1);
        XfEnc x_llen, x_qlen, x_sgen, x_sqlt;
        x_llen.init(a.arena + d->out_off[SFQ_S_USR_X],   d->out_cap[SFQ_S_USR_X],   XF_USR_X);
        x_qlen.init(a.arena + d->out_off[SFQ_S_USR_XQ],  d->out_cap[SFQ_S_USR_XQ],  XF_USR_XQ);
        x_sgen.init(a.arena + d->out_off[SFQ_S_USR_PFG], d->out_cap[SFQ_S_USR_PFG], XF_USR_PFG);
        x_sqlt.init(a.arena + d->out_off[SFQ_S_USR_PFQ], d->out_cap[SFQ_S_USR_PFQ], XF_USR_PFQ);
        const u32 solid = d->solid;
        u32 c_llen = d->llen, c_pfg = 0, c_pfq = 0;            // the running values carried from window to window
        u64 i_llen = 0, i_qlen = 0, i_sgen = 0, i_sqlt = 0;
        const u32 two_id = d->two_id;
        u32 odd_plus = 0;                                      // block format: a '+' line the reference would not give back (dev_common.h)
        for (u32 k0 = 0; k0 < d->nrec; k0 += 64) {
            const u32 m = d->nrec - k0 < 64 ? d->nrec - k0 : 64;
            const bool in = lane < m;
            const u64 r = d->rec0 + k0 + (in ? lane : 0);
            const u64 g0 = a.line_off[4 * r + 1], g1 = a.line_off[4 * r + 2] - 1;
            const u64 q0 = a.line_off[4 * r + 3], q1 = a.line_off[4 * r + 4] - 1;
            const u32 sl_len = (u32)(g1 - g0) - solid;
            const u32 ql = (q1 - q0) >= solid ? (u32)(q1 - q0) - solid : 0;
            const u32 cg = solid ? (u32)a.fq[g0] : 0u, cq = solid ? (u32)a.fq[q0] : 0u;
            if (a.lossless && in && !plus_line_is_regular(a.fq, a.line_off, r, two_id)) odd_plus = 1;
            const u32 p_len = wave_shr1(sl_len, c_llen), p_cg = wave_shr1(cg, c_pfg), p_cq = wave_shr1(cq, c_pfq);
            const u64 mL = __ballot(in && sl_len != p_len), mQ = __ballot(in && ql != sl_len);
            const u64 mG = __ballot(in && solid && cg != p_cg), mS = __ballot(in && solid && cq != p_cq);
            u64 mx = mL | mQ | mG | mS;
            while (mx) {
                const u32 bit = (u32)__ffsll((long long)mx) - 1u;
                mx &= mx - 1;
                const u64 rcnt = rec_count_of(a, d->rec0 + k0 + bit, d->rec0);
                if ((mG >> bit) & 1) { const u32 c = rl(cg, bit); if (lane == 0) { x_sgen.put(pw, rcnt - i_sgen); x_sgen.put_chr(pw, c); } i_sgen = rcnt; }   // usrs.cpp:323-327
                if ((mL >> bit) & 1) { const u32 v = rl(sl_len, bit); if (lane == 0) { x_llen.put(pw, rcnt - i_llen); x_llen.put(pw, v); } i_llen = rcnt; }   // usrs.cpp:342-343
                if ((mS >> bit) & 1) { const u32 c = rl(cq, bit); if (lane == 0) { x_sqlt.put(pw, rcnt - i_sqlt); x_sqlt.put_chr(pw, c); } i_sqlt = rcnt; }   // usrs.cpp:356-360
                if ((mQ >> bit) & 1) { const u32 v = rl(ql, bit); if (lane == 0) { x_qlen.put(pw, rcnt - i_qlen); x_qlen.put(pw, v); } i_qlen = rcnt; }     // usrs.cpp:371-372
            }
            c_llen = rl(sl_len, m - 1); c_pfg = rl(cg, m - 1); c_pfq = rl(cq, m - 1);
        }
        if (lane == 0) {
            d->size[SFQ_S_USR_X]   = x_llen.finish(pw);
            d->size[SFQ_S_USR_XQ]  = x_qlen.finish(pw);
            d->size[SFQ_S_USR_PFG] = x_sgen.finish(pw);
            d->size[SFQ_S_USR_PFQ] = x_sqlt.finish(pw);
            if (x_llen.sink.pos > x_llen.sink.cap || x_qlen.sink.pos > x_qlen.sink.cap ||
                x_sgen.sink.pos > x_sgen.sink.cap || x_sqlt.sink.pos > x_sqlt.sink.cap) atomicMax(&d->status, (u32)(-SFQ_E_OVERFLOW));
            if (x_llen.rc.err | x_qlen.rc.err | x_sgen.rc.err | x_sqlt.rc.err) atomicMax(&d->status, (u32)(-SFQ_E_CORRUPT));
        }
        if (odd_plus) atomicMax(&d->status, (u32)(-SFQ_E_UNSUPPORTED));
    }
}
void launch_usr_encode_w(const ModelArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_usr_encode_w, dim3(a.nbatch), dim3(64), 0, st, a);
}
