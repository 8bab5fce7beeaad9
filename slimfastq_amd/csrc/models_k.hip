// models_k.hip -- two-chain kernels: one wavefront codes TWO record blocks at once.
//
// Measured on MI355X (profiles/r01c): these kernels are bound by instruction ISSUE -- about 4.2 cycles per
// wave-instruction per SIMD whatever its type -- and the serial range-coder chain (stage 3 of models_w.hip)
// is the largest share: ~19 instructions per symbol executed by a whole wave for ONE chain, every lane
// computing the same value.  Here a wave owns two blocks (two table slots).  Stages 1-2 (contexts, rows,
// triples) run for a 64-symbol window of block A, then of block B; stage 3 then walks BOTH windows in one
// instruction stream, lanes 0-31 carrying block A's coder state and lanes 32-63 block B's.  The triples go
// through LDS (one ds_read_b128 per step, broadcast within each half) instead of four v_readlane.
// The bytes are the same as models_w.hip / models_l.hip / the reference: only the schedule differs.
#include "kernels.h"
#include "dev_wave.h"

#define NEUTRAL_TRIPLE make_uint4(0u, 1u, 1u, 0xFFFFFFFFu)     // range /= 1, low += 0, range *= 1: a no-op step

// reciprocal for the multiply-high divide: m = floor(2^32 / tot) (estimate at most 1 below); tot = 1 (only the
// neutral step) keeps 2^32 - 1, which the same single fix-up handles
__device__ __forceinline__ u32 recip_exact(u32 tot) {
    const u32 m0 = 0xFFFFFFFFu / tot;
    return (tot != 1 && (0xFFFFFFFFu - m0 * tot) == tot - 1) ? m0 + 1 : m0;
}

// ---- stage 3 for two chains: RCoder (coder.hpp) state per half-wave, all on the vector unit --------------------
struct DualCoder {
    u64 lo;            // RCoder::low   (equal in the 32 lanes of a half)
    u32 vr;            // RCoder::range
    u32 win;           // output window: lane j of a half holds byte (pos & ~31) + j
    u32 pos, cap;      // bytes produced / region size
    u8* outp;
    u32 err;
    __device__ __forceinline__ void reset(bool mine, u8* p, u32 c) {        // start a new stream on one half
        if (mine) { lo = 0; vr = 0xFFFFFFFFu; win = 0; pos = 0; cap = c; outp = p; }   // coder.hpp:34-39
    }
    __device__ __forceinline__ void put(bool pred, u32 byte, u32 l5) {      // FilerSave::put for the lanes with pred
        if (pred && l5 == (pos & 31)) win = byte;
        if (pred) pos++;
        if (pred && (pos & 31) == 0) {
            const u32 at = pos - 32 + l5;
            if (at < cap) outp[at] = (u8)win;
        }
    }
    __device__ __forceinline__ void renorm(u32 l5) {                        // coder.hpp:74-80, per half
        int guard = 0;
#pragma nounroll
        do {
            const bool pred = vr < RC_TOP;
            if (pred && ((lo ^ (lo + vr)) >> 56)) vr = (((u32)lo | (RC_TOP - 1)) - (u32)lo);
            put(pred, (u32)(lo >> 56), l5);
            if (pred) { vr <<= 8; lo <<= 8; }
            if (++guard > 12) { err = 1; if (vr < RC_TOP) vr = 0xFFFFFFFFu; break; }
        } while (__any(vr < RC_TOP));
    }
    // walk nmax steps; trip[h][k] = {cum, freq, tot, recip}; steps past a half's own count are neutral
    __device__ __forceinline__ void run(const uint4 (*trip)[64], u32 nmax, u32 h, u32 l5) {
#pragma nounroll
        for (u32 k = 0; k < nmax; k++) {
            const uint4 t = trip[h][k];
            u32 r = __umulhi(vr, t.w);                                       // r = range / tot (coder.hpp:68)
            const u32 rem = vr - r * t.z;
            r += rem >= t.z ? 1u : 0u;
            lo += (u64)t.x * r;                                              // coder.hpp:69 (cum * r < range: no wrap)
            vr = r * t.y;                                                    // coder.hpp:70
            if (__any(vr < RC_TOP)) renorm(l5);
        }
    }
    __device__ __forceinline__ void done(bool mine, u32 l5) {               // coder.hpp:52-61 + flush of the window
        for (int i = 0; i < 8; i++) { put(mine, (u32)(lo >> 56), l5); if (mine) lo <<= 8; }
        const u32 pend = pos & 31;
        if (mine && l5 < pend) { const u32 at = pos - pend + l5; if (at < cap) outp[at] = (u8)win; }
    }
};

// =========================================================================================================
// base encode, two blocks per wave: GenSave::save_x + normalize_gen (gens.cpp:91-159)
// =========================================================================================================
struct GenCur {                       // one block in flight (everything uniform)
    u32 b, active;
    BlockDesc* d;
    u32* tab;
    u64 rec0; u32 nrec, k;            // record cursor
    u32 base, llen, qlen;             // window cursor inside the record
    const u8* gp; const u8* qp;
    u32 carry, mask, solid;
    u64 genofs, ns_index, nn_index;   // g_genofs_count, m_last.{Ns,Nn}_index (block-relative)
    u32 n_byte; int bad;
};

__device__ __forceinline__ void gen_load_record(const ModelArgs& a, GenCur& c) {
    const u64 r = c.rec0 + c.k;
    const u64 g0 = a.line_off[4 * r + 1] + c.solid, g1 = a.line_off[4 * r + 2] - 1;
    const u64 q0 = a.line_off[4 * r + 3] + c.solid, q1 = a.line_off[4 * r + 4] - 1;
    c.llen = g1 > g0 ? (u32)(g1 - g0) : 0; c.qlen = q1 > q0 ? (u32)(q1 - q0) : 0;
    c.gp = a.fq + g0; c.qp = a.fq + q0;
    c.base = 0;
    c.carry = 0x007616c7u;                                                            // gens.cpp:139
}

// one 64-base window of cursor c -> trip[h][0..63]; returns the number of real steps, advances the cursor
__device__ __forceinline__ u32 gen_window(const ModelArgs& a, GenCur& c, u32 h, uint4 (*trip)[64], XfEnc* xf, const PwTab& pw, u32 lane) {
    const u32 m = c.llen - c.base < 64 ? c.llen - c.base : 64;
    const u32 idx = c.base + lane;
    const bool in = lane < m;
    const u32 gch = in ? c.gp[idx] : 'A';
    const u32 qch = (in && idx < c.qlen) ? c.qp[idx] : 40u;                           // gens.cpp:153
    const u32 n = gencode_w(gch);                                                     // normalize_gen gens.cpp:116-136
    const bool bad_n = in && n == 4, bad_q = in && qch == '!';
    if (__ballot(in && n > 4)) c.bad = SFQ_E_GENCHAR;
    const u32 code = n & 3u;                                                          // N is coded as 0 (A)
    const u64 mN = __ballot(bad_n), mQ = __ballot(bad_q);
    u64 mx = mN | mQ;
    while (mx) {                                                                      // bad_q_or_bad_n gens.cpp:91-114, in order
        const u32 bit = (u32)__ffsll((long long)mx) - 1u;
        mx &= mx - 1;
        const u64 pos = c.genofs + bit + 1;
        const bool is_n = (mN >> bit) & 1, is_q = (mQ >> bit) & 1;
        if (!is_n) {
            if (lane == 0) xf[1].put(pw, pos - c.nn_index);
            c.nn_index = pos;
        } else {
            const u32 ch = rl(gch, bit);
            if (!c.n_byte) c.n_byte = ch;
            if (ch != c.n_byte) c.bad = SFQ_E_GENCHAR;
            if (!is_q) { if (lane == 0) xf[0].put(pw, pos - c.ns_index); c.ns_index = pos; }
        }
    }
    c.genofs += m;
    // contexts: a 32-bit shift register of 2-bit codes; lane k sees the codes of lanes < k, then the carry
    u32 w = wave_shr1(code, 0u);
    w |= wave_shr1(w, 0u) << 2;
    w |= shfl_up0(w, 2, lane) << 4;
    w |= shfl_up0(w, 4, lane) << 8;
    w |= shfl_up0(w, 8, lane) << 16;
    const u32 ctx = ((lane < 16 ? c.carry << (2 * lane) : 0u) | w) & c.mask;
    c.carry = (rl(w, 63) << 2) | rl(code, 63);
    // rows: gather, update, scatter; a context repeated inside the window chains in order
    u32 row = in ? c.tab[ctx] : 0u;
    const u32 key = in ? ((ctx << 6) | lane) : (0x80000000u | (lane << 6) | lane);
    const u32 sk = bitonic_sort64(key, lane);
    const u32 skp = (u32)__shfl_up((int)sk, 1, 64);
    const bool dup = lane > 0 && (sk >> 6) == (skp >> 6);
    u32 cum = 0, freq = 1, tot = 1;
    if (!__ballot(dup)) {
        const u32 nrow = b2_model(row, code, cum, freq, tot);
        if (in) c.tab[ctx] = nrow;
    } else {
        for (u32 j = 0; j < m; j++) {
            const u32 cc = rl(ctx, j), s = rl(code, j);
            u32 cj, fj, tj;
            const u32 nrow = b2_model(c.tab[cc], s, cj, fj, tj);
            if (lane == 0) c.tab[cc] = nrow;
            if (lane == j) { cum = cj; freq = fj; tot = tj; }
        }
    }
    trip[h][lane] = in ? make_uint4(cum, freq, tot, recip_exact(tot)) : NEUTRAL_TRIPLE;
    // advance
    c.base += 64;
    if (c.base >= c.llen) {
        c.k++;
        if (c.k < c.nrec) gen_load_record(a, c);
    }
    return m;
}

__global__ __launch_bounds__(64) void k_gen_encode_k(ModelArgs a, u32* ticket) {
    __shared__ uint4 trip[2][64];
    __shared__ XfEnc xfs[2][2];                        // [half][0 = gen.Ns, 1 = gen.Nn]: side-stream coders, used by lane 0
    const u32 lane = threadIdx.x, h = lane >> 5, l5 = lane & 31;
    DualCoder dc; dc.lo = 0; dc.vr = 0xFFFFFFFFu; dc.win = 0; dc.pos = 0; dc.cap = 0; dc.outp = nullptr; dc.err = 0;
    GenCur cur[2];
    cur[0].active = cur[1].active = 0;
    PwTab pw[2];
    for (u32 hh = 0; hh < 2; hh++) {
        const size_t slot = (size_t)blockIdx.x * 2 + hh;
        pw[hh].slots = a.p_slots + slot * PR_ROWS * PW_NSYM; pw[hh].hdr = a.p_hdr + slot * PR_ROWS; pw[hh].epoch = 0;
        cur[hh].tab = a.g_tab + (slot << a.g_bits);
    }
    bool drained = false;                              // the ticket counter ran out
    for (;;) {
        // (re)fill idle halves with new blocks
#pragma unroll
        for (u32 hh = 0; hh < 2; hh++) {
            GenCur& c = cur[hh];
            if (c.active || drained) continue;
            const u32 b = next_block(ticket);
            if (b >= a.nblocks) { drained = true; continue; }
            BlockDesc* d = &a.blocks[b];
            c.b = b; c.d = d; c.active = 1;
            c.rec0 = d->rec0; c.nrec = d->nrec; c.k = 0;
            c.solid = d->solid; c.mask = (1u << d->gen_bits) - 1u;
            c.genofs = 0; c.ns_index = 0; c.nn_index = 0; c.n_byte = 0; c.bad = 0;
            pw[hh].epoch = EPOCH_L(a.epoch_base + b + 1);
            // Base2Ranger rows start at 3,3,3,3 (base2_ranger.hpp:68-71)
            const u32 n4 = 1u << (a.g_bits - 2);
#pragma unroll 4
            for (u32 i = lane; i < n4; i += 64) reinterpret_cast<uint4*>(c.tab)[i] = make_uint4(B2_INIT, B2_INIT, B2_INIT, B2_INIT);
            if (lane == 0) {
                xfs[hh][0].init(a.arena + d->out_off[SFQ_S_GEN_NS], d->out_cap[SFQ_S_GEN_NS], XF_GEN_NS);
                xfs[hh][1].init(a.arena + d->out_off[SFQ_S_GEN_NN], d->out_cap[SFQ_S_GEN_NN], XF_GEN_NN);
            }
            dc.reset(h == hh, a.arena + d->out_off[SFQ_S_GEN], d->out_cap[SFQ_S_GEN]);
            gen_load_record(a, c);
        }
        if (!(cur[0].active | cur[1].active)) break;
        __syncthreads();                               // one wave per workgroup: orders the LDS traffic across lanes
        // stages 1-2: one window of each active block
        u32 nmax = 0;
#pragma unroll
        for (u32 hh = 0; hh < 2; hh++) {
            if (cur[hh].active) { const u32 n = gen_window(a, cur[hh], hh, trip, xfs[hh], pw[hh], lane); nmax = n > nmax ? n : nmax; }
            else trip[hh][lane] = NEUTRAL_TRIPLE;
        }
        __syncthreads();
        // stage 3: both chains
        dc.run(trip, nmax, h, l5);
        // blocks that ran out of records: flush, publish sizes
#pragma unroll
        for (u32 hh = 0; hh < 2; hh++) {
            GenCur& c = cur[hh];
            if (!c.active || c.k < c.nrec) continue;
            dc.done(h == hh, l5);
            const u32 size = rl(dc.pos, hh * 32), cap = rl(dc.cap, hh * 32), cerr = rl(dc.err, hh * 32);
            if (lane == 0) {
                BlockDesc* d = c.d;
                d->n_byte = c.n_byte;
                d->size[SFQ_S_GEN] = size;
                d->size[SFQ_S_GEN_NS] = xfs[hh][0].finish(pw[hh]);
                d->size[SFQ_S_GEN_NN] = xfs[hh][1].finish(pw[hh]);
                if (size > cap || xfs[hh][0].sink.pos > xfs[hh][0].sink.cap || xfs[hh][1].sink.pos > xfs[hh][1].sink.cap) atomicMax(&d->status, (u32)(-SFQ_E_OVERFLOW));
                if (cerr | xfs[hh][0].rc.err | xfs[hh][1].rc.err) atomicMax(&d->status, (u32)(-SFQ_E_CORRUPT));
                if (c.bad) atomicMax(&d->status, (u32)(-c.bad));
            }
            if (h == hh) dc.err = 0;
            c.active = 0;
        }
    }
}
void launch_gen_encode_k(const ModelArgs& a, u32* ticket, hipStream_t st) {
    // grid: one workgroup per PAIR of table slots
    hipLaunchKernelGGL(k_gen_encode_k, dim3((a.nbatch + 1) / 2), dim3(64), 0, st, a, ticket);
}
