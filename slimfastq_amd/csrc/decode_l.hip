// decode_l.hip -- lane-per-block decode kernels + FASTQ re-assembly.
//
// Decoding is serial per chain by nature (the next context needs the previous symbol), so one lane
// owns one block's chain, as in models_l.hip.  The reference interleaves rec.load / qlt.load /
// gen.load per record (usrs.cpp:555-571); here the chains are independent kernels that write staging
// buffers, and a last streaming kernel lays the 4-line records out (UsrLoad::save, usrs.cpp:512-535).
// Order: usr (line lengths) -> scans -> qlt | gen | rec side by side on three streams -> record sizes -> scan ->
// assemble (which applies the one rule that ties bases to qualities: quality '!' means N, gens.cpp:206-208).
#include <algorithm>
#include "kernels.h"
#include "dev_models.h"
#include "dev_rec_lane.h"

#define LAST_QLT 63u

// Active lanes per decode wavefront.  A wave waits for the slowest of its lanes' table reads, and the chains are
// few (one per block): fewer lanes per wave = more waves, each waiting on fewer reads.
static u32 decode_lanes() { return 16; }      // measured: 64 -> 16 lanes per wave = -18 % decode time

struct DSlot {
    u32 b, epoch;
    u32* q_slots; RowHdr* q_hdr;
    PwTab pw;
    u32* g_tab;
};
__device__ __forceinline__ bool dslot_init(const ModelArgs& a, DSlot& s) {
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
__device__ __forceinline__ void dset_status(BlockDesc* d, int code) { atomicMax(&d->status, (u32)(-code)); }
__device__ __forceinline__ ByteSrc stream_src(const DecodeArgs& a, const BlockDesc* d, u32 b, int s) {
    ByteSrc r; r.init(a.streams + a.blk_stream_off[(u64)b * SFQ_NSTREAMS + s], d->size[s]);
    return r;
}
__device__ __forceinline__ u32 calc_last_delta_d(u32& delta, u32 q, u32 q1, u32 q2) {   // qlts.hpp:62-74
    if (q1 > q) delta += q1 - q;
    u32 d3 = delta >> 3;
    return (q | ((q1 < q2 ? q2 : q1) << 6) | ((u32)(q1 == q2) << 12) | ((d3 < 7 ? d3 : 7) << 13)) & 0xFFFFu;
}

// ---- UsrLoad::update (usrs.cpp:471-510): per-record line lengths and SOLiD prefixes ----------------
// The records of a block that has no framing exceptions at all -- its four streams are empty: every record has the block's line
// length, no prefixes -- get theirs from a thread per RECORD (round 4: k_usr_decode_l below writes a block's 1024 records from one
// lane, four scattered stores each; 0.6 ms at the head of every decode with nothing else running).  Block format only
// (uniform blocks of block_reads records).
__device__ __forceinline__ bool usr_streams_empty(const BlockDesc* d) {
    return (d->size[SFQ_S_USR_X] | d->size[SFQ_S_USR_XQ] | d->size[SFQ_S_USR_PFG] | d->size[SFQ_S_USR_PFQ]) == 0;
}
__global__ __launch_bounds__(256) void k_usr_fill(DecodeArgs a, u64 nrec) {
    const u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrec) return;
    BlockDesc* d = &a.m.blocks[r / a.block_reads];
    if (!usr_streams_empty(d)) return;
    u32 llen = d->llen;
    if (llen > a.max_line) { dset_status(d, SFQ_E_CORRUPT); llen = 0; }
    a.slen[r] = llen; a.qlen[r] = llen; a.pfg[r] = 0; a.pfq[r] = 0;
}
void launch_usr_fill(const DecodeArgs& a, u64 nrec, hipStream_t st) {
    if (nrec) hipLaunchKernelGGL(k_usr_fill, dim3((u32)((nrec + 255) / 256)), dim3(256), 0, st, a, nrec);
}
__global__ __launch_bounds__(64) void k_usr_decode_l(DecodeArgs a, u32 prefilled) {
    DSlot sl;
    if (!dslot_init(a.m, sl)) return;
    BlockDesc* d = &a.m.blocks[sl.b];
    if (prefilled && usr_streams_empty(d)) return;           // (k_usr_fill has written this block's records)
    XfDec x_llen, x_qlen, x_sgen, x_sqlt;
    { ByteSrc s = stream_src(a, d, sl.b, SFQ_S_USR_X);   x_llen.init(s.p, s.n, XF_USR_X); }
    { ByteSrc s = stream_src(a, d, sl.b, SFQ_S_USR_XQ);  x_qlen.init(s.p, s.n, XF_USR_XQ); }
    { ByteSrc s = stream_src(a, d, sl.b, SFQ_S_USR_PFG); x_sgen.init(s.p, s.n, XF_USR_PFG); }
    { ByteSrc s = stream_src(a, d, sl.b, SFQ_S_USR_PFQ); x_sqlt.init(s.p, s.n, XF_USR_PFQ); }
    u64 i_llen = x_llen.get(sl.pw), i_qlen = x_qlen.get(sl.pw), i_sgen = x_sgen.get(sl.pw), i_sqlt = x_sqlt.get(sl.pw);
    const u32 solid = d->solid;
    u32 llen = d->llen, qlen = llen, pfg = 0, pfq = 0, bad = 0;
    for (u32 k = 0; k < d->nrec; k++) {
        const u64 r = d->rec0 + k, rcnt = rec_count_of(a.m, r, d->rec0);
        if (i_llen == rcnt) { llen = (u32)x_llen.get(sl.pw); qlen = llen; i_llen += x_llen.get(sl.pw); }
        if (i_qlen == rcnt) { qlen = (u32)x_qlen.get(sl.pw); i_qlen += x_qlen.get(sl.pw); }
        else if (qlen != llen) qlen = llen;
        if (solid && i_sgen == rcnt) { pfg = x_sgen.get_chr(sl.pw); i_sgen += x_sgen.get(sl.pw); }
        if (solid && i_sqlt == rcnt) { pfq = x_sqlt.get_chr(sl.pw); i_sqlt += x_sqlt.get(sl.pw); }
        if (llen > a.max_line || qlen > a.max_line) { bad = 1; llen = qlen = 0; }
        a.slen[r] = llen; a.qlen[r] = qlen; a.pfg[r] = (u8)pfg; a.pfq[r] = (u8)pfq;
    }
    if (bad | x_llen.rc.err | x_qlen.rc.err | x_sgen.rc.err | x_sqlt.rc.err) dset_status(d, SFQ_E_CORRUPT);
}
void launch_usr_decode_l(const DecodeArgs& a, hipStream_t st, u32 prefilled) {
    hipLaunchKernelGGL(k_usr_decode_l, dim3((a.m.nbatch + 63) / 64), dim3(64), 0, st, a, prefilled);
}

// ---- QltLoad::load_1/2/3 (qlts.cpp:163-234) ------------------------------------------------------------
// Log64Ranger::get (log64_ranger.hpp:114-138) by one lane on a row in the WAVE layout (models_w.hip WaveRow:
// 64 dwords {total, iend | count << 16, epoch tag, pad, slots 0..59} + a 16-byte overflow row): the header and
// the first four slots are one 32-byte read, so a symbol that sits in the front of its row -- the usual case --
// costs one memory round trip.  A stale row starts from the shared prior row (format 7) or from zeros.
__device__ __forceinline__ u32* l64w_slot(u32* row, u32* ovf, u32 i) { return i < 60 ? row + 4 + i : ovf + (i - 60); }
__device__ __forceinline__ u32 l64w_get_lane(u32* row, u32* ovf, u32 epoch, const u32* prow, const u32* povf, RcDec& rc, ByteSrc& src) {
    const uint4 hq = *reinterpret_cast<const uint4*>(row);          // {total, iend | count << 16, epoch, pad}
    uint4 q = *reinterpret_cast<const uint4*>(row + 4);             // slots 0..3, same 64-byte sector
    u32 total, iend, count;
    if (hq.z == epoch) { total = hq.x; iend = hq.y & 0xffffu; count = (hq.y >> 16) & 0xffu; }
    else if (prow) {
        const uint4 ph = *reinterpret_cast<const uint4*>(prow);
        total = ph.x; iend = ph.y & 0xffffu; count = 0;
        const u32 n60 = iend < 60 ? iend : 60;
        for (u32 k = 0; k < n60; k += 4) *reinterpret_cast<uint4*>(row + 4 + k) = *reinterpret_cast<const uint4*>(prow + 4 + k);
        if (iend > 60) *reinterpret_cast<uint4*>(ovf) = *reinterpret_cast<const uint4*>(povf);
        q = *reinterpret_cast<const uint4*>(prow + 4);
    } else { total = 0; iend = 0; count = 0; }
    const u32 prob = rc.get_freq(total + 64);
    u32 i = 0, sumf = 0, s = 0;
    bool found = false;
    while (!found && i < 64) {
        if (i) q = *reinterpret_cast<const uint4*>(i < 60 ? row + 4 + i : ovf);
        u32 e[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (found) break;
            const u32 idx = i + c;
            if (iend == idx) { e[c] = idx << 16; *l64w_slot(row, ovf, idx) = e[c]; iend++; }      // :124-125
            const u32 f1 = (e[c] & 0xffffu) + 1;
            if (sumf + f1 <= prob) sumf += f1; else { s = e[c]; i = idx; found = true; }
        }
        if (!found) i += 4;
    }
    if (!found) { rc.err = 1; i = 63; s = ovf[3]; sumf -= (s & 0xffffu) + 1; }
    rc.decode(src, sumf, (s & 0xffffu) + 1);
    const u32 sym = (s >> 16) & 0xffu;
    // update_freq (log64_ranger.hpp:69-87)
    u32 f = s & 0xffffu;
    bool upd = true;
    if (f > (u32)((1 << 16) - 64 - 6)) {
        if (i == 0 && f + 20u > total) upd = false;
        else {
            u32 tt = 0;
            for (u32 k = 0; k < iend; k++) { u32* p = l64w_slot(row, ovf, k); const u32 nf = (*p & 0xffffu) >> 1; *p = (*p & 0xffff0000u) | nf; tt += nf; }
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
                u32* pp = l64w_slot(row, ovf, i - 1);
                const u32 pv = *pp;
                if (f > (pv & 0xffffu)) { *pp = ns; *l64w_slot(row, ovf, i) = pv; placed = true; }     // down_level :56-67
            }
        }
        if (!placed) *l64w_slot(row, ovf, i) = ns;
    }
    *reinterpret_cast<uint4*>(row) = make_uint4(total, iend | (count << 16), epoch, 0u);
    return sym;
}

__global__ __launch_bounds__(64) void k_qlt_decode_l(DecodeArgs a) {
    DSlot sl;
    if (!dslot_init(a.m, sl)) return;
    BlockDesc* d = &a.m.blocks[sl.b];
    ByteSrc src = stream_src(a, d, sl.b, SFQ_S_QLT);
    RcDec rc; rc.init(src);
    const int level = a.m.level;
    u32* const qo = reinterpret_cast<u32*>(sl.q_hdr);                   // overflow rows of the wave layout
    const u32 epoch_w = EPOCH_W(a.m.epoch_base + sl.b + 1);
    for (u64 r = d->rec0; r < d->rec0 + d->nrec; r++) {
        const u32 n = a.qlen[r];
        u8* p = a.qual_stage + a.qoff[r];
        u32 last = 0, delta = 5, q1 = 0, q2 = 0, di = 0;
        for (u32 i = 0; i < n; i++) {
            u32 b = l64w_get_lane(sl.q_slots + (size_t)last * L64_NSYM, qo + (size_t)last * 4, epoch_w,
                                  a.m.prior_w ? a.m.prior_w + (size_t)last * L64_NSYM : nullptr, a.m.prior_wovf + (size_t)last * 4, rc, src);
            if (b == LAST_QLT) b = sl.pw.get(PR_EXQ_ROW, rc, src);                          // qlts.cpp:168-171
            p[i] = (u8)('!' + b);
            if (level == 1)      last = (b | (last << 6)) & 0xFFFu;
            else if (level == 2) last = (b | (last << 6)) & 0xFFFFu;
            else if (++di & 1) { last = calc_last_delta_d(delta, b, q1, q2); q2 = b; }
            else               { last = calc_last_delta_d(delta, b, q2, q1); q1 = b; }
        }
    }
    if (rc.err) dset_status(d, SFQ_E_CORRUPT);
}
void launch_qlt_decode_l(const DecodeArgs& a, hipStream_t st) {
    const u32 L = decode_lanes();
    hipLaunchKernelGGL(k_qlt_decode_l, dim3((a.m.nbatch + L - 1) / L), dim3(L), 0, st, a);
}

// ---- GenLoad::load_x + normalize_gen (gens.cpp:200-249) ---------------------------------------------------
__global__ __launch_bounds__(64) void k_gen_decode_l(DecodeArgs a) {
    DSlot sl;
    if (!dslot_init(a.m, sl)) return;
    BlockDesc* d = &a.m.blocks[sl.b];
    ByteSrc src = stream_src(a, d, sl.b, SFQ_S_GEN);
    RcDec rc; rc.init(src);
    XfDec x_ns, x_nn, x_lc;
    { ByteSrc s = stream_src(a, d, sl.b, SFQ_S_GEN_NS); x_ns.init(s.p, s.n, XF_GEN_NS); }
    { ByteSrc s = stream_src(a, d, sl.b, SFQ_S_GEN_NN); x_nn.init(s.p, s.n, XF_GEN_NN); }
    { ByteSrc s = stream_src(a, d, sl.b, SFQ_S_GEN_LC); x_lc.init(s.p, s.n, XF_GEN_LC); }
    u64 ns_index = x_ns.get(sl.pw), nn_index = x_nn.get(sl.pw);                             // gens.cpp:187-188
    u64 lc_index = x_lc.get(sl.pw);                                                         // "gen.lc" (block format): absent = 0 = never
    const u32 n_byte = d->n_byte ? d->n_byte : 'N';                                         // gens.cpp:169
    const u32 code = d->solid ? 0x33323130u /* "0123" */ : 0x54474341u /* "ACGT" */;        // gens.cpp:173-178
    const u32 mask = (1u << d->gen_bits) - 1u;
    u64 genofs = 0;
    for (u64 r = d->rec0; r < d->rec0 + d->nrec; r++) {
        const u32 llen = a.slen[r];
        u8* g = a.seq_stage + a.soff[r];
        u32 last = 0x007616c7u & mask;
        u32 row = llen ? sl.g_tab[last] : 0u;
        for (u32 i = 0; i < llen; i++) {
            // the next context is (last << 2) + b: its four candidate rows are one aligned 16-byte piece, fetched
            // while this base is being decoded (the context needs at least 2 bits: gen_bits >= 2)
            const u32 nbase = (last << 2) & mask;
            const uint4 cand = *reinterpret_cast<const uint4*>(sl.g_tab + nbase);
            u32 b;
            const u32 nrow = b2_get(row, rc, src, b);
            sl.g_tab[last] = nrow;
            u32 ch = (code >> (8 * b)) & 0xff;
            const u32 nlast = nbase + b;
            row = b == 0 ? cand.x : b == 1 ? cand.y : b == 2 ? cand.z : cand.w;
            if (nlast == last) row = nrow;                                                  // the row just updated is its own successor
            last = nlast;
            // normalize_gen gens.cpp:200-213: a listed "not N" position keeps its base whatever its quality; an N
            // whose quality is not '!' is listed; every other base under quality '!' is N.  The last rule needs the
            // decoded qualities, so it is applied when the record is assembled (k_assemble): a "not N" base is
            // staged with bit 7 set, and the two decoders run side by side.  (The encoder lists an N only where the
            // quality is not '!', gens.cpp:91-114, so looking the list up first consumes it at the same positions.)
            genofs++;
            if (nn_index == genofs) { nn_index += x_nn.get(sl.pw); ch |= 0x80u; }
            else if (ns_index == genofs) { ch = n_byte; ns_index += x_ns.get(sl.pw); }
            if (lc_index == genofs) { ch |= 0x20u; lc_index += x_lc.get(sl.pw); }               // a lowercase base (k_assemble carries the bit over to an N made by a quality '!')
            g[i] = (u8)ch;
        }
    }
    if (rc.err | x_ns.rc.err | x_nn.rc.err | x_lc.rc.err) dset_status(d, SFQ_E_CORRUPT);
}
void launch_gen_decode_l(const DecodeArgs& a, hipStream_t st) {
    const u32 L = decode_lanes();
    hipLaunchKernelGGL(k_gen_decode_l, dim3((a.m.nbatch + L - 1) / L), dim3(L), 0, st, a);
}

// ---- RecLoad::load (recs.cpp:374-461) ----------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_rec_decode_l(DecodeArgs a) {
    DSlot sl;
    if (!dslot_init(a.m, sl)) return;
    BlockDesc* d = &a.m.blocks[sl.b];
    RecAdaptiveDec cd;
    cd.pw = sl.pw; cd.src = stream_src(a, d, sl.b, SFQ_S_REC); cd.rc.init(cd.src);
    XfDec x_rec;
    { ByteSrc s = stream_src(a, d, sl.b, SFQ_S_REC_X); x_rec.init(s.p, s.n, XF_REC_X); }
    rec_decode_lane(a, d, d->rec0, d->nrec, sl.b, cd, x_rec, sl.pw);
}
void launch_rec_decode_l(const DecodeArgs& a, hipStream_t st) {
    const u32 L = decode_lanes();
    hipLaunchKernelGGL(k_rec_decode_l, dim3((a.m.nbatch + L - 1) / L), dim3(L), 0, st, a);
}

// ---- frozen-table mode: the N rules applied to bases that chains.hip has staged (gens.cpp:200-213) -------------
// gen.Ns lists the N positions whose quality is not '!' (-> the N byte), gen.Nn the real bases under quality '!'
// (-> bit 7, which k_assemble reads as "keep this base"); both as gaps over the block's 1-based base index.
__global__ __launch_bounds__(64) void k_gen_exc_decode_l(DecodeArgs a) {
    DSlot sl;
    if (!dslot_init(a.m, sl)) return;
    BlockDesc* d = &a.m.blocks[sl.b];
    XfDec x_ns, x_nn, x_lc;
    { ByteSrc s = stream_src(a, d, sl.b, SFQ_S_GEN_NS); x_ns.init(s.p, s.n, XF_GEN_NS); }
    { ByteSrc s = stream_src(a, d, sl.b, SFQ_S_GEN_NN); x_nn.init(s.p, s.n, XF_GEN_NN); }
    { ByteSrc s = stream_src(a, d, sl.b, SFQ_S_GEN_LC); x_lc.init(s.p, s.n, XF_GEN_LC); }
    const u32 n_byte = d->n_byte ? d->n_byte : 'N';                                         // gens.cpp:169
    u8* const g = a.seq_stage + a.soff[d->rec0];
    const u64 nb = a.soff[d->rec0 + d->nrec] - a.soff[d->rec0];
    u32 bad = 0;
    for (u64 at = x_ns.get(sl.pw); at; ) {                                                  // gens.cpp:187
        if (at > nb) { bad = 1; break; }
        g[at - 1] = (u8)n_byte;
        const u64 gap = x_ns.get(sl.pw);
        if (!gap) break;
        at += gap;
    }
    for (u64 at = x_nn.get(sl.pw); at; ) {                                                  // gens.cpp:188
        if (at > nb) { bad = 1; break; }
        g[at - 1] |= 0x80u;
        const u64 gap = x_nn.get(sl.pw);
        if (!gap) break;
        at += gap;
    }
    for (u64 at = x_lc.get(sl.pw); at; ) {                                                  // "gen.lc": lowercase bases
        if (at > nb) { bad = 1; break; }
        g[at - 1] |= 0x20u;
        const u64 gap = x_lc.get(sl.pw);
        if (!gap) break;
        at += gap;
    }
    if (bad | x_ns.rc.err | x_nn.rc.err | x_lc.rc.err) dset_status(d, SFQ_E_CORRUPT);
}
void launch_gen_exc_decode_l(const DecodeArgs& a, hipStream_t st) {
    const u32 L = decode_lanes();
    hipLaunchKernelGGL(k_gen_exc_decode_l, dim3((a.m.nbatch + L - 1) / L), dim3(L), 0, st, a);
}

// ---- UsrLoad::save (usrs.cpp:512-535): '@'hdr \n [pf]bases \n '+'[hdr] \n [pf]quals \n --------------------
__global__ __launch_bounds__(256) void k_record_sizes(DecodeArgs a, u64 nrec, u32* rsize) {
    u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrec) return;
    const BlockDesc* d = &a.m.blocks[a.block_reads ? r / a.block_reads : 0];
    const u32 h = a.hlen[r], s = d->solid;
    rsize[r] = 1 + h + 1 + s + a.slen[r] + 1 + 1 + (d->two_id ? h : 0) + 1 + s + a.qlen[r] + 1;
}
void launch_record_sizes(const DecodeArgs& a, u64 nrec, u32* rsize, hipStream_t st) {
    hipLaunchKernelGGL(k_record_sizes, dim3((u32)((nrec + 255) / 256)), dim3(256), 0, st, a, nrec, rsize);
}
// a wave copies n bytes, a dword per lane and step (global loads and stores need no alignment on gfx9), the last 0..3 singly
__device__ __forceinline__ void wave_copy(u8* dst, const u8* src, u32 n, u32 lane) {
    const u32 nd = n >> 2;
    for (u32 i = lane; i < nd; i += 64) *reinterpret_cast<u32*>(dst + 4 * i) = *reinterpret_cast<const u32*>(src + 4 * i);
    if (lane < (n & 3u)) dst[4 * nd + lane] = src[4 * nd + lane];
}
__device__ __forceinline__ u32 merge_n(u32 c, u32 q, u32 nb4) {         // four bases: bit 7 = "keep this base" (gen.Nn), else a quality '!' makes it the N byte
    const u32 keep = ((c >> 7) & 0x01010101u) * 0xFFu;
    const u32 x = q ^ 0x21212121u;
    const u32 bang = ((~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u) >> 7) * 0xFFu;
    const u32 low = c & 0x20202020u & ((c & 0x40404040u) >> 1);          // a lowercase letter ("gen.lc"): the N byte it turns into is lowercase too
    return (c & 0x7F7F7F7Fu & keep) | (~keep & ((bang & (nb4 | low)) | (~bang & c)));
}
// A wave lays out `rpw` consecutive records (64 where there are millions of them, one where they are few and long: the host
// keeps the grid at tens of thousands of waves).  Their bounds are fetched a record per lane (coalesced) and handed round with
// readlane, so a record costs no round trip of its own for them; and a record is five small copies whose first steps (64
// dwords each -- a whole 150-base line) are loaded together, for the NEXT record before the current one is stored: the
// memory round trips of consecutive records overlap.  What is longer than 256 bytes goes through the loops behind.
struct AsmRec { u32 h, sl, ql, s, two, nb, pfg, pfq; const u8* hp; const u8* sp; const u8* qp; u8* o; };
struct AsmFirst { u32 vh, vs, vq, vqq; };
__device__ __forceinline__ u32 bc32(u32 v, u32 k) { return (u32)__builtin_amdgcn_readlane((int)v, (int)k); }
__device__ __forceinline__ u64 bc64(u64 v, u32 k) { return (u64)bc32((u32)v, k) | ((u64)bc32((u32)(v >> 32), k) << 32); }
__device__ __forceinline__ AsmFirst asm_first(const AsmRec& r, u32 lane) {
    const u32 hd = r.h >> 2, md = (r.sl < r.ql ? r.sl : r.ql) >> 2, qd = r.ql >> 2;
    AsmFirst f; f.vh = f.vs = f.vq = f.vqq = 0;
    if (lane < hd) f.vh = *reinterpret_cast<const u32*>(r.hp + 4 * lane);
    if (lane < md) { f.vs = *reinterpret_cast<const u32*>(r.sp + 4 * lane); f.vq = *reinterpret_cast<const u32*>(r.qp + 4 * lane); }
    if (lane < qd) f.vqq = lane < md ? f.vq : *reinterpret_cast<const u32*>(r.qp + 4 * lane);
    return f;
}
__global__ __launch_bounds__(64) void k_assemble(DecodeArgs a, u64 nrec, u32 rpw, const u64* __restrict__ roff, u8* __restrict__ out) {
    const u32 lane = threadIdx.x;
    const u64 r0 = (u64)blockIdx.x * rpw;
    if (r0 >= nrec) return;
    const u32 cnt = (u32)(nrec - r0 < rpw ? nrec - r0 : rpw);
    // the bounds of record r0 + lane
    const u64 rl = r0 + (lane < cnt ? lane : cnt - 1);
    const BlockDesc* d = &a.m.blocks[a.block_reads ? rl / a.block_reads : 0];
    const u32 m_h = a.hlen[rl], m_sl = a.slen[rl], m_ql = a.qlen[rl];
    const u64 m_ho = a.hoff[rl], m_so = a.soff[rl], m_qo = a.qoff[rl], m_ro = roff[rl];
    const u32 m_s = d->solid, m_two = d->two_id, m_nb = d->n_byte ? d->n_byte : 'N';
    const u32 m_pf = m_s ? ((u32)a.pfg[rl] | ((u32)a.pfq[rl] << 8)) : 0u;
    auto rec = [&](u32 k) {
        AsmRec r;
        r.h = bc32(m_h, k); r.sl = bc32(m_sl, k); r.ql = bc32(m_ql, k); r.s = bc32(m_s, k); r.two = bc32(m_two, k); r.nb = bc32(m_nb, k);
        const u32 pf = bc32(m_pf, k); r.pfg = pf & 0xffu; r.pfq = pf >> 8;
        r.hp = a.hdr_stage + bc64(m_ho, k); r.sp = a.seq_stage + bc64(m_so, k); r.qp = a.qual_stage + bc64(m_qo, k);
        r.o = out + bc64(m_ro, k);
        return r;
    };
    AsmRec r = rec(0);
    AsmFirst f = asm_first(r, lane);
    for (u32 k = 0; k < cnt; k++) {
        AsmRec rn = r; AsmFirst fn = f;
        if (k + 1 < cnt) { rn = rec(k + 1); fn = asm_first(rn, lane); }          // the next record's loads, before this one's stores
        const u32 h = r.h, s = r.s, sl = r.sl, ql = r.ql, two = r.two, n_byte = r.nb;
        const u8* __restrict__ hp = r.hp; const u8* __restrict__ sp = r.sp; const u8* __restrict__ qp = r.qp;
        const u32 nb4 = n_byte * 0x01010101u;
        u8* const o_h = r.o + 1;                                     // '@' hdr
        u8* const o_s = o_h + h + 1 + s;                             // '\n' [pf] bases
        u8* const o_2 = o_s + sl + 2;                                // '\n' '+' [hdr]
        u8* const o_q = o_2 + (two ? h : 0) + 1 + s;                 // '\n' [pf] qualities
        const u32 hd = h >> 2, md = (sl < ql ? sl : ql) >> 2, qd = ql >> 2;
        if (lane < hd) { *reinterpret_cast<u32*>(o_h + 4 * lane) = f.vh; if (two) *reinterpret_cast<u32*>(o_2 + 4 * lane) = f.vh; }
        if (lane < md) *reinterpret_cast<u32*>(o_s + 4 * lane) = merge_n(f.vs, f.vq, nb4);
        if (lane < qd) *reinterpret_cast<u32*>(o_q + 4 * lane) = f.vqq;
        // what is longer than 256 bytes
        for (u32 i = lane + 64; i < hd; i += 64) { const u32 v = *reinterpret_cast<const u32*>(hp + 4 * i); *reinterpret_cast<u32*>(o_h + 4 * i) = v; if (two) *reinterpret_cast<u32*>(o_2 + 4 * i) = v; }
        for (u32 i = lane + 64; i < md; i += 64) *reinterpret_cast<u32*>(o_s + 4 * i) = merge_n(*reinterpret_cast<const u32*>(sp + 4 * i), *reinterpret_cast<const u32*>(qp + 4 * i), nb4);
        for (u32 i = lane + 64; i < qd; i += 64) *reinterpret_cast<u32*>(o_q + 4 * i) = *reinterpret_cast<const u32*>(qp + 4 * i);
        // the ends: the last 0..3 bytes of each copy (the bases: whatever lies behind the shorter of the two lines), the fixed characters
        if (lane < (h & 3u)) { const u8 c = hp[4 * hd + lane]; o_h[4 * hd + lane] = c; if (two) o_2[4 * hd + lane] = c; }
        for (u32 i = 4 * md + lane; i < sl; i += 64) {
            const u32 c = sp[i];
            o_s[i] = (u8)((c & 0x80u) ? (c & 0x7fu) : (i < ql && qp[i] == '!') ? (n_byte | (c & 0x20u & ((c & 0x40u) >> 1))) : c);
        }
        if (lane < (ql & 3u)) o_q[4 * qd + lane] = qp[4 * qd + lane];
        if (lane == 0) {
            o_h[-1] = '@';
            o_h[h] = '\n'; if (s) o_h[h + 1] = (u8)r.pfg;
            o_s[sl] = '\n'; o_s[sl + 1] = '+';
            u8* e2 = o_2 + (two ? h : 0);
            e2[0] = '\n'; if (s) e2[1] = (u8)r.pfq;
            o_q[ql] = '\n';
        }
        r = rn; f = fn;
    }
}
// The same for short records, four at a time: sixteen lanes per record, sixteen bytes per lane and step (any alignment), a
// copy's last 1..15 bytes as one more sixteen-byte piece that ends where the copy ends (or singly, where the whole copy is
// shorter than that).  k_assemble above moves a dword per lane and has a 60-byte header keep 15 of its 64 lanes busy: 4.2 ms for
// 3.7 GB at the end of every decode, nothing beside it.
__device__ __forceinline__ uint4 ld16u(const u8* p) { const u32* q = reinterpret_cast<const u32*>(p); return make_uint4(q[0], q[1], q[2], q[3]); }
__device__ __forceinline__ void st16u(u8* p, const uint4& v) { u32* q = reinterpret_cast<u32*>(p); q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w; }
__global__ __launch_bounds__(64) void k_assemble4(DecodeArgs a, u64 nrec, u32 rpw, const u64* __restrict__ roff, u8* __restrict__ out) {
    const u32 lane = threadIdx.x, grp = lane >> 4, sub = lane & 15u;
    const u64 r0 = (u64)blockIdx.x * rpw;
    if (r0 >= nrec) return;
    const u32 cnt = (u32)(nrec - r0 < rpw ? nrec - r0 : rpw);
    const u64 rl = r0 + (lane < cnt ? lane : cnt - 1);                     // the bounds of record r0 + lane, handed to the record's sixteen lanes below
    const BlockDesc* d = &a.m.blocks[a.block_reads ? rl / a.block_reads : 0];
    const u32 m_h = a.hlen[rl], m_sl = a.slen[rl], m_ql = a.qlen[rl];
    const u64 m_ho = a.hoff[rl], m_so = a.soff[rl], m_qo = a.qoff[rl], m_ro = roff[rl];
    const u32 m_s = d->solid, m_two = d->two_id, m_nb = d->n_byte ? d->n_byte : 'N';
    const u32 m_pf = m_s ? ((u32)a.pfg[rl] | ((u32)a.pfq[rl] << 8)) : 0u;
    auto sh32 = [&](u32 v, u32 k) { return (u32)__shfl((int)v, (int)k, 64); };
    auto sh64 = [&](u64 v, u32 k) { return (u64)sh32((u32)v, k) | ((u64)sh32((u32)(v >> 32), k) << 32); };
    for (u32 k0 = 0; k0 < cnt; k0 += 4) {
        const u32 k = k0 + grp;
        const u32 kk = k < cnt ? k : cnt - 1;
        const u32 h = sh32(m_h, kk), sl = sh32(m_sl, kk), ql = sh32(m_ql, kk), s = sh32(m_s, kk), two = sh32(m_two, kk), n_byte = sh32(m_nb, kk), pf = sh32(m_pf, kk);
        const u8* __restrict__ hp = a.hdr_stage + sh64(m_ho, kk);
        const u8* __restrict__ sp = a.seq_stage + sh64(m_so, kk);
        const u8* __restrict__ qp = a.qual_stage + sh64(m_qo, kk);
        u8* const o = out + sh64(m_ro, kk);
        if (k >= cnt) continue;
        const u32 nb4 = n_byte * 0x01010101u;
        u8* const o_h = o + 1;                                       // '@' hdr
        u8* const o_s = o_h + h + 1 + s;                             // '\n' [pf] bases
        u8* const o_2 = o_s + sl + 2;                                // '\n' '+' [hdr]
        u8* const o_q = o_2 + (two ? h : 0) + 1 + s;                 // '\n' [pf] qualities
        // header (once or twice)
        for (u32 i = sub; i < (h >> 4); i += 16) { const uint4 v = ld16u(hp + 16 * i); st16u(o_h + 16 * i, v); if (two) st16u(o_2 + 16 * i, v); }
        if (h & 15u) {
            if (h >= 16u) { if (sub == 15u) { const uint4 v = ld16u(hp + h - 16); st16u(o_h + h - 16, v); if (two) st16u(o_2 + h - 16, v); } }
            else if (sub < h) { const u8 c = hp[sub]; o_h[sub] = c; if (two) o_2[sub] = c; }
        }
        // bases, with the qualities' '!' -> N (gens.cpp:91-114 on the way back) over the part both lines have
        const u32 md = sl < ql ? sl : ql;
        auto merged = [&](u32 at) {
            const uint4 c = ld16u(sp + at), q = ld16u(qp + at);
            return make_uint4(merge_n(c.x, q.x, nb4), merge_n(c.y, q.y, nb4), merge_n(c.z, q.z, nb4), merge_n(c.w, q.w, nb4));
        };
        for (u32 i = sub; i < (md >> 4); i += 16) st16u(o_s + 16 * i, merged(16 * i));
        if (md & 15u) {
            if (md >= 16u) { if (sub == 14u) st16u(o_s + md - 16, merged(md - 16)); }
            else if (sub < md) { const u32 c = sp[sub]; o_s[sub] = (u8)((c & 0x80u) ? (c & 0x7fu) : (qp[sub] == '!') ? (n_byte | (c & 0x20u & ((c & 0x40u) >> 1))) : c); }
        }
        for (u32 i = md + sub; i < sl; i += 16) { const u32 c = sp[i]; o_s[i] = (u8)((c & 0x80u) ? (c & 0x7fu) : c); }       // (a base line longer than its quality line)
        // qualities
        for (u32 i = sub; i < (ql >> 4); i += 16) st16u(o_q + 16 * i, ld16u(qp + 16 * i));
        if (ql & 15u) {
            if (ql >= 16u) { if (sub == 13u) st16u(o_q + ql - 16, ld16u(qp + ql - 16)); }
            else if (sub < ql) o_q[sub] = qp[sub];
        }
        if (sub == 0) {
            o_h[-1] = '@';
            o_h[h] = '\n'; if (s) o_h[h + 1] = (u8)(pf & 0xffu);
            o_s[sl] = '\n'; o_s[sl + 1] = '+';
            u8* e2 = o_2 + (two ? h : 0);
            e2[0] = '\n'; if (s) e2[1] = (u8)(pf >> 8);
            o_q[ql] = '\n';
        }
    }
}
// format 6 with oversize records: the size of every record of the file in file order -- a kept record's from rsize (through
// rec_map), an oversize one's from its four raw lines and its '@'
__global__ __launch_bounds__(256) void k_over_sizes(const u32* __restrict__ rec_map, const u32* __restrict__ rsize, u64 n_kept, const u64* __restrict__ no,
                                                    const u64* __restrict__ piece, u32 n_over, u32* __restrict__ size_all) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < n_kept) size_all[rec_map[i]] = rsize[i];
    else if (i - n_kept < n_over) {
        const u64 j = i - n_kept;
        size_all[no[j] - 1] = (u32)(1 + piece[(j * 4 + 0) * 2 + 1] + piece[(j * 4 + 1) * 2 + 1] + piece[(j * 4 + 2) * 2 + 1] + piece[(j * 4 + 3) * 2 + 1]);
    }
}
void launch_over_sizes(const u32* rec_map, const u32* rsize, u64 n_kept, const u64* no, const u64* piece, u32 n_over, u32* size_all, hipStream_t st) {
    const u64 n = n_kept + n_over;
    if (n) hipLaunchKernelGGL(k_over_sizes, dim3((u32)((n + 255) / 256)), dim3(256), 0, st, rec_map, rsize, n_kept, no, piece, n_over, size_all);
}
__global__ __launch_bounds__(256) void k_gather_u64(const u64* __restrict__ src, const u32* __restrict__ idx, u64 n, u64* __restrict__ dst) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
void launch_gather_u64(const u64* src, const u32* idx, u64 n, u64* dst, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_gather_u64, dim3((u32)((n + 255) / 256)), dim3(256), 0, st, src, idx, n, dst);
}
void launch_assemble(const DecodeArgs& a, u64 nrec, const u64* roff, u8* out, hipStream_t st) {
    const u32 rpw = (u32)std::max<u64>(1, std::min<u64>(64, nrec / 32768));
    if (rpw >= 4) hipLaunchKernelGGL(k_assemble4, dim3((u32)((nrec + rpw - 1) / rpw)), dim3(64), 0, st, a, nrec, rpw, roff, out);      // many records: short ones
    else hipLaunchKernelGGL(k_assemble, dim3((u32)((nrec + rpw - 1) / rpw)), dim3(64), 0, st, a, nrec, rpw, roff, out);
}
