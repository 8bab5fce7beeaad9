// dev_walk.h -- what every chain kernel shares: the chain geometry (which records, or which segment of a record, a chain holds) and a
// lane's walk through its text sixteen bytes at a time.  (chains.hip, gm.hip)
#pragma once
#include "kernels.h"
#include "dev_chain.h"

// ---- chain geometry ---------------------------------------------------------------------------------------------
// A chain is chain_reads whole records of a block -- or, where records are LONG (tens of kilobases: a call of 60 k reads has 60 k
// records and a lane that walks one of 50 kb alone takes as long as the rest of the call), a SEGMENT of one record: a record of
// M = max(bases, qualities) symbols is cut into n = ceil(M / seg_len) segments of ceil(M / n) symbols each; segment s of the
// quality line and segment s of the base line are chain seg_off[r] + s of their streams.  A segment starts as a line does: the
// quality context at its initial state (qlts.cpp:109-112), the base context at the seed (gens.cpp:139).
struct ChainPos { u32 b; u64 r0; u32 nrec; u64 sub_lo, sub_len; u32 seg, nseg; };
__device__ __forceinline__ ChainPos chain_pos(const ChainArgs& a, u32 c) {
    ChainPos p;
    p.sub_lo = 0; p.sub_len = 0; p.seg = 0; p.nseg = 1;
    if (a.seg_len) {
        const u64 r = a.seg_rec[c];
        p.b = (u32)(r / a.block_reads); p.r0 = r; p.nrec = 1;
        p.seg = (u32)(c - a.seg_off[r]); p.nseg = (u32)(a.seg_off[r + 1] - a.seg_off[r]);
        return p;                                          // (sub_lo / sub_len: seg_range, once the record's M is at hand)
    }
    p.b = c / a.geo.cpb;
    const u32 j = c - p.b * a.geo.cpb;
    const BlockDesc* d = &a.m.blocks[p.b];
    const u32 k0 = j * a.geo.chain_reads;
    p.nrec = k0 < d->nrec ? (d->nrec - k0 < a.geo.chain_reads ? d->nrec - k0 : a.geo.chain_reads) : 0u;
    p.r0 = d->rec0 + k0;
    return p;
}
__device__ __forceinline__ u32 seg_count_of(u64 M, u32 seg_len) { const u64 n = (M + seg_len - 1) / seg_len; return n ? (u32)n : 1u; }
// the symbols [sub_lo, sub_lo + sub_len) of a line that segment p.seg of p.nseg covers, M = max(bases, qualities) of the record
__device__ __forceinline__ void seg_range(ChainPos& p, u64 M) {
    const u64 L = (M + p.nseg - 1) / p.nseg;
    p.sub_len = L ? L : 1;
    p.sub_lo = (u64)p.seg * p.sub_len;
}
// M of record r from the text's line index (the lines as the walkers see them: without a SOLiD prefix character)
__device__ __forceinline__ u64 rec_symbols(const u64* line_off, u64 r, u32 solid) {
    const u64 g0 = line_off[4 * r + 1] + solid, g1 = line_off[4 * r + 2] - 1, q0 = line_off[4 * r + 3] + solid, q1 = line_off[4 * r + 4] - 1;
    const u64 gl = g1 > g0 ? g1 - g0 : 0, ql = q1 > q0 ? q1 - q0 : 0;
    return gl > ql ? gl : ql;
}
__device__ __forceinline__ void chain_seg_encode(const ChainArgs& a, ChainPos& p) {        // (encode: the segment's range from the text)
    if (a.seg_len) seg_range(p, rec_symbols(a.m.line_off, p.r0, a.m.blocks[p.b].solid));
}
// the first chain of block b
__device__ __forceinline__ u64 block_chain0(const ChainArgs& a, const ChainGeoArgs& geo, u32 b) {
    return a.seg_len ? a.seg_off[a.m.blocks[b].rec0] : (u64)b * geo.cpb;
}
__device__ __forceinline__ u64 block_chain1(const ChainArgs& a, const ChainGeoArgs& geo, u32 b) {
    if (a.seg_len) return a.seg_off[a.m.blocks[b].rec0 + a.m.blocks[b].nrec];
    const u64 e = (u64)(b + 1) * geo.cpb;
    return e < geo.nchains ? e : geo.nchains;
}
// a chain's output region inside its block's region of the scratch arena: proportional to the text before it (a segment: its
// record's text in equal parts)
__device__ __forceinline__ u8* chain_region(const ChainArgs& a, const ChainPos& p, int stream, u32 num, u32 den, u32& cap) {
    const BlockDesc* d = &a.m.blocks[p.b];
    const u64 t0 = a.m.line_off[4 * d->rec0];
    u64 tc = a.m.line_off[4 * p.r0], te = a.m.line_off[4 * (p.r0 + p.nrec)];
    if (a.seg_len) { const u64 len = te - tc, base = tc; tc = base + len * p.seg / p.nseg; te = base + len * (p.seg + 1) / p.nseg; }
    const u64 lo = ((tc - t0) * num / den + 3) & ~3ull, hi = ((te - t0) * num / den) & ~3ull;        // (dword-aligned: LaneEncB::drain stores 16 bytes at such addresses)
    cap = hi > lo ? (u32)(hi - lo) : 0u;
    return a.m.arena + d->out_off[stream] + lo;
}
// 16 text bytes of a lane: aligned loads, nothing read outside [fq, fq_end)
__device__ __forceinline__ uint4 load16(const u8* fq, u64 nbytes, u64 at) {          // any alignment (global loads need none on gfx9)
    const u8* p = fq + at;
    if (at + 16 <= nbytes) {
        const u32* q = reinterpret_cast<const u32*>(p);               // (a non-temporal load here: the call 0.7 ms slower, round 4)
        return make_uint4(q[0], q[1], q[2], q[3]);
    }
    u32 w[4] = {0, 0, 0, 0};
    for (u32 i = 0; i < 16; i++) if (at + i < nbytes) w[i >> 2] |= (u32)p[i] << ((i & 3) * 8);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// ---- a lane's text, piece by piece -----------------------------------------------------------------------------------
// A chain's symbols are the LINE-th lines (1 = bases, 3 = qualities) of records [r0, r0 + nrec), minus a SOLiD prefix
// character.  The lane takes them sixteen bytes at a time; next() describes the coming piece from
// the line bounds alone, so the caller can have the piece after the one it is working on in flight, and the bounds of
// a line are themselves fetched a record ahead: no memory round trip sits on the lane's critical path.
struct Piece { u64 at; u32 j0, j1, rk; bool valid, newline; };  // bytes [j0, j1) of the 16 at offset `at` are the lane's, of the walk's record rk; newline: they start a line
struct LineWalk {
    const u64* line_off; const u64* st_off; const u32* st_len;   // FASTQ text (line_off) or the decoder's staged lines (st_off / st_len)
    const u8* buf; u64 nbytes;
    u64 r0; u32 nrec, k, line, solid;
    u64 pos, end, npos, nend;
    u64 sub_lo, sub_len;         // a part of the line only: bytes [sub_lo, sub_lo + sub_len) of it (sub_len = 0: all of it)
    bool fresh;
    __device__ __forceinline__ void bounds(u32 kk, u64& b0, u64& b1) const {
        const u64 r = r0 + kk;
        if (st_off) { b0 = st_off[r]; b1 = b0 + st_len[r]; }
        else { b0 = line_off[4 * r + line] + solid; b1 = line_off[4 * r + line + 1] - 1; }
        if (sub_len) {
            if (b1 < b0) b1 = b0;
            const u64 lo = b0 + sub_lo;
            b0 = lo < b1 ? lo : b1;
            b1 = b0 + sub_len < b1 ? b0 + sub_len : b1;
        }
    }
    __device__ __forceinline__ void init(const ChainArgs& a, u64 r0_, u32 nrec_, u32 line_, u32 solid_, u64 sub_lo_ = 0, u64 sub_len_ = 0) {
        sub_lo = sub_lo_; sub_len = sub_len_;
        line_off = a.m.line_off; st_off = a.st_off; st_len = a.st_len;
        buf = st_off ? a.st_buf : a.m.fq; nbytes = st_off ? a.st_bytes : a.nbytes;
        r0 = r0_; nrec = nrec_; line = line_; solid = solid_; k = 0; pos = end = 0; npos = nend = 0; fresh = false;
        if (nrec) bounds(0, npos, nend);
    }
    __device__ __forceinline__ Piece next() {
        while (pos >= end && k < nrec) {
            pos = npos; end = nend; k++; fresh = true;
            if (k < nrec) bounds(k, npos, nend);                   // used a whole line later
            if (end < pos) end = pos;
        }
        // line-relative pieces: the next (up to) sixteen bytes of the line, wherever they lie in memory -- only a line's
        // LAST piece is short, so a model may run its state over all sixteen positions without masks (what the state
        // becomes behind a line's last symbol does not matter: the next piece starts a line)
        Piece p; p.valid = pos < end; p.newline = false; p.at = 0; p.j0 = 0; p.j1 = 0; p.rk = k - 1;
        if (p.valid) {
            p.at = pos;
            p.j1 = (u32)((end - pos) < 16 ? (end - pos) : 16);
            p.newline = fresh; fresh = false;
            pos += p.j1;
        }
        return p;
    }
    __device__ __forceinline__ uint4 fetch(const Piece& p) const {
        if (!p.valid) return make_uint4(0, 0, 0, 0);
        return load16(buf, nbytes, p.at);
    }
};
__device__ __forceinline__ u32 piece_byte(const uint4& w, u32 j) {          // j is a compile-time constant where this is used
    const u32 word = j < 4 ? w.x : j < 8 ? w.y : j < 12 ? w.z : w.w;
    return (word >> ((j & 3) * 8)) & 0xffu;
}

