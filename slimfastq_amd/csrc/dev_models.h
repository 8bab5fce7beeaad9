// dev_models.h -- the adaptive frequency tables ("rangers") as lane-serial device code:
//   Base2Ranger  (base2_ranger.hpp)   4 symbols, 4 x u8 counts in one dword
//   Log64Ranger  (log64_ranger.hpp)   64 symbols, STEP 6
//   PowerRanger / PowerRangerU (power_ranger.hpp) 256 symbols, STEP 14, varint wrapper
// Rows live in HBM as dword slots (freq | sym << 16) + a RowHdr; see dev_common.h.
#pragma once
#include "dev_coder.h"

// ---------------------------------------------------------------------------------------------------
// Base2Ranger.  Row = freq[0..3] in the four bytes of a dword, initial 0x03030303 (base2_ranger.hpp:68-71).
// ---------------------------------------------------------------------------------------------------
#define B2_INIT 0x03030303u

__device__ __forceinline__ u32 b2_update(u32 v, u32 sym) {       // base2_ranger.hpp:60-66, normalize 48-53
    u32 f = (v >> (8 * sym)) & 0xff;
    if (f > 254) v = ((v & ~0x01010101u) >> 1) | (v & 0x01010101u);
    return v + (1u << (8 * sym));
}
__device__ __forceinline__ u32 b2_put(u32 v, RcEnc& rc, ByteSink& s, u32 sym) {   // base2_ranger.hpp:74-84
    u32 f0 = v & 0xff, f1 = (v >> 8) & 0xff, f2 = (v >> 16) & 0xff, f3 = v >> 24;
    u32 total = (f0 + f1) + (f2 + f3);
    u32 cum = sym == 0 ? 0u : sym == 1 ? f0 : sym == 2 ? f0 + f1 : f0 + f1 + f2;
    u32 f = (v >> (8 * sym)) & 0xff;
    rc.encode(s, cum, f, total);
    return b2_update(v, sym);
}
template <typename SRC>
__device__ __forceinline__ u32 b2_get(u32 v, RcDec& rc, SRC& s, u32& sym_out) {   // base2_ranger.hpp:86-104
    u32 f0 = v & 0xff, f1 = (v >> 8) & 0xff, f2 = (v >> 16) & 0xff, f3 = v >> 24;
    u32 total = (f0 + f1) + (f2 + f3);
    u32 prob = rc.get_freq(total);
    u32 sym, cum, f;
    if (f0 > prob)                { sym = 0; cum = 0;            f = f0; }
    else if (f0 + f1 > prob)      { sym = 1; cum = f0;           f = f1; }
    else if (f0 + f1 + f2 > prob) { sym = 2; cum = f0 + f1;      f = f2; }
    else                          { sym = 3; cum = f0 + f1 + f2; f = f3; if (prob >= total) rc.err = 1; }  // reference: assert(i<4)
    rc.decode(s, cum, f);
    sym_out = sym;
    return b2_update(v, sym);
}

// ---------------------------------------------------------------------------------------------------
// Log64Ranger / PowerRanger share one scheme (log64_ranger.hpp:51-138, power_ranger.hpp:49-131);
// they differ in NSYM / STEP / MAX_FREQ and in the saturation slack (20 vs 256).
// ---------------------------------------------------------------------------------------------------
struct Triple { u32 cum, freq, tot; };    // RCoder::Encode(cumFreq, freq, totFreq) coder.hpp:66

template <int NSYM, int STEP, int MAXF, int SAT>
struct Ranger {
    // open a row: a stale epoch means the all-zero state of a fresh table
    static __device__ __forceinline__ void open(const RowHdr* hp, u32 epoch, u32& total, u32& iend, u32& count) {
        RowHdr h = *hp;
        if (h.epoch == epoch) { total = h.total; iend = h.iend; count = h.count; }
        else { total = 0; iend = 0; count = 0; }
    }
    static __device__ __forceinline__ void close(RowHdr* hp, u32 epoch, u32 total, u32 iend, u32 count) {
        RowHdr h; h.total = total; h.iend = (u16)iend; h.count = (u8)count; h.pad = 0; h.epoch = epoch; h.pad2 = 0;
        *hp = h;
    }
    // update_freq (log64_ranger.hpp:69-87 / power_ranger.hpp:66-84); s = slot i as loaded.
    static __device__ __forceinline__ void update(u32* slots, u32 i, u32 s, u32& total, u32 iend, u32& count) {
        u32 f = s & 0xffff;
        if (f > (u32)(MAXF - STEP)) {
            if (i == 0 && f + (u32)SAT > total) return;
            u32 t = 0;                                       // normalize
            for (u32 k = 0; k < iend; k++) {
                u32 v = slots[k];
                u32 nf = (v & 0xffff) >> 1;
                slots[k] = (v & 0xffff0000u) | nf;
                t += nf;
            }
            total = t;
            f >>= 1;
        }
        f += STEP;
        total += STEP;
        u32 ns = (s & 0xffff0000u) | f;
        if (i != 0) {
            count = (count + 1) & 0xff;
            if ((count & 0xf) == 0) {
                u32 pv = slots[i - 1];
                if (f > (pv & 0xffff)) {                     // down_level
                    slots[i - 1] = ns;
                    slots[i] = pv;
                    return;
                }
            }
        }
        slots[i] = ns;
    }
    // the model half of put (log64_ranger.hpp:98-107,111 / power_ranger.hpp:91-100,103): find the symbol's
    // slot, return what RCoder::Encode needs, update the row.
    static __device__ __forceinline__ Triple model(u32* slots, RowHdr* hp, u32 epoch, u32 sym, u32& err) {
        u32 total, iend, count;
        open(hp, epoch, total, iend, count);
        if (iend <= sym) { for (u32 k = iend; k <= sym; k++) slots[k] = k << 16; iend = sym + 1; }
        u32 i = 0, sumf = 0, s;
        for (;;) {
            s = slots[i];
            if ((s >> 16) == sym) break;
            sumf += s & 0xffff;
            if (++i >= (u32)NSYM) { err = 1; i = NSYM - 1; s = slots[i]; break; }   // unreachable on sane tables
        }
        Triple t; t.cum = sumf + i; t.freq = (s & 0xffff) + 1; t.tot = total + NSYM;
        update(slots, i, s, total, iend, count);
        close(hp, epoch, total, iend, count);
        return t;
    }
    // put (log64_ranger.hpp:98-112 / power_ranger.hpp:91-104)
    static __device__ __forceinline__ void put(u32* slots, RowHdr* hp, u32 epoch, RcEnc& rc, ByteSink& snk, u32 sym) {
        Triple t = model(slots, hp, epoch, sym, rc.err);
        rc.encode(snk, t.cum, t.freq, t.tot);
    }
    // get (log64_ranger.hpp:114-138 / power_ranger.hpp:106-130)
    static __device__ __forceinline__ u32 get(u32* slots, RowHdr* hp, u32 epoch, RcDec& rc, ByteSrc& src) {
        u32 total, iend, count;
        open(hp, epoch, total, iend, count);
        u32 vtot = total + NSYM;
        u32 prob = rc.get_freq(vtot);
        u32 i, sumf = 0, s = 0;
        for (i = 0; i < (u32)NSYM; i++) {
            if (iend == i) { slots[i] = i << 16; iend++; }
            s = slots[i];
            u32 f1 = (s & 0xffff) + 1;
            if (sumf + f1 <= prob) sumf += f1; else break;
        }
        if (i >= (u32)NSYM) { rc.err = 1; i = NSYM - 1; sumf -= (s & 0xffff) + 1; }
        rc.decode(src, sumf, (s & 0xffff) + 1);
        u32 sym = s >> 16;
        update(slots, i, s, total, iend, count);
        close(hp, epoch, total, iend, count);
        return sym & 0xff;
    }
};

typedef Ranger<64, 6, (1 << 16) - 64, 20>    Log64;   // log64_ranger.hpp:37-42, 72

// Format-7 warm start: the first touch of a quality row in a block copies the shared prior row
// (prior.hip) into the block's private row instead of leaving it all-zero.
__device__ __forceinline__ void l64_touch(u32* slots, RowHdr* hp, u32 epoch, const u32* pslots, const RowHdr* php) {
    if (!pslots || hp->epoch == epoch) return;
    RowHdr h = *php;
    for (u32 k = 0; k < h.iend; k++) slots[k] = pslots[k];
    h.epoch = epoch;
    *hp = h;
}
typedef Ranger<256, 14, (1 << 15) - 32, 256> Power;   // power_ranger.hpp:37-41, 70

// Log64Ranger::get (log64_ranger.hpp:114-138) by one lane with 16-byte accesses; a stale row starts from the
// shared prior row (format 7) or from zeros.
__device__ __forceinline__ u32 l64_get_lane(u32* slots, RowHdr* hp, u32 epoch, const u32* pslots, const RowHdr* php, RcDec& rc, ByteSrc& src) {
    const uint4 hq = *reinterpret_cast<const uint4*>(hp);          // {total, iend | count<<16, epoch, pad}
    u32 total, iend, count;
    if (hq.z == epoch) { total = hq.x; iend = hq.y & 0xffffu; count = (hq.y >> 16) & 0xffu; }
    else if (pslots) {
        const uint4 ph = *reinterpret_cast<const uint4*>(php);
        total = ph.x; iend = ph.y & 0xffffu; count = 0;
        for (u32 k = 0; k < iend; k += 4) *reinterpret_cast<uint4*>(slots + k) = *reinterpret_cast<const uint4*>(pslots + k);
    } else { total = 0; iend = 0; count = 0; }
    const u32 vtot = total + 64;
    const u32 prob = rc.get_freq(vtot);
    u32 i = 0, sumf = 0, s = 0;
    bool found = false;
    while (!found && i < 64) {
        const uint4 q = *reinterpret_cast<const uint4*>(slots + i);
        u32 e[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (found) break;
            const u32 idx = i + c;
            if (iend == idx) { e[c] = idx << 16; slots[idx] = e[c]; iend++; }          // :124-125
            const u32 f1 = (e[c] & 0xffffu) + 1;
            if (sumf + f1 <= prob) sumf += f1; else { s = e[c]; i = idx; found = true; }
        }
        if (!found) i += 4;
    }
    if (!found) { rc.err = 1; i = 63; s = slots[63]; sumf -= (s & 0xffffu) + 1; }
    rc.decode(src, sumf, (s & 0xffffu) + 1);
    const u32 sym = (s >> 16) & 0xffu;
    Log64::update(slots, i, s, total, iend, count);
    uint4 nh; nh.x = total; nh.y = iend | (count << 16); nh.z = epoch; nh.w = 0;
    *reinterpret_cast<uint4*>(hp) = nh;
    return sym;
}

// A block slot's PowerRanger rows.
struct PwTab {
    u32*    slots;   // [PR_ROWS][256]
    RowHdr* hdr;     // [PR_ROWS]
    u32     epoch;
    __device__ __forceinline__ void put(u32 row, RcEnc& rc, ByteSink& s, u32 sym) const {
        Power::put(slots + (size_t)row * PW_NSYM, hdr + row, epoch, rc, s, sym);
    }
    __device__ __forceinline__ u32 get(u32 row, RcDec& rc, ByteSrc& s) const {
        return Power::get(slots + (size_t)row * PW_NSYM, hdr + row, epoch, rc, s);
    }
    // PowerRangerU::put_u (power_ranger.hpp:138-163); rows row0 .. row0+13.  One loop, one inlined put: the
    // byte sequence is 1 byte (<= 0x7f), 2 bytes (< 0x7ffe), 0xff 0xfe + 4 LE bytes, or 0xff 0xff + 8 LE bytes.
    __device__ void put_u(u32 row0, RcEnc& rc, ByteSink& s, u64 num) const {
        const u32 n = num <= 0x7f ? 1u : num < 0x7ffe ? 2u : num < (1ULL << 32) ? 6u : 10u;
#pragma nounroll
        for (u32 j = 0; j < n; j++) {
            u32 row, sym;
            if (j == 0)      { row = row0;     sym = n == 1 ? (u32)num : n == 2 ? (0xff & (0x80 | (u32)(num >> 8))) : 0xffu; }
            else if (j == 1) { row = row0 + 1; sym = n == 2 ? (0xff & (u32)num) : n == 6 ? 0xfeu : 0xffu; }
            else             { row = row0 + (n == 6 ? 2 : 6) + (j - 2); sym = 0xff & (u32)(num >> (8 * (j - 2))); }
            put(row, rc, s, sym);
        }
    }
    // PowerRangerU::get_u (power_ranger.hpp:165-190)
    __device__ u64 get_u(u32 row0, RcDec& rc, ByteSrc& s) const {
        u64 num = 0; u32 n = 1;                                     // (one loop around one get, as put_u above: code size)
#pragma nounroll
        for (u32 j = 0; j < n; j++) {
            const u32 row = j < 2 ? row0 + j : row0 + (n == 6 ? 2u : 6u) + (j - 2);
            const u32 b = get(row, rc, s);
            if (j == 0) { num = b; n = b > 0x7f ? 2u : 1u; }
            else if (j == 1) {
                num = (num << 8) | b;
                if (num < 0xfffe) num &= 0x7fff;
                else { n = num == 0xfffe ? 6u : 10u; num = 0; }
            } else num |= (u64)b << (8 * (j - 2));
        }
        return num;
    }
};

// XFileSave (xfile.cpp:40-74): a lazily created side stream with its own coder; rows row0..row0+14.
struct XfEnc {
    RcEnc rc;
    ByteSink sink;
    u32 row0;
    u32 opened;
    __device__ __forceinline__ void init(u8* p, u32 cap, u32 xf) {
        sink.p = p; sink.pos = 0; sink.cap = cap; row0 = PR_XF_BASE + xf * PR_XF_ROWS; opened = 0; rc.init();
    }
    __device__ __forceinline__ void put(const PwTab& t, u64 gap) { opened = 1; t.put_u(row0, rc, sink, gap); }        // xfile.cpp:66-69
    __device__ __forceinline__ void put_chr(const PwTab& t, u32 c) { opened = 1; t.put(row0 + 14, rc, sink, c); }     // xfile.cpp:71-74
    __device__ __forceinline__ void put_str(const PwTab& t, const u8* p, u32 len) {                                  // xfile.cpp:95-99
        put(t, len);
        for (u32 j = 0; j < len; j++) t.put(row0 + 14, rc, sink, p[j]);
    }
    // ~XFileSave: terminator + flush, only if the stream exists (xfile.cpp:40-47). Returns its size.
    __device__ __forceinline__ u32 finish(const PwTab& t) {
        if (!opened) return 0;
        put(t, 0);
        rc.done(sink);
        return sink.pos;
    }
};

// XFileLoad (xfile.cpp:76-106): an absent stream reads as 0 forever.
struct XfDec {
    RcDec rc;
    ByteSrc src;
    u32 row0;
    u32 valid;
    __device__ __forceinline__ void init(const u8* p, u32 n, u32 xf) {
        src.init(p, n); row0 = PR_XF_BASE + xf * PR_XF_ROWS; valid = n > 0;
        if (valid) rc.init(src); else { rc.low = rc.code = 0; rc.range = 0xFFFFFFFFu; rc.err = 0; }
    }
    __device__ __forceinline__ u64 get(const PwTab& t) { return valid ? t.get_u(row0, rc, src) : 0; }
    __device__ __forceinline__ u32 get_chr(const PwTab& t) { return valid ? t.get(row0 + 14, rc, src) : 0; }
};
