// dev_chain.h -- lane-per-chain coding with frozen tables (block format 7, sfq_params.tables = SFQ_TABLES_FROZEN).
//
// The adaptive kernels (models_w.hip / models_k.hip) spend a whole wavefront on one or two serial chains and
// read-modify-write a private table row per symbol.  Here the learning is taken out of the per-symbol loop:
// the frequency tables are built by counting passes (a transmitted sample for qualities and headers, the
// earlier "generations" of the same file for bases), frozen while a chain is coded, and shared read-only by
// every chain.  With nothing private but the range coder's state, ONE LANE carries a chain: 64 independent
// chains per wavefront, no cross-lane work, table lookups from LDS / L2.  The arithmetic per symbol is still
// RCoder::Encode / GetFreq / Decode (coder.hpp:66-102) on (cum, freq, tot) triples in the rangers' form
// (freq + 1 over total + NSYM); only WHEN the counts are taken differs.  oracle/sfq_oracle.c restates the rule.
//
// Chain stream framing (ours, not the reference's): RCoder's first four output bytes are always zero
// (low < 2^32 until four renormalisations have happened) and are not stored; the flush writes the five
// significant bytes of the smallest multiple of 2^24 that is >= low (it lies inside [low, low + range)), less
// those of the five that are trailing zeros -- a decoder reads zeros past the end, as FilerLoad::get does (filer.hpp:94-97).
#pragma once
#include "dev_coder.h"

// frozen row entry: cum | freq << 16 (both < 65536); the total of a row is 2^16
#define FZ_CUM(e)  ((e) & 0xFFFFu)
#define FZ_FREQ(e) ((e) >> 16)
#define FZ_MAX_TOT 65535u

__device__ __forceinline__ u32 fz_recip(u32 tot) {      // floor(2^32 / tot) for tot >= 2; tot == 1 keeps 2^32 - 1
    const u32 m0 = 0xFFFFFFFFu / tot;
    return (tot != 1 && (0xFFFFFFFFu - m0 * tot) == tot - 1) ? m0 + 1 : m0;
}

struct LaneEnc {
    u64 low; u32 range;
    u32 n;          // bytes produced so far, the four elided ones included
    u32 acc;        // the last up-to-4 bytes, oldest in the low byte once four are in
    u8* out; u32 cap;
    u32 err;
    __device__ __forceinline__ void init(u8* p, u32 c) { low = 0; range = 0xFFFFFFFFu; n = 0; acc = 0; out = p; cap = c; err = 0; }
    // byte k of the coder lands at out[k - 4]: the first four bytes are always zero (low < 2^56 until four bytes have
    // left) and are dropped with the first dword
    __device__ __forceinline__ void put(u32 byte) {
        acc = __builtin_amdgcn_alignbit(byte, acc, 8);             // (acc >> 8) | (byte << 24)
        n++;
        if ((n & 3u) == 0 && n > 4 && n - 4 <= cap) *reinterpret_cast<u32*>(out + n - 8) = acc;
    }
    __device__ __forceinline__ void renorm() {                              // coder.hpp:74-80
        int guard = 0;
#pragma nounroll
        while (range < RC_TOP) {
            if ((low ^ (low + range)) >> 56) range = (((u32)low | (RC_TOP - 1)) - (u32)low);
            put((u32)(low >> 56));
            range <<= 8; low <<= 8;
            if (++guard > 12) { err = 1; range = 0xFFFFFFFFu; break; }    // the reference would spin; every chain must drain
        }
    }
    // coder.hpp:66-73 with the divide as a multiply-high by recip = floor(2^32 / tot) plus one exact fix-up
    __device__ __forceinline__ void encode(u32 cum, u32 freq, u32 tot, u32 recip) {
        u32 r = __umulhi(range, recip);
        r += (range - r * tot) >= tot ? 1u : 0u;
        low += (u64)cum * r;                                 // cum * r < range: no 32-bit wrap (coder.hpp:69)
        range = r * freq;
        renorm();
    }
    // the same with tot = 2^16: range / tot is a shift
    __device__ __forceinline__ void encode16(u32 cum, u32 freq) {
        const u32 r = range >> 16;
        low += (u64)cum * r;
        range = r * freq;
        renorm();
    }
    // flush; returns the stream's size (the flush's own trailing zero bytes are dropped)
    __device__ __forceinline__ u32 finish() {
        const u64 v = (low + 0xFFFFFFull) & ~0xFFFFFFull;
        if (n < 4 && (v >> (32 + 8 * n))) err = 1;            // cannot happen: the elided bytes are zero (v < 2^(32 + 8 n))
        const u32 top5_lo = (u32)(v >> 24);                  // flush bytes 1..4 (byte 0 = v >> 56)
        u32 tz = 0;                                          // trailing zero bytes among the five
        if (top5_lo == 0) tz = (v >> 56) ? 4u : 5u; else tz = ((u32)__builtin_ctz(top5_lo)) >> 3;
        u64 t = v;
        for (int i = 0; i < 5; i++) { put((u32)(t >> 56)); t <<= 8; }
        const u32 stored = n - 4;                            // n >= 5
        const u32 pend = stored & 3u;                        // bytes still in acc (its top `pend` bytes)
        for (u32 i = 0; i < pend; i++) { const u32 at = stored - pend + i; if (at < cap) out[at] = (u8)(acc >> (8 * (4 - pend + i))); }
        if (stored > cap) err |= 2;
        return stored - (tz < stored ? tz : stored);
    }
};

// The same coder without branches in the per-symbol path: a renormalisation step is executed by every lane and takes
// effect where the lane needs it (selects, a shift by 0 or 8), its output byte goes to a per-lane ring in LDS (or to a
// dummy slot), and the ring is drained to the chain's region 16 bytes at a time, once per piece of text.  A lone
// wavefront pays ~20 cycles for every exec-mask branch (compare -> SALU -> branch -> VALU); with two or three
// wavefronts per SIMD -- all the chains of a call give -- nothing hides that, so the branches were most of the time.
// R = ring dwords per lane (a power of two, at most 8... any power of two); a lane's ring is 4 R contiguous bytes at a
// stride of 4 R + 4 bytes -- an odd number of dwords, so the lanes of a wavefront spread over the banks.  The byte
// stream is exactly LaneEnc's.
template <int THREADS, int R>
struct LaneEncB {
    static constexpr u32 RB = 4u * (u32)R;                 // ring bytes
    static constexpr u32 LDS_DWORDS = (u32)THREADS * ((u32)R + 1u);
    u64 low; u32 range;
    u32 q;          // ring position of the next byte: bytes produced + 12 (the first stored byte sits at q = 16)
    u32 dq;         // ring position drained so far (a multiple of 16)
    u8* ring;       // LDS: this lane's ring
    u8* out; u32 cap;
    u32 err;
    __device__ __forceinline__ void init(u32* lds_ring, u32 tid, u8* p, u32 c) {
        low = 0; range = 0xFFFFFFFFu; q = 12; dq = 16; ring = reinterpret_cast<u8*>(lds_ring) + tid * (RB + 4u); out = p; cap = c; err = 0;
    }
    // (masks, not selects: the compiler turns a select between two computed values back into a branch)
    // the byte always goes to the ring's next free position (nothing live is there); it counts where nm is all ones
    __device__ __forceinline__ void put_if(u32 nm, u32 byte) {
        ring[q & (RB - 1u)] = (u8)byte;
        q -= nm;                                                               // + 1 where nm = -1
    }
    __device__ __forceinline__ void step() {                                  // one round of coder.hpp:74-80, where range < TOP
        const u32 nm = range < RC_TOP ? ~0u : 0u;
        const u32 lo = (u32)low, hi = (u32)(low >> 32);
        // coder.hpp:76-77: [low, low + range) crosses a multiple of 2^56 -- with range < 2^24 only where bits 24..55 of low
        // are all ones, once in 2^32 renormalisations: a cheap necessary test for the whole wavefront, the fix behind it
        if (__any((hi | 0xFF000000u) == 0xFFFFFFFFu)) {
            const u32 thi = (u32)((low + range) >> 32);
            const u32 sm = ((thi ^ hi) >> 24) ? ~0u : 0u;
            const u32 alt = ~lo & (RC_TOP - 1);                                // (lo | (TOP - 1)) - lo
            range ^= (range ^ alt) & (nm & sm);
        }
        put_if(nm, hi >> 24);
        const u32 sh = 8u & nm;
        range <<= sh; low <<= sh;
    }
    __device__ __forceinline__ void renorm() {
        step();
        int guard = 0;
#pragma nounroll
        while (__any(range < RC_TOP)) {                                       // rare: a symbol of probability < 2^-8
            step();
            if (++guard > 12) { err = 1; range = 0xFFFFFFFFu; break; }
        }
    }
    // the arithmetic of encode() for the lanes whose mask vm is all ones; the others keep their state
    __device__ __forceinline__ void encode_if(u32 vm, u32 cum, u32 freq, u32 tot, u32 recip) {
        u32 r = __umulhi(range, recip);
        r += (range - r * tot) >= tot ? 1u : 0u;
        low += (u64)(cum & vm) * r;
        range ^= (range ^ (r * freq)) & vm;
        renorm();
    }
    __device__ __forceinline__ void encode16_if(u32 vm, u32 cum, u32 freq) {
        const u32 r = range >> 16;
        low += (u64)(cum & vm) * r;
        range ^= (range ^ (r * freq)) & vm;
        renorm();
    }
    __device__ __forceinline__ void encode_bits_if(u32 vm, u32 cum, u32 freq, u32 bits) {       // a row that totals 2^bits
        const u32 r = range >> bits;
        low += (u64)(cum & vm) * r;
        range ^= (range ^ (r * freq)) & vm;
        renorm();
    }
    __device__ __forceinline__ void encode(u32 cum, u32 freq, u32 tot, u32 recip) {
        u32 r = __umulhi(range, recip);
        r += (range - r * tot) >= tot ? 1u : 0u;
        low += (u64)cum * r;
        range = r * freq;
        renorm();
    }
    __device__ __forceinline__ void encode16(u32 cum, u32 freq) {
        const u32 r = range >> 16;
        low += (u64)cum * r;
        range = r * freq;
        renorm();
    }
    // 16-byte rows of the ring that are complete go to the chain's region (the ring holds 4 R bytes and at most 15 stay
    // behind: between two calls the coder may add 4 R - 15)
    __device__ __forceinline__ void drain() {
        if (!__any(q >= dq + 16u)) return;                                    // (one scalar branch for the wavefront)
        while (q >= dq + 16u) {                                                // (q starts below dq: the four elided bytes)
            const u32* r32 = reinterpret_cast<const u32*>(ring + (dq & (RB - 1u)));      // (RB is a multiple of 16: the row does not wrap)
            uint4 v;
            v.x = r32[0]; v.y = r32[1]; v.z = r32[2]; v.w = r32[3];
            const u32 at = dq - 16u;
            if (at + 16u <= cap) *reinterpret_cast<uint4*>(out + at) = v; else err |= 2;
            dq += 16u;
        }
    }
    // The ring and its drains as a plain byte sink (bases packed two bits each, chains.hip: no coder): no elided bytes -- the first byte put is the stream's first
    __device__ __forceinline__ void init_raw(u32* lds_ring, u32 tid, u8* p, u32 c) { init(lds_ring, tid, p, c); q = 16; }
    __device__ __forceinline__ u32 finish_raw() {
        drain();
        for (u32 pos = dq; pos < q; pos++) { const u32 at = pos - 16u; if (at < cap) out[at] = ring[pos & (RB - 1u)]; else err |= 2; }
        return q - 16u;
    }
    // flush; returns the stream's size (the flush's own trailing zero bytes are dropped)
    __device__ __forceinline__ u32 finish() {
        const u64 v = (low + 0xFFFFFFull) & ~0xFFFFFFull;
        const u32 n = q - 12u;
        if (n < 4 && (v >> (32 + 8 * n))) err = 1;            // cannot happen: the elided bytes are zero
        const u32 top5_lo = (u32)(v >> 24);
        u32 tz = 0;
        if (top5_lo == 0) tz = (v >> 56) ? 4u : 5u; else tz = ((u32)__builtin_ctz(top5_lo)) >> 3;
        u64 t = v;
        for (int i = 0; i < 5; i++) { put_if(~0u, (u32)(t >> 56)); t <<= 8; }
        drain();
        for (u32 pos = dq; pos < q; pos++) { const u32 at = pos - 16u; if (at < cap) out[at] = ring[pos & (RB - 1u)]; else err |= 2; }
        const u32 stored = q - 16u;                          // q >= 17
        return stored - (tz < stored ? tz : stored);
    }
};

// A lane's decoded bytes on their way to memory: a byte store per symbol makes every symbol a partial-line write (the
// decoders were bound by those: two of them side by side ran at a third of their speed alone); here sixteen bytes gather
// in registers and leave as one store (any alignment: the line starts where the record does).
// (Round 4: 205 k lanes each keep a partly written 128-byte line in L2 -- with every lane writing into one small region instead the
//  quality decoder alone takes 8.9 ms instead of 9.8.  Non-temporal stores do not help: the decode goes from 20.9 to 22.5 ms.)
struct LaneOut {
    u8* p; u32 n, acc; u32 w0, w1, w2;
    __device__ __forceinline__ void begin(u8* dst) { p = dst; n = 0; acc = 0; w0 = w1 = w2 = 0; }
    __device__ __forceinline__ void put(u32 byte) {
        acc = __builtin_amdgcn_alignbit(byte, acc, 8);             // (acc >> 8) | (byte << 24)
        n++;
        if ((n & 3u) == 0) {                                       // a dword is full: the first three of a row of sixteen bytes wait
            const u32 q = (n >> 2) & 3u;
#ifdef SFQ_EXP_OUT_LOCAL
            if (q == 0) *reinterpret_cast<uint4*>(p + ((n - 16) & 48u)) = make_uint4(w0, w1, w2, acc);
#else
            if (q == 0) *reinterpret_cast<uint4*>(p + n - 16) = make_uint4(w0, w1, w2, acc);
#endif
            w0 = q == 1 ? acc : w0; w1 = q == 2 ? acc : w1; w2 = q == 3 ? acc : w2;
        }
    }
    __device__ __forceinline__ void end() {                       // what is left: up to three dwords, up to three bytes
        const u32 full = (n >> 2) & 3u, base = n & ~15u;
        if (full > 0) *reinterpret_cast<u32*>(p + base) = w0;
        if (full > 1) *reinterpret_cast<u32*>(p + base + 4) = w1;
        if (full > 2) *reinterpret_cast<u32*>(p + base + 8) = w2;
        const u32 pend = n & 3u;
        for (u32 i = 0; i < pend; i++) p[n - pend + i] = (u8)(acc >> (8 * (4 - pend + i)));
    }
};

// The same, thirty-two bytes -- a whole sector of HBM -- a time, for lines that START on a sector (the quality decoder's: api.cpp pads their places).
// LaneOut's sixteen-byte stores at any alignment dirty a sector twice or three times, a symbol-step apart -- 60 us --, and a quarter of a million half
// written sectors do not wait in L2 that long: the quality decoder wrote 4.6 GB for 1.5 GB of text (profiles/r05z_pmc_summary.txt).  Two stores back to back
// fill a sector at once.
struct LaneOut32 {
    u8* p; u32 n, acc; u32 w0, w1, w2, w3, w4, w5, w6;
    __device__ __forceinline__ void begin(u8* dst) { p = dst; n = 0; acc = 0; w0 = w1 = w2 = w3 = w4 = w5 = w6 = 0; }
    __device__ __forceinline__ void put(u32 byte) {
        acc = __builtin_amdgcn_alignbit(byte, acc, 8);             // (acc >> 8) | (byte << 24)
        n++;
        if ((n & 3u) == 0) {                                       // a dword is full: the first seven of a row of thirty-two bytes wait
            const u32 q = (n >> 2) & 7u;
            if (q == 0) {
                *reinterpret_cast<uint4*>(p + n - 32) = make_uint4(w0, w1, w2, w3);
                *reinterpret_cast<uint4*>(p + n - 16) = make_uint4(w4, w5, w6, acc);
            }
            w0 = q == 1 ? acc : w0; w1 = q == 2 ? acc : w1; w2 = q == 3 ? acc : w2; w3 = q == 4 ? acc : w3;
            w4 = q == 5 ? acc : w4; w5 = q == 6 ? acc : w5; w6 = q == 7 ? acc : w6;
        }
    }
    __device__ __forceinline__ void end() {                       // what is left: up to seven dwords, up to three bytes
        const u32 full = (n >> 2) & 7u, base = n & ~31u;
        if (full > 0) *reinterpret_cast<u32*>(p + base) = w0;
        if (full > 1) *reinterpret_cast<u32*>(p + base + 4) = w1;
        if (full > 2) *reinterpret_cast<u32*>(p + base + 8) = w2;
        if (full > 3) *reinterpret_cast<u32*>(p + base + 12) = w3;
        if (full > 4) *reinterpret_cast<u32*>(p + base + 16) = w4;
        if (full > 5) *reinterpret_cast<u32*>(p + base + 20) = w5;
        if (full > 6) *reinterpret_cast<u32*>(p + base + 24) = w6;
        const u32 pend = n & 3u;
        for (u32 i = 0; i < pend; i++) p[n - pend + i] = (u8)(acc >> (8 * (4 - pend + i)));
    }
};

struct LaneDec {
    u64 low, code; u32 range;
    const u8* p; u32 pos, n;
    u64 cur, a0, a1;    // stream bytes fetched ahead: `have` of them in cur (next one in its low byte), then a0 and -- while `ahead` is 2 -- a1
    u32 have, ahead, fpos;      // fpos = stream position of the first byte not fetched yet
    u32 err;
    // stream bytes from position at, zeros past the end (FilerLoad::get returns 0 there, filer.hpp:94-97); nothing
    // outside [p, p + n) is touched
    __device__ __forceinline__ u64 fetch8(u32 at) const {
        if (at + 8 <= n) { const u32* q = reinterpret_cast<const u32*>(p + at); return (u64)q[0] | ((u64)q[1] << 32); }   // (no alignment needed on gfx9)
        u64 v = 0;
        for (u32 i = 0; i < 8; i++) if (at + i < n) v |= (u64)p[at + i] << (8 * i);
        return v;
    }
    __device__ __forceinline__ void fetch16(u32 at) {                  // sixteen bytes a load: a lane's refills are what the decoders' vector memory path is busy with
        if (at + 16 <= n) {
            const uint4 v = *reinterpret_cast<const uint4*>(p + at);
            a0 = (u64)v.x | ((u64)v.y << 32); a1 = (u64)v.z | ((u64)v.w << 32);
        } else { a0 = fetch8(at); a1 = fetch8(at + 8); }
    }
    __device__ __forceinline__ u32 get() {
        if (have == 0) {
            cur = a0; a0 = a1; have = 8;
            if (--ahead == 0) { fetch16(fpos); fpos += 16; ahead = 2; }     // the load issued here is used eight bytes later
        }
        const u32 b = (u32)cur & 0xffu;
        cur >>= 8; have--; pos++;
        return b;
    }
    __device__ __forceinline__ void init(const u8* ptr, u32 len) {
        p = ptr; pos = 0; n = len; low = 0; range = 0xFFFFFFFFu; err = 0;
        cur = fetch8(0); fetch16(8); have = 8; ahead = 2; fpos = 24;
        code = 0;
        for (int i = 0; i < 4; i++) code = (code << 8) | get();          // the four elided zero bytes, then four real ones
    }
    // coder.hpp:83-86
    __device__ __forceinline__ u32 get_freq(u32 tot, u32 recip) {
        u32 r = __umulhi(range, recip);
        r += (range - r * tot) >= tot ? 1u : 0u;
        if (r == 0) { err = 1; r = 1; }
        range = r;
        if (code >> 32) { err = 1; return 0; }
        return (u32)code / r;
    }
    __device__ __forceinline__ u32 get_freq16() {                     // tot = 2^16
        const u32 r = range >> 16;
        range = r;
        if (code >> 32) { err = 1; return 0; }
        const u32 q = (u32)code / r;
        if (q > 0xFFFFu) { err = 1; return 0xFFFFu; }
        return q;
    }
    // coder.hpp:88-102
    __device__ __forceinline__ void decode(u32 cum, u32 freq) {
        const u32 temp = cum * range;
        low += temp; code -= temp;
        range *= freq;
        int guard = 0;
        // (not unrolled: the guard bounds the trip count, and thirteen copies of the refill -- with its end-of-stream byte loads -- per
        //  decoded symbol site are what the compiler makes of that)
#pragma nounroll
        while (range < RC_TOP) {
            if ((low ^ (low + range)) >> 56) range = (((u32)low | (RC_TOP - 1)) - (u32)low);
            code = (code << 8) | get();
            range <<= 8; low <<= 8;
            if (++guard > 12) { err = 1; range = 0xFFFFFFFFu; break; }
        }
    }
};

// The quality / base decoders' lean coder (round 4; chains.hip k_qlt_decode_c has the account of what a symbol costs)
// floor(code / r), 256 <= r < 65536: the estimate is within one of the quotient (relative error of the conversion, the
// reciprocal and the product: 2^-22; quotient < 2^24), the remainder says which way
__device__ __forceinline__ u32 div_exact(u32 code, u32 r) {
    u32 q = (u32)((float)code * __builtin_amdgcn_rcpf((float)r));
    const i32 rem = (i32)(code - q * r);                    // (modulo 2^32: |code - q r| < 2 r)
    q -= rem < 0 ? 1u : 0u;
    q += rem >= (i32)r ? 1u : 0u;
    return q;
}
struct LaneDecQ {
    u64 low; u32 code, range;
    u64 cur; u32 nxt, nsh;      // stream bytes ahead: `have` of them in cur (the next one in its low byte), then the dword nxt >> nsh
    u32 have, fpos;             // fpos: the stream position behind nxt's bytes
    const u8* p; u32 n, nsafe; u32 err;
    // Four stream bytes from position at, zeros past the end (FilerLoad::get returns 0 there, filer.hpp:94-97): no branch and ONE
    // load that nothing touches until the bytes are wanted -- the address is held inside the stream (nsafe = n - 4), and what
    // then lies before `at` is shifted out (by sh bits) where the dword is USED.  (Two paths -- a dword where it fits, bytes at the
    // tail -- or the shift next to the load made the wave wait for the load where it was issued: a round trip to L2 on every symbol.)
    __device__ __forceinline__ void load4(u32 at, u32& raw, u32& sh) const {
        const u32 at_c = at < nsafe ? at : nsafe;
        const u32 ov = at - at_c;
        raw = *reinterpret_cast<const u32*>(p + at_c);                               // (no alignment needed on gfx9)
        sh = 8u * (ov < 4u ? ov : 4u);
    }
    static __device__ __forceinline__ u32 take(u32 raw, u32 sh) { return (u32)((u64)raw >> sh); }
    __device__ __forceinline__ void init(const u8* ptr, u32 len, const u8* spare /* four readable bytes somewhere */) {
        p = ptr; n = len; err = 0; low = 0; range = 0xFFFFFFFFu;
        u32 w0, w1;
        if (len >= 4) {
            nsafe = len - 4;
            u32 r0, s0, r1, s1;
            load4(0, r0, s0); load4(4, r1, s1); load4(8, nxt, nsh);
            w0 = take(r0, s0); w1 = take(r1, s1);
        } else {                                             // a stream of under four bytes: taken whole, nothing of it is loaded later
            w0 = 0;
            for (u32 i = 0; i < len; i++) w0 |= (u32)ptr[i] << (8 * i);
            w1 = 0; p = spare; n = 0; nsafe = 0; nxt = 0; nsh = 32;        // (load4(at >= 4) of this: zeros)
        }
        code = __builtin_bswap32(w0);                        // the four elided zero bytes, then four real ones (dev_chain.h)
        cur = (u64)w1; have = 4; fpos = 12;
    }
    // at least four bytes in cur: once per coded symbol, ahead of its (at most two, see renorm) masked steps
    __device__ __forceinline__ void top_up() {
        if (__any(have <= 4u)) {
            if (have <= 4u) {
                cur |= (u64)take(nxt, nsh) << (8u * have);
                have += 4u;
                load4(fpos, nxt, nsh); fpos += 4u;
            }
        }
    }
    __device__ __forceinline__ void step() {                 // one round of coder.hpp:93-100, where range < TOP
        const u32 nm = range < RC_TOP ? ~0u : 0u;
        const u32 lo = (u32)low, hi = (u32)(low >> 32);
        // coder.hpp:94-95: [low, low + range) crosses a multiple of 2^56 -- with range < 2^24 only where bits 24..55 of low are all
        // ones: a cheap necessary test for the whole wavefront, the exact one behind it
        if (__any((hi | 0xFF000000u) == 0xFFFFFFFFu)) {
            const u32 thi = (u32)((low + range) >> 32);
            const u32 sm = ((thi ^ hi) >> 24) ? ~0u : 0u;
            range ^= (range ^ (~lo & (RC_TOP - 1))) & (nm & sm);
        }
        const u32 sh = 8u & nm;
        code = (code << sh) | ((u32)cur & 0xffu & nm);
        range <<= sh; low <<= sh; cur >>= sh;
        have -= nm & 1u;
    }
    __device__ __forceinline__ void renorm() {
        step(); step();
        int guard = 0;
#pragma nounroll
        while (__any(range < RC_TOP)) {                      // rare: a symbol of probability below 2^-16
            top_up();
            step();
            if (++guard > 12) { err = 1; range = 0xFFFFFFFFu; break; }
        }
    }
    // coder.hpp:83-86 for a row that totals 2^16; r = range >> 16 stays in `range`'s place until decode()
    __device__ __forceinline__ u32 get_freq16(u32& r) {
        r = range >> 16;
        const u32 q = div_exact(code, r);
        if (q > 0xFFFFu) err = 1;                            // (code >= range: not a stream an encoder wrote)
        return q;
    }
    // the same for a row that totals 2^bits (bits <= 16)
    __device__ __forceinline__ u32 get_freq_bits(u32 bits, u32& r) {
        r = range >> bits;
        const u32 q = div_exact(code, r);
        if (q >> bits) err = 1;
        return q & ((1u << bits) - 1u);
    }
    // coder.hpp:83-86 for any total: range / tot as a multiply-high by recip = floor(2^32 / tot) plus one exact fix-up (dev_chain.h)
    __device__ __forceinline__ u32 get_freq(u32 tot, u32 recip, u32& r) {
        r = __umulhi(range, recip);
        r += (range - r * tot) >= tot ? 1u : 0u;
        const u32 q = div_exact(code, r);                    // (quotient < tot + 1: far inside div_exact's reach; r >= 2^24 / 1020)
        if (q >= tot) err = 1;
        return q;
    }
    // coder.hpp:88-102
    __device__ __forceinline__ void decode(u32 r, u32 cum, u32 freq) {
        const u32 temp = cum * r;
        low += temp; code -= temp;
        range = r * freq;
        renorm();
    }
    // the same where one step nearly always does (a base of a flat row: a quarter of the range, a byte every four bases)
    __device__ __forceinline__ void decode1(u32 r, u32 cum, u32 freq) {
        const u32 temp = cum * r;
        low += temp; code -= temp;
        range = r * freq;
        step();
        int guard = 0;
#pragma nounroll
        while (__any(range < RC_TOP)) {
            top_up();
            step();
            if (++guard > 12) { err = 1; range = 0xFFFFFFFFu; break; }
        }
    }
};

// Chain geometry: a block of block_reads records is cut into chains of chain_reads records; chain c of the call is
// chain c % cpb of block c / cpb (cpb = chains per block).  Every stream that is coded in chains (qlt, gen) has one
// range-coder stream per chain; a block's bytes of that stream are its chains' streams back to back (kernels.h ChainArgs).
