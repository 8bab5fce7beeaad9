// dev_wavepw.h -- wave-cooperative PowerRanger rows and XFile coders (power_ranger.hpp, xfile.cpp): every lane of a
// wavefront runs the same (uniform) range coder, the row's 256 slots lie four per lane -- found by ballot, summed by a DPP
// scan -- instead of a 256-step walk of dependent reads.  Shared by models_w.hip (encoders, exception passes) and
// decode_w.hip (the wave-per-block decoders).
#pragma once
#include "kernels.h"
#include "dev_models.h"
#include "dev_wave.h"

struct Sink0 {                 // uniform cursor; lane 0 stores
    u8* p; u32 pos, cap;
    __device__ __forceinline__ void put(u32 b) { if (threadIdx.x == 0 && pos < cap) p[pos] = (u8)b; pos++; }
};
struct RcEncU {                // RCoder (coder.hpp) on uniform values, one symbol at a time
    u64 low; u32 range, err;
    __device__ __forceinline__ void init() { low = 0; range = 0xFFFFFFFFu; err = 0; }
    __device__ __forceinline__ void encode(Sink0& s, u32 cum, u32 freq, u32 tot) {      // coder.hpp:66-81
        const u32 r = range / tot;
        low += (u64)(u32)(cum * r);
        range = r * freq;
        int guard = 0;
#pragma nounroll
        while (range < RC_TOP) {
            if ((low ^ (low + range)) >> 56) range = (((u32)low | (RC_TOP - 1)) - (u32)low);
            s.put((u32)(low >> 56));
            range <<= 8; low <<= 8;
            if (++guard > 12) { err = 1; range = 0xFFFFFFFFu; break; }
        }
    }
    __device__ __forceinline__ void done(Sink0& s) { for (int i = 0; i < 8; i++) { s.put((u32)(low >> 56)); low <<= 8; } }
};

// A block slot's PowerRanger rows with lane l holding slots 4l..4l+3 (power_ranger.hpp:36-131).
struct WavePw {
    u32* slots; RowHdr* hdr; u32 epoch;
    // rows [hrow0, hrow0 + hn) live in the wave's LDS instead (k_gen_exc_w: the two rows nearly every gap goes through --
    // a gap's row is read, updated and written back, and the next gap reads it again: round trips through L2 otherwise)
    u32* hslots = nullptr; RowHdr* hhdr = nullptr; u32 hrow0 = 0, hn = 0;
    // The header's accesses name their address space: through one generic pointer that is either, they are flat loads / stores, which
    // take the vector memory path even when the row is in LDS.  (The slots stay behind a selected generic pointer: the same split for
    // them -- a uniform branch around a ds_ and a global_ access inside the lanes' own conditions -- coded different bytes on the GPU
    // and was taken out again, unexplained.)
    typedef u32 v4 __attribute__((ext_vector_type(4)));
    __device__ __forceinline__ u32* row_slots(u32 row) const { const u32 k = row - hrow0; return k < hn ? hslots + (size_t)k * PW_NSYM : slots + (size_t)row * PW_NSYM; }
    __device__ __forceinline__ uint4 ld_slots(u32 row, u32 i0) const { return *reinterpret_cast<const uint4*>(row_slots(row) + i0); }
    __device__ __forceinline__ void st_slots(u32 row, u32 i0, const uint4& v) const { *reinterpret_cast<uint4*>(row_slots(row) + i0) = v; }
    __device__ __forceinline__ RowHdr ld_hdr(u32 row) const {                  // RowHdr as its four dwords (dev_common.h)
        const u32 k = row - hrow0; v4 r;
        if (k < hn) r = *(const __attribute__((address_space(3))) v4*)(hhdr + k);
        else        r = *(const __attribute__((address_space(1))) v4*)(hdr + row);
        RowHdr h; h.total = r.x; h.iend = (u16)(r.y & 0xffffu); h.count = (u8)((r.y >> 16) & 0xffu); h.pad = 0; h.epoch = r.z; h.pad2 = 0;
        return h;
    }
    __device__ __forceinline__ void st_hdr(u32 row, const RowHdr& h) const {
        const u32 k = row - hrow0; v4 r; r.x = h.total; r.y = (u32)h.iend | ((u32)h.count << 16); r.z = h.epoch; r.w = 0;
        if (k < hn) *(__attribute__((address_space(3))) v4*)(hhdr + k) = r;
        else        *(__attribute__((address_space(1))) v4*)(hdr + row) = r;
    }
    // hot row k starts fresh: epoch 0 is no block's (written the way ld_hdr reads it)
    __device__ __forceinline__ void fresh_hot(u32 k) const { v4 z; z.x = 0; z.y = 0; z.z = 0; z.w = 0; *(__attribute__((address_space(3))) v4*)(hhdr + k) = z; }
    __device__ __forceinline__ u32 comp(const uint4& v, u32 c) const { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }
    // update_freq (power_ranger.hpp:66-84) of slot i = 4 hl + c (value cur), then the row and its header back to HBM
    __device__ __forceinline__ void update(u32 row, u32 lane, u32 i, u32 hl, u32 c, u32 cur, u32 total, u32 iend, u32 count, uint4 v, u64 dirty) {
        const u32 i0 = 4 * lane;
        u32 f = cur & 0xffffu;
        bool upd = true;
        if (f > (u32)((1 << 15) - 32 - 14)) {                              // update_freq :66-84
            if (i == 0 && f + 256u > total) upd = false;
            else {
                if (i0 + 0 < iend) v.x = (v.x & 0xffff0000u) | ((v.x & 0xffffu) >> 1);     // normalize :49-52
                if (i0 + 1 < iend) v.y = (v.y & 0xffff0000u) | ((v.y & 0xffffu) >> 1);
                if (i0 + 2 < iend) v.z = (v.z & 0xffff0000u) | ((v.z & 0xffffu) >> 1);
                if (i0 + 3 < iend) v.w = (v.w & 0xffff0000u) | ((v.w & 0xffffu) >> 1);
                const u32 p2 = ((i0 + 0 < iend) ? (v.x & 0xffffu) : 0u) + ((i0 + 1 < iend) ? (v.y & 0xffffu) : 0u) +
                               ((i0 + 2 < iend) ? (v.z & 0xffffu) : 0u) + ((i0 + 3 < iend) ? (v.w & 0xffffu) : 0u);
                total = rl(wave_incl_scan(p2), 63);
                f >>= 1;
                dirty |= __ballot(i0 < iend);
            }
        }
        if (upd) {
            f += 14; total += 14;
            u32 ns = (cur & 0xffff0000u) | f;
            u32 at = i;                                                    // where ns lands
            if (i != 0) {
                count = (count + 1) & 0xffu;
                if ((count & 0xfu) == 0) {
                    const u32 pl = (i - 1) >> 2, pc = (i - 1) & 3;
                    const u32 pv = pc == 0 ? rl(v.x, pl) : pc == 1 ? rl(v.y, pl) : pc == 2 ? rl(v.z, pl) : rl(v.w, pl);
                    if (f > (pv & 0xffffu)) {                              // down_level :54-64
                        if (lane == hl) { if (c == 0) v.x = pv; else if (c == 1) v.y = pv; else if (c == 2) v.z = pv; else v.w = pv; }
                        at = i - 1;
                        dirty |= 1ull << hl;
                    }
                }
            }
            const u32 al = at >> 2, ac = at & 3;
            if (lane == al) { if (ac == 0) v.x = ns; else if (ac == 1) v.y = ns; else if (ac == 2) v.z = ns; else v.w = ns; }
            dirty |= 1ull << al;
        }
        if ((dirty >> lane) & 1) st_slots(row, i0, v);
        if (lane == 0) {
            RowHdr nh; nh.total = total; nh.iend = (u16)iend; nh.count = (u8)count; nh.pad = 0; nh.epoch = epoch; nh.pad2 = 0;
            st_hdr(row, nh);
        }
    }
    // PowerRanger::get (power_ranger.hpp:106-130): the slot whose cumulative range holds the coder's value -- every slot's
    // freq + 1 summed four per lane, a wave scan, the first lane past the value, then its four slots; the slots between
    // the old iend and the one found come into being on the way (:118-119).  Every lane runs the same (uniform) coder.
    template <typename SRC>
    __device__ __forceinline__ u32 get(u32 row, RcDec& rc, SRC& src, u32 lane) {
        const RowHdr h = ld_hdr(row);
        const bool live = rl(h.epoch, 0) == epoch;
        u32 total = live ? rl(h.total, 0) : 0u, iend = live ? rl((u32)h.iend, 0) : 0u, count = live ? rl((u32)h.count, 0) : 0u;
        const u32 prob = rc.get_freq(total + PW_NSYM);
        const u32 i0 = 4 * lane;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (i0 < iend) v = ld_slots(row, i0);
        if (i0 + 0 >= iend) v.x = (i0 + 0) << 16;                           // a slot not yet in the row: its own symbol, frequency 0
        if (i0 + 1 >= iend) v.y = (i0 + 1) << 16;
        if (i0 + 2 >= iend) v.z = (i0 + 2) << 16;
        if (i0 + 3 >= iend) v.w = (i0 + 3) << 16;
        const u32 f0 = (v.x & 0xffffu) + 1, f1 = (v.y & 0xffffu) + 1, f2 = (v.z & 0xffffu) + 1, f3 = (v.w & 0xffffu) + 1;
        const u32 lsum = (f0 + f1) + (f2 + f3);
        const u32 incl = wave_incl_scan(lsum);
        const u64 past = __ballot(incl > prob);
        u32 hl, c, sumf;
        if (past) {
            hl = (u32)__ffsll((long long)past) - 1u;
            sumf = rl(incl - lsum, hl);
            const u32 a0 = rl(f0, hl), a1 = rl(f1, hl), a2 = rl(f2, hl);
            c = 0;
            if (sumf + a0 <= prob) { sumf += a0; c = 1; if (sumf + a1 <= prob) { sumf += a1; c = 2; if (sumf + a2 <= prob) { sumf += a2; c = 3; } } }
        } else {                                                            // the value lies beyond the row's total: a corrupt stream
            rc.err = 1; hl = 63; c = 3; sumf = rl(incl, 63) - rl(f3, 63);
        }
        const u32 i = 4 * hl + c;
        u64 dirty = 0;
        if (i >= iend) { dirty |= __ballot(i0 + 3 >= iend && i0 <= i); iend = i + 1; }
        const u32 cur = c == 0 ? rl(v.x, hl) : c == 1 ? rl(v.y, hl) : c == 2 ? rl(v.z, hl) : rl(v.w, hl);
        rc.decode(src, sumf, (cur & 0xffffu) + 1);
        update(row, lane, i, hl, c, cur, total, iend, count, v, dirty);
        return (cur >> 16) & 0xffu;
    }
    // PowerRangerU::get_u (power_ranger.hpp:165-190)
    template <typename SRC>
    __device__ __forceinline__ u64 get_u(u32 row0, RcDec& rc, SRC& s, u32 lane) {
        // (one loop around ONE inlined get, as put_u below: four copies of the row search per number made these kernels 150-250 KB of code)
        u64 num = 0; u32 n = 1;
#pragma nounroll
        for (u32 j = 0; j < n; j++) {
            const u32 row = j < 2 ? row0 + j : row0 + (n == 6 ? 2u : 6u) + (j - 2);
            const u32 b = get(row, rc, s, lane);
            if (j == 0) { num = b; n = b > 0x7f ? 2u : 1u; }
            else if (j == 1) {
                num = (num << 8) | b;
                if (num < 0xfffe) num &= 0x7fff;
                else { n = num == 0xfffe ? 6u : 10u; num = 0; }
            } else num |= (u64)b << (8 * (j - 2));
        }
        return num;
    }
    // PowerRanger::put minus Encode: returns the triple, updates the row in HBM.  sym uniform, < 256.
    __device__ __forceinline__ Triple model(u32 row, u32 sym, u32 lane) {
        const RowHdr h = ld_hdr(row);
        const bool live = rl(h.epoch, 0) == epoch;
        u32 total = live ? rl(h.total, 0) : 0u, iend = live ? rl((u32)h.iend, 0) : 0u, count = live ? rl((u32)h.count, 0) : 0u;
        const u32 i0 = 4 * lane;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (i0 < iend) v = ld_slots(row, i0);                              // only the lanes that hold live slots
        u64 dirty = 0;                                                     // lanes to store back
        if (iend <= sym) {                                                 // :94-96
            if (i0 + 0 >= iend && i0 + 0 <= sym) v.x = (i0 + 0) << 16;
            if (i0 + 1 >= iend && i0 + 1 <= sym) v.y = (i0 + 1) << 16;
            if (i0 + 2 >= iend && i0 + 2 <= sym) v.z = (i0 + 2) << 16;
            if (i0 + 3 >= iend && i0 + 3 <= sym) v.w = (i0 + 3) << 16;
            dirty |= __ballot(i0 + 3 >= iend && i0 <= sym);
            iend = sym + 1;
        }
        const u32 mc = ((i0 + 0 < iend && (v.x >> 16) == sym) ? 1u : 0u) | ((i0 + 1 < iend && (v.y >> 16) == sym) ? 2u : 0u) |
                       ((i0 + 2 < iend && (v.z >> 16) == sym) ? 4u : 0u) | ((i0 + 3 < iend && (v.w >> 16) == sym) ? 8u : 0u);
        const u64 hit = __ballot(mc != 0);
        const u32 hl = (u32)__ffsll((long long)hit) - 1u;                  // :98 (exactly one slot holds sym)
        const u32 c = (u32)__ffs((int)rl(mc, hl)) - 1u;
        const u32 i = 4 * hl + c;
        const u32 part = ((i0 + 0 < i) ? (v.x & 0xffffu) : 0u) + ((i0 + 1 < i) ? (v.y & 0xffffu) : 0u) +
                         ((i0 + 2 < i) ? (v.z & 0xffffu) : 0u) + ((i0 + 3 < i) ? (v.w & 0xffffu) : 0u);
        const u32 sumf = i ? rl(wave_incl_scan(part), 63) : 0u;
        u32 cur = c == 0 ? rl(v.x, hl) : c == 1 ? rl(v.y, hl) : c == 2 ? rl(v.z, hl) : rl(v.w, hl);
        Triple t; t.cum = sumf + i; t.freq = (cur & 0xffffu) + 1; t.tot = total + PW_NSYM;               // :100
        update(row, lane, i, hl, c, cur, total, iend, count, v, dirty);
        return t;
    }
    __device__ __forceinline__ void put(u32 row, RcEncU& rc, Sink0& s, u32 sym, u32 lane) {
        const Triple t = model(row, rl(sym, 0), lane);
        rc.encode(s, t.cum, t.freq, t.tot);
    }
    // PowerRangerU::put_u (power_ranger.hpp:138-163), as one loop around one inlined put (see PwTab::put_u)
    __device__ __forceinline__ void put_u(u32 row0, RcEncU& rc, Sink0& s, u64 num, u32 lane) {
        const u32 n = num <= 0x7f ? 1u : num < 0x7ffe ? 2u : num < (1ULL << 32) ? 6u : 10u;
#pragma nounroll
        for (u32 j = 0; j < n; j++) {
            u32 row, sym;
            if (j == 0)      { row = row0;     sym = n == 1 ? (u32)num : n == 2 ? (0xff & (0x80 | (u32)(num >> 8))) : 0xffu; }
            else if (j == 1) { row = row0 + 1; sym = n == 2 ? (0xff & (u32)num) : n == 6 ? 0xfeu : 0xffu; }
            else             { row = row0 + (n == 6 ? 2 : 6) + (j - 2); sym = 0xff & (u32)(num >> (8 * (j - 2))); }
            put(row, rc, s, sym, lane);
        }
    }
};
struct XfEncW {                // XFileSave (xfile.cpp:40-74), wave-cooperative
    RcEncU rc; Sink0 sink; u32 row0, opened;
    __device__ __forceinline__ void init(u8* p, u32 cap, u32 xf) { sink.p = p; sink.pos = 0; sink.cap = cap; row0 = PR_XF_BASE + xf * PR_XF_ROWS; opened = 0; rc.init(); }
    __device__ __forceinline__ void put(WavePw& t, u64 gap, u32 lane) { opened = 1; t.put_u(row0, rc, sink, gap, lane); }
    __device__ __forceinline__ void put_str(WavePw& t, const u8* p, u32 len, u32 lane) {
        put(t, len, lane);
        for (u32 j = 0; j < len; j++) t.put(row0 + 14, rc, sink, p[j], lane);
    }
    __device__ __forceinline__ u32 finish(WavePw& t, u32 lane) {
        if (!opened) return 0;
        put(t, 0, lane);
        rc.done(sink);
        return sink.pos;
    }
};

struct XfDecW {                // XFileLoad (xfile.cpp:76-99), wave-cooperative
    RcDec rc; ByteSrc1 src; u32 row0, valid;
    __device__ __forceinline__ void init(const u8* p, u32 n, u32 xf) {
        src.init(p, n); row0 = PR_XF_BASE + xf * PR_XF_ROWS; valid = n > 0;
        if (valid) rc.init(src); else { rc.low = rc.code = 0; rc.range = 0xFFFFFFFFu; rc.err = 0; }
    }
    __device__ __forceinline__ u64 get(WavePw& t, u32 lane) { return valid ? t.get_u(row0, rc, src, lane) : 0; }
    __device__ __forceinline__ u32 get_chr(WavePw& t, u32 lane) { return valid ? t.get(row0 + 14, rc, src, lane) : 0; }       // xfile.cpp:101-106
};
