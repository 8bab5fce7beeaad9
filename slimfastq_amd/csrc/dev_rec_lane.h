// dev_rec_lane.h -- the header model (recs.cpp) as lane-serial device code, written once for every coder behind it:
// the adaptive PowerRanger rows of the lane-per-block kernels (models_l.hip, decode_l.hip: the reference's behaviour) and
// the frozen rows / the counting pass of the lane-per-chain kernels (chains.hip).
#pragma once
#include "kernels.h"
#include "dev_models.h"
#include "dev_rec.h"

// RecSave::save (recs.cpp:277-372) for records [rec0, rec0 + nrec) of a block whose first record is base_rec, by one
// lane.  The block's first header is the BASE: never coded (it travels as "rec.first", recs.cpp:68-75), it is what the
// first coded record is compared with.  The reference's case is rec0 == base_rec (a block = one chain); the frozen-table
// mode cuts a block's headers into several chains that all start from the block's base (chains.hip).
// C codes the symbols of the "rec" stream: put(row, byte) / put_u(row0, number) on PowerRanger row numbers
// (dev_common.h: field i uses rows (i + 1) * 16 + {0 type, 1 str, 2.. num}).  A header whose shape changed goes whole
// to the block's adaptive "rec.x" XFile -- or, for a coder with C::inband, into the chain itself: every record starts
// with a flag symbol (row REC_FLAG_ROW: 0 = fields follow, 1 = the length and the characters of the whole line follow).
#define REC_FLAG_ROW (65 * 16)        /* the rows of field 65: no header has that many fields (map_space stops at 64) */
template <typename C>
__device__ __forceinline__ void rec_encode_lane(const ModelArgs& a, u64 base_rec, u64 rec0, u32 nrec, C& cd, XfEnc& x_rec, const PwTab& xpw,
                                                u32& hdr_bytes_out, int& bad_out) {
    SpaceMap sm[2];
    u8  fkind[2][66];
    u64 fvalue[2][66];
    u32 cur = 0; int bad = 0;
    u64 last_index = 0;                       // m_last.index recs.hpp:54
    u32 hdr_bytes = 0, step = 0, coded = 0;
    const u8* prev = nullptr;
    {                                                                         // recs.cpp:279-287: the base
        const u64 h0 = a.line_off[4 * base_rec] + 1, h1 = a.line_off[4 * base_rec + 1] - 1;
        const u32 n = h1 > h0 ? (u32)(h1 - h0) : 0;
        const u8* buf = cd.stage(a.fq + h0, n, step++);
        if (!map_space(buf, n, sm[0])) bad = SFQ_E_FORMAT;
        for (int i = 0; i < 66; i++) { fkind[0][i] = 0; fkind[1][i] = 0; }
        prev = buf;
    }
    for (u32 k = 0; k < nrec && !bad; k++) {
        const u64 r = rec0 + k;
        const u64 record_count = rec_count_of(a, r, base_rec);   // g_record_count, block-relative
        const u64 h0 = a.line_off[4 * r] + 1, h1 = a.line_off[4 * r + 1] - 1;
        const u32 n = h1 > h0 ? (u32)(h1 - h0) : 0;
        hdr_bytes += n;
        if (r == base_rec) continue;                                          // the base itself
        cd.record(coded++);
        const u8* buf = cd.stage(a.fq + h0, n, step++);  // the text itself, or the coder's faster copy of it
        const u32 prv = cur;
        cur ^= 1;
        if (!map_space(buf, n, sm[cur])) { bad = SFQ_E_FORMAT; break; }
        SpaceMap& mi = sm[cur]; SpaceMap& mp = sm[prv];
        bool shape = mi.len != mp.len;
        if (!shape) for (u32 i = 0; i < mi.len; i++) if (mi.str[i] != mp.str[i]) { shape = true; break; }
        if (a.lossless && mi.str[mi.len - 1] == 0) shape = true;              // a NUL inside: the fields behind it would be lost (dev_common.h)
        if constexpr (C::inband) cd.put(REC_FLAG_ROW, shape ? 1u : 0u);
        if (shape) {                                                          // recs.cpp:292-305
            if constexpr (C::inband) {
                cd.put_u(REC_FLAG_ROW + 2, n);
                for (u32 j = 0; j < n; j++) cd.put(REC_FLAG_ROW + 1, buf[j]);
            } else {
                x_rec.put(xpw, record_count - last_index);
                last_index = record_count;
                x_rec.put_str(xpw, buf, n);
            }
            for (int i = 0; i < 66; i++) fkind[cur][i] = 0;
            prev = buf;
            continue;
        }
        u64 map = 0;
        for (u32 i = 0; i < mi.len; i++)
            if (mi.wln[i] != mp.wln[i] || bytes_differ(buf + mi.off[i], prev + mp.off[i], mi.wln[i])) map |= 1ULL << i;
        cd.put_u(0 * 16 + 2, map);                                            // put_num(0, map) recs.cpp:313
        for (u32 i = 0; i < mi.len; i++) {
            if (map & (1ULL << i)) {
                const u8* bp = buf + mi.off[i];
                u64 fnum;
                u32 type = numberwang(bp, mi.wln[i], fnum, fkind[prv][i]);
                if (a.lossless && type != ST_STR && !rec_number_prints_back(type, mi.wln[i], bp[0])) type = ST_STR;
                const u32 rr = (i + 1) * 16;
                if (type == ST_STR) {                                         // recs.cpp:324-331
                    cd.put(rr + 0, type);
                    cd.put_u(rr + 2, mi.wln[i]);
                    for (u32 j = 0; j < mi.wln[i]; j++) cd.put(rr + 1, bp[j]);
                    fkind[cur][i] = 0;
                    continue;
                }
                u64 was = fkind[prv][i] ? fvalue[prv][i] : 0;                // recs.cpp:333-348
                u64 gap;
                fkind[cur][i] = (type < ST_STR || type >= ST_DGT_Z) ? 1 : 2;
                fvalue[cur][i] = fnum;
                if (fnum < was) { gap = was - fnum; type++; }
                else gap = fnum - was;
                cd.put(rr + 0, type);
                cd.put_u(rr + 2, gap);
            } else {
                fkind[cur][i] = fkind[prv][i];
                fvalue[cur][i] = fvalue[prv][i];
            }
        }
        prev = buf;
    }
    hdr_bytes_out = hdr_bytes; bad_out = bad;
}
// PowerRangerU::put_u's byte sequence (power_ranger.hpp:138-163) on any coder with put(row, byte): 1 byte (<= 0x7f),
// 2 bytes (< 0x7ffe), 0xff 0xfe + 4 LE bytes, or 0xff 0xff + 8 LE bytes, on rows row0 .. row0 + 13
template <typename C>
__device__ __forceinline__ void put_u_rows(C& cd, u32 row0, u64 num) {
    const u32 n = num <= 0x7f ? 1u : num < 0x7ffe ? 2u : num < (1ULL << 32) ? 6u : 10u;
#pragma nounroll
    for (u32 j = 0; j < n; j++) {
        u32 row, sym;
        if (j == 0)      { row = row0;     sym = n == 1 ? (u32)num : n == 2 ? (0xff & (0x80 | (u32)(num >> 8))) : 0xffu; }
        else if (j == 1) { row = row0 + 1; sym = n == 2 ? (0xff & (u32)num) : n == 6 ? 0xfeu : 0xffu; }
        else             { row = row0 + (n == 6 ? 2 : 6) + (j - 2); sym = 0xff & (u32)(num >> (8 * (j - 2))); }
        cd.put(row, sym);
    }
}
// PowerRangerU::get_u (power_ranger.hpp:165-190) on any coder with get(row)
template <typename C>
__device__ __forceinline__ u64 get_u_rows(C& cd, u32 row0) {
    // (one loop around ONE get: a coder's get is a table search plus a renormalisation with its refills -- four copies of it per number
    //  made the frozen header decoder 337 KB of code for a 64 KB instruction cache)
    u64 num = 0; u32 n = 1;
#pragma nounroll
    for (u32 j = 0; j < n; j++) {
        const u32 row = j < 2 ? row0 + j : row0 + (n == 6 ? 2u : 6u) + (j - 2);
        const u32 s = cd.get(row);
        if (j == 0) { num = s; n = s > 0x7f ? 2u : 1u; }
        else if (j == 1) {
            num = (num << 8) | s;
            if (num < 0xfffe) num &= 0x7fff;
            else { n = num == 0xfffe ? 6u : 10u; num = 0; }
        } else num |= (u64)s << (8 * (j - 2));
    }
    return num;
}
// the adaptive coder: the block's own PowerRanger rows (the reference's behaviour)
struct RecAdaptiveEnc {
    static constexpr bool inband = false;
    PwTab pw; RcEnc rc; ByteSink snk;
    __device__ __forceinline__ void record(u32) {}
    __device__ __forceinline__ const u8* stage(const u8* g, u32, u32) { return g; }
    __device__ __forceinline__ void put(u32 row, u32 sym) { pw.put(row, rc, snk, sym); }
    __device__ __forceinline__ void put_u(u32 row0, u64 num) { pw.put_u(row0, rc, snk, num); }
};

// RecLoad::load (recs.cpp:374-461; load_pre5 463-510) for the records of one block, by one lane.  C decodes the symbols
// of the "rec" stream: get(row) / get_u(row0) / err(); the "rec.x" exceptions come through the block's adaptive XFile rows.
struct DSpaceMap { u16 off[66]; u16 wln[66]; u8 str[66]; u32 len; };
__device__ __forceinline__ bool d_isword(u32 c) { return (c - '0' < 10u) || ((c | 0x20) - 'a' < 26u); }
__device__ bool d_map_space(const u8* p, u32 n, DSpaceMap& m) {       // recs.cpp:141-157
    m.len = 0; m.off[0] = 0;
    for (u32 i = 0; ; i++) {
        u32 c = i < n ? p[i] : '\n';
        if (!d_isword(c)) {
            m.wln[m.len] = (u16)(i - m.off[m.len]);
            m.str[m.len++] = (u8)c;
            m.off[m.len] = (u16)(i + 1);
            if (i >= n || c == 0) break;
            if (m.len > 64) return false;
        }
    }
    return m.len <= 64;
}
// sprintf("%lld" / "%llx" / "%llX") of a non-zero value (recs.cpp:453-456)
__device__ u32 fmt_dec(u8* b, u64 val) {
    u32 n = 0;
    i64 sv = (i64)val;
    u64 mag = sv < 0 ? (u64)0 - val : val;
    if (sv < 0) b[n++] = '-';
    u8 tmp[20]; u32 k = 0;
    while (mag) { tmp[k++] = (u8)('0' + mag % 10); mag /= 10; }
    while (k) b[n++] = tmp[--k];
    return n;
}
__device__ u32 fmt_hex(u8* b, u64 val, bool upper) {
    u32 n = 0; int sh = 60;
    while (sh > 0 && ((val >> sh) & 0xf) == 0) sh -= 4;
    for (; sh >= 0; sh -= 4) {
        u32 d = (u32)(val >> sh) & 0xf;
        b[n++] = (u8)(d < 10 ? '0' + d : (upper ? 'A' : 'a') + d - 10);
    }
    return n;
}
__device__ bool d_is_number(const u8* p, int len, i64& num) {          // recs.cpp:265-275
    if (*p == '0') return false;
    num = 0;
    for (int i = 0; i < len; i++) {
        if (p[i] - '0' < 10u) num = (num << 3) + (num << 1) + p[i] - '0';
        else return false;
    }
    return true;
}

template <typename C, typename X>
__device__ __forceinline__ void rec_decode_lane(const DecodeArgs& a, BlockDesc* d, u64 rec0, u32 nrec, u32 blk, C& cd, X& x_rec, const PwTab& xpw) {
    u64 index = 0;
    if constexpr (!C::inband) index = x_rec.get(xpw);                                       // recs.cpp:104
    u8* const stage = a.hdr_stage + a.hdr_stage_off[blk];
    const u64 cap = a.hdr_stage_cap[blk];
    u64 pos = 0;            // write cursor in stage
    DSpaceMap sm;
    u8  fkind[2][66];
    u64 fvalue[2][66];
    u32 cur = 0;
    int bad = 0;
    // the base: the block's first header (load_first_line recs.cpp:113-119); records [rec0, rec0 + nrec) of the block follow it
    const u8* prev = a.first_hdrs + d->first_hdr_off; u32 prev_n = d->first_hdr_len;
    for (int i = 0; i < 66; i++) { fkind[0][i] = 0; fkind[1][i] = 0; }
    if (prev_n > SFQ_MAX_ID_LLEN) { bad = SFQ_E_CORRUPT; nrec = 0; }                        // (also refused by sfq_decode_blocks)
    for (u32 k = 0; k < nrec; k++) {
        const u64 r = rec0 + k, rcnt = rec_count_of(a.m, r, d->rec0);
        // worst case for one header: every field regenerated at its longest (MAX_ID_LLEN) -> bounded check
        if (pos + SFQ_MAX_ID_LLEN + 2 > cap) { bad = SFQ_E_OVERFLOW; break; }
        u8* buf = stage + pos;
        u32 n = 0;
        if (r == d->rec0) {                                                                 // the base itself
            n = prev_n;
            for (u32 i = 0; i < n; i++) buf[i] = prev[i];
        } else {
            const u32 prv = cur;
            cur ^= 1;
            bool whole = false;
            if constexpr (C::inband) whole = cd.get(REC_FLAG_ROW) != 0; else whole = index == rcnt;
            if (whole && C::inband) {
                u64 len = cd.get_u(REC_FLAG_ROW + 2);
                if (len > SFQ_MAX_ID_LLEN) { bad = SFQ_E_CORRUPT; break; }
                for (u32 j = 0; j < (u32)len; j++) buf[j] = (u8)cd.get(REC_FLAG_ROW + 1);
                n = (u32)len;
                for (int i = 0; i < 66; i++) fkind[cur][i] = 0;
            } else if (whole) {                                                             // recs.cpp:386-393
                u64 len = x_rec.get(xpw);
                if (len > SFQ_MAX_ID_LLEN) { bad = SFQ_E_CORRUPT; break; }
                for (u32 j = 0; j < (u32)len; j++) buf[j] = (u8)x_rec.get_chr(xpw);
                n = (u32)len;
                index += x_rec.get(xpw);
                for (int i = 0; i < 66; i++) fkind[cur][i] = 0;
            } else {
                if (!d_map_space(prev, prev_n, sm)) { bad = SFQ_E_CORRUPT; break; }
                const u64 map = cd.get_u(0 * 16 + 2);
                u8* b = buf;
                for (u32 i = 0; i < sm.len; i++) {
                    // every copy below is bounded by what is left of the staging slice's per-header headroom (MAX_ID_LLEN + 2):
                    // a field is copied (wln bytes), regenerated (<= 21 bytes) or read as a string (checked where it is read)
                    if ((u64)(b - buf) + sm.wln[i] + 64 > SFQ_MAX_ID_LLEN) { bad = SFQ_E_CORRUPT; break; }
                    const u32 rr = (i + 1) * 16;
                    if (a.version >= 5) {                                                   // recs.cpp:403-459
                        if (!(map & (1ULL << i))) {
                            const u8* pp = prev + sm.off[i];
                            for (u32 j = 0; j < sm.wln[i]; j++) b[j] = pp[j];
                            b += sm.wln[i];
                            *b++ = sm.str[i];
                            fkind[cur][i] = fkind[prv][i];
                            fvalue[cur][i] = fvalue[prv][i];
                            continue;
                        }
                        const u32 type = cd.get(rr + 0);
                        if (type == ST_STR) {
                            u64 len = cd.get_u(rr + 2);
                            if (len > SFQ_MAX_ID_LLEN || (u64)(b - buf) + len + 64 > SFQ_MAX_ID_LLEN) { bad = SFQ_E_CORRUPT; break; }
                            for (u32 j = 0; j < (u32)len; j++) b[j] = (u8)cd.get(rr + 1);
                            b += len;
                            fkind[cur][i] = 0;
                            *b++ = sm.str[i];
                            continue;
                        }
                        const u64 pval = fkind[prv][i] == 0 ? 0 : fvalue[prv][i];
                        const u64 gap = cd.get_u(rr + 2);
                        if (type > ST_DLT_Z) { bad = SFQ_E_CORRUPT; break; }
                        const bool less = type == ST_DLT || type == ST_HLT || type == ST_HLT_Z || type == ST_HLTC ||
                                          type == ST_HLTC_Z || type == ST_DLT_Z;
                        const u64 val = less ? pval - gap : pval + gap;
                        const bool deci = type < ST_STR || type >= ST_DGT_Z;
                        const bool lead = type == ST_HGT_Z || type == ST_HLT_Z || type == ST_HGTC_Z || type == ST_HLTC_Z ||
                                          type == ST_DGT_Z || type == ST_DLT_Z;
                        const bool upper = type >= ST_HGTC && type <= ST_HLTC_Z;
                        fkind[cur][i] = deci ? 1 : 2;
                        fvalue[cur][i] = val;
                        if (val == 0) *b++ = '0';                                            // recs.cpp:453-454
                        else {
                            if (lead) *b++ = '0';
                            b += deci ? fmt_dec(b, val) : fmt_hex(b, val, upper);
                        }
                        *b++ = sm.str[i];
                    } else {                                                                // load_pre5 recs.cpp:463-510
                        if (map & (1ULL << i)) {
                            const u32 type = cd.get(rr + 0);
                            if (type == ST_DGT || type == ST_DLT) {
                                i64 pval = 0;
                                d_is_number(prev + sm.off[i], sm.wln[i], pval);
                                const i64 gap = (i64)cd.get_u(rr + 2);
                                const i64 val = type == ST_DGT ? pval + gap : pval - gap;
                                if (val == 0) *b++ = '0'; else b += fmt_dec(b, (u64)val);
                            } else if (type == ST_STR) {
                                u64 len = cd.get_u(rr + 2);
                                if (len > SFQ_MAX_ID_LLEN || (u64)(b - buf) + len + 64 > SFQ_MAX_ID_LLEN) { bad = SFQ_E_CORRUPT; break; }
                                for (u32 j = 0; j < (u32)len; j++) b[j] = (u8)cd.get(rr + 1);
                                b += len;
                            } else { bad = SFQ_E_CORRUPT; break; }
                        } else {
                            const u8* pp = prev + sm.off[i];
                            for (u32 j = 0; j < sm.wln[i]; j++) b[j] = pp[j];
                            b += sm.wln[i];
                        }
                        *b++ = sm.str[i];
                    }
                }
                if (bad) break;
                n = (u32)(b - buf) - 1;                                                     // recs.cpp:460
            }
        }
        buf[n] = '\n';
        a.hlen[r] = n; a.hoff[r] = a.hdr_stage_off[blk] + pos;
        prev = buf; prev_n = n;
        pos += (u64)n + 1;
        if (cd.err() | x_rec.rc.err) { bad = SFQ_E_CORRUPT; break; }
    }
    if (bad) {
        atomicMax(&d->status, (u32)(-bad));
        // leave the remaining records empty so the assembly stays in bounds
        for (u32 k = 0; k < nrec; k++) { const u64 r = rec0 + k; if (a.hoff[r] == ~0ULL) { a.hlen[r] = 0; a.hoff[r] = a.hdr_stage_off[blk]; } }
    }
}
struct RecAdaptiveDec {
    static constexpr bool inband = false;
    PwTab pw; RcDec rc; ByteSrc src;
    __device__ __forceinline__ u32 get(u32 row) { return pw.get(row, rc, src); }
    __device__ __forceinline__ u64 get_u(u32 row0) { return pw.get_u(row0, rc, src); }
    __device__ __forceinline__ u32 err() const { return rc.err; }
};
