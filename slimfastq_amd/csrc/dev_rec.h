// dev_rec.h -- header-model helpers shared by the lane-per-block, wave-per-block and chain kernels: cutting a header line
// into fields and typing a field.  Written from the behaviour table of SURVEY.md row a13 (what the reference's tokeniser and
// field typing DO, RecBase::map_space recs.cpp:141-157, numberwang recs.cpp:192-262), as a tokeniser with a running field start
// and a one-pass classifier over character classes.
#pragma once
#include "dev_common.h"

enum {  // the type symbols of the "rec" stream: a format table (recs.cpp:159-190)
    ST_DGT = 0, ST_DLT = 1, ST_STR = 2, ST_HGT = 3, ST_HLT = 4, ST_HGT_Z = 5, ST_HLT_Z = 6,
    ST_HGTC = 7, ST_HLTC = 8, ST_HGTC_Z = 9, ST_HLTC_Z = 10, ST_DGT_Z = 11, ST_DLT_Z = 12
};
struct SpaceMap { u16 off[66]; u16 wln[66]; u8 str[66]; u32 len; };   // per field: where it starts, how long it is, the separator behind it
__device__ __forceinline__ bool isdig(u32 c) { return c - '0' < 10u; }
__device__ __forceinline__ bool isword(u32 c) { return isdig(c) || ((c | 0x20) - 'a' < 26u); }   // a field character: letter or digit

// The fields of the header text p[0..n).  A field ends at every character that is neither a letter nor a digit; the line's
// '\n' behind the text closes the last field, a NUL inside closes it early.  false: more fields than the header model has
// rows for (64).
static __device__ bool map_space(const u8* p, u32 n, SpaceMap& m) {
    u32 nf = 0, start = 0;
    m.off[0] = 0;
    for (u32 i = 0; i <= n; i++) {
        const u32 c = i < n ? p[i] : '\n';
        if (isword(c)) continue;
        m.wln[nf] = (u16)(i - start);
        m.str[nf] = (u8)c;
        start = i + 1;
        m.off[++nf] = (u16)start;
        if (nf > 64) { m.len = nf; return false; }
        if (c == 0) break;
    }
    m.len = nf;
    return true;
}

// A field's type and value, in ONE pass over its characters.  Every character falls in a class -- decimal digit, a-f, A-F,
// anything else --; the classes seen are ORed together and both readings of the field, decimal and hexadecimal, are carried
// along.  The type then follows from the class mask, the length and what the field was BEFORE (pctype 2: it was hexadecimal,
// and hexadecimal is sticky):
//   "00..."                              cannot be printed back from a number: string
//   digits only, not sticky-hex          decimal (a single leading zero is part of the type); a decimal reading that wrapped
//                                        around 2^64 on the way: string
//   else                                 hexadecimal where the field has at most 16 characters, all hex digits, its letters of
//                                        ONE case (upper case is part of the type); string otherwise
// at(j) = character j of the field; at(len), the separator behind it, is readable.
#define FC_DEC 1u
#define FC_LOW 2u
#define FC_UPP 4u
#define FC_OTHER 8u
template <typename AT>
__device__ __forceinline__ u32 field_type(AT&& at, u32 len, u64& num, u32 pctype) {
    num = 0;
    const u32 z = at(0) == '0' ? 1u : 0u;
    if (z && at(1) == '0') return ST_STR;
    u32 seen = 0; u64 dec = 0, hex = 0; bool wrapped = false;
    for (u32 j = z; j < len; j++) {
        const u32 c = at(j);
        const u32 d = c - '0', lo = c - 'a', up = c - 'A';
        const u32 cls = d < 10u ? FC_DEC : lo < 6u ? FC_LOW : up < 6u ? FC_UPP : FC_OTHER;
        const u32 nib = d < 10u ? d : lo < 6u ? lo + 10u : up + 10u;
        if (seen <= FC_DEC && cls == FC_DEC) { const u64 t = dec * 10u + d; wrapped = wrapped || t < dec; dec = t; }   // while only digits have been seen
        seen |= cls;
        hex = (hex << 4) | (nib & 15u);
    }
    if (pctype != 2) {
        if (wrapped) return ST_STR;
        if (seen <= FC_DEC) { num = dec; return z ? ST_DGT_Z : ST_DGT; }
    }
    if (len > 16 || (seen & FC_OTHER) || (seen & (FC_LOW | FC_UPP)) == (FC_LOW | FC_UPP)) return ST_STR;
    num = hex;
    return (seen & FC_UPP) ? (z ? ST_HGTC_Z : ST_HGTC) : (z ? ST_HGT_Z : ST_HGT);
}
// the same over a field in memory (p[len] is the separator)
static __device__ u32 numberwang(const u8* p, int len, u64& num, u32 pctype) {
    return field_type([p](u32 j) -> u32 { return p[j]; }, (u32)len, num, pctype);
}
__device__ __forceinline__ bool bytes_differ(const u8* x, const u8* y, u32 n) {
    for (u32 i = 0; i < n; i++) if (x[i] != y[i]) return true;
    return false;
}
