// dev_rec.h -- header-model helpers shared by the lane-per-block and wave-per-block kernels:
// tokenising (RecBase::map_space, recs.cpp:141-157) and field typing (numberwang, recs.cpp:192-262).
#pragma once
#include "dev_common.h"

enum {  // recs.cpp:159-190
    ST_DGT = 0, ST_DLT = 1, ST_STR = 2, ST_HGT = 3, ST_HLT = 4, ST_HGT_Z = 5, ST_HLT_Z = 6,
    ST_HGTC = 7, ST_HLTC = 8, ST_HGTC_Z = 9, ST_HLTC_Z = 10, ST_DGT_Z = 11, ST_DLT_Z = 12
};
struct SpaceMap { u16 off[66]; u16 wln[66]; u8 str[66]; u32 len; };   // recs.hpp:68-73
__device__ __forceinline__ bool isword(u32 c) { return (c - '0' < 10u) || ((c | 0x20) - 'a' < 26u); }   // recs.cpp:139
__device__ __forceinline__ bool isdig(u32 c) { return c - '0' < 10u; }

// map_space recs.cpp:141-157 over text p[0..n) (the terminator '\n' is at p[n]).  false = > 64 separators.
static __device__ bool map_space(const u8* p, u32 n, SpaceMap& m) {
    m.len = 0; m.off[0] = 0;
    for (u32 i = 0; ; i++) {
        u32 c = i < n ? p[i] : '\n';
        if (!isword(c)) {
            m.wln[m.len] = (u16)(i - m.off[m.len]);
            m.str[m.len++] = (u8)c;
            m.off[m.len] = (u16)(i + 1);
            if (i >= n || c == 0) break;
            if (m.len > 64) return false;
        }
    }
    return m.len <= 64;
}
// numberwang recs.cpp:192-262.  p[len] is readable (separator).
static __device__ u32 numberwang(const u8* p, int len, u64& num, u32 pctype) {
    int i = 0;
    const bool has_z = p[0] == '0';
    if (has_z) if (p[++i] == '0') return ST_STR;
    u32 caps = 0;
    num = 0;
    while (pctype != 2) {
        if (i >= len) return has_z ? ST_DGT_Z : ST_DGT;
        u32 c = p[i];
        if (isdig(c)) {
            u64 tnum = (num << 3) + (num << 1) + c - '0';
            i++;
            if (tnum < num) return ST_STR;
            num = tnum;
            continue;
        }
        if ((c | 0x20) < 'a' || (c | 0x20) > 'f') return ST_STR;
        caps = 1 + (c < 'a');
        i = has_z;
        num = 0;
        break;
    }
    if (len > 16) return ST_STR;
    for (; i < len; i++) {
        u32 c = p[i], nib;
        if (isdig(c)) nib = c - '0';
        else if (c >= 'a' && c <= 'f') { if (caps == 2) return ST_STR; caps = 1; nib = 10 + (c - 'a'); }
        else if (c >= 'A' && c <= 'F') { if (caps == 1) return ST_STR; caps = 2; nib = 10 + (c - 'A'); }
        else return ST_STR;
        num = (num << 4) + nib;
    }
    return caps == 2 ? (has_z ? ST_HGTC_Z : ST_HGTC) : (has_z ? ST_HGT_Z : ST_HGT);
}
__device__ __forceinline__ bool bytes_differ(const u8* x, const u8* y, u32 n) {
    for (u32 i = 0; i < n; i++) if (x[i] != y[i]) return true;
    return false;
}

