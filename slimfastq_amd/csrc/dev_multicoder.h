// dev_multicoder.h -- stage 3 for K record blocks in one wavefront: the range coder's state lives per lane GROUP
// (group h = lane / (64/K) carries block h's chain), the (cum, freq, tot, reciprocal) steps come through LDS.
// Used by the multi-chain kernels (models_k.hip: bases, models_w.hip: qualities).
#pragma once
#include "dev_coder.h"
#include "dev_wave.h"

#define NEUTRAL_TRIPLE make_uint4(0u, 1u, 1u, 0xFFFFFFFFu)     // range /= 1, low += 0, range *= 1: a no-op step

// reciprocal for the multiply-high divide: m = floor(2^32 / tot) (estimate at most 1 below); tot = 1 (only the
// neutral step) keeps 2^32 - 1, which the same single fix-up handles
__device__ __forceinline__ u32 recip_exact(u32 tot) {
    const u32 m0 = 0xFFFFFFFFu / tot;
    return (tot != 1 && (0xFFFFFFFFu - m0 * tot) == tot - 1) ? m0 + 1 : m0;
}
// per-chain scalars live one per lane (lane j = chain j): read with readlane, written with a select
#define CGET(reg, j) rl(reg, j)
#define CSET(reg, j, val) do { const u32 v_ = (val); reg = (threadIdx.x == (j)) ? v_ : reg; } while (0)

// ---- stage 3 for K chains: RCoder (coder.hpp) state per lane group, all on the vector unit -------------------
struct MultiCoder {
    u64 lo;            // RCoder::low   (equal in all lanes of a group)
    u32 vr;            // RCoder::range
    u32 acc;           // the last up-to-4 output bytes, oldest in the low byte once full
    u32 pos, cap;      // bytes produced / region size
    u8* outp;          // 16-byte aligned region
    u32 err;
    __device__ __forceinline__ void reset(bool mine, u8* p, u32 c) {        // start a new stream on one group
        if (mine) { lo = 0; vr = 0xFFFFFFFFu; acc = 0; pos = 0; cap = c; outp = p; }   // coder.hpp:34-39
    }
    __device__ __forceinline__ void put(bool pred, u32 byte, bool lead) {   // FilerSave::put for the lanes with pred
        if (pred) {
            acc = (acc >> 8) | (byte << 24);
            pos++;
            if ((pos & 3) == 0 && lead && pos <= cap) *reinterpret_cast<u32*>(outp + pos - 4) = acc;
        }
    }
    // one pass of coder.hpp:74-80 on the lanes that need it, written with selects (no divergent control flow: the
    // compiler's version of the branchy form spent a third of its instructions moving loop-carried registers)
    __device__ __forceinline__ void renorm_step(bool lead) {
        const bool pred = vr < RC_TOP;
        const u32 lo32 = (u32)lo, hi32 = (u32)(lo >> 32);
        const u64 sum = lo + vr;
        const bool clamp = ((((u32)(sum >> 32)) ^ hi32) >> 24) != 0;       // (low ^ (low + range)) >> 56
        const u32 vrc = clamp ? (~lo32 & (RC_TOP - 1)) : vr;               // ((u32)low | (TOP-1)) - (u32)low
        const u32 nacc = __builtin_amdgcn_alignbit(hi32 >> 24, acc, 8);    // (acc >> 8) | (byte << 24)
        const u32 npos = pos + 1;
        if (pred && lead && (npos & 3) == 0 && npos <= cap) *reinterpret_cast<u32*>(outp + npos - 4) = nacc;
        acc = pred ? nacc : acc;
        pos = pred ? npos : pos;
        vr = pred ? (vrc << 8) : vr;
        lo = pred ? (lo << 8) : lo;
    }
    // walk nmax steps; trip[h][k] = {cum, freq, tot, recip}; steps past a chain's own count are neutral
    __device__ __forceinline__ void run(const uint4 (*trip)[64], u32 nmax, u32 h, bool lead) {
#pragma nounroll
        for (u32 k = 0; k < nmax; k++) {
            const uint4 t = trip[h][k];
            u32 r = __umulhi(vr, t.w);                                       // r = range / tot (coder.hpp:68)
            const u32 rem = vr - r * t.z;
            r += rem >= t.z ? 1u : 0u;
            lo += (u64)t.x * r;                                              // coder.hpp:69 (cum * r < range: no wrap)
            vr = r * t.y;                                                    // coder.hpp:70
            int guard = 0;
#pragma nounroll
            while (__any(vr < RC_TOP)) {
                renorm_step(lead);
                // the reference spins forever if the clamp yields range 0; every chain here must drain
                if (++guard > 13) { err = 1; if (vr < RC_TOP) vr = 0xFFFFFFFFu; break; }
            }
        }
    }
    __device__ __forceinline__ void done(bool mine, bool lead) {            // coder.hpp:52-61 + the bytes still in acc
        for (int i = 0; i < 8; i++) { put(mine, (u32)(lo >> 56), lead); if (mine) lo <<= 8; }
        const u32 pend = pos & 3;
        if (mine && lead) for (u32 i = 0; i < pend; i++) { const u32 at = pos - pend + i; if (at < cap) outp[at] = (u8)(acc >> (8 * (4 - pend + i))); }
    }
};

