// dev_wave.h -- wave-level building blocks shared by the wave-per-block kernels (models_w.hip, models_k.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "dev_common.h"
#include "dev_models.h"

// ---- wave primitives (gfx950 = wave64, GFX9 DPP controls) --------------------------------------------
#define DPP_ROW_SHR(n)  (0x110 + (n))
#define DPP_WAVE_SHR1   0x138
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143

__device__ __forceinline__ u32 rl(u32 v, u32 lane) { return (u32)__builtin_amdgcn_readlane((int)v, (int)lane); }
// write a uniform value into one lane (this clang has no writelane builtin: compare + select)
__device__ __forceinline__ u32 wl(u32 old, u32 val, u32 lane) { return (threadIdx.x == lane) ? val : old; }
// lane 0's value.  Deliberately NOT readfirstlane: hipcc may sink a readfirstlane into a divergent
// select (`lane == k ? rfl(x) : y`), where it would read lane k instead; readlane(.., 0) ignores EXEC.
__device__ __forceinline__ u32 rfl(u32 v) { return (u32)__builtin_amdgcn_readlane((int)v, 0); }

// inclusive prefix sum across the 64 lanes (6 DPP adds); lane 63 ends up with the wave total
__device__ __forceinline__ u32 wave_incl_scan(u32 x) {
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, DPP_ROW_SHR(1), 0xf, 0xf, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, DPP_ROW_SHR(2), 0xf, 0xf, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, DPP_ROW_SHR(4), 0xf, 0xf, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, DPP_ROW_SHR(8), 0xf, 0xf, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, DPP_ROW_BCAST15, 0xa, 0xf, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, DPP_ROW_BCAST31, 0xc, 0xf, false);
    return x;
}
// value of the previous lane; lane 0 receives `first`
__device__ __forceinline__ u32 wave_shr1(u32 x, u32 first) {
    return (u32)__builtin_amdgcn_update_dpp((int)first, (int)x, DPP_WAVE_SHR1, 0xf, 0xf, false);
}

// Persistent launch: a workgroup (= one wave = one table slot) takes blocks off a shared ticket counter until
// none are left, so the grid never exceeds the table slots and long and short blocks balance themselves.
__device__ __forceinline__ u32 next_block(u32* ticket) {
    u32 b = 0;
    if (threadIdx.x == 0) b = atomicAdd(ticket, 1u);
    return rl(b, 0);
}


__device__ __forceinline__ u32 wave_incl_scan_max(u32 x) {
    u32 y;
    y = (u32)__builtin_amdgcn_update_dpp(0, (int)x, DPP_ROW_SHR(1), 0xf, 0xf, false); x = x > y ? x : y;
    y = (u32)__builtin_amdgcn_update_dpp(0, (int)x, DPP_ROW_SHR(2), 0xf, 0xf, false); x = x > y ? x : y;
    y = (u32)__builtin_amdgcn_update_dpp(0, (int)x, DPP_ROW_SHR(4), 0xf, 0xf, false); x = x > y ? x : y;
    y = (u32)__builtin_amdgcn_update_dpp(0, (int)x, DPP_ROW_SHR(8), 0xf, 0xf, false); x = x > y ? x : y;
    y = (u32)__builtin_amdgcn_update_dpp(0, (int)x, DPP_ROW_BCAST15, 0xa, 0xf, false); x = x > y ? x : y;
    y = (u32)__builtin_amdgcn_update_dpp(0, (int)x, DPP_ROW_BCAST31, 0xc, 0xf, false); x = x > y ? x : y;
    return x;
}


// ---- base-model helpers ------------------------------------------------------------------------------------
__device__ __forceinline__ u32 gencode_w(u32 c) {                      // gens.cpp:72-77
    const u32 l = c | 0x20u;
    u32 n = 0x10u;
    n = (l == 'a' || c == '0') ? 0u : n;
    n = (l == 'c' || c == '1') ? 1u : n;
    n = (l == 'g' || c == '2') ? 2u : n;
    n = (l == 't' || c == '3') ? 3u : n;
    n = (l == 'n' || c == '.') ? 4u : n;
    return n;
}
__device__ __forceinline__ u32 shfl_up0(u32 x, u32 d, u32 lane) {      // lane-d's value, 0 for lanes < d
    const u32 y = (u32)__shfl_up((int)x, d, 64);
    return lane >= d ? y : 0u;
}
__device__ __forceinline__ u32 bitonic_sort64(u32 key, u32 lane) {
#pragma unroll
    for (u32 k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (u32 j = k >> 1; j > 0; j >>= 1) {
            const u32 other = (u32)__shfl_xor((int)key, j, 64);
            const bool up = (lane & k) == 0;
            const bool lower = (lane & j) == 0;
            const u32 lo = key < other ? key : other, hi = key < other ? other : key;
            key = (lower == up) ? lo : hi;
        }
    }
    return key;
}
// Base2Ranger::put minus the Encode call, on a row value (base2_ranger.hpp:74-84)
__device__ __forceinline__ u32 b2_model(u32 v, u32 sym, u32& cum, u32& freq, u32& tot) {
    const u32 f0 = v & 0xff, f1 = (v >> 8) & 0xff, f2 = (v >> 16) & 0xff, f3 = v >> 24;
    tot = (f0 + f1) + (f2 + f3);
    cum = sym == 0 ? 0u : sym == 1 ? f0 : sym == 2 ? f0 + f1 : f0 + f1 + f2;
    freq = (v >> (8 * sym)) & 0xff;
    return b2_update(v, sym);
}


// A stream's bytes for wave-UNIFORM code (every lane the same stream position): 256 bytes at a time, a dword per lane, the
// 256 behind them fetched while these are used -- no memory round trip sits on the coder's renormalisation.  Zeros past the
// end (FilerLoad::get, filer.hpp:94-97); nothing outside [p, p + n) is touched.
struct WaveSrc {
    const u8* p; u32 n, pos, lane;
    u32 cur, nxt;                                          // per lane: dword `lane` of the window at pos & ~255, and of the next one
    __device__ __forceinline__ u32 fetch(u32 base) const {
        const u32 at = base + 4u * lane;
        if (at + 4u <= n) return *reinterpret_cast<const u32*>(p + at);       // (vector loads need no alignment on gfx9)
        u32 v = 0;
        for (u32 k = 0; k < 4; k++) if (at + k < n) v |= (u32)p[at + k] << (8 * k);
        return v;
    }
    __device__ __forceinline__ void init(const u8* ptr, u32 len) { p = ptr; n = len; pos = 0; lane = threadIdx.x & 63u; cur = fetch(0); nxt = fetch(256); }
    __device__ __forceinline__ u32 get() {
        const u32 w = rl(cur, (pos >> 2) & 63u);
        const u32 b = (w >> ((pos & 3u) * 8u)) & 0xffu;
        pos++;
        if ((pos & 255u) == 0) { cur = nxt; nxt = fetch(pos + 256u); }
        return b;
    }
};
