// dev_rice.h -- adaptive Rice codes for the base-exception lists of the frozen-table mode (exc.hip has the story): the code's
// parameter, a wave's bit sink (the encoder: models_w.hip k_gen_exc_w<true>) and a lane's bit source (the decoder: exc.hip).
#pragma once
#include "dev_common.h"

#define RICE_A0 256u
#define RICE_KMAX 24u
#define RICE_ESC 32u

__device__ __forceinline__ u32 rice_k(u64 A, u32 N) {                  // the smallest k with (N << k) >= A, at most RICE_KMAX
    if (A <= N) return 0;
    const int la = 64 - __clzll((long long)A), ln = 32 - __clz((int)N);
    u32 k = la > ln ? (u32)(la - ln) : 0u;                              // N << k has as many bits as A: at most one step short
    if (((u64)N << k) < A) k++;
    return k < RICE_KMAX ? k : RICE_KMAX;
}
// A wave's bit sink (models_w.hip k_gen_exc_w<true>: a wavefront per block): every lane holds the same state, lane 0 stores.
struct RiceWU {
    u8* p; u32 pos, cap; u64 acc; u32 nbits; u64 A; u32 N; u32 opened, ovf;
    __device__ __forceinline__ void init(u8* dst, u32 c) { p = dst; pos = 0; cap = c; acc = 0; nbits = 0; A = RICE_A0; N = 1; opened = 0; ovf = 0; }
    __device__ __forceinline__ void bits(u32 v, u32 n, u32 lane) {     // n <= 32 (fewer than 32 bits wait in acc)
        const u64 m = n >= 32 ? 0xFFFFFFFFull : ((1ull << n) - 1ull);
        acc |= ((u64)v & m) << nbits;
        nbits += n;
        if (nbits >= 32) {
            if (pos + 4 <= cap) { if (lane == 0) *reinterpret_cast<u32*>(p + pos) = (u32)acc; } else ovf = 1;      // (any alignment)
            pos += 4; acc >>= 32; nbits -= 32;
        }
    }
    __device__ __forceinline__ void put(u64 v, u32 lane) {
        opened = 1;
        const u32 k = rice_k(A, N);
        const u64 q = v >> k;
        if (q < RICE_ESC) { bits((u32)((1ull << q) - 1ull), (u32)q, lane); bits(0, 1, lane); bits((u32)v, k, lane); }
        else { bits(0xFFFFFFFFu, 32, lane); bits((u32)v & 0xFFFFFu, 20, lane); bits((u32)(v >> 20) & 0xFFFFFu, 20, lane); }
        A += v; N++;
        if (N >= 32) { A >>= 1; N >>= 1; }
    }
    __device__ __forceinline__ u32 finish(u32 lane) {                  // the stream's size
        if (!opened) return 0;
        put(0, lane);
        while (nbits) {
            if (pos < cap) { if (lane == 0) p[pos] = (u8)acc; } else ovf = 1;
            pos++; acc >>= 8; nbits = nbits > 8 ? nbits - 8 : 0;
        }
        return pos;
    }
};

struct RiceR {                 // a lane's bit source: zeros behind the stream's end
    const u8* p; u32 n, pos; u64 acc; u32 nbits; u64 A; u32 N; u32 valid, err;
    __device__ __forceinline__ u32 word(u32 at) const {               // four stream bytes from `at`, nothing read outside [p, p + n)
        if (at + 4 <= n) return *reinterpret_cast<const u32*>(p + at);
        u32 v = 0;
        for (u32 i = 0; i < 4; i++) if (at + i < n) v |= (u32)p[at + i] << (8 * i);
        return v;
    }
    __device__ __forceinline__ void init(const u8* src, u32 len) { p = src; n = len; pos = 0; acc = 0; nbits = 0; A = RICE_A0; N = 1; valid = len > 0; err = 0; }
    __device__ __forceinline__ void fill() { if (nbits <= 32) { acc |= (u64)word(pos) << nbits; pos += 4; nbits += 32; } }    // at least 33 bits behind it
    __device__ __forceinline__ u32 take(u32 k) {                      // k <= 32
        const u64 m = k >= 32 ? 0xFFFFFFFFull : ((1ull << k) - 1ull);
        const u32 v = (u32)(acc & m);
        acc >>= k; nbits -= k;
        return v;
    }
    __device__ __forceinline__ u64 get() {                            // the next gap; 0 = the list's end (or no list at all)
        if (!valid) return 0;
        if (pos > n + 16) { err = 1; valid = 0; return 0; }           // (a stream without its end)
        const u32 k = rice_k(A, N);
        fill();
        const u32 low = (u32)acc;
        u32 q = low == 0xFFFFFFFFu ? 32u : (u32)__ffs((int)~low) - 1u;
        u64 v;
        if (q < RICE_ESC) {
            take(q + 1);
            fill();
            v = ((u64)q << k) | take(k);
        } else {
            take(32);
            fill(); v = take(20);
            fill(); v |= (u64)take(20) << 20;
        }
        if (!v) {
            // (the list's end must lie INSIDE the stream: zeros are read behind it, and a stream cut short would end its list there
            //  silently -- its last N's missing -- ADVICE, round 4)
            if ((u64)pos * 8u - nbits > (u64)n * 8u) err = 1;
            valid = 0; return 0;
        }
        A += v; N++;
        if (N >= 32) { A >>= 1; N >>= 1; }
        return v;
    }
};
