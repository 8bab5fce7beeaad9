// dev_coder.h -- the carry-less range coder (reference: coder.hpp) as lane-serial device code.
// One instance = one serial coder chain.  All arithmetic is unsigned integer and bit-exact with the
// reference: u64 low/code, u32 range, u32 wrap of cum*range (coder.hpp:68-70, 88-92).
#pragma once
#include <hip/hip_runtime.h>
#include "dev_common.h"

#define RC_TOP (1u << 24)   // coder.hpp:24

// Byte sink replacing FilerSave::put (filer.hpp:70-75): a bounded region of the scratch arena.
// pos keeps counting past cap so the caller can detect (and size) an overflow.
struct ByteSink {
    u8* p;
    u32 pos;
    u32 cap;
    __device__ __forceinline__ void put(u8 b) {
        if (pos < cap) p[pos] = b;
        pos++;
    }
};

// Byte source replacing FilerLoad::get (filer.hpp:94-97): 0 past the end.  The decoders are one serial chain per
// lane and every renormalisation needs the next stream byte, so the bytes are fetched eight at a time: one memory
// round trip per eight bytes instead of one per byte (the last seven bytes of a stream are read singly, so
// nothing beyond the stream is ever touched).
struct ByteSrc {
    const u8* p;
    u32 pos;
    u32 n;
    u64 buf;      // bytes already fetched, next one in the low byte
    u32 have;     // how many of them
    __device__ __forceinline__ void init(const u8* ptr, u32 len) { p = ptr; pos = 0; n = len; buf = 0; have = 0; }
    __device__ __forceinline__ u32 get() {
        if (have == 0) {
            if (pos + 8 <= n) {
                const u32* q = reinterpret_cast<const u32*>(p + pos);      // global loads need no alignment on gfx9
                buf = (u64)q[0] | ((u64)q[1] << 32);
                have = 8;
            } else { buf = pos < n ? p[pos] : 0u; have = 1; }
        }
        const u32 b = (u32)buf & 0xffu;
        buf >>= 8; have--;
        pos++;
        return b;
    }
};

// The same for code that runs wave-UNIFORM (every lane the same stream position): there the compiler turns the eight-byte
// fetch above into scalar loads, and scalar loads drop the low two address bits -- an unaligned stream would be read from
// the wrong place.  Byte loads have no such trap.
struct ByteSrc1 {
    const u8* p;
    u32 pos;
    u32 n;
    __device__ __forceinline__ void init(const u8* ptr, u32 len) { p = ptr; pos = 0; n = len; }
    __device__ __forceinline__ u32 get() { const u32 b = pos < n ? p[pos] : 0u; pos++; return b; }
};

struct RcEnc {
    u64 low;
    u32 range;
    u32 err;
    __device__ __forceinline__ void init() { low = 0; range = 0xFFFFFFFFu; err = 0; }   // coder.hpp:34-39

    // coder.hpp:66-81
    __device__ __forceinline__ void encode(ByteSink& s, u32 cum, u32 freq, u32 tot) {
        u32 r = range / tot;
        low += (u64)(u32)(cum * r);
        range = r * freq;
        int guard = 0;
#pragma nounroll
        while (range < RC_TOP) {
            if ((low ^ (low + range)) >> 56) range = (((u32)low | (RC_TOP - 1)) - (u32)low);
            s.put((u8)(low >> 56));
            range <<= 8;
            low <<= 8;
            // the reference spins forever if the clamp yields range 0; every chain here must drain
            if (++guard > 12) { err = 1; range = 0xFFFFFFFFu; break; }
        }
    }
    // coder.hpp:52-61
    __device__ __forceinline__ void done(ByteSink& s) {
        for (int i = 0; i < 8; i++) { s.put((u8)(low >> 56)); low <<= 8; }
    }
};

struct RcDec {
    u64 low, code;
    u32 range;
    u32 err;
    // coder.hpp:41-49
    template <typename SRC>
    __device__ __forceinline__ void init(SRC& s) {
        low = 0; range = 0xFFFFFFFFu; code = 0; err = 0;
        for (int i = 0; i < 8; i++) code = (code << 8) | s.get();
    }
    // coder.hpp:83-86.  code < 2^32 in every well-formed stream; the 64-bit divide keeps corrupt ones defined.
    __device__ __forceinline__ u32 get_freq(u32 tot) {
        range /= tot;
        if (range == 0) { err = 1; range = 1; }
        if ((code >> 32) == 0) return (u32)code / range;
        return (u32)(code / range);
    }
    // coder.hpp:88-102
    template <typename SRC>
    __device__ __forceinline__ void decode(SRC& s, u32 cum, u32 freq) {
        u32 temp = cum * range;
        low += temp;
        code -= temp;
        range *= freq;
        int guard = 0;
#pragma nounroll
        while (range < RC_TOP) {
            if ((low ^ (low + range)) >> 56) range = (((u32)low | (RC_TOP - 1)) - (u32)low);
            code = (code << 8) | s.get();
            range <<= 8;
            low <<= 8;
            if (++guard > 12) { err = 1; range = 0xFFFFFFFFu; break; }
        }
    }
};
