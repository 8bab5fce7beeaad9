// gm.hip -- the bases of the block format's frozen mode under the generation MATCH model (round 5).
//
// The reference's base model (gens.cpp:36-44, base2_ranger.hpp) is a table of 2^24 rows addressed by the last twelve bases: one
// random memory access per base -- 1.5 G of them per 10 M-read call, and the chip serves 50-60 G random 64-byte sectors a second
// from tables beyond its L2 whatever their layout (profiles/r02a_atomics_*): 30 ms of a call whose other models take 13.  What such
// a table learns from reads that overlap is what an EARLIER READ of the same place says outright, so a chain of generation g
// follows a pointer into the bases of the generations before it and reads its predictions SEQUENTIALLY:
//   stage    the call's base lines as letters (N as 'A', gens.cpp:116-136), a '\n' behind every line: position soff[r] + i = base i
//            of record r, soff[r] = (bases before record r) + r.  The decoder's staged bases are exactly this; the encoder makes a copy
//            (k_gm_stage) so that both sides speak of the same positions;
//   index    2^tb entries of 8 bytes.  Over the records the counting passes take (every stride-th of a generation), the k-mer of 16
//            bases ending at base i (i + 1 >= 16, i + 1 < len) is hashed, h = kmer * 0x9E3779B97F4A7C15; a quarter of them (top two
//            bits zero) go to entry (h >> (62 - tb)) & (2^tb - 1) as  min(entry, p << 24 | check),  p = the position of base i + 1,
//            check = the next 24 bits of h: the EARLIEST occurrence stays whatever the order of the atomics, and a chain of
//            generation g takes an entry only if p lies below its generation's first position -- one table serves every generation
//            of an encode at once, and a decoder that has indexed the generations before g sees exactly the entries g may take;
//   walk     sfq_oracle.c "the generation MATCH model" has the rule: with a pointer the base under it is coded at 4096 - 3 Fo[m] of
//            4096 (m = bases matched in a row), a miss resets m, a second miss within eight bases or a '\n' drops the pointer;
//            without one the base is coded flat and the k-mers of the sampled quarter are looked up; a pointer found behind base i
//            predicts from base i + 1 + GM_D on (GM_D = 1: a decoder reads the entry behind base i, looks at it behind base i + 1 and
//            has the base between them to hide the round trip; 4 was measured no faster -- a lone wave's base takes 1 us, as long as
//            a round trip -- and costs 0.08 bit a base more).
// Encode in two passes -- k_gm_plan: a RECORD per lane (ten million lanes: the lookups' round trips hide behind each other) writes
// a token per base (pointer or not, Fo level, predicted base); k_gm_code: a CHAIN per lane codes bases + tokens, no random access.
// Decode: k_gm_decode_c, a chain per lane, generation by generation, k_gm_insert behind each.
// The rule is restated in oracle/sfq_oracle.c (sfqo_gm_encode_chains / _segs) and compared bit for bit (tests/test_frozen_tables.py).
#include <algorithm>
#include "kernels.h"
#include "dev_chain.h"
#include "dev_walk.h"

#define GM_K 16u
#define GM_D 1u                          // bases between a lookup and its pointer's first prediction
#define GM_DROP 8u
#define GM_MCAP 31u
#define GM_HASH 0x9E3779B97F4A7C15ull
#define GM_EMPTY (~0ull)
#define GM_BITS 12u                      // every row totals 4096: range / total is a shift

__device__ __forceinline__ u32 gm_level(u32 m) { return m < 4u ? 0u : m < 8u ? 1u : m < 16u ? 2u : 3u; }
__device__ __forceinline__ u32 gm_fo_of_level(u32 lv) { return (0x10182840u >> (8u * lv)) & 0xffu; }       // 64, 40, 24, 16
// a staged letter -> its code ("ACGT": bits 1..2 of the letter are 0, 1, 3, 2; "0123": the low two bits)
__device__ __forceinline__ u32 gm_code(u32 byte, u32 solid) {
    const u32 x = (byte >> 1) & 3u, sm = 0u - (solid & 1u);             // (masks, not a select: `solid` differs from lane to lane, and the
    return (byte & 3u & sm) | ((x ^ (x >> 1)) & ~sm);                    //  compiler makes an exec-mask branch of a select between two computed values)
}
__device__ __forceinline__ u32 gm_byte_at(const uint4& w, u32 idx) {          // idx 0..15, lane-variable: two selects and a 64-bit shift
    const u64 lo = (u64)w.x | ((u64)w.y << 32), hi = (u64)w.z | ((u64)w.w << 32);      // (written with selects of dwords the compiler keeps the window in LDS)
    const u64 h = (idx & 8u) ? hi : lo;
    return (u32)(h >> ((idx & 7u) * 8u)) & 0xffu;
}
// a '\n' among bytes 0 .. GM_D of the window (GM_D <= 7)
__device__ __forceinline__ bool gm_newline_ahead(const uint4& w) {
    constexpr u64 keep = GM_D >= 7u ? ~0ull : ((1ull << (8u * (GM_D + 1u))) - 1ull);
    const u64 a = (((u64)w.x | ((u64)w.y << 32)) ^ 0x0A0A0A0A0A0A0A0Aull) | ~keep;              // a zero byte = a '\n' among the bytes kept
    return ((a - 0x0101010101010101ull) & ~a & 0x8080808080808080ull) != 0ull;
}
// sixteen bytes at `at` of a buffer that is readable up to cap + 16: ONE load whatever the place (the address is held at cap; what a
// window holds beyond the buffer's last line is never looked at -- a pointer stops at its line's '\n').  A load with a second path for the
// buffer's tail made the compiler wait for every window where it is issued: the round trip each prefetch is there to hide.
__device__ __forceinline__ uint4 gm_ld16(const u8* __restrict__ buf, u64 at, u64 cap) {
    const u32* q = reinterpret_cast<const u32*>(buf + (at < cap ? at : cap));          // (no alignment needed on gfx9)
    return make_uint4(q[0], q[1], q[2], q[3]);
}
__device__ __forceinline__ u64 gm_hash(u32 kmer) { return (u64)kmer * GM_HASH; }
__device__ __forceinline__ bool gm_sampled(u64 h) { return (h >> 62) == 0; }
__device__ __forceinline__ u64 gm_slot(u64 h, u32 tb) { return (h >> (62u - tb)) & ((1ull << tb) - 1ull); }
__device__ __forceinline__ u32 gm_check(u64 h, u32 tb) { return (u32)(h >> (38u - tb)) & 0xFFFFFFu; }

// ---- encoder: the stage ---------------------------------------------------------------------------------------------
// base-line lengths of records [0, n)
__global__ __launch_bounds__(256) void k_gm_lens(const u64* __restrict__ line_off, const BlockDesc* __restrict__ blocks, u32 block_reads, u64 n, u32* __restrict__ slen) {
    const u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const u32 solid = blocks[r / block_reads].solid;
    const u64 g0 = line_off[4 * r + 1] + solid, g1 = line_off[4 * r + 2] - 1;
    slen[r] = g1 > g0 ? (u32)(g1 - g0) : 0u;
}
// soff[r] = boff[r] + r  (a sentinel behind every line before it), r in [0, n]
__global__ __launch_bounds__(256) void k_gm_soff(const u64* __restrict__ boff, u64 n, u64* __restrict__ soff) {
    const u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
    if (r <= n) soff[r] = boff[r] + r;
}
// the '\n' behind every line of records [0, n) (decoder: before anything is decoded)
__global__ __launch_bounds__(256) void k_gm_sentinels(u8* __restrict__ stage, const u64* __restrict__ soff, const u32* __restrict__ slen, u64 n) {
    const u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
    if (r < n) stage[soff[r] + slen[r]] = '\n';
}
// records [r0, r1): sixteen lanes per record copy the line sixteen bytes a lane and step, letter by letter as the decoder will restore
// it ("ACGT" / "0123", N-like as 'A': a 256-entry table in LDS), mark the record for the exception pass (an N, a lowercase base) and
// report an illegal character (gens.cpp:125-126).  (A wavefront per record, four bytes a lane: 6.0 ms per 10 M reads; this: see DESIGN.)
__global__ __launch_bounds__(256) void k_gm_stage(ModelArgs m, u32 block_reads, u64 r0, u64 r1, u64 nbytes, u8* __restrict__ stage, const u64* __restrict__ soff, const u32* __restrict__ slen, u8* exc_flag) {
    __shared__ u16 lut[2][256];                               // [solid][character] -> letter | flags << 8 (4 N-like, 0x10 illegal, 0x20 lowercase)
    for (u32 i = threadIdx.x; i < 512; i += 256) {
        const u32 c = i & 255u, solid = i >> 8;
        const u32 cd = gen_code_of(c) | (is_lower_base(c) ? 0x20u : 0u);
        lut[solid][c] = (u16)((((solid ? 0x33323130u : 0x54474341u) >> (8u * (cd & 3u))) & 0xffu) | ((cd & 0x34u) << 8));
    }
    __syncthreads();
    const u32 sub = threadIdx.x & 15u;
    const u64 ngroups = (u64)gridDim.x * 16u;
    for (u64 r = r0 + (u64)blockIdx.x * 16u + (threadIdx.x >> 4); r < r1; r += ngroups) {
        const u32 b = (u32)(r / block_reads);
        const u32 solid = m.blocks[b].solid;
        const u64 src = m.line_off[4 * r + 1] + solid;
        u8* dst = stage + soff[r];
        const u32 len = slen[r];
        const u16* const lt = lut[solid];
        u32 odd = 0;
        for (u32 i = sub * 16u; i < len; i += 256u) {
            const uint4 v = load16(m.fq, nbytes, src + i);
            const u32 k = len - i < 16u ? len - i : 16u;
            u32 o[4];
#pragma unroll
            for (u32 t = 0; t < 16; t++) {
                const u32 e = lt[piece_byte(v, t)];
                if (t < k) odd |= e;
                if ((t & 3u) == 0) o[t >> 2] = e & 0xffu; else o[t >> 2] |= (e & 0xffu) << ((t & 3u) * 8u);
            }
            if (k == 16u) { u32* d32 = reinterpret_cast<u32*>(dst + i); d32[0] = o[0]; d32[1] = o[1]; d32[2] = o[2]; d32[3] = o[3]; }
            else for (u32 t = 0; t < k; t++) dst[i + t] = (u8)(o[t >> 2] >> ((t & 3u) * 8u));
        }
        if (sub == 0) dst[len] = '\n';
        // the sixteen lanes' flags together (xor-shuffles inside the group of sixteen)
#pragma unroll
        for (int d = 8; d > 0; d >>= 1) odd |= (u32)__shfl_xor((int)odd, d, 64);
        if (sub == 0) {
            if ((odd & 0x3400u) && exc_flag) exc_flag[r] = 1;
            if (odd & 0x1000u) atomicMax(&m.blocks[b].status, (u32)(-SFQ_E_GENCHAR));
        }
    }
}

// ---- the index ------------------------------------------------------------------------------------------------------
// the counted records of blocks [b0, b1) (every stride-th; a lane per record, or per stretch of seg_len bases of a long line with
// the sixteen bases before it run through the k-mer alone), read from the stage
__global__ __launch_bounds__(256) void k_gm_insert(ChainArgs a, u32 b0, u32 b1, u32 stride, u32 seg_len, u32 segs, u64* __restrict__ T, u32 tb) {
    const u64 first = a.m.blocks[b0].rec0, endr = a.m.blocks[b1 - 1].rec0 + a.m.blocks[b1 - 1].nrec;
    const u64 id = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 r = first + (id / segs) * stride;
    const u32 seg = (u32)(id % segs);
    const bool live = r < endr;
    u32 solid = 0, len = 0;
    if (live) { solid = a.m.blocks[(u32)(r / a.block_reads)].solid; len = a.st_len[r]; }
    const u32 warm = (seg_len && seg) ? 16u : 0u;
    LineWalk lw;
    if (seg_len) lw.init(a, r, live ? 1u : 0u, 1, 0, (u64)seg * seg_len - warm, (u64)seg_len + warm);
    else lw.init(a, r, live ? 1u : 0u, 1, 0);
    const u64 line0 = live ? a.st_off[r] : 0;
    u32 kmer = 0, seen = 0;
    Piece pc = lw.next();
    uint4 w = lw.fetch(pc);
    while (__any(pc.valid)) {
        const Piece pn = lw.next();
        const uint4 wn = lw.fetch(pn);
#pragma unroll
        for (u32 j = 0; j < 16; j++) {
            if (j < pc.j1) {
                kmer = (kmer << 2) | gm_code(piece_byte(w, j), solid);
                seen++;
                const u64 p = pc.at + j + 1;                               // the position of the base behind the k-mer
                if (seen >= GM_K && seen > warm && p - line0 < len && !(p >> 40)) {
                    const u64 h = gm_hash(kmer);
                    if (gm_sampled(h)) atomicMin((unsigned long long*)&T[gm_slot(h, tb)], (unsigned long long)((p << 24) | gm_check(h, tb)));
                }
            }
        }
        pc = pn; w = wn;
    }
}

// ---- the walk ----------------------------------------------------------------------------------------------------------------
// One routine for the encoder's plan, the verdict's price and the decoder: the state of a lane, what it predicts for base i
// (gm_predict), and what base i does to it (gm_update).  WITHOUT BRANCHES but for the loads: a lane that walks its chain alone -- the
// early generations of a decode are a few thousand lanes on a chip that holds half a million -- pays 20-40 cycles for every exec-mask
// branch, and the first version of this walk, written with ifs, had 37 of them per base: 0.9 us a base, 6.5 ms for a generation of
// 3 328 chains that decode nothing but flat bases.  The entry of a lookup is read behind base i and looked at behind base i + 1 --
// before anything could depend on it but the lookup of base i + 1, which the rule forbids while a pointer is pending -- so the order
// of events is the oracle's, and a lone wave has a base's worth of instructions between the load and its use.
struct GmCosts { u16 hit[4], miss[4]; };          // 1/1024 bit: a predicted base that comes / does not come, by Fo level
// P: the type of a stage position -- u32 where the stage is below 4 GiB (every call of up to 4 G bases: half the 64-bit arithmetic of a
// step gone), u64 else
template <typename P>
struct GmWalk {
    u32 kmer, m, pend_at, ask_check;
    u32 have, asked;                              // masks: all ones / zero
    P ptr, swb, pend_p; u64 ent;
    uint4 sw, swn;                                // the sixteen bytes at swb, the sixteen behind them (asked for halfway through sw)
    __device__ __forceinline__ void reset() {
        kmer = 0; m = 0; pend_at = ~0u; ask_check = 0; have = 0; asked = 0; ptr = 0; swb = 0; pend_p = 0; ent = GM_EMPTY;
        sw = make_uint4(0, 0, 0, 0); swn = make_uint4(0, 0, 0, 0);
    }
};
__device__ __forceinline__ u64 selp(u32 mask, u64 a, u64 b) { return (a & (u64)(i64)(i32)mask) | (b & ~(u64)(i64)(i32)mask); }
__device__ __forceinline__ u32 selp(u32 mask, u32 a, u32 b) { return (a & mask) | (b & ~mask); }
// what the walk says of base i: have (mask), the predicted base e, its Fo level lv; fo / fm are the frequencies of a base that is not /
// that is the predicted one (both 1024 without a pointer)
template <typename P>
__device__ __forceinline__ void gm_predict(GmWalk<P>& W, u32 i, const u8* __restrict__ stage, u64 cap, u32 solid, u32& e, u32& lv, u32& fo, u32& fm) {
    // a pointer whose time has come (its sixteen bytes are in sw since the entry was looked at): not across a line's end
    const u32 act = W.pend_at == i ? ~0u : 0u;
    const u32 ok = act & (gm_newline_ahead(W.sw) ? 0u : ~0u);
    W.have |= ok;
    W.m = ok ? GM_K : W.m;
    W.ptr = selp(ok, (P)(W.pend_p + GM_D), W.ptr);
    W.pend_at |= act;
    // the window: taken over where the pointer has left it, the one behind it asked for halfway through
    u32 o = (u32)W.ptr - (u32)W.swb;
    const u32 sh = W.have & (o >= 16u ? ~0u : 0u);
    W.sw.x = sh ? W.swn.x : W.sw.x; W.sw.y = sh ? W.swn.y : W.sw.y; W.sw.z = sh ? W.swn.z : W.sw.z; W.sw.w = sh ? W.swn.w : W.sw.w;
    W.swb += sh & 16u;
    o -= sh & 16u;
    const bool want = W.have && o == 8u;
    if (__any(want)) { if (want) W.swn = gm_ld16(stage, W.swb + 16u, cap); }
    const u32 sb = gm_byte_at(W.sw, o & 15u);
    W.have &= sb == '\n' ? 0u : ~0u;
    e = gm_code(sb, solid);
    lv = gm_level(W.m);
    fo = W.have ? gm_fo_of_level(lv) : 1024u;
    fm = W.have ? 4096u - 3u * fo : 1024u;
}
// base i was b (n = the line's bases, lim = the first stage position of the lane's generation)
template <typename P>
__device__ __forceinline__ void gm_update(GmWalk<P>& W, u32 i, u32 n, u32 b, u32 e, P lim, const u64* __restrict__ T, u32 tb, const u8* __restrict__ stage, u64 cap) {
    const u32 hit = b == e ? ~0u : 0u;
    const u32 drop = W.have & ~hit & (W.m < GM_DROP ? ~0u : 0u);
    W.m = hit ? (W.m < GM_MCAP ? W.m + 1u : W.m) : 0u;                  // (without a pointer m does not matter)
    W.have &= ~drop;
    W.ptr += W.have & 1u;
    W.kmer = (W.kmer << 2) | b;
    // the entry read behind the base before this one
    const u32 found = W.asked & ~W.have & (W.ent != GM_EMPTY ? ~0u : 0u) & ((u32)(W.ent & 0xFFFFFFull) == W.ask_check ? ~0u : 0u) & ((W.ent >> 24) < (u64)lim ? ~0u : 0u);
    W.asked = 0;
    if (__any(found != 0u)) {
        if (found) {                                                     // read behind base i - 1: the pointer predicts from base i + GM_D on
            W.pend_at = i + GM_D; W.pend_p = (P)(W.ent >> 24); W.swb = W.pend_p;
            W.sw = gm_ld16(stage, W.pend_p, cap);
        }
    }
    const u32 elig = (lim != 0 && !W.have && W.pend_at == ~0u && i + 1u >= GM_K && i + 1u + GM_D < n) ? ~0u : 0u;
    if (__any(elig != 0u)) {
        const u64 h = gm_hash(W.kmer);
        const u32 samp = elig & (gm_sampled(h) ? ~0u : 0u);
        if (samp) W.ent = T[gm_slot(h, tb)];
        W.ask_check = gm_check(h, tb); W.asked = samp;
    }
}
// One line (or segment) of n bases at stage position q0, the bases read from the stage (encoder).
// PRICE: the cost in 1/1024 bit is returned, nothing written; else a token per base at tok[q0 + i]:
//   0 = coded flat;  0x80 | level << 2 | e = predicted base e at Fo level `level`
template <bool PRICE, typename P>
__device__ __forceinline__ u64 gm_plan_line(const u8* __restrict__ stage, u64 stage_bytes, u64 q0, u32 n, P lim, u32 solid, const u64* __restrict__ T, u32 tb,
                                            u8* __restrict__ tok, const GmCosts& gc) {
    GmWalk<P> W; W.reset();
    const u64 hitc = (u64)gc.hit[0] | ((u64)gc.hit[1] << 16) | ((u64)gc.hit[2] << 32) | ((u64)gc.hit[3] << 48);
    const u64 misc = (u64)gc.miss[0] | ((u64)gc.miss[1] << 16) | ((u64)gc.miss[2] << 32) | ((u64)gc.miss[3] << 48);
    u64 cost = 0;
    for (u32 i0 = 0; i0 < n; i0 += 16u) {
        const uint4 w = gm_ld16(stage, q0 + i0, stage_bytes);
        const u32 cnt = n - i0 < 16u ? n - i0 : 16u;
        u32 tk[4] = {0, 0, 0, 0};
#pragma unroll
        for (u32 j = 0; j < 16; j++) {
            if (j < cnt) {
                const u32 i = i0 + j;
                const u32 b = gm_code(piece_byte(w, j), solid);
                u32 e, lv, fo, fm;
                gm_predict(W, i, stage, stage_bytes, solid, e, lv, fo, fm);
                if (PRICE) cost += W.have ? (u32)((b == e ? hitc : misc) >> (16u * lv)) & 0xffffu : 2048u;
                else tk[j >> 2] |= (W.have & (0x80u | (lv << 2) | e)) << ((j & 3u) * 8u);
                gm_update(W, i, n, b, e, lim, T, tb, stage, stage_bytes);
            }
        }
        if (!PRICE) {
            u8* t = tok + q0 + i0;
            if (cnt == 16u) { u32* t32 = reinterpret_cast<u32*>(t); t32[0] = tk[0]; t32[1] = tk[1]; t32[2] = tk[2]; t32[3] = tk[3]; }
            else for (u32 j = 0; j < cnt; j++) t[j] = (u8)(tk[j >> 2] >> ((j & 3u) * 8u));
        }
    }
    return cost;
}
// the generation of block b, and the first stage position of that generation
__device__ __forceinline__ u64 gm_limit(const ChainArgs& a, const u64* soff, u32 b) {
    if (!a.g_ngen) return 0;
    u32 g = 0;
    while (g + 1 < a.g_ngen && b >= a.g_bound[g + 1]) g++;
    return soff[a.m.blocks[a.g_bound[g]].rec0];
}
// the plan: lane = record (lanes [0, nrec)), or -- a.seg_len != 0 -- lane = chain (a segment of one record)
__global__ __launch_bounds__(256) void k_gm_plan(ChainArgs a, u64 nlanes, const u8* __restrict__ stage, u64 stage_bytes, const u64* __restrict__ soff, const u32* __restrict__ slen,
                                                 const u64* __restrict__ T, u32 tb, u8* __restrict__ tok) {
    const u64 id = (u64)blockIdx.x * 256 + threadIdx.x;
    if (id >= nlanes) return;
    GmCosts gc = {};
    u64 r = id, q0; u32 n;
    if (a.seg_len) {
        ChainPos cp = chain_pos(a, (u32)id);
        chain_seg_encode(a, cp);
        r = cp.r0;
        const u32 len = slen[r];
        const u64 lo = cp.sub_lo < len ? cp.sub_lo : len;
        n = (u32)(len - lo < cp.sub_len ? len - lo : cp.sub_len); q0 = soff[r] + lo;
    } else { n = slen[r]; q0 = soff[r]; }
    const u32 b = (u32)(r / a.block_reads);
    const u64 lim = gm_limit(a, soff, b);
    if (stage_bytes >> 32) gm_plan_line<false, u64>(stage, stage_bytes, q0, n, lim, a.m.blocks[b].solid, T, tb, tok, gc);
    else gm_plan_line<false, u32>(stage, stage_bytes, q0, n, (u32)lim, a.m.blocks[b].solid, T, tb, tok, gc);
}
// the verdict's price: every step-th record from r0 below r1, a lane per STRETCH of GM_PRICE_STRETCH bases of it -- each walked as a
// line of its own (no pointer, no k-mer at its start: the rule's, sfq_oracle.c; a lane per 30 kb read kept every chain of a long-read
// call waiting 35 ms for the verdict) --, all under the limit soff[lim_rec]; cost[0] += 1/1024 bits, cost[1] += bases
#define GM_PRICE_STRETCH 256u
__global__ __launch_bounds__(256) void k_gm_price(ChainArgs a, u64 r0, u64 r1, u64 step, u32 segs, u64 lim_rec, const u8* __restrict__ stage, u64 stage_bytes, const u64* __restrict__ soff,
                                                  const u32* __restrict__ slen, const u64* __restrict__ T, u32 tb, GmCosts gc, u64* cost) {
    const u64 id = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 r = r0 + (id / segs) * step;
    const u32 lo = (u32)(id % segs) * GM_PRICE_STRETCH;
    u64 c = 0; u32 n = 0;
    const u64 lim = soff[lim_rec];
    if (r < r1 && lo < slen[r]) {
        n = slen[r] - lo < GM_PRICE_STRETCH ? slen[r] - lo : GM_PRICE_STRETCH;
        c = gm_plan_line<true, u64>(stage, stage_bytes, soff[r] + lo, n, lim, a.m.blocks[(u32)(r / a.block_reads)].solid, T, tb, nullptr, gc);
    }
    u64 nb = n;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { c += __shfl_xor(c, d, 64); nb += __shfl_xor(nb, d, 64); }
    if ((threadIdx.x & 63) == 0 && nb) { atomicAdd((unsigned long long*)cost, (unsigned long long)c); atomicAdd((unsigned long long*)cost + 1, (unsigned long long)nb); }
}

// ---- encoder: the chains ----------------------------------------------------------------------------------------------------
#define GM_RING 8
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_gm_code(ChainArgs a, const u8* __restrict__ tok) {
    __shared__ u32 ring[LaneEncB<THREADS, GM_RING>::LDS_DWORDS];
    const u32 c = blockIdx.x * THREADS + threadIdx.x;
    const bool live = c < a.geo.nchains;
    ChainPos cp; cp.b = 0; cp.r0 = 0; cp.nrec = 0; cp.sub_lo = cp.sub_len = 0; cp.seg = 0; cp.nseg = 1;
    if (live) { cp = chain_pos(a, c); chain_seg_encode(a, cp); }
    const BlockDesc* d = &a.m.blocks[cp.b];
    const u32 solid = live ? d->solid : 0u;
    LaneEncB<THREADS, GM_RING> rc; u32 cap = 0;
    u8* outp = live ? chain_region(a, cp, SFQ_S_GEN, 3, 4, cap) : nullptr;
    rc.init(ring, threadIdx.x, outp, cap);
    LineWalk lw; lw.init(a, cp.r0, cp.nrec, 1, 0, cp.sub_lo, cp.sub_len);          // (the stage: a.st_*)
    Piece pc = lw.next();
    uint4 w = lw.fetch(pc);
    uint4 tw = gm_ld16(tok, pc.at, a.st_bytes);
    while (__any(pc.valid)) {
        const Piece pn = lw.next();
        const uint4 wn = lw.fetch(pn);
        const uint4 twn = gm_ld16(tok, pn.at, a.st_bytes);
        const u32 len = pc.j1;
#pragma unroll
        for (u32 j = 0; j < 16; j++) {
            const u32 b = gm_code(piece_byte(w, j), solid), t = piece_byte(tw, j);
            const u32 e = t & 3u, hv = t >> 7;
            const u32 fo = hv ? gm_fo_of_level((t >> 2) & 3u) : 1024u;
            const u32 fm = hv ? 4096u - 3u * fo : 1024u;
            const u32 cum = b * fo + (b > e ? fm - fo : 0u);
            const u32 freq = b == e ? fm : fo;
            rc.encode_bits_if(j < len ? ~0u : 0u, cum, freq, GM_BITS);
            if ((j & 3u) == 3u) rc.drain();
        }
        pc = pn; w = wn; tw = twn;
    }
    if (live) {
        a.csz[c] = rc.finish();
        if (rc.err & 2) atomicMax(&a.m.blocks[cp.b].status, (u32)(-SFQ_E_OVERFLOW));
        else if (rc.err) atomicMax(&a.m.blocks[cp.b].status, (u32)(-SFQ_E_CORRUPT));
    }
}

// ---- decoder ----------------------------------------------------------------------------------------------------------------
// the chains [c0, c1) of one generation; soff[lim_rec] = the generation's first stage position.  What a lookup needs from memory is in flight
// while the bases between it and its use are decoded: the entry for one base, the sixteen bytes at the pointer for four.
template <int THREADS, typename P>
__global__ __launch_bounds__(THREADS) void k_gm_decode_c(ChainArgs a, DecodeArgs da, u32 c0, u32 c1, u64 lim_rec, const u64* __restrict__ T, u32 tb, u64 stage_bytes) {
    __builtin_amdgcn_s_setprio(3);            // the generations are the decode's critical path and a few thousand lanes each: ahead of the quality decoder's waves on a SIMD (28.9 -> 27.2 ms per 10 M genome-sampled reads)
    const u32 c = c0 + blockIdx.x * THREADS + threadIdx.x;
    if (c >= c1) return;
    const P lim = (P)da.soff[lim_rec];
    ChainPos cp = chain_pos(a, c);
    const BlockDesc* d = &a.m.blocks[cp.b];
    LaneDecQ rc; rc.init(da.streams + a.coff[c], a.csz[c], reinterpret_cast<const u8*>(a.qesc));
    const u32 solid = d->solid;
    const u32 alphabet = solid ? 0x33323130u /* "0123" */ : 0x54474341u /* "ACGT" */;    // gens.cpp:173-178
    const u8* stage = da.seq_stage;
    u32 n_next = cp.nrec ? da.slen[cp.r0] : 0u; u64 off_next = cp.nrec ? da.soff[cp.r0] : 0ull;
    if (a.seg_len) {                                         // a segment of one record: its part of the base line
        const u32 ql = da.qlen[cp.r0];
        seg_range(cp, ql > n_next ? ql : n_next);
        const u64 lo = cp.sub_lo < n_next ? cp.sub_lo : n_next;
        n_next = (u32)(n_next - lo < cp.sub_len ? n_next - lo : cp.sub_len); off_next += lo;
    }
    for (u32 k = 0; k < cp.nrec; k++) {
        const u32 n = n_next; const u64 off = off_next;
        if (k + 1 < cp.nrec) { n_next = da.slen[cp.r0 + k + 1]; off_next = da.soff[cp.r0 + k + 1]; }
        LaneOut out; out.begin(da.seq_stage + off);
        GmWalk<P> W; W.reset();
        for (u32 i = 0; i < n; i++) {
            u32 e, lv, fo, fm;
            gm_predict(W, i, stage, stage_bytes, solid, e, lv, fo, fm);
            rc.top_up();
            u32 r;
            const u32 q = rc.get_freq_bits(GM_BITS, r);
            const u32 f0 = e == 0u ? fm : fo, f1 = e == 1u ? fm : fo, f2 = e == 2u ? fm : fo;
            const u32 k1 = f0, k2 = f0 + f1, k3 = k2 + f2;
            const u32 b = (q >= k1 ? 1u : 0u) + (q >= k2 ? 1u : 0u) + (q >= k3 ? 1u : 0u);
            const u32 cum = (b > 0u ? f0 : 0u) + (b > 1u ? f1 : 0u) + (b > 2u ? f2 : 0u);
            rc.decode(r, cum, b == e ? fm : fo);
            out.put((alphabet >> (8u * b)) & 0xffu);
            gm_update(W, i, n, b, e, lim, T, tb, stage, stage_bytes);
        }
        if (!a.seg_len) out.put('\n');                        // the line's sentinel (chains that are segments: k_gm_sentinels has written them)
        out.end();
    }
    if (rc.err) atomicMax(&a.m.blocks[cp.b].status, (u32)(-SFQ_E_CORRUPT));
}

// ---- launches -----------------------------------------------------------------------------------------------------------------
void launch_gm_lens(const u64* line_off, const BlockDesc* blocks, u32 block_reads, u64 n, u32* slen, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_gm_lens, dim3((u32)((n + 255) / 256)), dim3(256), 0, st, line_off, blocks, block_reads, n, slen);
}
void launch_gm_soff(const u64* boff, u64 n, u64* soff, hipStream_t st) {
    hipLaunchKernelGGL(k_gm_soff, dim3((u32)((n + 256) / 256)), dim3(256), 0, st, boff, n, soff);
}
void launch_gm_sentinels(u8* stage, const u64* soff, const u32* slen, u64 n, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_gm_sentinels, dim3((u32)((n + 255) / 256)), dim3(256), 0, st, stage, soff, slen, n);
}
void launch_gm_stage(const ModelArgs& m, u32 block_reads, u64 r0, u64 r1, u64 nbytes, u8* stage, const u64* soff, const u32* slen, u8* exc_flag, hipStream_t st) {
    if (r1 <= r0) return;
    const u32 grid = (u32)std::min<u64>((r1 - r0 + 15) / 16, 1u << 20);
    hipLaunchKernelGGL(k_gm_stage, dim3(grid), dim3(256), 0, st, m, block_reads, r0, r1, nbytes, stage, soff, slen, exc_flag);
}
void launch_gm_insert(const ChainArgs& a, u32 b0, u32 b1, u64 nrec_range, u32 max_line, u64* T, u32 tb, hipStream_t st) {
    if (!nrec_range) return;
    const u32 stride = gen_count_stride(nrec_range);
    u32 seg_len = 0;                                            // (as launch_gen_count: short stretches while the lanes fit the chip at once)
    if (max_line >= 64u) {
        const u64 nsel = (nrec_range + stride - 1) / stride;
        for (u32 sl = 32u; sl < max_line && sl <= 512u; sl *= 2u)
            if (nsel * ((max_line + sl - 1) / sl) <= 524288ull) { seg_len = sl; break; }
    }
    if (!seg_len && max_line > 1024u) seg_len = 512u;
    const u32 segs = seg_len ? (max_line + seg_len - 1) / seg_len : 1u;
    const u64 lanes = ((nrec_range + stride - 1) / stride) * segs;
    hipLaunchKernelGGL(k_gm_insert, dim3((u32)((lanes + 255) / 256)), dim3(256), 0, st, a, b0, b1, stride, seg_len, segs, T, tb);
}
void launch_gm_plan(const ChainArgs& a, u64 nlanes, const u8* stage, u64 stage_bytes, const u64* soff, const u32* slen, const u64* T, u32 tb, u8* tok, hipStream_t st) {
    if (nlanes) hipLaunchKernelGGL(k_gm_plan, dim3((u32)((nlanes + 255) / 256)), dim3(256), 0, st, a, nlanes, stage, stage_bytes, soff, slen, T, tb, tok);
}
void launch_gm_price(const ChainArgs& a, u64 r0, u64 r1, u64 step, u32 max_line, u64 lim_rec, const u8* stage, u64 stage_bytes, const u64* soff, const u32* slen, const u64* T, u32 tb,
                     const u16* costs /* hit[4], miss[4] */, u64* cost, hipStream_t st) {
    if (r1 <= r0 || !step) return;
    GmCosts gc; for (int i = 0; i < 4; i++) { gc.hit[i] = costs[i]; gc.miss[i] = costs[4 + i]; }
    const u32 segs = max_line ? (max_line + GM_PRICE_STRETCH - 1) / GM_PRICE_STRETCH : 1u;
    const u64 lanes = ((r1 - r0 + step - 1) / step) * segs;
    hipLaunchKernelGGL(k_gm_price, dim3((u32)((lanes + 255) / 256)), dim3(256), 0, st, a, r0, r1, step, segs, lim_rec, stage, stage_bytes, soff, slen, T, tb, gc, cost);
}
void launch_gm_code(const ChainArgs& a, const u8* tok, hipStream_t st) {
    constexpr int TH = 256;
    if (a.geo.nchains) hipLaunchKernelGGL(k_gm_code<TH>, dim3((a.geo.nchains + TH - 1) / TH), dim3(TH), 0, st, a, tok);
}
void launch_gm_decode_c(const ChainArgs& a, const DecodeArgs& da, u32 c0, u32 c1, u64 lim_rec, const u64* T, u32 tb, u64 stage_bytes, hipStream_t st) {
    constexpr int TH = 256;
    if (c1 > a.geo.nchains) c1 = a.geo.nchains;
    if (c1 <= c0) return;
    if (stage_bytes >> 32) hipLaunchKernelGGL((k_gm_decode_c<TH, u64>), dim3((c1 - c0 + TH - 1) / TH), dim3(TH), 0, st, a, da, c0, c1, lim_rec, T, tb, stage_bytes);
    else hipLaunchKernelGGL((k_gm_decode_c<TH, u32>), dim3((c1 - c0 + TH - 1) / TH), dim3(TH), 0, st, a, da, c0, c1, lim_rec, T, tb, stage_bytes);
}
