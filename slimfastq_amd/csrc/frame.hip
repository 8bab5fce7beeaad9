// frame.hip -- device-side FASTQ framing and packing.
//
// The reference frames records on one host core with a 1 MiB sliding buffer (UsrSave::get_record,
// usrs.cpp:303-390).  Here the whole text is resident in HBM and framing is a newline index, built in ONE pass over
// the text (k_frame: line start offsets, the '@' / '+' prefix checks of usrs.cpp:311,346, the marks of the exception pass),
// then the per-record line limits (usrs.hpp:34-36) from the index alone and one descriptor per record block (the
// first-record analysis of UsrSave::determine_record, usrs.cpp:186-267).  All of it is streaming, HBM-bound work.
#include "kernels.h"

#define HIP_KCHECK() do { } while (0)

// Block-wide exclusive scan of one u32 per thread (256 threads = 4 waves). Returns the exclusive
// prefix; *total receives the block sum.
__device__ __forceinline__ u32 block_excl_scan_256(u32 v, u32* lds /* >= 8 u32 */, u32* total) {
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u32 o = __shfl_up(incl, d, 64);
        if (lane >= (u32)d) incl += o;
    }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (u32 w = 0; w < 4; w++) { u32 s = lds[w]; if (w < wave) base += s; tot += s; }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// =========================================================================================================
// framing in ONE pass (round 4)
//
// k_count_newlines + the scan + k_write_newlines read the text twice and issued 1.76e9 wave instructions per 3.7 GB call (the second
// kernel: a block scan per 4 KiB, a divergent loop per dword for the offsets, byte-wise attribution of the marks) -- 2.4 ms, and
// k_validate_records another 0.8 ms of scattered byte reads.  k_frame reads the text once:
//   * a thread owns 64 CONTIGUOUS bytes (four 16-byte loads): one newline count, one block scan per 16 KiB tile;
//   * a tile's place in the file comes from a decoupled look-back over the tiles before it (tstat: flag | value in one 64-bit
//     word, so no fence is needed) instead of a counting pass and a scan kernel;
//   * newlines, '!' candidates and odd-base candidates are 64-bit masks per thread (a SWAR test per dword, four flag bits gathered
//     with one multiply), so offsets and marks are a few bit operations per LINE END in the window, not per byte;
//   * the '@' / '+' checks of UsrSave::get_record (usrs.cpp:311, 346) ride on the line ends: the byte behind a newline is in cache.
// The line index must be sized before the number of lines is known: the caller guesses (cap entries), the kernel never writes
// past it and reports the count; a text of shorter lines than guessed is framed again with the exact size.
// =========================================================================================================
struct FrameOut { u64 nlines; u32 guard_tripped; u32 ticket; };      // (zeroed by the caller: ticket = the tile number a starting workgroup takes)
__device__ __forceinline__ u32 flags4(u32 m) { return (m * 0x00204081u) >> 28; }      // the 0x80 flags of four bytes as four bits
__device__ __forceinline__ u64 wave_sum_u64(u64 v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
#define TS_AGG  (1ull << 62)
#define TS_PFX  (2ull << 62)
#define TS_VAL  ((1ull << 62) - 1)
// A workgroup takes a TILE of FRAME_TILE bytes in FRAME_WIN sub-tiles of 16 KiB.  Round one LOADS the text -- coalesced, 16 bytes a
// lane, a wave's 64 lanes one KiB -- and keeps of every 64-byte window five 64-bit masks in LDS: newlines, '@', '+', '!' candidates,
// odd-base candidates (40 bytes per 64 of text: 40 KiB a tile); its newline count is what the tiles behind it wait for.  Round two --
// the tile's place in the file known from the look-back -- runs FROM THE MASKS: a thread takes a window, writes offsets and marks per
// line end, and checks the '@' / '+' behind a line end in the masks too.  The text is read ONCE (round 5).
// (Round 4 read it twice -- the second time "out of L2", which the counters did not bear out: TCC_MISS 6.1e7 of 7.1e7 requests, 7.4 GB
//  fetched per 3.7 GB of text after the guide's gfx950 correction: 256 workgroups x 128 KiB in flight are four times the L2s.  Measured
//  on the way, per 3.7 GB: a thread loading its own 64 contiguous bytes straight from memory 1.9-2.0 ms; one round with every window's
//  masks kept in registers: 158-288 VGPRs.)
// Tiles are numbered by a TICKET taken when a workgroup starts, not by blockIdx: a tile's predecessors in the look-back are then
// always running or through, whatever order the hardware starts workgroups in (ADVICE, round 4).
#ifndef FRAME_WIN
#define FRAME_WIN 4
#endif
#define FRAME_SUB 16384u
#define FRAME_TILE (FRAME_SUB * FRAME_WIN)
#define FRAME_NW (FRAME_TILE / 64u)      /* windows of a tile */
__device__ __forceinline__ u32 eq_flags(u32 x, u32 c4) {    // 0x80 where the byte equals the one c4 repeats four times
    const u32 y = x ^ c4;
    return ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu);
}
__device__ __forceinline__ u32 nl_flags(u32 x) { return eq_flags(x, 0x0a0a0a0au); }
__device__ __forceinline__ u32 mask16(u32 fx, u32 fy, u32 fz, u32 fw) { return flags4(fx) | (flags4(fy) << 4) | (flags4(fz) << 8) | (flags4(fw) << 12); }
template <bool MARKS>
__global__ __launch_bounds__(256) void k_frame(const u8* __restrict__ fq, u64 n, u64* __restrict__ tstat, u64* __restrict__ line_off, u64 cap,
                                               u32* __restrict__ status, u8* __restrict__ exc_flag, u64 ecap, FrameOut* __restrict__ fo) {
    constexpr u32 NM = MARKS ? 5u : 3u;
    __shared__ u64 mk[NM][FRAME_NW];                         // [newline, '@', '+', '!' candidate, odd-base candidate][window]
    __shared__ u32 wtot[4];
    __shared__ u32 s_last[4];
    __shared__ u64 s_base;
    __shared__ u32 s_tile;
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(&fo->ticket, 1u);
    __syncthreads();
    const u64 tile = s_tile;
    const u64 tb = tile * FRAME_TILE;
    // a sub-tile's text, coalesced: piece j of a thread = bytes [sb + 4096 j + 16 tid, + 16); bytes at or past n read as 0
    auto fetch = [&](u64 sb, uint4 (&v)[4]) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u64 pos = sb + 4096u * j + 16u * tid;
            if (pos + 16 <= n) { const u32* q = reinterpret_cast<const u32*>(fq + pos); v[j] = make_uint4(q[0], q[1], q[2], q[3]); }      // (no alignment needed on gfx9)
            else {
                u32 w[4] = {0, 0, 0, 0};
#pragma nounroll
                for (u32 b = 0; b < 16; b++) if (pos + b < n) w[b >> 2] |= (u32)fq[pos + b] << (8 * (b & 3));
                v[j] = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
    };
    // ---- round 1: the text, once -- the windows' masks into LDS, the tile's newlines counted -------------------------------------------
    u32 cnt = 0;
    {
        uint4 nx[4];
        fetch(tb, nx);
#pragma nounroll
        for (u32 wi = 0; wi < FRAME_WIN; wi++) {
            uint4 v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = nx[j];
            if (wi + 1 < FRAME_WIN) fetch(tb + (u64)FRAME_SUB * (wi + 1), nx);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // piece j of thread tid lies in window 64 j + tid / 4 of the sub-tile, at bytes 16 (tid % 4)
                const u32 W = wi * 256u + 64u * j + (tid >> 2), q = tid & 3u;
                const uint4 x = v[j];
                const u32 nl = mask16(nl_flags(x.x), nl_flags(x.y), nl_flags(x.z), nl_flags(x.w));
                cnt += (u32)__popc(nl);
                reinterpret_cast<u16*>(&mk[0][W])[q] = (u16)nl;
                reinterpret_cast<u16*>(&mk[1][W])[q] = (u16)mask16(eq_flags(x.x, 0x40404040u), eq_flags(x.y, 0x40404040u), eq_flags(x.z, 0x40404040u), eq_flags(x.w, 0x40404040u));
                reinterpret_cast<u16*>(&mk[2][W])[q] = (u16)mask16(eq_flags(x.x, 0x2b2b2b2bu), eq_flags(x.y, 0x2b2b2b2bu), eq_flags(x.z, 0x2b2b2b2bu), eq_flags(x.w, 0x2b2b2b2bu));
                if constexpr (MARKS) {
                    // '!' candidates: a byte b with (b & 0x5e) == 0 -- in a quality line (0x21 .. 0x7e) that is '!' alone   (no carry between bytes: 0x5e + 0x7f < 0x100)
                    // odd-base candidates: bit 3 (N, '.') or bits 5 and 6 (lowercase) -- every N-like or lowercase base, and no A C G T 0 1 2 3
#define FR_BANG(w) (~(((w) & 0x5e5e5e5eu) + 0x7f7f7f7fu) & 0x80808080u)
#define FR_ODD(w) ((((w) << 4) | (((w) << 1) & ((w) << 2))) & 0x80808080u)
                    reinterpret_cast<u16*>(&mk[3][W])[q] = (u16)mask16(FR_BANG(x.x), FR_BANG(x.y), FR_BANG(x.z), FR_BANG(x.w));
                    reinterpret_cast<u16*>(&mk[4][W])[q] = (u16)mask16(FR_ODD(x.x), FR_ODD(x.y), FR_ODD(x.z), FR_ODD(x.w));
#undef FR_BANG
#undef FR_ODD
                }
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) cnt += (u32)__shfl_xor((int)cnt, d, 64);
    if (lane == 0) wtot[wave] = cnt;
    __syncthreads();
    const u32 total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    // ---- the tile among the file's (decoupled look-back) ------------------------------------------------------------------------------
    if (wave == 0) {
        if (lane == 0) __hip_atomic_store(&tstat[tile], (tile == 0 ? TS_PFX : TS_AGG) | (u64)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u64 excl = 0;
        if (tile) {
            long long j = (long long)tile - 1;
            u32 spins = 0; bool tripped = false;
            for (;;) {
                const long long idx = j - (long long)lane;
                u64 word = TS_PFX;                                                  // (before the file's first tile: a prefix of 0)
                if (idx >= 0) word = __hip_atomic_load(&tstat[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__any((word >> 62) == 0)) {                                     // a tile before this one has not published yet
                    if (++spins > (1u << 22)) { tripped = true; break; }            // (never in a sound launch; every wave must end)
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                const u64 pm = __ballot((word >> 62) == 2);
                if (pm) {
                    const u32 first = (u32)__ffsll((long long)pm) - 1u;             // the nearest tile that knows its prefix
                    excl += wave_sum_u64(lane <= first ? (word & TS_VAL) : 0ull);
                    break;
                }
                excl += wave_sum_u64(word & TS_VAL);
                j -= 64;
            }
            if (tripped && lane == 0) fo->guard_tripped = 1;
            if (lane == 0) __hip_atomic_store(&tstat[tile], TS_PFX | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            s_base = excl;
            if (tile + 1 == gridDim.x) fo->nlines = excl + total;
        }
    }
    __syncthreads();
    u64 run = s_base;                                        // newlines before the sub-tile at hand
    if (tile == 0 && tid == 0) line_off[0] = 0;
    // does a line start with the sub-tile at hand (thread 0's business)?  With the file it does; elsewhere the byte before says
    u32 carry_nl = 0;
    if (tid == 0) carry_nl = tb == 0 ? 1u : (tb <= n && fq[tb - 1] == '\n') ? 1u : 0u;
    // ---- round 2: offsets, the '@' / '+' checks, the marks -- from the masks -----------------------------------------------------------
#pragma nounroll
    for (u32 wi = 0; wi < FRAME_WIN; wi++) {
        const u64 sb = tb + (u64)FRAME_SUB * wi;
        if (sb >= n) break;                                  // (the same for every thread)
        const u32 W = wi * 256u + tid;
        const u64 w0 = sb + 64u * tid;
        const u64 nlm = mk[0][W], atm = mk[1][W], plm = mk[2][W];
        u64 bang = 0, odd = 0;
        if constexpr (MARKS) {
            const u64 live = w0 >= n ? 0ull : (n - w0) >= 64 ? ~0ull : ((1ull << (n - w0)) - 1);      // (bytes past the end read as 0: a '!' candidate)
            bang = mk[3][W] & live; odd = mk[4][W] & live;
        }
        // this window's newlines among its sub-tile's
        const u32 c = (u32)__popcll(nlm);
        u32 incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 o = (u32)__shfl_up((int)incl, d, 64); if (lane >= (u32)d) incl += o; }
        if (lane == 63) { wtot[wave] = incl; s_last[wave] = (u32)(nlm >> 63); }
        __syncthreads();
        u32 bw = 0, tw = 0;
#pragma unroll
        for (u32 k = 0; k < 4; k++) { const u32 t = wtot[k]; if (k < wave) bw += t; tw += t; }
        u64 k = run + bw + incl - c;                         // newlines before the window = the number of the line its first byte lies in
        run += tw;
        // usrs.cpp:311, 346: a record's first line starts with '@', its third with '+'.  A line that starts WITH the window (the
        // byte before it, the window before's last, is a newline -- handed over from lane to lane, wave to wave, sub-tile to sub-tile):
        u32 prev_nl = (u32)__shfl_up((int)(u32)(nlm >> 63), 1, 64);
        if (lane == 0) prev_nl = wave ? s_last[wave - 1] : carry_nl;
        if (tid == 0) carry_nl = s_last[3];
        if (prev_nl && w0 < n && ((u32)k & 1u) == 0u && !((((u32)k & 2u) ? plm : atm) & 1ull)) atomicMax(status, (u32)(-SFQ_E_FORMAT));
        if (w0 < n) {
            u64 m = nlm;
            u64 below = 0;                                   // the window's bytes up to the line end looked at last
            while (m) {
                const u32 i = (u32)__ffsll((long long)m) - 1u;
                m &= m - 1;
                const u64 upto = (1ull << i) - 1;            // bytes of the window before this line end
                if constexpr (MARKS) {
                    const u64 seg = upto & ~below;           // ... that belong to the line k
                    const u32 type = (u32)k & 3u;
                    if (((type == 3u && (bang & seg)) || (type == 1u && (odd & seg))) && (k >> 2) < ecap) exc_flag[k >> 2] = 1;
                }
                below = upto | (1ull << i);
                k++;                                         // the line that starts behind this newline
                const u64 start = w0 + i + 1;
                if (k <= cap) line_off[k] = start;
                // ... and a line that starts inside it: the byte behind the line end, in the '@' / '+' masks
                if (i < 63u && start < n) {
                    const u32 type = (u32)k & 3u;
                    if ((type & 1u) == 0u && !(((type ? plm : atm) >> (i + 1u)) & 1ull)) atomicMax(status, (u32)(-SFQ_E_FORMAT));
                }
            }
            if constexpr (MARKS) {                           // what lies behind the window's last line end (or the whole window)
                const u64 seg = ~below;
                const u32 type = (u32)k & 3u;
                if (((type == 3u && (bang & seg)) || (type == 1u && (odd & seg))) && (k >> 2) < ecap) exc_flag[k >> 2] = 1;
            }
        }
        __syncthreads();                                     // (wtot is written again in the next round)
    }
}
u32 frame_tiles(u64 n) { return (u32)((n + FRAME_TILE - 1) / FRAME_TILE); }
void launch_frame(const u8* fq, u64 n, u64* tstat /* [frame_tiles(n)], zeroed */, u64* line_off, u64 cap, u32* status, u8* exc_flag, u64 ecap, void* frame_out /* 16 bytes, zeroed */, hipStream_t st) {
    const u32 tiles = frame_tiles(n);
    if (exc_flag) hipLaunchKernelGGL(k_frame<true>, dim3(tiles), dim3(256), 0, st, fq, n, tstat, line_off, cap, status, exc_flag, ecap, reinterpret_cast<FrameOut*>(frame_out));
    else hipLaunchKernelGGL(k_frame<false>, dim3(tiles), dim3(256), 0, st, fq, n, tstat, line_off, cap, status, exc_flag, ecap, reinterpret_cast<FrameOut*>(frame_out));
}
// the per-record checks that need the line index alone (the '@' / '+' prefixes are k_frame's): line limits (usrs.hpp:34-36) and the
// call's longest header and base line
__global__ __launch_bounds__(256) void k_validate_lines(const u64* __restrict__ line_off, u64 nrec, u32 max_hdr, u32 max_line, u32* status) {
    const u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
    const u32 lane = threadIdx.x & 63;
    u32 hl = 0, gl = 0, bad = 0, over = 0, hmin = 0;
    // a record's five line starts: its own four as two 16-byte loads (coalesced: thread after thread), the next record's first from the
    // next lane (the wave's last lane reads it)
    u64 l0 = 0, l1 = 0, l2 = 0, l3 = 0, l4 = 0;
    if (r < nrec) {
        const ulonglong2 a01 = *reinterpret_cast<const ulonglong2*>(line_off + 4 * r), a23 = *reinterpret_cast<const ulonglong2*>(line_off + 4 * r + 2);
        l0 = a01.x; l1 = a01.y; l2 = a23.x; l3 = a23.y;
    }
    l4 = ((u64)(u32)__shfl_down((int)(u32)(l0 >> 32), 1, 64) << 32) | (u32)__shfl_down((int)(u32)l0, 1, 64);
    if (r < nrec && (lane == 63 || r + 1 == nrec)) l4 = line_off[4 * r + 4];
    if (r < nrec) {
        // the reference diverts longer lines to raw "oversize" streams (usrs.cpp:313-317, 333-337, 366-367): max_line = 0xfffe
        // where its format is written; the block format codes base / quality lines of any length the usual way
        if ((l1 - l0 - 2) > max_hdr || (l2 - l1 - 1) > max_line || (l3 - l2 - 2) > 0x1ffe || (l4 - l3 - 1) > max_line) bad = (u32)(-SFQ_E_UNSUPPORTED);
        else if (l2 - l1 - 1 == 0) bad = (u32)(-SFQ_E_UNSUPPORTED);            // empty base line: usrs.cpp:217-222 mis-frames it
        hl = (u32)(l1 - l0 - 2); gl = (u32)(l2 - l1 - 1);
        hmin = ~hl;                                                            // (the shortest header, as a maximum: status[4] starts at 0)
        // status[3] != 0: some record may be over format 6's line limits (usrs.hpp:34-36; a SOLiD line may be one longer -- the
        // oversize pass decides exactly)
        over = ((l1 - l0 - 2) > 0x1ffe || (l2 - l1 - 1) > 0xfffe || (l4 - l3 - 1) > 0xfffe) ? 1u : 0u;
        if ((l1 - l0) < 2 || (l3 - l2) < 2) bad = (u32)(-SFQ_E_FORMAT);        // (a header or '+' line without its prefix: k_frame has flagged it too)
    }
#pragma unroll
    for (int dd = 32; dd > 0; dd >>= 1) {
        const u32 o1 = (u32)__shfl_xor((int)hl, dd, 64), o2 = (u32)__shfl_xor((int)gl, dd, 64), o3 = (u32)__shfl_xor((int)bad, dd, 64), o4 = (u32)__shfl_xor((int)over, dd, 64), o5 = (u32)__shfl_xor((int)hmin, dd, 64);
        hl = o1 > hl ? o1 : hl; gl = o2 > gl ? o2 : gl; bad = o3 > bad ? o3 : bad; over |= o4; hmin = o5 > hmin ? o5 : hmin;
    }
    if (lane == 0) {
        if (bad) atomicMax(status, bad);
        if (hl > status[1]) atomicMax(status + 1, hl);      // (the plain read only spares atomics that cannot raise it)
        if (gl > status[2]) atomicMax(status + 2, gl);
        if (over) status[3] = 1;
        if (hmin > status[4]) atomicMax(status + 4, hmin);   // ~(the shortest header's length)
    }
}
void launch_validate_lines(const u64* line_off, u64 nrec, u32 max_hdr, u32 max_line, u32* status, hipStream_t st) {
    hipLaunchKernelGGL(k_validate_lines, dim3((u32)((nrec + 255) / 256)), dim3(256), 0, st, line_off, nrec, max_hdr, max_line, status);
}

// out[0] = max(out[0], max of v[0..n)); a thread takes sixteen values, a wave one atomic
__global__ __launch_bounds__(256) void k_max_u32(const u32* __restrict__ v, u64 n, u32* out) {
    const u64 i0 = ((u64)blockIdx.x * 256 + threadIdx.x) * 16;
    u32 m = 0;
    if (i0 + 16 <= n) {
        const uint4* q = reinterpret_cast<const uint4*>(v + i0);          // (v is a device allocation: 16-byte aligned at i0 % 4 == 0)
#pragma unroll
        for (int j = 0; j < 4; j++) { const uint4 w = q[j]; m = max(max(m, max(w.x, w.y)), max(w.z, w.w)); }
    } else {
        for (u64 i = i0; i < n; i++) m = max(m, v[i]);
    }
#pragma unroll
    for (int dd = 32; dd > 0; dd >>= 1) { const u32 o = (u32)__shfl_xor((int)m, dd, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
void launch_max_u32(const u32* v, u64 n, u32* out, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_max_u32, dim3((u32)((n + 4095) / 4096)), dim3(256), 0, st, v, n, out);
}

// A fingerprint of a device-resident text: 65 536 sixteen-byte pieces spread evenly over it, each hashed with its number, the
// hashes XORed (api.cpp: is the text at this address still the one sfq_count_priors framed?).  out[0] starts at zero.
__global__ __launch_bounds__(256) void k_text_fingerprint(const u8* __restrict__ fq, u64 n, u64* out) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    const u64 span = n > 16 ? n - 16 : 0;
    const u64 pos = span >= 65535ull * 16 ? (u64)i * (span / 65535ull) : (u64)i * 16;
    u64 h = 0;
    if (pos + 16 <= n) {
        const u32* q = reinterpret_cast<const u32*>(fq + pos);             // (global loads need no alignment on gfx9)
        u32 x = q[0] * 0x9E3779B1u;
        x = (x ^ q[1]) * 0x85EBCA77u; x = (x ^ q[2]) * 0xC2B2AE3Du; x = (x ^ q[3]) * 0x27D4EB2Fu;
        h = ((u64)x << 32 | (x ^ (x >> 15))) * (2ull * i + 1);
    } else if (pos < n) {
        for (u64 p = pos; p < n; p++) h = (h ^ fq[p]) * 0x100000001B3ull;
        h *= 2ull * i + 1;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) h ^= __shfl_xor(h, d, 64);
    if ((threadIdx.x & 63) == 0 && h) atomicXor((unsigned long long*)out, (unsigned long long)h);
}
void launch_text_fingerprint(const u8* fq, u64 n, u64* out, hipStream_t st) {
    hipLaunchKernelGGL(k_text_fingerprint, dim3(256), dim3(256), 0, st, fq, n, out);
}

// ---- generic exclusive scan u32 -> u64 -------------------------------------------------------------
#define SCAN_TILE 1024u
__global__ __launch_bounds__(256) void k_scan_tile_sums(const u32* in, u64 n, u64* sums, u32 pad) {
    __shared__ u32 lds[8];
    u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * 4;
    u32 s = 0;
    for (int j = 0; j < 4; j++) if (base + j < n) s += (in[base + j] + pad) & ~pad;
    u32 total;
    block_excl_scan_256(s, lds, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
__global__ __launch_bounds__(256) void k_scan_sums(u64* sums, u64 ntiles) {   // single workgroup, in place, exclusive
    __shared__ u64 carry_s;
    __shared__ u64 wsum[4];
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (u64 i0 = 0; i0 < ntiles; i0 += 256) {
        u64 i = i0 + threadIdx.x;
        u64 v = i < ntiles ? sums[i] : 0;
        u64 incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            u64 o = __shfl_up(incl, d, 64);
            if (lane >= (u32)d) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        u64 base = carry_s;
        for (u32 w = 0; w < wave; w++) base += wsum[w];
        if (i < ntiles) sums[i] = base + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = base + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[ntiles] = carry_s;
}
__global__ __launch_bounds__(256) void k_scan_tiles(const u32* in, u64 n, const u64* sums, u64* out, u64 ntiles, u32 pad) {
    __shared__ u32 lds[8];
    u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * 4;
    u32 v[4]; u32 s = 0;
    for (int j = 0; j < 4; j++) { v[j] = base + j < n ? (in[base + j] + pad) & ~pad : 0; s += v[j]; }
    u32 total;
    u32 ex = block_excl_scan_256(s, lds, &total);
    u64 run = sums[blockIdx.x] + ex;
    for (int j = 0; j < 4; j++) { if (base + j < n) out[base + j] = run; run += v[j]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = sums[ntiles];
}
// pad: 0, or 2^k - 1 -- every value rounded up to a multiple of 2^k first (places of lines that start on 32-byte sectors)
void launch_scan_u32(const u32* in, u64* out, u64 n, u64* tmp, hipStream_t st, u32 pad) {
    u64 ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (ntiles == 0) ntiles = 1;
    hipLaunchKernelGGL(k_scan_tile_sums, dim3((u32)ntiles), dim3(256), 0, st, in, n, tmp, pad);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, st, tmp, ntiles);
    hipLaunchKernelGGL(k_scan_tiles, dim3((u32)ntiles), dim3(256), 0, st, in, n, (const u64*)tmp, out, ntiles, pad);
}

// ---- format 6: the reference's OVERSIZE records (usrs.cpp:269-301) -----------------------------------------------------
// A record whose header has more than 8190 bytes or whose base line (less a SOLiD prefix) more than 65534 never reaches
// the reference's models: UsrSave::get_record (usrs.cpp:313-318, 333-338) sends its four lines raw to "usr.lrec" /
// "usr.lgen" / "usr.lqlt" and goes on to the next record; the record still counts (g_record_count).  Here: flags per
// record, the text WITHOUT those records (what the model kernels code, through the usual framing), the map from a kept
// record to its number in the file (the exception streams count in those), and the list of the oversize ones.
// first[0] = the first record that passes determine_record's test (usrs.cpp:203-229: header within 8190 bytes, raw base line
// of 1..65535): the file's llen / usr.solid / usr.2id come from it
__global__ __launch_bounds__(256) void k_over_first(const u64* __restrict__ line_off, u64 nrec, u32* first) {
    const u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrec) return;
    const u64 l0 = line_off[4 * r], l1 = line_off[4 * r + 1], l2 = line_off[4 * r + 2];
    if ((l1 - l0 - 2) <= 0x1ffe && (l2 - l1 - 1) >= 1 && (l2 - l1 - 1) <= 0xffff) atomicMin(first, (u32)r);
}
// usrs.cpp:241-259 on that record: a digit 0..3 before any of a, c, g, t means SOLiD colour space
__global__ void k_over_solid(const u8* __restrict__ fq, const u64* __restrict__ line_off, u64 r, u32* out) {
    const u64 l1 = line_off[4 * r + 1], l2 = line_off[4 * r + 2];
    const u32 llen = (u32)(l2 - l1 - 1);
    u32 solid = 0;
    for (u32 i = 1; i < llen; i++) {
        const u32 c = fq[l1 + i] | 0x20;
        if (c >= '0' && c <= '3') { solid = 1; break; }
        if (c == 'a' || c == 'c' || c == 'g' || c == 't') break;
    }
    *out = solid;
}
// flags[r] = 1: record r is oversize; kbytes[r] = the bytes of a kept record.  A quality line over the limit beside a base
// line within it is refused: the reference has by then written the record's "usr.x" / "usr.pfg" exceptions under a number
// its decoder never looks them up at (usrs.cpp:340-345 run before 372-373) -- its own archive of such a file does not decode.
__global__ __launch_bounds__(256) void k_over_flags(const u64* __restrict__ line_off, u64 nrec, u32 solid, u32* __restrict__ flags,
                                                    u32* __restrict__ kbytes, u32* status) {
    const u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrec) return;
    const u64 l0 = line_off[4 * r], l1 = line_off[4 * r + 1], l2 = line_off[4 * r + 2], l3 = line_off[4 * r + 3], l4 = line_off[4 * r + 4];
    const u64 hl = l1 - l0 - 2, gl = l2 - l1 - 1, ql = l4 - l3 - 1;
    const bool over = hl > 0x1ffe || gl > 0xfffeull + solid;
    if (!over && ql > 0xfffeull + solid) atomicMax(status, (u32)(-SFQ_E_UNSUPPORTED));
    flags[r] = over ? 1u : 0u;
    kbytes[r] = over ? 0u : (u32)(l4 - l0);
}
// a wave per record: a kept record's text goes to its place in the filtered text, its file number to rec_map; an oversize
// record's number to over_list
__global__ __launch_bounds__(256) void k_over_split(const u8* __restrict__ fq, const u64* __restrict__ line_off, u64 nrec, const u32* __restrict__ flags,
                                                    const u64* __restrict__ fpos, const u64* __restrict__ koff, u8* __restrict__ filt,
                                                    u32* __restrict__ rec_map, u32* __restrict__ over_list) {
    const u64 r = (u64)blockIdx.x * 4 + (threadIdx.x >> 6);
    const u32 lane = threadIdx.x & 63;
    if (r >= nrec) return;
    if (flags[r]) { if (lane == 0) over_list[fpos[r]] = (u32)r; return; }
    if (lane == 0) rec_map[r - fpos[r]] = (u32)r;
    const u64 l0 = line_off[4 * r], n = line_off[4 * r + 4] - l0;
    const u8* src = fq + l0; u8* dst = filt + koff[r];
    for (u64 i = lane; i < n; i += 64) dst[i] = src[i];
}
void launch_over_first(const u64* line_off, u64 nrec, u32* first, hipStream_t st) {
    hipLaunchKernelGGL(k_over_first, dim3((u32)((nrec + 255) / 256)), dim3(256), 0, st, line_off, nrec, first);
}
void launch_over_solid(const u8* fq, const u64* line_off, u64 r, u32* out, hipStream_t st) {
    hipLaunchKernelGGL(k_over_solid, dim3(1), dim3(1), 0, st, fq, line_off, r, out);
}
void launch_over_flags(const u64* line_off, u64 nrec, u32 solid, u32* flags, u32* kbytes, u32* status, hipStream_t st) {
    hipLaunchKernelGGL(k_over_flags, dim3((u32)((nrec + 255) / 256)), dim3(256), 0, st, line_off, nrec, solid, flags, kbytes, status);
}
void launch_over_split(const u8* fq, const u64* line_off, u64 nrec, const u32* flags, const u64* fpos, const u64* koff, u8* filt, u32* rec_map, u32* over_list, hipStream_t st) {
    hipLaunchKernelGGL(k_over_split, dim3((u32)((nrec + 3) / 4)), dim3(256), 0, st, fq, line_off, nrec, flags, fpos, koff, filt, rec_map, over_list);
}
// decode side: flags[number - 1] = 1 for the oversize records' 1-based numbers; rec_map[k] = the file number (0-based) of the k-th kept record
__global__ __launch_bounds__(256) void k_over_mark(const u64* __restrict__ over_no, u32 n_over, u64 total, u32* __restrict__ flags, u32* status) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_over) return;
    const u64 no = over_no[i];
    if (no == 0 || no > total || (i && over_no[i - 1] >= no)) { atomicMax(status, (u32)(-SFQ_E_CORRUPT)); return; }
    flags[no - 1] = 1u;
}
__global__ __launch_bounds__(256) void k_over_map(const u32* __restrict__ flags, const u64* __restrict__ fpos, u64 total, u32* __restrict__ rec_map) {
    const u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
    if (r >= total || flags[r]) return;
    rec_map[r - fpos[r]] = (u32)r;
}
void launch_over_mark(const u64* over_no, u32 n_over, u64 total, u32* flags, u32* status, hipStream_t st) {
    if (n_over) hipLaunchKernelGGL(k_over_mark, dim3((n_over + 255) / 256), dim3(256), 0, st, over_no, n_over, total, flags, status);
}
void launch_over_map(const u32* flags, const u64* fpos, u64 total, u32* rec_map, hipStream_t st) {
    hipLaunchKernelGGL(k_over_map, dim3((u32)((total + 255) / 256)), dim3(256), 0, st, flags, fpos, total, rec_map);
}

// ---- block descriptors: UsrSave::determine_record (usrs.cpp:186-267) on each block's first record ----
__global__ __launch_bounds__(64) void k_block_prepare(const u8* fq, const u64* line_off, u64 nrec, u32 block_reads,
                                                      BlockDesc* blocks, u32 nblocks, i32 gen_bits) {
    u32 b = blockIdx.x * 64 + threadIdx.x;
    if (b >= nblocks) return;
    BlockDesc d;
    d.rec0 = (u64)b * block_reads;
    u64 rend = d.rec0 + block_reads; if (rend > nrec) rend = nrec;
    d.nrec = (u32)(rend - d.rec0);
    const u64 r = d.rec0;
    const u64 l0 = line_off[4 * r], l1 = line_off[4 * r + 1], l2 = line_off[4 * r + 2], l3 = line_off[4 * r + 3];
    d.first_hdr_off = l0 + 1;
    d.first_hdr_len = (u32)(l1 - l0 - 2);
    u32 llen = (u32)(l2 - l1 - 1);
    u32 two_id = 0;
    for (u64 p = l2 + 1; p + 1 < l3; p++) if (fq[p] != ' ') two_id = 1;      // usrs.cpp:233-236
    u32 solid = 0;
    for (u32 i = 1; i < llen; i++) {                                          // usrs.cpp:239-257
        u32 c = fq[l1 + i] | 0x20;
        if (c >= '0' && c <= '3') { solid = 1; break; }
        if (c == 'a' || c == 'c' || c == 'g' || c == 't') break;
    }
    if (solid) llen--;
    d.llen = llen; d.solid = (u8)solid; d.two_id = (u8)two_id; d.gen_bits = (u8)gen_bits; d.pad = 0;
    d.status = 0; d.n_byte = 0; d.extra_hi = 0; d.hdr_bytes = 0;
    // scratch-arena regions: sized from the block's text bytes (overflow is detected, never silent)
    const u64 t0 = l0, t1 = line_off[4 * rend];
    const u64 bb = t1 - t0;
    u64 off = (t0 * 17 / 2 + (u64)b * 1024 + 15) & ~15ull;     // the caps below add up to 8.25 bb + 704, 16-byte rounding included < 8.5 bb + 1024 (api.cpp sizes the arena)
    const u32 caps[SFQ_NSTREAMS] = {
        (u32)(bb + bb / 2 + 64),   // rec
        (u32)(bb * 3 / 4 + 64),    // gen
        (u32)(2 * bb + 64),        // qlt (a quality over 62 costs an escape: up to 4 bytes a symbol, twice the text)
        (u32)(bb / 2 + 64),        // gen.Ns
        (u32)(bb / 2 + 64),        // gen.Nn
        (u32)(bb + 64),            // rec.x
        (u32)(bb / 2 + 64),        // usr.x
        (u32)(bb / 2 + 64),        // usr.x.q
        (u32)(bb / 4 + 64),        // usr.pfg
        (u32)(bb / 4 + 64),        // usr.pfq
        (u32)(bb / 2 + 64),        // gen.lc
        0, 0, 0 };                 // usr.lrec / usr.lgen / usr.lqlt: the oversize pass has regions of its own (api.cpp)
    for (int s = 0; s < SFQ_NSTREAMS; s++) {
        d.size[s] = 0; d.out_off[s] = off; d.out_cap[s] = caps[s];
        off += (caps[s] + 15u) & ~15u;
    }
    blocks[b] = d;
}
void launch_block_prepare(const u8* fq, const u64* line_off, u64 nrec, u32 block_reads, BlockDesc* blocks, u32 nblocks,
                          u64 nbytes, i32 level, i32 gen_bits, hipStream_t st) {
    (void)nbytes; (void)level;
    hipLaunchKernelGGL(k_block_prepare, dim3((nblocks + 63) / 64), dim3(64), 0, st, fq, line_off, nrec, block_reads, blocks, nblocks, gen_bits);
}

__global__ __launch_bounds__(256) void k_fill_u32(u32* p, u64 n, u32 v) {
    u64 i = ((u64)blockIdx.x * 256 + threadIdx.x) * 4;
    const u64 stride = (u64)gridDim.x * 256 * 4;
    for (; i < n; i += stride) {
        if (i + 4 <= n) *reinterpret_cast<uint4*>(p + i) = make_uint4(v, v, v, v);
        else for (u64 j = i; j < n; j++) p[j] = v;
    }
}
void launch_fill_u32(u32* p, u64 n, u32 v, hipStream_t st) {
    if (!n) return;
    u64 nb = (n / 4 + 255) / 256; if (nb > 8192) nb = 8192; if (nb == 0) nb = 1;
    hipLaunchKernelGGL(k_fill_u32, dim3((u32)nb), dim3(256), 0, st, p, n, v);
}

// ---- packing: per-block stream sizes -> offsets -> compact copy ---------------------------------------
// One workgroup per stream walks the blocks in order (nblocks is in the thousands).
__global__ __launch_bounds__(256) void k_block_stream_offsets(BlockDesc* blocks, u32 nblocks, u64* blk_stream_off, u64* stream_total, u32 s0) {
    __shared__ u64 carry_s;
    __shared__ u64 wsum[4];
    const u32 s = s0 + blockIdx.x;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (u32 i0 = 0; i0 < nblocks; i0 += 256) {
        u32 i = i0 + threadIdx.x;
        u64 v = i < nblocks ? blocks[i].size[s] : 0;
        u64 incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            u64 o = __shfl_up(incl, d, 64);
            if (lane >= (u32)d) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        u64 base = carry_s;
        for (u32 w = 0; w < wave; w++) base += wsum[w];
        if (i < nblocks) blk_stream_off[(u64)i * SFQ_NSTREAMS + s] = base + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = base + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) stream_total[s] = carry_s;
}
void launch_block_stream_offsets(BlockDesc* blocks, u32 nblocks, u64* blk_stream_off, u64* stream_total, u32 s0, u32 s1, hipStream_t st) {
    hipLaunchKernelGGL(k_block_stream_offsets, dim3(s1 - s0), dim3(256), 0, st, blocks, nblocks, blk_stream_off, stream_total, s0);
}
// Where each stream starts in the caller's buffer, and whether the packing may run at all: the blocks' worst status and the
// output's size against the caller's room are looked at HERE, so that the host need not come back between the sizes and the
// packing (that round trip was 0.8 ms of every call's tail).  gate[0] = 1: go; gate[1] = worst status.
__global__ __launch_bounds__(256) void k_stream_gate(const BlockDesc* __restrict__ blocks, u32 nblocks, const u64* __restrict__ stream_total, u64 out_cap,
                                                     u64* __restrict__ stream_base, u32* __restrict__ gate) {
    __shared__ u32 wmax[4];
    u32 worst = 0;
    for (u32 b = threadIdx.x; b < nblocks; b += 256) worst = max(worst, blocks[b].status);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) worst = max(worst, (u32)__shfl_xor((int)worst, d, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = worst;
    __syncthreads();
    if (threadIdx.x == 0) {
        worst = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        u64 run = 0;
        for (u32 s = 0; s < SFQ_NSTREAMS; s++) { stream_base[s] = run; run += stream_total[s]; }
        gate[1] = worst;
        gate[0] = (worst == 0 && run <= out_cap) ? 1u : 0u;
    }
}
void launch_stream_gate(const BlockDesc* blocks, u32 nblocks, const u64* stream_total, u64 out_cap, u64* stream_base, u32* gate, hipStream_t st) {
    hipLaunchKernelGGL(k_stream_gate, dim3(1), dim3(256), 0, st, blocks, nblocks, stream_total, out_cap, stream_base, gate);
}
// grid = (nblocks, SFQ_NSTREAMS); stream_base[s] = offset of stream s in out
__global__ __launch_bounds__(256) void k_compact(const BlockDesc* blocks, const u8* arena, const u64* blk_stream_off,
                                                 const u64* stream_base, u8* out, u32 skip_streams, const u32* gate) {
    if (gate && !gate[0]) return;
    const u32 b = blockIdx.x, s = blockIdx.y;
    const u32 n = blocks[b].size[s];
    if (!n || ((skip_streams >> s) & 1)) return;
    const u8* src = arena + blocks[b].out_off[s];
    u8* dst = out + stream_base[s] + blk_stream_off[(u64)b * SFQ_NSTREAMS + s];
    // src is 16-byte aligned; dst is not.  Copy dwords where dst allows, bytes at the edges.
    u32 head = (u32)((4 - ((uintptr_t)dst & 3)) & 3); if (head > n) head = n;
    for (u32 i = threadIdx.x; i < head; i += 256) dst[i] = src[i];
    const u32 nd = (n - head) / 4;
    u32* d4 = reinterpret_cast<u32*>(dst + head);
    for (u32 i = threadIdx.x; i < nd; i += 256) {
        const u8* p = src + head + 4 * (u64)i;
        d4[i] = (u32)p[0] | (u32)p[1] << 8 | (u32)p[2] << 16 | (u32)p[3] << 24;
    }
    for (u32 i = head + nd * 4 + threadIdx.x; i < n; i += 256) dst[i] = src[i];
}
void launch_compact(const BlockDesc* blocks, u32 nblocks, const u8* arena, const u64* blk_stream_off,
                    const u64* stream_base, u8* out, u32 skip_streams, hipStream_t st, const u32* gate) {
    hipLaunchKernelGGL(k_compact, dim3(nblocks, SFQ_NSTREAMS), dim3(256), 0, st, blocks, arena, blk_stream_off, stream_base, out, skip_streams, gate);
}

// ---- first headers ("rec.first", recs.cpp:68-75): one per block, gathered into a blob -----------------
__global__ __launch_bounds__(256) void k_first_hdr_lens(const BlockDesc* blocks, u32 nblocks, u32* lens) {
    u32 b = blockIdx.x * 256 + threadIdx.x;
    if (b < nblocks) lens[b] = blocks[b].first_hdr_len;
}
__global__ __launch_bounds__(64) void k_gather_first_hdrs(const BlockDesc* blocks, u32 nblocks, const u8* fq,
                                                          const u64* blob_off, u8* blob, u64 cap) {
    u32 b = blockIdx.x;                       // one wave per block
    if (b >= nblocks) return;
    const u32 n = blocks[b].first_hdr_len;
    const u64 o = blob_off[b];
    if (o + n > cap) return;
    const u8* src = fq + blocks[b].first_hdr_off;
    for (u32 i = threadIdx.x; i < n; i += 64) blob[o + i] = src[i];
}
void launch_first_hdr_lens(const BlockDesc* blocks, u32 nblocks, u32* lens, hipStream_t st) {
    hipLaunchKernelGGL(k_first_hdr_lens, dim3((nblocks + 255) / 256), dim3(256), 0, st, blocks, nblocks, lens);
}
void launch_gather_first_hdrs(const BlockDesc* blocks, u32 nblocks, const u8* fq, const u64* blob_off, u8* blob, u64 cap, hipStream_t st) {
    hipLaunchKernelGGL(k_gather_first_hdrs, dim3(nblocks), dim3(64), 0, st, blocks, nblocks, fq, blob_off, blob, cap);
}
