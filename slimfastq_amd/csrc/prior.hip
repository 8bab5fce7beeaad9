// prior.hip -- warm start for the quality model (block format 7 only; DESIGN.md "warm start").
//
// The reference keeps ONE adaptive state per file, so cutting a file into independent blocks costs it
// its learning again in every block (+25 % on the quality stream at 1024-record blocks).  Format 7 lets
// every block start from a shared PRIOR instead of from all-zero rows:
//   1. histogram: count (context, symbol) over a sample of the records (the first PRIOR_SYMBOLS of each: the
//                 prior is transmitted, so any deterministic sample will do)      -- parallel, atomics
//   2. rows     : per context, symbols ordered by count and frequencies scaled -- one wave per context
// The prior is stored once in the archive ("qlt.pri"); encoder and decoder build identical tables from
// it, and a block's first touch of a row copies the prior row instead of starting from zero.  With it a
// 256-record block codes as well as the reference's whole-file state (oracle/sfq_oracle.c restates the
// rule for the tests).  One block and no prior stays byte-identical to the reference.
#include "kernels.h"
#include "dev_wave.h"

// ---- 1. histogram --------------------------------------------------------------------------------------
__device__ __forceinline__ u32 p_calc_last_delta(u32& delta, u32 q, u32 q1, u32 q2) {   // qlts.hpp:62-74
    if (q1 > q) delta += q1 - q;
    u32 d3 = delta >> 3;
    return (q | ((q1 < q2 ? q2 : q1) << 6) | ((u32)(q1 == q2) << 12) | ((d3 < 7 ? d3 : 7) << 13)) & 0xFFFFu;
}
// One lane per sampled record walks its quality line with the model's context function and counts
// (ctx, symbol).  Plain global atomics serialise on the few very hot counters (memory-side atomics:
// ~30 ns each on one address), so every workgroup first aggregates in an LDS hash table and flushes each
// distinct key once.
// (context, symbol) pairs are mostly rare -- the 38 k symbols of a workgroup hold 6-8 k distinct ones -- so the table has to have
// room for all of them WITHOUT filling up: with 8192 slots of key + count (8 bytes) it ran 75-95 % full and spent 75 LDS instructions
// per symbol on probe chains (3.2 ms alone, 6-10 ms beside the other kernels; the quality chains wait for this).  A slot is now ONE
// dword -- the key's 22 bits over a 10-bit count -- so the same 64 KiB hold 16384 of them; a count that reaches 256 is moved on to
// the global counter by the lane that saw it get there (the few keys that take most of the symbols do that a few times per workgroup).
// Counting a key that gives up its place in a crowded table directly in memory is no way out: the counters many workgroups add to
// serialise in L2, and a two-slot cache in front of them took 30 ms.
#ifndef HIST_S
#define HIST_S 16384u
#endif
#ifndef HIST_T
#define HIST_T 256
#endif
#define HIST_SLOTS HIST_S
#define HIST_EMPTY 0xFFFFFFFFu
#define HIST_CBITS 10u
#define HIST_CMASK ((1u << HIST_CBITS) - 1u)
#define HIST_HARVEST 256u
// (the table is named as LDS and hist as global memory: through generic pointers the compiler folds the two adds into one flat atomic)
typedef __attribute__((address_space(3))) u32 lds_u32;
typedef __attribute__((address_space(1))) u32 glb_u32;
__device__ __forceinline__ void hist_add(lds_u32* slots, glb_u32* hist, u32 key) {
    u32 slot = ((key * 2654435761u) >> 16) & (HIST_SLOTS - 1);
    for (int probe = 0; probe < 128; probe++) {
        u32 v = slots[slot];
        if (v == HIST_EMPTY) {
            u32 expect = HIST_EMPTY;
            if (__hip_atomic_compare_exchange_strong(&slots[slot], &expect, (key << HIST_CBITS) | 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return;
            v = expect;                                   // another lane's key got there first: maybe ours
        }
        if ((v >> HIST_CBITS) == key) {                   // (a slot keeps its key)
            // The count must never carry into the key.  Every time its low eight bits wrap, exactly one lane sees them at 255 and
            // takes 256 off -- however late that lands behind the other lanes' adds (a few keys take most symbols of binned
            // qualities: all 256 lanes queue on one slot), it lands once per wrap, so the count stays below 256 x (1 + subs in flight).
            // And a lane that READS a count of 768 or more counts in memory instead: 255 other lanes can add before it, no more.
            if ((v & HIST_CMASK) >= 768u) { __hip_atomic_fetch_add(&hist[key], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
            const u32 old = __hip_atomic_fetch_add(&slots[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((old & (HIST_HARVEST - 1u)) == HIST_HARVEST - 1u) {
                __hip_atomic_fetch_sub(&slots[slot], HIST_HARVEST, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&hist[key], HIST_HARVEST, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        slot = (slot + 1) & (HIST_SLOTS - 1);
    }
    __hip_atomic_fetch_add(&hist[key], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // table crowded: count directly
}
template <u32 THREADS, u32 READS_PER_LANE>
__global__ __launch_bounds__(THREADS) void k_qlt_hist(const u8* __restrict__ fq, u64 nbytes, const u64* __restrict__ line_off,
                                                 const BlockDesc* __restrict__ blocks, u32 block_reads,
                                                 u64 nrec, u32 step, int level, u32 cap, u32* __restrict__ hist) {
#ifndef NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    __shared__ u32 slots[HIST_SLOTS];
    for (u32 i = threadIdx.x; i < HIST_SLOTS; i += THREADS) slots[i] = HIST_EMPTY;
    __syncthreads();
    for (u32 rr = 0; rr < READS_PER_LANE; rr++) {
        const u64 r = (((u64)blockIdx.x * READS_PER_LANE + rr) * THREADS + threadIdx.x) * step;
        if (r >= nrec) break;
        const u32 solid = blocks[r / block_reads].solid;
        const u64 q0 = line_off[4 * r + 3] + solid, q1e = line_off[4 * r + 4] - 1;
        const u32 nfull = q1e > q0 ? (u32)(q1e - q0) : 0;
        const u32 n = nfull < cap ? nfull : cap;             // the head of a long record stands for the rest
        // read the line as aligned dwords, four at a time: one memory latency per 16 symbols
        const u64 a0 = q0 & ~3ull;
        const u32 skip = (u32)(q0 - a0);
        const u32* wp = reinterpret_cast<const u32*>(fq + a0);
        const u64 wmax = (nbytes - a0) / 4;              // whole dwords available from a0
        u32 last = 0, delta = 5, q1 = 0, q2 = 0, di = 0;
        u32 done = 0;                                    // symbols consumed
        for (u64 w = 0; done < n; w += 4) {
            u32 wd[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (w + k < wmax) wd[k] = wp[w + k];
                else { wd[k] = 0; for (int j = 0; j < 4; j++) { const u64 at = a0 + (w + k) * 4 + j; if (at < nbytes) wd[k] |= (u32)fq[at] << (8 * j); } }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const u32 pos = (u32)(w + k) * 4 + j;    // byte index from a0
                    if (pos < skip || done >= n) continue;
                    const u32 b = (u32)(u8)(((wd[k] >> (8 * j)) & 0xff) - '!');
                    hist_add((lds_u32*)slots, (glb_u32*)hist, last * 64 + (b < 63u ? b : 63u));
                    if (level == 1)      last = (b | (last << 6)) & 0xFFFu;
                    else if (level == 2) last = (b | (last << 6)) & 0xFFFFu;
                    else if (++di & 1) { last = p_calc_last_delta(delta, b, q1, q2); q2 = b; }
                    else               { last = p_calc_last_delta(delta, b, q2, q1); q1 = b; }
                    done++;
                }
            }
        }
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < HIST_SLOTS; i += THREADS) { const u32 v = slots[i]; if (v != HIST_EMPTY && (v & HIST_CMASK)) atomicAdd(&hist[v >> HIST_CBITS], v & HIST_CMASK); }
}
// The same counts for LONG records, a WAVE per sampled record (round 4): a lane that walks the 4096 sampled symbols of a long read
// alone is the sample's time -- 5.5 ms of a 19.6 ms call over 60 k reads of 10-50 kb.  Here 64 consecutive symbols are taken at
// once, a lane each (one coalesced load), and the model's contexts -- functions of the three symbols before and, at levels 3 and 4,
// of the running sum of the quality drops (qlts.hpp:52-74) -- come from lane shifts and a wave prefix sum: the same (context,
// symbol) pairs as the lane's walk, so the same counts.
__global__ __launch_bounds__(256) void k_qlt_hist_w(const u8* __restrict__ fq, const u64* __restrict__ line_off,
                                                    const BlockDesc* __restrict__ blocks, u32 block_reads,
                                                    u64 nrec, u32 step, int level, u32 cap, u32* __restrict__ hist) {
    __shared__ u32 slots[HIST_SLOTS];
    for (u32 i = threadIdx.x; i < HIST_SLOTS; i += 256) slots[i] = HIST_EMPTY;
    __syncthreads();
    const u32 lane = threadIdx.x & 63;
    const u64 r = ((u64)blockIdx.x * 4 + (threadIdx.x >> 6)) * step;
    if (r < nrec) {
        const u32 solid = blocks[r / block_reads].solid;
        const u64 q0 = line_off[4 * r + 3] + solid, q1e = line_off[4 * r + 4] - 1;
        const u32 nfull = q1e > q0 ? (u32)(q1e - q0) : 0;
        const u32 n = nfull < cap ? nfull : cap;
        const u32 mask = level == 1 ? 0xFFFu : 0xFFFFu;
        u32 c1 = 0, c2 = 0, c3 = 0, dsum = 5;                // the three symbols before the step's first, the drops summed so far (+ 5)
        for (u32 base = 0; base < n; base += 64) {
            const u32 i = base + lane;
            const bool in = i < n;
            const u32 b = in ? (u32)(u8)(fq[q0 + i] - '!') : 0u;
            const u32 b1 = wave_shr1(b, c1), b2 = wave_shr1(b1, c2), b3 = wave_shr1(b2, c3);
            u32 ctx;
            if (level <= 2) ctx = (b1 | (b2 << 6) | (b3 << 12)) & mask;                       // qlts.hpp:52-57, unrolled
            else {
                const u32 t = b1 > b ? b1 - b : 0u;                                           // this symbol's drop: counts for the ones behind it
                const u32 incl = wave_incl_scan(t);
                const u32 delta = dsum + incl - t;
                const u32 d3 = delta >> 3;
                ctx = (b1 | ((b2 < b3 ? b3 : b2) << 6) | ((u32)(b2 == b3) << 12) | ((d3 < 7 ? d3 : 7) << 13)) & 0xFFFFu;
                if (i == 0) ctx = 0;                                                          // qlts.cpp:109-112: a line starts from context 0
                dsum += rl(incl, 63);
            }
            if (in) hist_add((lds_u32*)slots, (glb_u32*)hist, ctx * 64 + (b < 63u ? b : 63u));
            c1 = rl(b, 63); c2 = rl(b1, 63); c3 = rl(b2, 63);
        }
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < HIST_SLOTS; i += 256) { const u32 v = slots[i]; if (v != HIST_EMPTY && (v & HIST_CMASK)) atomicAdd(&hist[v >> HIST_CBITS], v & HIST_CMASK); }
}
void launch_qlt_hist(const u8* fq, u64 nbytes, const u64* line_off, const BlockDesc* blocks, u32 block_reads, u64 nrec, u32 step,
                     int level, u32 cap, u32* hist, hipStream_t st) {
    const u64 nsamp = (nrec + step - 1) / step;
    // long records: a wave per sampled record, four to a workgroup and its table
    if (nbytes / (nrec ? nrec : 1) > 4000) {
        hipLaunchKernelGGL(k_qlt_hist_w, dim3((u32)((nsamp + 3) / 4)), dim3(256), 0, st, fq, line_off, blocks, block_reads, nrec, step, level, cap, hist);
    } else {
        // (a record per lane: the kernel's time is one lane's walk -- with four records per lane and a 64 KiB table, two workgroups
        //  per CU, it took 4.6 ms of every call whatever the sample's size)
        const u64 per_wg = HIST_T;
        hipLaunchKernelGGL((k_qlt_hist<HIST_T, 1>), dim3((u32)((nsamp + per_wg - 1) / per_wg)), dim3(HIST_T), 0, st, fq, nbytes, line_off, blocks, block_reads, nrec, step, level, cap, hist);
    }
}

// ---- 2. rows -------------------------------------------------------------------------------------------
// The rule (also oracle/sfq_oracle.c sfqo_qlt_prior_rows): iend = highest seen symbol + 1; freq = (6 * count) >> s
// with the smallest s that brings the context's largest to <= 32000; symbols below iend ordered by (freq desc,
// symbol asc); total = sum of freq; count = 0.  Output, 66 dwords per context: slot[64] (freq | sym << 16),
// total, iend -- the exchange form the host packs into "qlt.pri" -- plus the two device layouts.
__device__ __forceinline__ u32 pr_sort64(u32 key, u32 lane) {
#pragma unroll
    for (u32 k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (u32 j = k >> 1; j > 0; j >>= 1) {
            const u32 other = (u32)__shfl_xor((int)key, j, 64);
            const bool up = (lane & k) == 0, lower = (lane & j) == 0;
            const u32 lo = key < other ? key : other, hi = key < other ? other : key;
            key = (lower == up) ? lo : hi;
        }
    }
    return key;
}
__global__ __launch_bounds__(256) void k_prior_rows(const u32* hist, u32 q_rows, u32* rows66, u32* w_rows, u32* w_ovf,
                                                   u32* l_slots, RowHdr* l_hdr) {
    const u32 lane = threadIdx.x & 63;
    const u32 ctx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ctx >= q_rows) return;
    const u32 c = hist[(size_t)ctx * 64 + lane];
    const u64 seen = __ballot(c != 0);
    const u32 iend = seen ? 64u - (u32)__clzll((long long)seen) : 0u;
    u32 mx = c;                                                       // largest count of the context
#pragma unroll
    for (int dd = 32; dd > 0; dd >>= 1) { const u32 o = (u32)__shfl_xor((int)mx, dd, 64); mx = o > mx ? o : mx; }
    u32 sh = 0;
    while ((((u64)mx * 6) >> sh) > 32000) sh++;
    const u32 fq = (u32)(((u64)c * 6) >> sh);                          // this symbol's scaled frequency (<= 32000)
    // ascending sort of (~freq, symbol) = scaled frequency descending, symbol ascending; lanes beyond iend last
    const u32 key = lane < iend ? (((0xFFFFu - fq) << 6) | lane) : (0xFFFFFFC0u | lane);
    const u32 sk = pr_sort64(key, lane);
    const u32 f = lane < iend ? 0xFFFFu - (sk >> 6) : 0u;
    const u32 sym = sk & 63u;
    const u32 slot = lane < iend ? (f | (sym << 16)) : 0u;
    u32 tot = lane < iend ? f : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) tot += (u32)__shfl_xor((int)tot, d, 64);
    // exchange form
    u32* r = rows66 + (size_t)ctx * 66;
    r[lane] = slot;
    if (lane == 0) { r[64] = tot; r[65] = iend; }
    // lane-per-block layout (dev_common.h RowHdr); epoch 0 never matches a running block
    l_slots[(size_t)ctx * 64 + lane] = slot;
    if (lane == 0) { RowHdr h; h.total = tot; h.iend = (u16)iend; h.count = 0; h.pad = 0; h.epoch = 0; h.pad2 = 0; l_hdr[ctx] = h; }
    // wave layout (models_w.hip WaveRow): dwords 0..2 header, 4..63 slots 0..59, overflow slots 60..63
    u32* w = w_rows + (size_t)ctx * 64;
    if (lane < 60) w[4 + lane] = slot; else w_ovf[(size_t)ctx * 4 + (lane - 60)] = slot;
    if (lane == 0) { w[0] = tot; w[1] = iend; w[2] = 0; w[3] = 0; }
}
// The rows the prior lists (iend != 0), back to back and in context order, for the host that packs "qlt.pri" from them: the
// dense table is 17 MB, of which a call's sample fills a few thousand rows -- the copy of all of it held a hardware queue for half
// a millisecond and was the tail of every small call (round 4).  list: [0] = n, [4 + 67 i ..] = context, then the row's 66 words.
// (two steps: which rows are listed -- a thread per context, all over the chip: the flags sit 264 bytes apart in the table, and
//  one workgroup reading all 65 536 of them took 0.7 ms --, then ONE workgroup turns the flags into places, in place)
__global__ __launch_bounds__(256) void k_prior_flags(const u32* __restrict__ rows66, u32 q_rows, u32* __restrict__ slot) {
    const u32 c = blockIdx.x * 256 + threadIdx.x;
    if (c < q_rows) slot[c] = rows66[(size_t)c * 66 + 65] != 0 ? 1u : 0u;
}
__global__ __launch_bounds__(1024) void k_prior_slots(u32 q_rows, u32* __restrict__ slot, u32* __restrict__ list) {
    __shared__ u32 wsum[16];
    const u32 t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const u32 per = (((q_rows + 1023u) / 1024u) + 3u) & ~3u, c0 = t * per;            // contexts per thread, a multiple of 4 (q_rows is a power of two >= 4096)
    u32 mine = 0;
    for (u32 k = 0; k < per; k += 4) {
        if (c0 + k + 4 <= q_rows) { const uint4 v = *reinterpret_cast<const uint4*>(slot + c0 + k); mine += v.x + v.y + v.z + v.w; }
        else for (u32 j = 0; j < 4; j++) if (c0 + k + j < q_rows) mine += slot[c0 + k + j];
    }
    u32 inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 o = (u32)__shfl_up((int)inc, d, 64); if (lane >= (u32)d) inc += o; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    u32 before = 0, all = 0;
    for (u32 w = 0; w < 16; w++) { const u32 v = wsum[w]; before += w < wave ? v : 0u; all += v; }
    u32 at = before + inc - mine;
    for (u32 k = 0; k < per; k++) {
        const u32 c = c0 + k;
        if (c >= q_rows) break;
        const bool on = slot[c] != 0;
        slot[c] = on ? at : ~0u;
        at += on ? 1u : 0u;
    }
    if (t == 0) { list[0] = all; list[1] = list[2] = list[3] = 0; }
}
__global__ __launch_bounds__(256) void k_prior_gather(const u32* __restrict__ rows66, u32 q_rows, const u32* __restrict__ slot, u32* __restrict__ list) {
    const u32 lane = threadIdx.x & 63;
    const u32 c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= q_rows) return;
    const u32 s = slot[c];
    if (s == ~0u) return;
    const u32* r = rows66 + (size_t)c * 66;
    u32* o = list + 4 + (size_t)s * 67;
    if (lane == 0) o[0] = c;
    o[1 + lane] = r[lane];
    if (lane < 2) o[65 + lane] = r[64 + lane];
}
void launch_prior_list(const u32* rows66, u32 q_rows, u32* slot, u32* list, hipStream_t st) {
    hipLaunchKernelGGL(k_prior_flags, dim3((q_rows + 255) / 256), dim3(256), 0, st, rows66, q_rows, slot);
    hipLaunchKernelGGL(k_prior_slots, dim3(1), dim3(1024), 0, st, q_rows, slot, list);
    hipLaunchKernelGGL(k_prior_gather, dim3((q_rows + 3) / 4), dim3(256), 0, st, rows66, q_rows, (const u32*)slot, list);
}
void launch_prior_rows(const u32* hist, u32 q_rows, u32* rows66, u32* w_rows, u32* w_ovf, u32* l_slots, RowHdr* l_hdr, hipStream_t st) {
    hipLaunchKernelGGL(k_prior_rows, dim3((q_rows + 3) / 4), dim3(256), 0, st, hist, q_rows, rows66, w_rows, w_ovf, l_slots, l_hdr);
}

// decode side: the host unpacked "qlt.pri" into the exchange form; spread it into the device layouts
__global__ __launch_bounds__(256) void k_prior_spread(const u32* rows66, u32 q_rows, u32* w_rows, u32* w_ovf, u32* l_slots, RowHdr* l_hdr) {
    const u32 lane = threadIdx.x & 63;
    const u32 ctx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ctx >= q_rows) return;
    const u32* r = rows66 + (size_t)ctx * 66;
    const u32 slot = r[lane], tot = r[64], iend = r[65];
    l_slots[(size_t)ctx * 64 + lane] = slot;
    if (lane == 0) { RowHdr h; h.total = tot; h.iend = (u16)iend; h.count = 0; h.pad = 0; h.epoch = 0; h.pad2 = 0; l_hdr[ctx] = h; }
    u32* w = w_rows + (size_t)ctx * 64;
    if (lane < 60) w[4 + lane] = slot; else w_ovf[(size_t)ctx * 4 + (lane - 60)] = slot;
    if (lane == 0) { w[0] = tot; w[1] = iend; w[2] = 0; w[3] = 0; }
}
// rows66[ctxs[i]] = rows[i] (66 words each): the rows a "qlt.pri" lists, into the zeroed dense table
__global__ __launch_bounds__(256) void k_prior_scatter(const u32* __restrict__ ctxs, const u32* __restrict__ rows, u32 n, u32* __restrict__ rows66) {
    const u32 i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    u32* dst = rows66 + (size_t)ctxs[i] * 66;
    const u32* src = rows + (size_t)i * 66;
    dst[lane] = src[lane];
    if (lane < 2) dst[64 + lane] = src[64 + lane];
}
void launch_prior_scatter(const u32* ctxs, const u32* rows, u32 n, u32* rows66, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_prior_scatter, dim3((n + 3) / 4), dim3(256), 0, st, ctxs, rows, n, rows66);
}
void launch_prior_spread(const u32* rows66, u32 q_rows, u32* w_rows, u32* w_ovf, u32* l_slots, RowHdr* l_hdr, hipStream_t st) {
    hipLaunchKernelGGL(k_prior_spread, dim3((q_rows + 3) / 4), dim3(256), 0, st, rows66, q_rows, w_rows, w_ovf, l_slots, l_hdr);
}
