// chains.hip -- the kernels of the frozen-table mode (dev_chain.h): a coding chain per lane over tables that do not change
// while chains are coded; the quality, base and header streams of the block format.
//
//   k_qlt_frozen_rows   prior rows (prior.hip, exchange form) -> dense direct-indexed rows {cum | freq << 16}[64]
//   k_hot_*             (optional) the rows the sample saw most, as an LDS image for the quality kernels' workgroups
//   k_qlt_encode_c      QltSave::save_1/2/3's symbol walk (qlts.cpp:74-136), one chain per lane, rows frozen
//   k_qlt_decode_c      QltLoad::load_1/2/3 (qlts.cpp:163-234), likewise
//   k_gen_count / k_gen_rows / k_gen_encode_c / k_gen_decode_c   bases: see the section below
//   k_rec_count(_f) / k_rec_frozen_rows                          the header prior's counting pass and rows
//   k_rec_tokens -> k_rec_code                                   headers: a record per lane makes the symbols, a chain per lane codes them
//   k_rec_encode_f / k_rec_encode_c                              headers, a chain per lane all the way (what the token step leaves)
//   k_rec_dsym -> k_rec_dtext                                    headers back: a chain per lane decodes the symbols, a record per lane rebuilds the texts
//   k_rec_decode_f / k_rec_decode_c                              RecLoad::load (recs.cpp:374-461), a chain per lane all the way (what those two leave)
//   k_chain_block_sizes / k_compact_chains   a block's chain streams packed back to back
//
// Text is read 16 bytes per lane at a time (any alignment).
#include "kernels.h"
#include "dev_chain.h"
#include "dev_walk.h"
// a dword of text at any address: vector global loads need no alignment on gfx9, SCALAR loads drop the address's low bits -- and a plain u32 pointer lets the
// compiler take one where the address is wave-uniform (the record that lane 0 of a wave stages alone)
typedef u32 __attribute__((aligned(1))) u32_any;

#define LAST_QLT 63u
#define QROW_BYTES 256u                 // 64 entries x 4 bytes
#define B2_INIT 0x03030303u             // Base2Ranger row of a fresh table (base2_ranger.hpp:68-71)

// ---- frozen quality rows -------------------------------------------------------------------------------------
// rows66: per context 64 slots (freq | sym << 16), total, iend (prior.hip k_prior_rows; the decoder unpacks "qlt.pri" into
// the same form).  A frozen row lists all 64 symbols in symbol order with the Log64Ranger's weights -- freq + 1 over
// total + 64 (log64_ranger.hpp:109) -- rescaled to a total of exactly 2^16, so that RCoder::Encode's range / tot
// (coder.hpp:68) is a shift:  x[s] = f[s] + 1,  g[s] = max(1, floor(x[s] * 65536 / sum x)),  and the difference
// 65536 - sum g goes to the largest g (the first of them).  Entry s = cum(g[0..s)) | g[s] << 16  (g <= 65473).
// A context the sample never saw has the uniform row (g = 1024).  Rows are indexed by the context itself: q_rows x 256 B,
// of which a file touches a few thousand rows (L2-resident).
// qdec (decode only, may be null): the decoder's form of a row, QDEC_ROW u16 -- the cum of every 8th symbol (8 of them: one
// 16-byte piece), then the cums of symbols 1 .. 64 (cum[64], the row's total 65536, is stored as 0): group k = the sixteen bytes
// that hold cum[8 k + 1 .. 8 k + 8].  A decoder finds a symbol with two 16-byte fetches -- the eighth of the row, then the
// symbol in it -- and has its cum and the NEXT one in hand: the group's first cum is the coarse entry it has just found.
#define QDEC_ROW 72u
// lane s stores cum[s] where the decoder's form wants it; lane 63 also the total behind the last symbol (65536 = 0)
__device__ __forceinline__ void qdec_store(u16* row, u32 s, u32 cum) {
    if (s) row[8 + s - 1] = (u16)cum;
    if ((s & 7u) == 0) row[s >> 3] = (u16)cum;
    if (s == 63) row[8 + 63] = 0;
}
__global__ __launch_bounds__(256) void k_qlt_frozen_rows(const u32* __restrict__ rows66, u32 q_rows, u32* __restrict__ qrows, u16* __restrict__ qdec) {
    const u32 lane = threadIdx.x & 63;
    const u32 ctx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ctx >= q_rows) return;
    const u32* r = rows66 + (size_t)ctx * 66;
    const u32 slot = r[lane], iend = r[65];
    if (iend == 0) {
        qrows[(size_t)ctx * 64 + lane] = (lane << 10) | (1024u << 16);
        if (qdec) qdec_store(qdec + (size_t)ctx * QDEC_ROW, lane, lane << 10);
        return;
    }
    // slot order -> symbol order: lane j sends its frequency to lane sym(j) (lanes >= iend hold no slot: they send 0 to themselves)
    const u32 sym = slot >> 16, fslot = lane < iend ? (slot & 0xffffu) : 0u;
    const u32 f = (u32)__builtin_amdgcn_ds_permute((int)((lane < iend ? sym : lane) * 4), (int)fslot);
    const u32 x = f + 1;
    u32 S = x;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) S += (u32)__shfl_xor((int)S, d, 64);
    u32 g = (x << 16) / S;                                  // x <= 65536: no overflow
    g = g ? g : 1u;
    u32 sum = g, best = (g << 6) | (63u - lane);             // largest g, lowest symbol first
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        sum += (u32)__shfl_xor((int)sum, d, 64);
        const u32 o = (u32)__shfl_xor((int)best, d, 64); best = o > best ? o : best;
    }
    if (lane == 63u - (best & 63u)) g += 65536u - sum;
    u32 incl = g;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 o = (u32)__shfl_up((int)incl, d, 64); if (lane >= (u32)d) incl += o; }
    qrows[(size_t)ctx * 64 + lane] = (incl - g) | (g << 16);
    if (qdec) qdec_store(qdec + (size_t)ctx * QDEC_ROW, lane, incl - g);
}
void launch_qlt_frozen_rows(const u32* rows66, u32 q_rows, u32* qrows, u16* qdec, hipStream_t st) {
    hipLaunchKernelGGL(k_qlt_frozen_rows, dim3((q_rows + 3) / 4), dim3(256), 0, st, rows66, q_rows, qrows, qdec);
}

// The rows the quality chains keep in LDS (BASELINE north_star: "ranger probability tables are staged in LDS"), picked on the
// device from the sample's counts -- no host round trip between the sample and the chains.  What bounds the chain kernel is
// the CU's vector memory path: a symbol's row entry is a gather whose 64 lanes touch 64 different lines (~128 cycles per wave
// instruction); an entry that lives in LDS costs three LDS reads instead.  The HOT IMAGE, one per call:
//   map   [q_rows / 32] x { u32 bits, u32 rank }   bit c & 31 of word c >> 5: context c is staged; its row = rank + the bits below it
//   rows  [K] x u16[QH_ROW_U16]                    cum of symbols 0 .. QH_SYMS (an entry's freq is the next cum minus its own)
// A context is eligible when its frozen row came from a prior row of at most QH_SYMS symbols (qualities up to 'P'); a symbol
// >= QH_SYMS of a staged context, and every other context, is read from the table in L2 as before.  The K contexts the sample
// saw most are staged (ties by count cut off together); where a row comes from does not show in the stream.
#define QH_SYMS 48u
#define QH_ROW_U16 50u
#define QH_MAP_BYTES(q_rows) ((q_rows) / 32u * 8u)
#define QH_EMPTY 0xFFFFFFFFu
// ctot[c] = symbols the sample saw in context c (0: not eligible)
__global__ __launch_bounds__(256) void k_hot_totals(const u32* __restrict__ hist, const u32* __restrict__ rows66, u32 q_rows, u32* __restrict__ ctot) {
    const u32 lane = threadIdx.x & 63;
    const u32 ctx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ctx >= q_rows) return;
    u32 c = hist[(size_t)ctx * 64 + lane];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += (u32)__shfl_xor((int)c, d, 64);
    const u32 iend = rows66[(size_t)ctx * 66 + 65];
    if (lane == 0) ctot[ctx] = (iend != 0 && iend <= QH_SYMS) ? c : 0u;
}
// one workgroup: the smallest threshold T with at most `want` contexts of ctot >= T; then the map (info[0] = rows staged)
__global__ __launch_bounds__(1024) void k_hot_select(const u32* __restrict__ ctot, u32 q_rows, u32 want, uint2* __restrict__ map, u32* __restrict__ info) {
    __shared__ u32 red[16];
    __shared__ u32 wbits[2048];                                            // the map's words: 32 contexts each
    const u32 t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const u32 nwords = q_rows / 32u;
    u32 c[64];                                                             // context k * 1024 + t: a wave reads 64 neighbours at a time
    u32 mx = 0;
#pragma unroll
    for (u32 k = 0; k < 64; k++) {
        const u32 ctx = k * 1024u + t;
        c[k] = ctx < q_rows ? ctot[ctx] : 0u;
        mx = c[k] > mx ? c[k] : mx;
    }
    auto block_sum = [&](u32 v) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += (u32)__shfl_xor((int)v, d, 64);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        u32 s2 = 0;
        for (u32 w = 0; w < 16; w++) s2 += red[w];
        return s2;
    };
    auto block_max = [&](u32 v) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { const u32 o = (u32)__shfl_xor((int)v, d, 64); v = o > v ? o : v; }
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        u32 s2 = 0;
        for (u32 w = 0; w < 16; w++) s2 = red[w] > s2 ? red[w] : s2;
        return s2;
    };
    u32 lo = 1, hi = block_max(mx) + 1u;                                 // count(>= hi) = 0 <= want
    while (lo < hi) {                                                      // the smallest T in [lo, hi] with count(>= T) <= want
        const u32 mid = lo + (hi - lo) / 2u;
        u32 n = 0;
#pragma unroll
        for (u32 k = 0; k < 64; k++) n += c[k] >= mid;
        if (block_sum(n) <= want) hi = mid; else lo = mid + 1u;
    }
    const u32 T = lo;
#pragma unroll
    for (u32 k = 0; k < 64; k++) {                                         // a ballot = the two map words of 64 neighbouring contexts
        const u64 m = __ballot(c[k] >= T);
        if (lane == 0) { wbits[k * 32u + wave * 2u] = (u32)m; wbits[k * 32u + wave * 2u + 1u] = (u32)(m >> 32); }
    }
    __syncthreads();
    const u32 b0 = 2u * t < nwords ? wbits[2u * t] : 0u, b1 = 2u * t + 1u < nwords ? wbits[2u * t + 1u] : 0u;
    const u32 mine = (u32)__popc(b0) + (u32)__popc(b1);
    u32 inc = mine;                                                        // ranks: a scan over the threads' two words
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 o = (u32)__shfl_up((int)inc, d, 64); if (lane >= (u32)d) inc += o; }
    __syncthreads();
    if (lane == 63) red[wave] = inc;
    __syncthreads();
    u32 before = 0, all = 0;
    for (u32 w = 0; w < 16; w++) { const u32 v = red[w]; before += w < wave ? v : 0u; all += v; }
    const u32 rank = before + inc - mine;
    if (t == 0) info[0] = all;
    if (2u * t < nwords) map[2u * t] = make_uint2(b0, rank);
    if (2u * t + 1u < nwords) map[2u * t + 1u] = make_uint2(b1, rank + (u32)__popc(b0));
}
// the staged rows: the cum of symbols 0 .. QH_SYMS of every staged context, at its place
__global__ __launch_bounds__(256) void k_hot_image(const uint2* __restrict__ map, const u32* __restrict__ qrows, u32 q_rows, u16* __restrict__ rows) {
    const u32 lane = threadIdx.x & 63;
    const u32 ctx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ctx >= q_rows) return;
    const uint2 mr = map[ctx >> 5];
    const u32 bit = 1u << (ctx & 31u);
    if (!(mr.x & bit)) return;
    const u32 slot = mr.y + (u32)__popc(mr.x & (bit - 1u));
    if (lane < QH_ROW_U16) rows[(size_t)slot * QH_ROW_U16 + lane] = lane <= QH_SYMS ? (u16)FZ_CUM(qrows[(size_t)ctx * 64 + lane]) : (u16)0;
}
// img: [QH_MAP_BYTES(q_rows)] map, then [want] rows; ctot: [q_rows] scratch; info[0] = rows staged
void launch_hot_rows(const u32* hist, const u32* rows66, const u32* qrows, u32 q_rows, u32 want, u32* ctot, u8* img, u32* info, hipStream_t st) {
    hipLaunchKernelGGL(k_hot_totals, dim3((q_rows + 3) / 4), dim3(256), 0, st, hist, rows66, q_rows, ctot);
    hipLaunchKernelGGL(k_hot_select, dim3(1), dim3(1024), 0, st, (const u32*)ctot, q_rows, want, reinterpret_cast<uint2*>(img), info);
    hipLaunchKernelGGL(k_hot_image, dim3((q_rows + 3) / 4), dim3(256), 0, st, reinterpret_cast<const uint2*>(img), qrows, q_rows,
                       reinterpret_cast<u16*>(img + QH_MAP_BYTES(q_rows)));
}

// The decoder's image: the same map; a staged row = the COARSE list of the decoder's form of the row (QDEC_ROW above: the cums
// of symbols 0, 8, .., 56 -- sixteen bytes).  Round 4: alone on the chip the quality decoder takes the same 10.5-12 ms with 3 or
// with 6 waves a SIMD and with its rows in L2 or in LDS -- what it runs at is the rate at which ONE per-CU unit serves sixteen
// bytes to each of a wave's 64 lanes from 64 different places: ~100 cycles a wave either through the vector memory path or out
// of LDS (bank conflicts), and a symbol needs two such reads.  So the two reads go to the two units: the coarse list of (nearly)
// every context the prior names out of LDS -- up to 6000 lists of sixteen bytes and the map are 110 KiB (9000 fit, and made the call slower: 23.9 ms against 20.0 -- the other decoders' workgroups then find no LDS beside it) --, the group of eight cums
// from the table in L2.  The contexts the prior gives the most weight are staged first (the choice is free: staging does not
// show in the text).
#define QHD_ROW_U16 8u
#define QHD_MAX_ROWS 6000u
__global__ __launch_bounds__(256) void k_hot_totals_prior(const u32* __restrict__ rows66, u32 q_rows, u32* __restrict__ ctot) {
    const u32 ctx = blockIdx.x * 256 + threadIdx.x;
    if (ctx >= q_rows) return;
    const u32 iend = rows66[(size_t)ctx * 66 + 65];
    ctot[ctx] = iend != 0 ? rows66[(size_t)ctx * 66 + 64] + 1u : 0u;
}
__global__ __launch_bounds__(256) void k_hot_image_dec(const uint2* __restrict__ map, const u16* __restrict__ qdec, u32 q_rows, u16* __restrict__ rows) {
    const u32 ctx = blockIdx.x * 256 + threadIdx.x;
    if (ctx >= q_rows) return;
    const uint2 mr = map[ctx >> 5];
    const u32 bit = 1u << (ctx & 31u);
    if (!(mr.x & bit)) return;
    const u32 slot = mr.y + (u32)__popc(mr.x & (bit - 1u));
    *reinterpret_cast<uint4*>(rows + (size_t)slot * QHD_ROW_U16) = *reinterpret_cast<const uint4*>(qdec + (size_t)ctx * QDEC_ROW);
}
void launch_hot_rows_dec(const u32* rows66, const u16* qdec, u32 q_rows, u32 want, u32* ctot, u8* img, u32* info, hipStream_t st) {
    hipLaunchKernelGGL(k_hot_totals_prior, dim3((q_rows + 255) / 256), dim3(256), 0, st, rows66, q_rows, ctot);
    hipLaunchKernelGGL(k_hot_select, dim3(1), dim3(1024), 0, st, (const u32*)ctot, q_rows, want, reinterpret_cast<uint2*>(img), info);
    hipLaunchKernelGGL(k_hot_image_dec, dim3((q_rows + 255) / 256), dim3(256), 0, st, reinterpret_cast<const uint2*>(img), qdec, q_rows,
                       reinterpret_cast<u16*>(img + QH_MAP_BYTES(q_rows)));
}
u32 hot_rows_dec_max(void) { return QHD_MAX_ROWS; }

// segments per record: lane per record
__global__ __launch_bounds__(256) void k_seg_count(const u64* __restrict__ line_off, const BlockDesc* __restrict__ blocks, u32 block_reads, u64 nrec, u32 seg_len, u32* __restrict__ nseg) {
    const u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrec) return;
    nseg[r] = seg_count_of(rec_symbols(line_off, r, blocks[r / block_reads].solid), seg_len);
}
__global__ __launch_bounds__(256) void k_seg_count_dec(const u32* __restrict__ slen, const u32* __restrict__ qlen, u64 nrec, u32 seg_len, u32* __restrict__ nseg) {
    const u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrec) return;
    nseg[r] = seg_count_of(slen[r] > qlen[r] ? slen[r] : qlen[r], seg_len);
}
__global__ __launch_bounds__(256) void k_seg_fill(const u64* __restrict__ seg_off, u64 nrec, u32* __restrict__ seg_rec) {
    const u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrec) return;
    for (u64 c = seg_off[r]; c < seg_off[r + 1]; c++) seg_rec[c] = (u32)r;
}
void launch_seg_count(const u64* line_off, const BlockDesc* blocks, u32 block_reads, u64 nrec, u32 seg_len, u32* nseg, hipStream_t st) {
    hipLaunchKernelGGL(k_seg_count, dim3((u32)((nrec + 255) / 256)), dim3(256), 0, st, line_off, blocks, block_reads, nrec, seg_len, nseg);
}
void launch_seg_count_dec(const u32* slen, const u32* qlen, u64 nrec, u32 seg_len, u32* nseg, hipStream_t st) {
    hipLaunchKernelGGL(k_seg_count_dec, dim3((u32)((nrec + 255) / 256)), dim3(256), 0, st, slen, qlen, nrec, seg_len, nseg);
}
void launch_seg_fill(const u64* seg_off, u64 nrec, u32* seg_rec, hipStream_t st) {
    hipLaunchKernelGGL(k_seg_fill, dim3((u32)((nrec + 255) / 256)), dim3(256), 0, st, seg_off, nrec, seg_rec);
}

// =========================================================================================================
// quality encode: one chain per lane
// =========================================================================================================
// LDS staging of the hottest rows: the workgroup copies the call's hot image (above) into LDS once; a symbol whose context
// is staged reads its entry there (map word, two cums), the others gather it from the L2-resident table.
#ifndef QLT_STEP
#define QLT_STEP 2u      /* symbols whose row entries are looked up at a time, a step ahead of the coder (16 / QLT_STEP steps a piece) */
#endif
#define QLT_RING 8       // ring dwords per lane: 15 bytes may wait for their row of 16, four symbols add at most 4 x (2 + 2 escape)
// MARK: the chains mark the records with a '!' for the exception pass (a.exc_flag; the framing does that where it can)
template <int THREADS, bool LDS, bool MARK>
__global__ __launch_bounds__(THREADS) void k_qlt_encode_c(ChainArgs a) {
    __shared__ u32 ring[LaneEncB<THREADS, QLT_RING>::LDS_DWORDS];
    extern __shared__ u32 lds[];                              // the hot image: map, then rows
    const uint2* const lmap = reinterpret_cast<const uint2*>(lds);
    const u16* const lrows = reinterpret_cast<const u16*>(lds + QH_MAP_BYTES(a.q_rows) / 4u);
    if constexpr (LDS) {
        const u32 nd = (QH_MAP_BYTES(a.q_rows) + a.qh_info[0] * QH_ROW_U16 * 2u + 3u) / 4u;
        const u32* src = reinterpret_cast<const u32*>(a.qh_img);
        for (u32 i = threadIdx.x; i < nd; i += THREADS) lds[i] = src[i];
        __syncthreads();
    }
    const u32 c = blockIdx.x * THREADS + threadIdx.x;
    const bool live = c < a.geo.nchains;
    ChainPos cp; cp.b = 0; cp.r0 = 0; cp.nrec = 0; cp.sub_lo = cp.sub_len = 0; cp.seg = 0; cp.nseg = 1;
    if (live) { cp = chain_pos(a, c); chain_seg_encode(a, cp); }
    const BlockDesc* d = &a.m.blocks[cp.b];
    LaneEncB<THREADS, QLT_RING> rc; u32 cap = 0;
    u8* outp = live ? chain_region(a, cp, SFQ_S_QLT, 2, 1, cap) : nullptr;
    rc.init(ring, threadIdx.x, outp, cap);
    const int level = a.m.level;
    const u32 mask12 = level == 1 ? 0xFFFu : 0xFFFFu;
    LineWalk lw; lw.init(a, cp.r0, cp.nrec, 3, live ? d->solid : 0u, cp.sub_lo, cp.sub_len);
    u32 last = 0, p1 = 0, p2 = 0, delta = 5;
    u32 extra = 0;
    // (a) the contexts of a piece's symbols depend on the text alone: all of its row entries are fetched at once -- and a
    // piece AHEAD of the one being coded, so the gathers' latency hides behind sixteen coder steps.  No branches and no
    // masks on the model's state: whatever follows the lane's bytes in a line's last piece runs through the same
    // instructions and is dropped by the coder's mask.
    // The row entries of QLT_STEP symbols at a time (round 5b; a piece's sixteen before): a piece's steps are looked up in turn, each while the step before
    // it is coded, so sixteen + QLT_STEP entries are live where sixteen + sixteen were.  (Tried on top: the lookup in two halves a coder step apart -- the
    // model's walk and the map word, then the row's place and the entries --, so that neither of the two dependent LDS reads is waited for: 92 bytes of
    // scratch a lane instead of 52, the call 13.4 ms instead of 11.4.  The kernel is at its registers' end.)
    const __amdgpu_buffer_rsrc_t qtab = __builtin_amdgcn_make_buffer_rsrc((void*)a.qrows, 0, (int)(a.q_rows * 256u), 0x00020000);      // (raw, bounds-checked: q_rows x 64 dwords)
    auto look8 = [&](const Piece& p, const uint4& f, const u32 j0, u32 (&e)[QLT_STEP], u32 (&eg)[QLT_STEP], u32& lowest, u32& top) {
        if (j0 == 0 && p.newline) { last = 0; p1 = p2 = 0; delta = 5; }          // qlts.cpp:109-112
        const u32 len = p.j1;
        lowest = 255; top = 0;
#pragma unroll
        for (u32 jj = 0; jj < QLT_STEP; jj++) {
            const u32 j = j0 + jj;
            const u32 vm = j < len ? ~0u : 0u;
            const u32 b = (piece_byte(f, j) - '!') & 0xffu;
            const u32 sym = b < LAST_QLT ? b : LAST_QLT;
            if constexpr (MARK) lowest = min(lowest, b | ~vm);
            top = max(top, b & vm);
            if constexpr (LDS) {
                const uint2 mr = lmap[last >> 5];
                const u32 bit = 1u << (last & 31u);
                // No branch between the image and the table (round 5b): of 64 lanes one nearly always misses the image, so the `else` ran for every symbol --
                // exec masks, a 64-bit address, the branch: 13 instructions of the kernel's 90 a symbol, 0.7 ms of the call.  The table is read through a
                // BUFFER descriptor instead: a lane that hit asks for offset 0xFFFFFFFF, which the bounds check answers with 0 without touching memory;
                // the image's entry is masked the other way (a lane that missed reads entry 0) and the two are ORed.
                const u32 hm = ((mr.x & bit) && sym < QH_SYMS) ? ~0u : 0u;
                const u32 ridx = ((mr.y + (u32)__popc(mr.x & (bit - 1u))) * QH_ROW_U16 + sym) & hm;
                const u32 c0 = lrows[ridx], c1 = lrows[ridx + 1];
                const u32 g = (u32)__builtin_amdgcn_raw_buffer_load_b32(qtab, (int)(((last << 8) | (sym << 2)) | hm), 0, 0);
                e[jj] = (c0 | ((c1 - c0) << 16)) & hm; eg[jj] = g;                 // (ORed where the entry is used: the table's answer need not be there before)
            } else { e[jj] = a.qrows[(size_t)last * 64 + sym]; eg[jj] = 0; }
            if (level <= 2) last = (b | (last << 6)) & mask12;                           // qlts.hpp:52-57
            else {                                                                       // qlts.hpp:62-74
                delta += max(p1, b) - b;                                                 // if (p1 > b) delta += p1 - b
                const u32 d3 = delta >> 3;
                last = (b | ((p1 < p2 ? p2 : p1) << 6) | ((u32)(p1 == p2) << 12) | ((d3 < 7 ? d3 : 7) << 13)) & 0xFFFFu;
                p2 = p1; p1 = b;
            }
        }
    };
    // (b) the serial part: the range coder, eight symbols
    auto code8 = [&](const Piece& p, const uint4& f, const u32 j0, const u32 (&e)[QLT_STEP], u32 lowest, u32 top) {
        const u32 len = p.j1;
        // a '!' marks the record for the pass over the N / quality-0 exceptions (k_gen_exc_w)
        if constexpr (MARK) if (lowest == 0 && p.valid) a.exc_flag[cp.r0 + p.rk] = 1;
        if (!__any(top >= LAST_QLT)) {
#pragma unroll
            for (u32 jj = 0; jj < QLT_STEP; jj++) {
                rc.encode16_if(j0 + jj < len ? ~0u : 0u, FZ_CUM(e[jj]), FZ_FREQ(e[jj]));
                if (((j0 + jj) & 3u) == 3u) rc.drain();
            }
        } else {
            // a quality over 62 somewhere in the wave: the escape symbol, then the raw value through the frozen escape row (qlts.cpp:80-86)
#pragma unroll
            for (u32 jj = 0; jj < QLT_STEP; jj++) {
                const u32 j = j0 + jj;
                const u32 vm = j < len ? ~0u : 0u;
                rc.encode16_if(vm, FZ_CUM(e[jj]), FZ_FREQ(e[jj]));
                const u32 b = (piece_byte(f, j) - '!') & 0xffu;
                const u32 em = (b >= LAST_QLT ? ~0u : 0u) & vm;
                if (__any(em != 0)) {
                    const u32 ee = a.qesc[b];
                    rc.encode16_if(em, FZ_CUM(ee), FZ_FREQ(ee));
                    extra -= em;
                }
                if ((j & 3u) == 3u) rc.drain();
            }
        }
    };
    Piece pc = lw.next();
    uint4 w = lw.fetch(pc);
    Piece pn = lw.next();
    uint4 wn = lw.fetch(pn);
    constexpr u32 NS = 16u / QLT_STEP;
    // (whether a step of the piece holds a quality over 62 ANYWHERE in the wave -- the escape's path -- is one bit a step in a scalar register; the lanes'
    //  own maxima, a word a step, were eight registers of a kernel that has none to spare)
    u32 E[NS][QLT_STEP], lows[MARK ? NS : 1u], esc = 0;
#pragma unroll
    for (u32 k = 0; k < NS; k++) {
        u32 G[QLT_STEP], lowt, topt;
        look8(pc, w, k * QLT_STEP, E[k], G, lowt, topt);
#pragma unroll
        for (u32 j = 0; j < QLT_STEP; j++) E[k][j] |= G[j];
        if (__any(topt >= LAST_QLT)) esc |= 1u << k;
        if constexpr (MARK) lows[k] = lowt;
    }
    while (__any(pc.valid)) {
        const Piece pnn = lw.next();                  // the text two pieces ahead, the rows a step ahead
        const uint4 wnn = lw.fetch(pnn);
        u32 escn = 0;
#pragma unroll
        for (u32 k = 0; k < NS; k++) {
            u32 T[QLT_STEP], G[QLT_STEP], lowt, topt;
            look8(pn, wn, k * QLT_STEP, T, G, lowt, topt);       // the next piece's step k, while this piece's is coded
            code8(pc, w, k * QLT_STEP, E[k], MARK ? lows[MARK ? k : 0u] : 255u, ((esc >> k) & 1u) ? LAST_QLT : 0u);
#pragma unroll
            for (u32 j = 0; j < QLT_STEP; j++) E[k][j] = T[j] | G[j];
            if (__any(topt >= LAST_QLT)) escn |= 1u << k;
            if constexpr (MARK) lows[k] = lowt;
        }
        esc = escn;
        pc = pn; w = wn; pn = pnn; wn = wnn;
    }
    if (live) {
        const u32 size = rc.finish();
        a.csz[c] = size;
        if (extra) atomicAdd(&a.m.blocks[cp.b].extra_hi, extra);
        if (rc.err & 2) atomicMax(&a.m.blocks[cp.b].status, (u32)(-SFQ_E_OVERFLOW));
        else if (rc.err) atomicMax(&a.m.blocks[cp.b].status, (u32)(-SFQ_E_CORRUPT));
    }
}
template <int T>
static void launch_qlt_lds(const ChainArgs& a, u32 dyn, hipStream_t st) {
    static u32 allowed = 0;
    if (dyn > allowed) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_qlt_encode_c<T, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); allowed = dyn; }
    hipLaunchKernelGGL((k_qlt_encode_c<T, true, false>), dim3((a.geo.nchains + T - 1) / T), dim3(T), dyn, st, a);
}
void launch_qlt_encode_c(const ChainArgs& a, hipStream_t st) {
    if (a.exc_flag) { hipLaunchKernelGGL((k_qlt_encode_c<256, false, true>), dim3((a.geo.nchains + 255) / 256), dim3(256), 0, st, a); return; }   // (chains that mark: no caller does)
    if (a.q_hot) {
        // One workgroup of 1024 lanes per CU shares the image (a table per 256 lanes would hold a quarter of the rows).  205 k chains
        // are 200 such workgroups: 56 CUs carry no quality chains -- and that is where the other kernels get their work done.  With
        // workgroups of 832 lanes on 247 CUs the quality chains took as long and the header coder 6.1 ms instead of 2.5.
        const u32 dyn = QH_MAP_BYTES(a.q_rows) + a.q_hot * QH_ROW_U16 * 2u;
        launch_qlt_lds<1024>(a, dyn, st);
    } else hipLaunchKernelGGL((k_qlt_encode_c<256, false, false>), dim3((a.geo.nchains + 255) / 256), dim3(256), 0, st, a);
}

// =========================================================================================================
// quality decode: one chain per lane
// =========================================================================================================
// Round 4.  The whole decode call runs at the chip's instruction-issue limit (10.8e9 wave instructions at ~4 cycles each on
// 1024 SIMDs = the 22 ms it took; profiles/r03z_pmc_summary.txt), and this kernel was half of them: 179 VALU + 41 SALU a symbol.
// What a symbol costs now:
//   * GetFreq's divide (coder.hpp:85) is a float reciprocal and ONE exact fix-up (div_exact below) instead of the compiler's
//     32-bit division sequence;
//   * the symbol search is two binary searches over packed u16 pairs -- three compares and five selects in the coarse list,
//     three compares and six selects in the group, the last select leaving (cum, next) in one dword (QDEC_ROW above);
//   * the coder keeps `code` in 32 bits (code < range while the stream is sound; what is not is reported) and renormalises
//     without branches: two masked steps, a loop only behind them (a symbol of probability below 2^-16);
//   * stream bytes come through a 64-bit shift register topped up four bytes at a time, the next four always in flight.
template <int THREADS, bool LDS>
__global__ __launch_bounds__(THREADS) void k_qlt_decode_c(ChainArgs a, DecodeArgs da) {
#ifdef PRIO_QDEC
    __builtin_amdgcn_s_setprio(PRIO_QDEC);
#endif
    extern __shared__ u32 lds[];                              // the decoder's image: map, then the staged contexts' coarse lists
    const uint2* const lmap = reinterpret_cast<const uint2*>(lds);
    const u16* const lrows = reinterpret_cast<const u16*>(lds + QH_MAP_BYTES(a.q_rows) / 4u);
    if constexpr (LDS) {
        const u32 nd = (QH_MAP_BYTES(a.q_rows) + a.qh_info[0] * QHD_ROW_U16 * 2u + 3u) / 4u;
        const u32* src = reinterpret_cast<const u32*>(a.qh_img);
        for (u32 i = threadIdx.x; i < nd; i += THREADS) lds[i] = src[i];
        __syncthreads();
    }
    const u32 c = blockIdx.x * THREADS + threadIdx.x;
    if (c >= a.geo.nchains) return;
    LaneDecQ rc; rc.init(da.streams + a.coff[c], a.csz[c], reinterpret_cast<const u8*>(a.qesc));
    const int level = a.m.level;
    const u32 mask12 = level == 1 ? 0xFFFu : 0xFFFFu;
    ChainPos cp = chain_pos(a, c);
    u32 n_next = cp.nrec ? da.qlen[cp.r0] : 0u; u64 off_next = cp.nrec ? da.qoff[cp.r0] : 0ull;      // a record's length and place, a record ahead
    if (a.seg_len) {                                         // a segment of one record: its part of the quality line
        const u32 sl = da.slen[cp.r0];
        seg_range(cp, sl > n_next ? sl : n_next);
        const u64 lo = cp.sub_lo < n_next ? cp.sub_lo : n_next;
        n_next = (u32)(n_next - lo < cp.sub_len ? n_next - lo : cp.sub_len); off_next += lo;
    }
    for (u32 k = 0; k < cp.nrec; k++) {
        const u32 n = n_next; const u64 off = off_next;
        if (k + 1 < cp.nrec) { n_next = da.qlen[cp.r0 + k + 1]; off_next = da.qoff[cp.r0 + k + 1]; }
#ifdef SFQ_EXP_OUT_LOCAL               /* scratch experiment: every lane's output into a small region that stays in L2 (the text comes out wrong) */
        LaneOut32 out; out.begin(da.qual_stage + (size_t)(c & 8191u) * 64u + 0 * off);
#else
        LaneOut32 out; out.begin(da.qual_stage + off);
#endif
        u32 last = 0, p1 = 0, p2 = 0, delta = 5;
        for (u32 i = 0; i < n; i++) {
            // largest s with cum[s] <= prob (cum is increasing: every g >= 1): the eighth of the row from the coarse list, then the
            // symbol among its eight -- two dependent fetches
            const u16* qd = a.qdec + (size_t)last * QDEC_ROW;
            u32 hot = ~0u;                                                  // the row's place in the LDS image, if it is staged
            if constexpr (LDS) {
                const uint2 mr = lmap[last >> 5];
#ifndef SFQ_EXP_QDEC_FLAT
                asm volatile("" :: "v"(mr.x), "v"(mr.y));                   // (both words in one read, not the rank behind a branch on the bits)
#endif
                const u32 bit = 1u << (last & 31u);
                if (mr.x & bit) hot = (mr.y + (u32)__popc(mr.x & (bit - 1u))) * QHD_ROW_U16;
            }
            uint4 cv;
#ifdef SFQ_EXP_QDEC_FLAT
            if (hot != ~0u) cv = *reinterpret_cast<const uint4*>(lrows + hot); else cv = *reinterpret_cast<const uint4*>(qd);
#else
            // two loads, not one through a generic pointer (what `staged ? LDS : table` compiles to: a flat load, which takes the
            // vector memory path for all 64 lanes): every lane reads LDS -- a lane whose row is not staged reads list 0 and drops
            // it --, the few lanes without a staged row gather theirs
            if constexpr (LDS) {
                const uint4 cl = *reinterpret_cast<const uint4*>(lrows + (hot != ~0u ? hot : 0u));
                uint4 cg = make_uint4(0, 0, 0, 0);
                if (hot == ~0u) cg = *reinterpret_cast<const uint4*>(qd);
                asm volatile("" :: "v"(cl.x), "v"(cl.y), "v"(cl.z), "v"(cl.w));        // (the LDS read is used whatever the lane: it stays an LDS read)
                const bool st = hot != ~0u;
                cv.x = st ? cl.x : cg.x; cv.y = st ? cl.y : cg.y; cv.z = st ? cl.z : cg.z; cv.w = st ? cl.w : cg.w;
            } else cv = *reinterpret_cast<const uint4*>(qd);
#endif
            rc.top_up();                                                    // (behind the row's fetch: the two loads travel together)
            u32 r;
            const u32 prob = rc.get_freq16(r);
            const bool t4 = (cv.z & 0xffffu) <= prob;
            const u32 A0 = t4 ? cv.z : cv.x, A1 = t4 ? cv.w : cv.y;
            const bool t2 = (A1 & 0xffffu) <= prob;
            const u32 B = t2 ? A1 : A0;
            const bool t1 = (B >> 16) <= prob;
            const u32 k8 = (t4 ? 4u : 0u) | (t2 ? 2u : 0u) | (t1 ? 1u : 0u);
            const u32 ff0 = t1 ? B >> 16 : B & 0xffffu;                      // cum[8 k8]
            // the group: w.x = (ff1, ff2), w.y = (ff3, ff4), w.z = (ff5, ff6), w.w = (ff7, ff8); ff[j] = cum[8 k8 + j]
            const uint4 w = *reinterpret_cast<const uint4*>(qd + 8 + k8 * 8u);
            // five in a row (x, y0 = two, y1 = two), then three (x, z = two), then the pair (cum, next)
            const bool u4 = (w.y >> 16) <= prob;
            const u32 x = u4 ? w.y >> 16 : ff0, y0 = u4 ? w.z : w.x, y1 = u4 ? w.w : w.y;
            const bool u2 = (y0 >> 16) <= prob;
            const u32 x2 = u2 ? y0 >> 16 : x, z = u2 ? y1 : y0;
            const bool u1 = (z & 0xffffu) <= prob;
            const u32 win = u1 ? z : (x2 | (z << 16));                      // cum | next << 16
            const u32 s = k8 * 8u + ((u4 ? 4u : 0u) | (u2 ? 2u : 0u) | (u1 ? 1u : 0u));
            const u32 cum = win & 0xffffu;
            rc.decode(r, cum, ((win >> 16) - cum) & 0xffffu);                               // (a next of 0 is the total, 65536)
            u32 b = s;
            if (s == LAST_QLT) {                                            // qlts.cpp:168-171: the raw value through the escape row,
                rc.top_up();                                                // whose 256 values are equally likely (api.cpp build_qesc)
                u32 re;
                const u32 pe = rc.get_freq16(re);
                b = pe >> 8; b = b > 255u ? 255u : b;
                rc.decode(re, b << 8, 256u);
            }
            out.put(('!' + b) & 0xffu);
            if (level <= 2) last = (b | (last << 6)) & mask12;
            else {
                delta += max(p1, b) - b;                                    // if (p1 > b) delta += p1 - b
                const u32 d3 = delta >> 3;
                last = (b | ((p1 < p2 ? p2 : p1) << 6) | ((u32)(p1 == p2) << 12) | ((d3 < 7 ? d3 : 7) << 13)) & 0xFFFFu;
                p2 = p1; p1 = b;
            }
        }
        out.end();
    }
    if (rc.err) atomicMax(&a.m.blocks[cp.b].status, (u32)(-SFQ_E_CORRUPT));
}
void launch_qlt_decode_c(const ChainArgs& a, const DecodeArgs& da, hipStream_t st) {
    if (a.q_hot) {
        constexpr int T = 1024;                                // one workgroup per CU shares the image
        const u32 dyn = QH_MAP_BYTES(a.q_rows) + a.q_hot * QHD_ROW_U16 * 2u;
        static u32 allowed = 0;
        if (dyn > allowed) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_qlt_decode_c<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); allowed = dyn; }
        hipLaunchKernelGGL((k_qlt_decode_c<T, true>), dim3((a.geo.nchains + T - 1) / T), dim3(T), dyn, st, a, da);
    } else {
        constexpr int T = 256;
        hipLaunchKernelGGL((k_qlt_decode_c<T, false>), dim3((a.geo.nchains + T - 1) / T), dim3(T), 0, st, a, da);
    }
}

// =========================================================================================================
// packing: a block's chain streams back to back
// =========================================================================================================
// blocks[b].size[stream] = sum of its chains' sizes: a wave per block, a chain per lane (a thread per block summed its
// chains one dependent load after the other: 1.1 ms of the call's tail for 10 k blocks)
__global__ __launch_bounds__(256) void k_chain_block_sizes(ChainArgs a, ChainGeoArgs geo, int stream, const u32* csz, const u32* rhb) {
    const u32 b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= a.m.nblocks) return;
    u32 sum = 0, hb = 0;
    // (the header chains have a geometry of their own, rgeo: never segments)
    const bool seg = a.seg_len && stream != SFQ_S_REC;
    const u64 c0 = seg ? block_chain0(a, geo, b) : (u64)b * geo.cpb;
    u64 c1 = seg ? block_chain1(a, geo, b) : c0 + geo.cpb;
    if (c1 > geo.nchains) c1 = geo.nchains;
    for (u64 cc = c0 + lane; cc < c1; cc += 64) { sum += csz[cc]; if (rhb) hb += rhb[cc]; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { sum += (u32)__shfl_xor((int)sum, d, 64); hb += (u32)__shfl_xor((int)hb, d, 64); }
    if (lane == 0) {
        a.m.blocks[b].size[stream] = sum;
        if (rhb) a.m.blocks[b].hdr_bytes = hb;
    }
}
void launch_chain_block_sizes(const ChainArgs& a, const ChainGeoArgs& geo, int stream, const u32* csz, const u32* rhb, hipStream_t st) {
    hipLaunchKernelGGL(k_chain_block_sizes, dim3((a.m.nblocks + 3) / 4), dim3(256), 0, st, a, geo, stream, csz, rhb);
}
// one wave per chain: region -> its place in the packed stream
__global__ __launch_bounds__(256) void k_compact_chains(ChainArgs a, ChainGeoArgs geo, int stream, u32 num, u32 den, const u32* csz,
                                                        const u64* blk_stream_off, const u64* stream_base, u8* out, const u32* gate) {
    const u32 c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= geo.nchains) return;
    if (gate && !gate[0]) return;                          // (frame.hip k_stream_gate: a failed block or too little room -- nothing is written)
    const bool seg = a.seg_len && stream != SFQ_S_REC;    // (the header chains have a geometry of their own: never segments)
    if (!seg) a.seg_len = 0;
    a.geo = geo;                                           // chain_pos / chain_region read the geometry from `a`
    ChainPos cp = chain_pos(a, c);
    const u32 n = csz[c];
    if (!n) return;
    u32 before = 0;
    for (u64 cc = block_chain0(a, geo, cp.b) + lane; cc < c; cc += 64) before += csz[cc];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) before += (u32)__shfl_xor((int)before, d, 64);
    u32 cap;
    const u8* src = chain_region(a, cp, stream, num, den, cap);
    u8* dst = out + stream_base[stream] + blk_stream_off[(u64)cp.b * SFQ_NSTREAMS + stream] + before;
    // sixteen bytes a lane and step (any alignment on either side); a byte a lane took 0.85 ms of every call's tail for 0.74 GB
    const u32 n16 = n >> 4;
    for (u32 i = lane; i < n16; i += 64) {
        const u32* q = reinterpret_cast<const u32*>(src + 16 * i);
        const uint4 v = make_uint4(q[0], q[1], q[2], q[3]);
        u32* w = reinterpret_cast<u32*>(dst + 16 * i);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    }
    for (u32 i = (n16 << 4) + lane; i < n; i += 64) dst[i] = src[i];
}
void launch_compact_chains(const ChainArgs& a, const ChainGeoArgs& geo, int stream, u32 num, u32 den, const u32* csz, const u64* blk_stream_off,
                           const u64* stream_base, u8* out, hipStream_t st, const u32* gate) {
    hipLaunchKernelGGL(k_compact_chains, dim3((geo.nchains + 3) / 4), dim3(256), 0, st, a, geo, stream, num, den, csz, blk_stream_off, stream_base, out, gate);
}

// =========================================================================================================
// bases
//
// The reference's base model is ONE file-wide table of Base2Ranger rows (gens.cpp:36-44, base2_ranger.hpp) that
// learns the genome as coverage accumulates; independent blocks cannot (1.17x the reference's bytes on
// genome-sampled reads).  Format 7 cuts the call's blocks into GENERATIONS (the first 1/64 of the blocks, then
// growing geometrically): a chain of generation g codes with the rows built from the exact counts of every base of
// generations < g, frozen while the generation is coded; a decoder rebuilds the same rows from what it has already
// decoded.  Counting is integer atomics (commutative: the counts do not depend on scheduling); a row is
// f[i] = 3 + GEN_STEP * n[i], halved the reference's way -- (f >> 1) | (f & 1), base2_ranger.hpp:48-53 -- until every
// f[i] <= 255.  Generations 0 and 1 code with the initial row (3, 3, 3, 3); whether the tables are used at all
// is decided once, from the cost generation 1 would have had under the rows of generation 0 (api.cpp).
// N / quality-0 exceptions (gens.cpp:91-136) do not touch the model (an N is coded as 'A'): a wave per block
// scans for them and codes the reference's gen.Ns / gen.Nn side streams.
// =========================================================================================================
// per-lane walk over the base lines of records [r0, r0 + nrec): look(ctx) for every base of a piece first (so that a
// caller can start its table lookups together), then code(j, code) for each base in order.  The bases come from the FASTQ
// text or from the decoder's staged bases (ChainArgs::st_*).  N is coded as 0 (gens.cpp:116-136).
// seg_len != 0: only the seg-th stretch of seg_len bases of the (one) record; the sixteen bases before it are run
// through the context alone, so that the contexts of the stretch are what a walk from the line's start would give
template <typename LOOK, typename CODE>
__device__ __forceinline__ void walk_bases(const ChainArgs& a, u64 r0, u32 nrec, u32 solid, u32 mask, LOOK&& look, CODE&& code, u32 seg = 0, u32 seg_len = 0) {
    const u32 warm = (seg_len && seg) ? 16u : 0u;                  // (seg_len is at least 16)
    LineWalk lw;
    if (seg_len) lw.init(a, r0, nrec, 1, solid, (u64)seg * seg_len - warm, (u64)seg_len + warm);
    else lw.init(a, r0, nrec, 1, solid);
    u32 last = 0, seen = 0;
    Piece pc = lw.next();
    uint4 w = lw.fetch(pc);
    while (__any(pc.valid)) {
        const Piece pn = lw.next();
        const uint4 wn = lw.fetch(pn);
        if (pc.newline) last = 0x007616c7u;                        // gens.cpp:139 (a stretch behind the line's start overwrites it in its warm-up)
        u32 codes = 0;                                             // 2 bits per base of the piece
        const u32 seen0 = seen;
#pragma unroll
        for (u32 j = 0; j < 16; j++) {
            if (j >= pc.j0 && j < pc.j1) {
                const u32 cd = gen_code_of(piece_byte(w, j)) & 3u;
                if (seen >= warm) look(j, last & mask);
                seen++;
                last = (last << 2) | cd;
                codes |= cd << (2 * j);
            }
        }
        seen = seen0;
#pragma unroll
        for (u32 j = 0; j < 16; j++) if (j >= pc.j0 && j < pc.j1) { if (seen >= warm) code(j, (codes >> (2 * j)) & 3u); seen++; }
        pc = pn; w = wn;
    }
}

// The encoder's walk, without branches inside a piece: the codes come from a 256-entry table in LDS (code | 4 for an
// N-like character | 0x10 for an illegal one | 0x20 for a lowercase one), look(j, ctx) runs for all 16 positions (a position outside the lane's
// bytes repeats the context before it, so its lookup is harmless), code(j, code, valid) likewise with a flag, and
// piece_end() once per piece.
template <typename LOOK, typename CODE, typename PEND>
__device__ __forceinline__ u32 walk_bases_b(const ChainArgs& a, u64 r0, u32 nrec, u32 solid, u32 mask, const u8* lut, LOOK&& look, CODE&& code, PEND&& piece_end, u8* exc_flag,
                                            u64 sub_lo = 0, u64 sub_len = 0) {
    LineWalk lw; lw.init(a, r0, nrec, 1, solid, sub_lo, sub_len);
    u32 last = 0, illegal = 0;
    Piece pc = lw.next();
    uint4 w = lw.fetch(pc);
    while (__any(pc.valid)) {
        const Piece pn = lw.next();
        const uint4 wn = lw.fetch(pn);
        if (pc.newline) last = 0x007616c7u;                        // gens.cpp:139
        const uint4& f = w;                                        // (a line's last piece carries other bytes behind the lane's)
        const u32 len = pc.j1;
        u32 cd4[16];
#pragma unroll
        for (u32 j = 0; j < 16; j++) cd4[j] = lut[piece_byte(f, j)];
        u32 codes = 0, odd = 0;
#pragma unroll
        for (u32 j = 0; j < 16; j++) {
            look(j, last & mask);
            last = (last << 2) | (cd4[j] & 3u);                    // (behind the line's last base the context no longer matters)
            codes |= (cd4[j] & 3u) << (2 * j);
            odd |= cd4[j] & (j < len ? ~0u : 0u);
        }
        // an N, a lowercase base or an illegal character marks the record for the pass over the N / quality-0 / case exceptions (k_gen_exc_w)
        if (exc_flag && (odd & 0x34u) && pc.valid) exc_flag[r0 + pc.rk] = 1;
        illegal |= odd & 0x10u;                                    // (unexpected genome char, gens.cpp:125-126: reported by the chain itself)
#pragma unroll
        for (u32 j = 0; j < 16; j++) {
            code(j, (codes >> (2 * j)) & 3u, j < len ? ~0u : 0u);
            if ((j & 3u) == 3u) piece_end();                       // every four bases (the size of LaneEncB's ring follows from this)
        }
        pc = pn; w = wn;
    }
    return illegal;
}

// The walk of a chain without a model (round 5): quad(sym, k) for every four bases of a line -- sym = their codes, the first in the
// low bits; k = 4, fewer (or none: the quad lies behind the line's end) in a line's last piece.
template <typename QUAD>
__device__ __forceinline__ u32 walk_bases_q(const ChainArgs& a, u64 r0, u32 nrec, u32 solid, const u8* lut, QUAD&& quad, u8* exc_flag, u64 sub_lo = 0, u64 sub_len = 0) {
    LineWalk lw; lw.init(a, r0, nrec, 1, solid, sub_lo, sub_len);
    u32 illegal = 0;
    Piece pc = lw.next();
    uint4 w = lw.fetch(pc);
    while (__any(pc.valid)) {
        const Piece pn = lw.next();
        const uint4 wn = lw.fetch(pn);
        const u32 len = pc.j1;
        u32 codes = 0, odd = 0;
#pragma unroll
        for (u32 j = 0; j < 16; j++) {
            const u32 cd = lut[piece_byte(w, j)];
            codes |= (cd & 3u) << (2 * j);
            odd |= cd & (j < len ? ~0u : 0u);
        }
        if (exc_flag && (odd & 0x34u) && pc.valid) exc_flag[r0 + pc.rk] = 1;
        illegal |= odd & 0x10u;
#pragma unroll
        for (u32 q = 0; q < 4; q++) {
            const u32 k = len > 4u * q ? (len - 4u * q < 4u ? len - 4u * q : 4u) : 0u;
            quad((codes >> (8u * q)) & 0xffu, k);
        }
        pc = pn; w = wn;
    }
    return illegal;
}

// counts of (context, base) over the records of blocks [b0, b1): one record per lane.  With `rows` given, also the
// cost (in 1/1024 bit) those bases would have under these rows: cost[0] += sum log2(tot) - log2(f[code])
// Long lines (seg_len != 0): a lane takes one stretch of seg_len bases of a record, lane id = record x segs + stretch --
// a lane per 30 kb read made this pass 95 ms of the long-read workload's 108.
// sub: 0 = every stride-th record; 2 = all of them but every GEN_PRE-th (what the pre-verdict's sample -- a launch with GEN_PRE times
// the stride, api.cpp gen_tables_begin -- has counted already).  do_count = 0 (with rows): nothing is counted, and cost[] receives the
// pre-verdict's statistic instead of the cost; do_count = 2 (with rows): the cost alone (the counting goes through the bins, below).
__global__ __launch_bounds__(256) void k_gen_count(ChainArgs a, u32 b0, u32 b1, u32 stride, u32 seg_len, u32 segs, u32* __restrict__ cnt, const u32* __restrict__ rows,
                                                  const u16* __restrict__ log2fp, u64* cost, u32 sub, u32 do_count) {
    const u64 first = a.m.blocks[b0].rec0, endr = a.m.blocks[b1 - 1].rec0 + a.m.blocks[b1 - 1].nrec;
    u32 mybases = 0;
    const u64 id = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 r = first + (id / segs) * stride;                    // every stride-th record of the generation (gen_count_stride)
    const u32 seg = (u32)(id % segs);
    const bool live = r < endr && (sub == 0 || (id / segs) % GEN_PRE != 0);            // sub == 2: the records the pre-verdict's sample left
    u32 solid = 0, mask = 0;
    if (live) {
        // the block of record r: blocks are uniform (block_reads records) except the last
        const u32 b = (u32)(r / a.block_reads);
        solid = a.m.blocks[b].solid; mask = (1u << a.m.blocks[b].gen_bits) - 1u;
    }
    u64 mycost = 0;
    u32 cx[16], rv[16];
    walk_bases(a, r, live ? 1u : 0u, solid, mask,
        [&](u32 j, u32 ctx) { cx[j] = ctx; rv[j] = rows ? rows[ctx] : 0u; },
        [&](u32 j, u32 code) {
            if (do_count == 1) atomicAdd(&cnt[((size_t)cx[j] << 2) | code], 1u);
            if (rows) {
                const u32 v = rv[j];
                const u32 f0 = v & 0xff, f1 = (v >> 8) & 0xff, f2 = (v >> 16) & 0xff, f3 = v >> 24;
                if (do_count) {
                    mybases++;
                    mycost += (u32)log2fp[(f0 + f1) + (f2 + f3)] - (u32)log2fp[(v >> (8 * code)) & 0xff];
                } else if (v != 0x03030303u) {
                    // the pre-verdict's statistic: over the bases whose context the sample has seen (m times), 4 x (times it saw THIS
                    // base) - m.  Bases that do not depend on their context give 0 on average, with variance 3 m.
                    const u32 m = ((f0 + f1) + (f2 + f3) - 12u) >> 2;
                    mybases += m;
                    mycost += (u64)(long long)((int)((v >> (8 * code)) & 0xff) - 3 - (int)m);
                }
            }
        }, seg, seg_len);
    if (rows) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) mycost += __shfl_xor(mycost, d, 64);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) mybases += (u32)__shfl_xor((int)mybases, d, 64);
        if ((threadIdx.x & 63) == 0 && mybases) {
            atomicAdd((unsigned long long*)cost, (unsigned long long)mycost);              // (two's complement: the pre-verdict's sum is signed)
            atomicAdd((unsigned long long*)cost + 1, (unsigned long long)mybases);
        }
    }
}
void launch_gen_count(const ChainArgs& a, u32 b0, u32 b1, u64 nrec_range, u32 max_line, u32* cnt, const u32* rows, const u16* log2fp, u64* cost, hipStream_t st, u32 sub, u32 do_count) {
    if (!nrec_range) return;
    u32 stride = gen_count_stride(nrec_range);
    if (sub == 1) { stride *= GEN_PRE; sub = 0; }             // (the sample alone: a launch of its lanes only)
    // lines up to 1 KiB: a lane per record; longer: a lane per stretch of 512 bases (the lanes past a record's end idle).
    // A SMALL generation (the first ones: 1/64 of the call each, whose passes every base chain waits for) is latency, not
    // throughput: its time is one lane's walk through its record, so the records are cut into the shortest stretches (of 32
    // bases or more, sixteen bases of warm-up before each) that still fit the chip's 524 288 lanes at once
    // (round 4: long lines too -- the pre-verdict's sample of a long-read call is a hundred records of 30 kb, and at 512 bases a
    //  lane its two passes took 1.2 + 1.5 ms with every chain of the call waiting for them)
    u32 seg_len = 0;
    if (max_line >= 64u) {
        const u64 nsel = (nrec_range + stride - 1) / stride;
        for (u32 sl = 32u; sl < max_line && sl <= 512u; sl *= 2u)
            if (nsel * ((max_line + sl - 1) / sl) <= 524288ull) { seg_len = sl; break; }
    }
    if (!seg_len && max_line > 1024u) seg_len = 512u;
    const u32 segs = seg_len ? (max_line + seg_len - 1) / seg_len : 1u;
    const u64 lanes = ((nrec_range + stride - 1) / stride) * segs;
    hipLaunchKernelGGL(k_gen_count, dim3((u32)((lanes + 255) / 256)), dim3(256), 0, st, a, b0, b1, stride, seg_len, segs, cnt, rows, log2fp, cost, sub, do_count);
}
// ---- counting through bins (round 5) --------------------------------------------------------------------------------
// A scattered atomic per base is what k_gen_count's counting costs: the chip retires 27 G of them a second whatever the table's
// size (profiles/r02a_atomics_*), 330 M per 10 M-read call -- a third of the time of every input whose bases can be learned.  The
// same counts without a global atomic per base, in two streaming passes:
//   k_gen_bin        the walk of k_gen_count, but a base's key (context << 2 | base) goes to the BIN of its context's slice of the
//                    table -- 2^slice_bits contexts, what pass two holds in LDS -- as a 16-bit remainder.  A workgroup collects a
//                    tile of keys in LDS, counts them per bin (LDS atomics), reserves room in every bin with ONE global atomic per
//                    bin and tile, and stores the remainders (a bin that is full sends its key to the table's counter directly:
//                    the counts are exact whatever the bins' capacity);
//   k_gen_bin_count  a workgroup per bin: the slice's counters in LDS (two u16 per word; the bin is taken 65 535 keys at a time, so
//                    a field cannot overflow), then added to the table's slice with coalesced 16-byte accesses -- and, where the
//                    caller wants the next generation's rows, those are written from the sums at once (k_gen_rows folded in).
// Counts are sums: the order in which keys reach a bin does not show.  Both coders call this through launch_gen_count_binned.
#define GB_T 256u                        // lanes of a binning workgroup
#define GB_ROUNDS 2u                     // pieces of sixteen bases a lane adds to a tile
#define GB_TILE (GB_T * 16u * GB_ROUNDS) // keys per tile
#define GB_SLICE_BITS 14u                // contexts per bin at most: 16 384 x 8 bytes of counters = 128 KiB of LDS
__host__ __device__ __forceinline__ u32 gb_slice_bits(u32 g_bits) { return g_bits < GB_SLICE_BITS ? g_bits : GB_SLICE_BITS; }

__global__ __launch_bounds__(GB_T) void k_gen_bin(ChainArgs a, u32 b0, u32 b1, u32 stride, u32 seg_len, u32 segs, u32 sub, u64 id0, u64 id1, u32 g_bits,
                                                  u16* __restrict__ bins, u32* __restrict__ fill, u32 cap, u32* __restrict__ cnt) {
    extern __shared__ u32 lds[];
    const u32 sb = gb_slice_bits(g_bits), nb = 1u << (g_bits - sb), sh = sb + 2u;
    u32* keys = lds;                     // [GB_TILE], laid out [round x 16 + j][lane]
    u32* hist = lds + GB_TILE;           // [nb] keys per bin of this tile; then the cursor inside the tile's reservation
    u32* base = hist + nb;               // [nb] where the tile's keys of a bin go in the bin
    const u32 tid = threadIdx.x;
    const u64 first = a.m.blocks[b0].rec0, endr = a.m.blocks[b1 - 1].rec0 + a.m.blocks[b1 - 1].nrec;
    const u64 id = id0 + (u64)blockIdx.x * GB_T + tid;
    const u64 r = first + (id / segs) * stride;
    const u32 seg = (u32)(id % segs);
    const bool live = id < id1 && r < endr && (sub == 0 || (id / segs) % GEN_PRE != 0);
    u32 solid = 0; const u32 mask = (1u << g_bits) - 1u;
    if (live) solid = a.m.blocks[(u32)(r / a.block_reads)].solid;
    const u32 warm = (seg_len && seg) ? 16u : 0u;
    LineWalk lw;
    if (seg_len) lw.init(a, r, live ? 1u : 0u, 1, solid, (u64)seg * seg_len - warm, (u64)seg_len + warm);
    else lw.init(a, r, live ? 1u : 0u, 1, solid);
    u32 last = 0, seen = 0;
    Piece pc = lw.next();
    uint4 w = lw.fetch(pc);
    for (;;) {
        for (u32 i = tid; i < nb; i += GB_T) hist[i] = 0;
        __syncthreads();
        int any = 0;
#pragma unroll
        for (u32 rd = 0; rd < GB_ROUNDS; rd++) {
            const Piece pn = lw.next();
            const uint4 wn = lw.fetch(pn);
            if (pc.newline) last = 0x007616c7u;                    // gens.cpp:139
#pragma unroll
            for (u32 j = 0; j < 16; j++) {
                u32 k = ~0u;
                if (j < pc.j1) {
                    const u32 cd = gen_code_of(piece_byte(w, j)) & 3u;
                    if (seen >= warm) k = ((last & mask) << 2) | cd;
                    seen++;
                    last = (last << 2) | cd;
                }
                keys[(rd * 16u + j) * GB_T + tid] = k;
                if (k != ~0u) atomicAdd(&hist[k >> sh], 1u);
            }
            any |= pc.valid ? 1 : 0;
            pc = pn; w = wn;
        }
        if (!__syncthreads_or(any)) break;
        for (u32 i = tid; i < nb; i += GB_T) { const u32 n = hist[i]; base[i] = n ? atomicAdd(&fill[i], n) : 0u; hist[i] = 0; }
        __syncthreads();
        for (u32 i = tid; i < GB_TILE; i += GB_T) {
            const u32 k = keys[i];
            if (k == ~0u) continue;
            const u32 b = k >> sh;
            const u32 pos = base[b] + atomicAdd(&hist[b], 1u);
            if (pos < cap) bins[(size_t)b * cap + pos] = (u16)(k & ((1u << sh) - 1u));
            else atomicAdd(&cnt[k], 1u);                           // the bin is full: straight to the counter
        }
        __syncthreads();
    }
}
// rows_out (may be null): the rows f = 3 + step x n of the contexts' sums, as k_gen_rows writes them
__device__ __forceinline__ u32 gen_row_of(uint4 n, u32 step) {
    u64 f0 = 3 + (u64)step * n.x, f1 = 3 + (u64)step * n.y, f2 = 3 + (u64)step * n.z, f3 = 3 + (u64)step * n.w;
    while ((f0 | f1 | f2 | f3) > 255) {                     // any of them > 255 (all are < 2^35)
        f0 = (f0 >> 1) | (f0 & 1); f1 = (f1 >> 1) | (f1 & 1); f2 = (f2 >> 1) | (f2 & 1); f3 = (f3 >> 1) | (f3 & 1);
    }
    return (u32)f0 | (u32)f1 << 8 | (u32)f2 << 16 | (u32)f3 << 24;
}
__global__ __launch_bounds__(1024) void k_gen_bin_count(const u16* __restrict__ bins, u32* __restrict__ fill, u32 cap, u32 g_bits, u32* __restrict__ cnt,
                                                        u32* __restrict__ rows_out, u32 step) {
    extern __shared__ u32 lds[];         // [2^sb][2]: word 0 = codes 0 | 1 << 16, word 1 = codes 2 | 3 << 16
    const u32 sb = gb_slice_bits(g_bits), nctx = 1u << sb, b = blockIdx.x, tid = threadIdx.x;
    u32 n = fill[b]; if (n > cap) n = cap;
    const u16* bin = bins + (size_t)b * cap;
    uint4* slice = reinterpret_cast<uint4*>(cnt + ((size_t)b << (sb + 2)));
    if (!n && !rows_out) return;
    for (u32 at = 0; ; at += 65535u) {
        for (u32 i = tid; i < 2u * nctx; i += 1024u) lds[i] = 0;
        __syncthreads();
        const u32 m = n - at < 65535u ? n - at : 65535u;
        for (u32 i = tid; i < m; i += 1024u) {
            const u32 k = bin[at + i];
            atomicAdd(&lds[k >> 1], 1u << ((k & 1u) * 16u));         // k = context << 2 | code: word (context, code >> 1), field code & 1
        }
        __syncthreads();
        const bool last_chunk = at + m >= n;
        for (u32 i = tid; i < nctx; i += 1024u) {
            const u32 w0 = lds[2 * i], w1 = lds[2 * i + 1];
            if (!(w0 | w1) && !(rows_out && last_chunk)) continue;
            uint4 c = slice[i];
            c.x += w0 & 0xffffu; c.y += w0 >> 16; c.z += w1 & 0xffffu; c.w += w1 >> 16;
            if (w0 | w1) slice[i] = c;
            if (rows_out && last_chunk) rows_out[((size_t)b << sb) + i] = gen_row_of(c, step);
        }
        if (last_chunk) break;
        __syncthreads();
    }
    if (tid == 0) fill[b] = 0;           // (ready for the next batch)
}
// the counts of launch_gen_count(a, b0, b1, ..., cnt, null, null, null, st, sub, 1) through bins; rows_out: the rows of the sums as well
// (then launch_gen_rows is not needed).  gb: scratch sized by gen_bins_plan(), gb.fill zeroed once by the caller.
GenBins gen_bins_plan(u32 g_bits, u64 max_keys) {
    // a pair of passes takes at most `batch` keys; a bin of a table whose contexts are hit evenly gets its share of them: the bins
    // hold twice that and a tile's worth of slack (what does not fit is counted by the atomics the bins replace)
    GenBins gb; gb.bins = nullptr; gb.fill = nullptr;
    gb.g_bits = g_bits;
    gb.batch = max_keys < GEN_BIN_BATCH ? (max_keys ? max_keys : 1) : GEN_BIN_BATCH;
    const u32 nb = 1u << (g_bits - gb_slice_bits(g_bits));
    gb.cap = (u32)((2ull * ((gb.batch + nb - 1) / nb) + 4096ull + 63ull) & ~63ull);
    gb.bins_bytes = (u64)gb.cap * 2ull * nb; gb.fill_bytes = 4ull * nb;
    return gb;
}
void launch_gen_count_binned(const ChainArgs& a, u32 b0, u32 b1, u64 nrec_range, u32 max_line, const GenBins& gb, u32* cnt, u32* rows_out, u32 step,
                             hipStream_t st, u32 sub) {
    const u32 g_bits = gb.g_bits, sb = gb_slice_bits(g_bits), nb = 1u << (g_bits - sb), cap = gb.cap;
    const u32 dyn2 = (2u << sb) * 4u;
    static u32 allowed = 0;
    if (dyn2 > 65536u && dyn2 > allowed) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gen_bin_count), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn2); allowed = dyn2; }
    if (nrec_range) {
        u32 stride = gen_count_stride(nrec_range);
        if (sub == 1) { stride *= GEN_PRE; sub = 0; }
        u32 seg_len = 0;                                            // (as launch_gen_count)
        if (max_line >= 64u) {
            const u64 nsel = (nrec_range + stride - 1) / stride;
            for (u32 sl = 32u; sl < max_line && sl <= 512u; sl *= 2u)
                if (nsel * ((max_line + sl - 1) / sl) <= 524288ull) { seg_len = sl; break; }
        }
        if (!seg_len && max_line > 1024u) seg_len = 512u;
        const u32 segs = seg_len ? (max_line + seg_len - 1) / seg_len : 1u;
        const u64 lanes = ((nrec_range + stride - 1) / stride) * segs;
        // lanes of one batch: their keys (a lane holds seg_len bases, or a whole line of up to max_line) fit the bins' batch
        const u64 per_lane = seg_len ? seg_len : (max_line ? max_line : 1u);
        u64 lanes_per_batch = gb.batch / per_lane; if (!lanes_per_batch) lanes_per_batch = 1;
        lanes_per_batch = (lanes_per_batch + GB_T - 1) / GB_T * GB_T;
        const u32 dyn1 = (GB_TILE + 2u * nb) * 4u;
        for (u64 id0 = 0; id0 < lanes; id0 += lanes_per_batch) {
            const u64 id1 = id0 + lanes_per_batch < lanes ? id0 + lanes_per_batch : lanes;
            const bool last = id1 >= lanes;
            hipLaunchKernelGGL(k_gen_bin, dim3((u32)((id1 - id0 + GB_T - 1) / GB_T)), dim3(GB_T), dyn1, st, a, b0, b1, stride, seg_len, segs, sub, id0, id1, g_bits, gb.bins, gb.fill, cap, cnt);
            hipLaunchKernelGGL(k_gen_bin_count, dim3(nb), dim3(1024), dyn2, st, (const u16*)gb.bins, gb.fill, cap, g_bits, cnt, last ? rows_out : nullptr, step);
        }
    } else if (rows_out) hipLaunchKernelGGL(k_gen_bin_count, dim3(nb), dim3(1024), dyn2, st, (const u16*)gb.bins, gb.fill, cap, g_bits, cnt, rows_out, step);
}

// rows from counts
__global__ __launch_bounds__(256) void k_gen_rows(const u32* __restrict__ cnt, u32* __restrict__ rows, u64 nctx, u32 step) {
    const u64 c = (u64)blockIdx.x * 256 + threadIdx.x;
    if (c >= nctx) return;
    const uint4 n = *reinterpret_cast<const uint4*>(cnt + c * 4);
    u64 f0 = 3 + (u64)step * n.x, f1 = 3 + (u64)step * n.y, f2 = 3 + (u64)step * n.z, f3 = 3 + (u64)step * n.w;
    while ((f0 | f1 | f2 | f3) > 255) {                     // any of them > 255 (all are < 2^35)
        f0 = (f0 >> 1) | (f0 & 1); f1 = (f1 >> 1) | (f1 & 1); f2 = (f2 >> 1) | (f2 & 1); f3 = (f3 >> 1) | (f3 & 1);
    }
    rows[c] = (u32)f0 | (u32)f1 << 8 | (u32)f2 << 16 | (u32)f3 << 24;
}
void launch_gen_rows(const u32* cnt, u32* rows, u64 nctx, u32 step, hipStream_t st) {
    hipLaunchKernelGGL(k_gen_rows, dim3((u32)((nctx + 255) / 256)), dim3(256), 0, st, cnt, rows, nctx, step);
}

// the rows a chain of block b codes with: null = the initial row
__device__ __forceinline__ const u32* gen_rows_of(const ChainArgs& a, u32 b) {
    if (!a.g_ngen) return nullptr;
    u32 g = 0;
    while (g + 1 < a.g_ngen && b >= a.g_bound[g + 1]) g++;            // generation g covers blocks [g_bound[g], g_bound[g + 1])
    return a.g_rows[g];
}

#define GEN_RING 8       // ring dwords per lane: 15 bytes may wait for their row of 16, four bases add at most 4 x 2
// FLAT: the call has no generation tables (api.cpp gen_tables_finish): only the initial row's path is compiled -- no row
// values in flight, no reciprocal table -- so the kernel holds fewer registers and less LDS beside the other chains' kernels
template <int THREADS, bool FLAT>
__global__ __launch_bounds__(THREADS) void k_gen_encode_c(ChainArgs a, u32 c0, u32 c1 /* the chains [c0, c1) */) {
    __shared__ u32 rcp[FLAT ? 1 : 1024];                      // reciprocals of the row totals (<= 1020)
    __shared__ u8 lut[256];                                   // character -> code (gen_code_of)
    __shared__ u32 ring[LaneEncB<THREADS, GEN_RING>::LDS_DWORDS];
    if constexpr (!FLAT) for (u32 i = threadIdx.x; i < 1024; i += THREADS) rcp[i] = i ? fz_recip(i) : 0u;
    for (u32 i = threadIdx.x; i < 256; i += THREADS) lut[i] = (u8)(gen_code_of(i) | (is_lower_base(i) ? 0x20u : 0u));    // 0x20: a lowercase base ("gen.lc")
    __syncthreads();
    const u32 c = c0 + blockIdx.x * THREADS + threadIdx.x;
    const bool live = c < c1;
    ChainPos cp; cp.b = 0; cp.r0 = 0; cp.nrec = 0; cp.sub_lo = cp.sub_len = 0; cp.seg = 0; cp.nseg = 1;
    if (live) { cp = chain_pos(a, c); chain_seg_encode(a, cp); }
    const BlockDesc* d = &a.m.blocks[cp.b];
    LaneEncB<THREADS, GEN_RING> rc; u32 cap = 0;
    u8* outp = live ? chain_region(a, cp, SFQ_S_GEN, 3, 4, cap) : nullptr;
    rc.init(ring, threadIdx.x, outp, cap);
    const u32* rows = (!FLAT && live) ? gen_rows_of(a, cp.b) : nullptr;
    u32 illegal = 0;
    if (a.flat_raw && (FLAT || !__any(rows != nullptr))) {
        // no model, no coder (round 5b, block format 10): a base is two bits whatever comes before it, and a range coder that is told so writes a byte per four
        // bases -- after a shift, a multiply, a renormalisation step and its carry test.  The chain's bases, four a byte (the first in the low bits), across
        // its records' ends; the last byte padded with zeros.  N-like bases code as 0, as with the coder (the exception lists restore them).
        rc.init_raw(ring, threadIdx.x, outp, cap);
        u32 acc = 0, nb = 0;
        illegal = walk_bases_q(a, cp.r0, cp.nrec, live ? d->solid : 0u, lut,
            [&](u32 sym, u32 k) {
                acc |= (sym & ((1u << (2u * k)) - 1u)) << nb;                 // (nb <= 6, 2 k <= 8: within 14 bits)
                nb += 2u * k;
                const u32 full = nb >= 8u ? ~0u : 0u;
                rc.put_if(full, acc & 0xffu);
                acc >>= 8u & full; nb -= 8u & full;
                rc.drain();
            },
            a.exc_flag, cp.sub_lo, cp.sub_len);
        if (live) {
            if (nb) rc.put_if(~0u, acc & 0xffu);
            a.csz[c] = rc.finish_raw();
            if (rc.err & 2) atomicMax(&a.m.blocks[cp.b].status, (u32)(-SFQ_E_OVERFLOW));
            if (illegal) atomicMax(&a.m.blocks[cp.b].status, (u32)(-SFQ_E_GENCHAR));
        }
        return;
    }
    if (a.flat_quads && (FLAT || !__any(rows != nullptr))) {
        // no model, four bases a symbol (round 5): a shift, a multiply and ONE renormalisation step per four bases -- a byte leaves per full
        // quad, exactly -- where the initial row's 3 of 12 took a divide and a step per base (0.94e9 of the default call's 6.6e9 wave instructions)
        illegal = walk_bases_q(a, cp.r0, cp.nrec, live ? d->solid : 0u, lut,
            [&](u32 sym, u32 k) { const u32 bits = 2u * k; rc.encode_bits_if(k ? ~0u : 0u, sym & ((1u << bits) - 1u), 1u, bits); rc.drain(); },
            a.exc_flag, cp.sub_lo, cp.sub_len);
    } else if (FLAT || !__any(rows != nullptr)) {
        // every lane of the wave codes with the initial row (3, 3, 3, 3): cum = 3 * code, freq 3 of 12, no lookups
        const u32 r12 = fz_recip(12u);
        illegal = walk_bases_b(a, cp.r0, cp.nrec, live ? d->solid : 0u, 0u, lut, [&](u32, u32) {},
            [&](u32, u32 code, u32 vm) { rc.encode_if(vm, 3u * code, 3u, 12u, r12); },
            [&]() { rc.drain(); }, a.exc_flag, cp.sub_lo, cp.sub_len);
    } else if constexpr (!FLAT) {
        // a lane whose generation has no rows yet reads the initial row from a one-entry table
        const u32* rp = rows ? rows : a.g_init;
        const u32 mask = (live && rows) ? (1u << d->gen_bits) - 1u : 0u;
        u32 rv[16];
        illegal = walk_bases_b(a, cp.r0, cp.nrec, live ? d->solid : 0u, mask, lut,
            [&](u32 j, u32 ctx) { rv[j] = rp[ctx]; },
            [&](u32 j, u32 code, u32 vm) {
                const u32 v = rv[j];
                const u32 f0 = v & 0xff, f1 = (v >> 8) & 0xff, f2 = (v >> 16) & 0xff, f3 = v >> 24;
                const u32 tot = (f0 + f1) + (f2 + f3);
                // cum = the sum of the bytes of v below byte `code`
                const u32 below = v & ((1u << (8 * code)) - 1u);
                const u32 cum = (below & 0xff) + ((below >> 8) & 0xff) + (below >> 16);
                rc.encode_if(vm, cum, (v >> (8 * code)) & 0xff, tot, rcp[tot]);          // base2_ranger.hpp:74-84 without the update
            },
            [&]() { rc.drain(); }, a.exc_flag, cp.sub_lo, cp.sub_len);
    }
    if (live) {
        a.csz[c] = rc.finish();
        if (rc.err & 2) atomicMax(&a.m.blocks[cp.b].status, (u32)(-SFQ_E_OVERFLOW));
        else if (rc.err) atomicMax(&a.m.blocks[cp.b].status, (u32)(-SFQ_E_CORRUPT));
        if (illegal) atomicMax(&a.m.blocks[cp.b].status, (u32)(-SFQ_E_GENCHAR));
    }
}
// the base chains [c0, c1) (c1 = 0: all of them); flat: every one of them codes with the initial row
void launch_gen_encode_c(const ChainArgs& a, hipStream_t st, u32 c0, u32 c1, bool flat) {
    constexpr int T = 256;
    if (!c1 || c1 > a.geo.nchains) c1 = a.geo.nchains;
    if (c1 <= c0) return;
    const dim3 grid((c1 - c0 + T - 1) / T);
    if (flat) hipLaunchKernelGGL((k_gen_encode_c<T, true>), grid, dim3(T), 0, st, a, c0, c1);
    else hipLaunchKernelGGL((k_gen_encode_c<T, false>), grid, dim3(T), 0, st, a, c0, c1);
}

// One base: the row's four frequencies -> the symbol under prob, its cum and freq (base2_ranger.hpp:86-104)
__device__ __forceinline__ u32 b2_pick(u32 v, LaneDecQ& rc, const u32* rcp) {
    const u32 f0 = v & 0xff, f1 = (v >> 8) & 0xff, f2 = (v >> 16) & 0xff, f3 = v >> 24;
    const u32 c1 = f0, c2 = f0 + f1, c3 = c2 + f2, tot = c3 + f3;
    u32 r;
    const u32 prob = rc.get_freq(tot, rcp[tot], r);
    const u32 b = (prob >= c1 ? 1u : 0u) + (prob >= c2 ? 1u : 0u) + (prob >= c3 ? 1u : 0u);
    const u32 cum = b == 0 ? 0u : b == 1 ? c1 : b == 2 ? c2 : c3;
    rc.decode(r, cum, (v >> (8 * b)) & 0xff);
    return b;
}
// decode the chains of blocks [b0, b1): GenLoad::load_x (gens.cpp:215-249) without the N rules (k_gen_exc_decode).
// With rows, a base's row is a fetch from a table far larger than the caches whose address the base before it decides:
// the walk of a chain is a chain of memory round trips.  The NEXT context's four candidate rows are one aligned 16-byte
// piece, fetched while this base is decoded (fetching the sixteen candidates two bases ahead was measured: four times the
// vector memory instructions cost more than the round trip they hide -- 158 ms against 95 per 10 M genome-sampled reads).
// (Also measured: the generation's counting pass folded into this kernel -- the atomics share the vector memory path
// with the row fetch every base waits for: 109 ms against 95.)
// Round 4: the quality decoder's lean coder (LaneDecQ); without rows a base is a divide by a twelfth of the range, two bits and
// a byte every four bases -- 94 -> ~60 instructions a base.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_gen_decode_c(ChainArgs a, DecodeArgs da, u32 c0, u32 c1 /* the chains [c0, c1) */) {
    __shared__ u32 rcp[1024];
    for (u32 i = threadIdx.x; i < 1024; i += THREADS) rcp[i] = i ? fz_recip(i) : 0u;
    __syncthreads();
    const u32 c = c0 + blockIdx.x * THREADS + threadIdx.x;
    if (c >= c1) return;
    ChainPos cp = chain_pos(a, c);
    const BlockDesc* d = &a.m.blocks[cp.b];
    LaneDecQ rc; rc.init(da.streams + a.coff[c], a.csz[c], reinterpret_cast<const u8*>(a.qesc));
    const u32* rows = gen_rows_of(a, cp.b);
    const u32 alphabet = d->solid ? 0x33323130u /* "0123" */ : 0x54474341u /* "ACGT" */;    // gens.cpp:173-178
    const u32 mask = (1u << d->gen_bits) - 1u;
    const u32 INIT = 0x007616c7u;                                                           // gens.cpp:139
    const u32 r12 = fz_recip(12u);
    u32 n_next = cp.nrec ? da.slen[cp.r0] : 0u; u64 off_next = cp.nrec ? da.soff[cp.r0] : 0ull;      // a record's length and place, a record ahead
    if (a.seg_len) {                                         // a segment of one record: its part of the base line
        const u32 ql = da.qlen[cp.r0];
        seg_range(cp, ql > n_next ? ql : n_next);
        const u64 lo = cp.sub_lo < n_next ? cp.sub_lo : n_next;
        n_next = (u32)(n_next - lo < cp.sub_len ? n_next - lo : cp.sub_len); off_next += lo;
    }
    // (block format 10, "chn.idx" flag bit 7: the chain is its bases, two bits each, four a byte -- no coder)
    const u8* const raw = da.streams + a.coff[c]; const u32 raw_n = a.csz[c];
    u32 racc = 0, rnb = 0, rat = 0, rbad = 0;
    for (u32 k = 0; k < cp.nrec; k++) {
        const u32 llen = n_next; const u64 off = off_next;
        if (k + 1 < cp.nrec) { n_next = da.slen[cp.r0 + k + 1]; off_next = da.soff[cp.r0 + k + 1]; }
        LaneOut32 out; out.begin(da.seq_stage + off);
        u32 last = INIT;
        if (a.flat_raw) {
            for (u32 i = 0; i < llen; i++) {
                if (rnb == 0) {                                                             // sixteen bases a load where four bytes are left
                    if (rat + 4u <= raw_n) { racc = *reinterpret_cast<const u32_any*>(raw + rat); rat += 4u; rnb = 32u; }
                    else if (rat < raw_n) { racc = raw[rat++]; rnb = 8u; }
                    else { rbad = 1; racc = 0; rnb = 32u; }
                }
                out.put((alphabet >> (8u * (racc & 3u))) & 0xffu);
                racc >>= 2; rnb -= 2u;
            }
        } else
        if (rows) {                                                                         // (gen_bits >= 2: the four candidates are in bounds)
            u32 v = rows[last & mask];
            for (u32 i = 0; i < llen; i++) {
                const uint4 cand = *reinterpret_cast<const uint4*>(rows + ((last << 2) & mask));
                rc.top_up();
                const u32 b = b2_pick(v, rc, rcp);
                out.put((alphabet >> (8 * b)) & 0xff);
                last = (last << 2) | b;
                v = b == 0 ? cand.x : b == 1 ? cand.y : b == 2 ? cand.z : cand.w;
            }
        } else if (a.flat_quads) {                                                          // no model, four bases a symbol ("chn.idx" flag bit 6)
            for (u32 i = 0; i < llen; i += 4u) {
                const u32 k = llen - i < 4u ? llen - i : 4u;
                rc.top_up();
                u32 r;
                const u32 S = rc.get_freq_bits(2u * k, r);
                rc.decode1(r, S, 1u);
                for (u32 j = 0; j < k; j++) out.put((alphabet >> (8u * ((S >> (2u * j)) & 3u))) & 0xffu);
            }
        } else {
            for (u32 i = 0; i < llen; i++) {                                                // the initial row (3, 3, 3, 3)
                rc.top_up();
                u32 r;
                const u32 prob = rc.get_freq(12u, r12, r);
                const u32 b = (prob >= 3u ? 1u : 0u) + (prob >= 6u ? 1u : 0u) + (prob >= 9u ? 1u : 0u);
                rc.decode1(r, 3u * b, 3u);
                out.put((alphabet >> (8 * b)) & 0xff);
            }
        }
        out.end();
    }
    if (a.flat_raw) { if (rbad || rat != raw_n || (rnb >= 8u) || racc) atomicMax(&a.m.blocks[cp.b].status, (u32)(-SFQ_E_CORRUPT)); }      // (every byte used, the padding zero)
    else
    if (rc.err) atomicMax(&a.m.blocks[cp.b].status, (u32)(-SFQ_E_CORRUPT));
}
void launch_gen_decode_c(const ChainArgs& a, const DecodeArgs& da, u32 c0, u32 c1, hipStream_t st) {
    constexpr int T = 256;
    if (c1 > a.geo.nchains) c1 = a.geo.nchains;
    if (c1 <= c0) return;
    hipLaunchKernelGGL(k_gen_decode_c<T>, dim3((c1 - c0 + T - 1) / T), dim3(T), 0, st, a, da, c0, c1);
}

// =========================================================================================================
// headers
//
// RecSave::save / RecLoad::load (dev_rec_lane.h: the reference's field-by-field model, lane-serial) with the symbols
// of the "rec" stream coded through FROZEN PowerRanger rows: a counting pass runs the same model over short runs of
// records spread over the call, the counts travel once as the header prior ("rec.pri"), and every header chain
// -- one per LANE -- codes with the rows built from it.  A row lists all 256 byte values with the ranger's weights
// (freq + 1 over total + 256, power_ranger.hpp:100; freq = 14 per hit) rescaled to a total of exactly 2^16, like the
// quality rows.  A block's headers are cut into chains of rgeo.chain_reads records; every chain starts from the
// block's first header (the base, stored once as "rec.first") with cold field types, so the chains of a block are
// independent.  A header whose shape changed is coded inside its chain (flag symbol + the whole line): no "rec.x".
// =========================================================================================================
#include "dev_rec_lane.h"
#include "dev_wave.h"

struct RecChainPos { u32 b; u64 r0; u32 nrec; };
__device__ __forceinline__ RecChainPos rec_chain_pos(const ChainArgs& a, u32 c) {
    RecChainPos p;
    p.b = c / a.rgeo.cpb;
    const u32 j = c - p.b * a.rgeo.cpb;
    const BlockDesc* d = &a.m.blocks[p.b];
    const u32 k0 = j * a.rgeo.chain_reads;
    p.nrec = k0 < d->nrec ? (d->nrec - k0 < a.rgeo.chain_reads ? d->nrec - k0 : a.rgeo.chain_reads) : 0u;
    p.r0 = d->rec0 + k0;
    return p;
}
// a header chain's output region inside its block's "rec" region (1.5 bytes per byte of text, frame.hip k_block_prepare)
__device__ __forceinline__ u8* rec_chain_region(const ChainArgs& a, const RecChainPos& p, u32& cap) {
    const BlockDesc* d = &a.m.blocks[p.b];
    const u64 t0 = a.m.line_off[4 * d->rec0], tc = a.m.line_off[4 * p.r0], te = a.m.line_off[4 * (p.r0 + p.nrec)];
    const u64 lo = ((tc - t0) * 3 / 2 + 3) & ~3ull, hi = ((te - t0) * 3 / 2) & ~3ull;             // as chain_region(.., 3, 2, ..): k_compact_chains reads through that
    cap = hi > lo ? (u32)(hi - lo) : 0u;
    return a.m.arena + d->out_off[SFQ_S_REC] + lo;
}

#define REC_HBUF 128u                    // bytes of LDS per staged header (two per lane: the current and the previous one)
#define REC_LDS_ROWS 8u                  // frozen rows staged in LDS per wave (1 KiB each): a handful of rows take nearly all symbols
struct RecFrozenEnc {
    static constexpr bool inband = true;
    const u32* rows; LaneEnc rc;
    u8* lbuf;                            // this lane's two header buffers in LDS
    const u16* lmap; const u32* lrows;   // LDS: row -> staged slot (0xFFFF = not staged), the staged rows
    __device__ __forceinline__ void record(u32) {}
    // headers are scanned byte by byte several times (tokens, field compare, number typing): a copy in LDS costs one
    // round of loads instead of a memory round trip per byte.  (n + 1 bytes: the separator behind the text is read too.)
    __device__ __forceinline__ const u8* stage(const u8* g, u32 n, u32 k) {
        if (n + 1 > REC_HBUF) return g;
        u8* dst = lbuf + (k & 1u) * REC_HBUF;
        const u32 nw = (n + 4) / 4;                                     // covers bytes 0..n; the text goes on behind the header line
#pragma unroll 8
        for (u32 i = 0; i < nw; i++) reinterpret_cast<u32*>(dst)[i] = reinterpret_cast<const u32_any*>(g)[i];     // (vector global loads need no alignment on gfx9)
        return dst;
    }
    __device__ __forceinline__ void put(u32 row, u32 sym) {
        const u32 slot = lmap[row];
        const u32 e = slot != 0xFFFFu ? lrows[slot * 256 + sym] : rows[(size_t)row * 256 + sym];
        rc.encode16(FZ_CUM(e), FZ_FREQ(e));
    }
    __device__ __forceinline__ void put_u(u32 row0, u64 num) { put_u_rows(*this, row0, num); }
};
struct RecCountEnc {
    static constexpr bool inband = true;
    u32* cnt; u32 kc;
    __device__ __forceinline__ void record(u32 k) { kc = k; }
    __device__ __forceinline__ const u8* stage(const u8* g, u32, u32) { return g; }
    __device__ __forceinline__ void put(u32 row, u32 sym) { atomicAdd(&cnt[(size_t)row * 256 + sym], kc >= 1u ? 1u : 0u); }    // (see RecFastCount::put)
    __device__ __forceinline__ void put_u(u32 row0, u64 num) { put_u_rows(*this, row0, num); }
};
// counting pass: lane i walks records [i * stride, i * stride + run): the first is the run's base
__global__ __launch_bounds__(64) void k_rec_count(ModelArgs a, u64 nrec, u64 stride, u32 run, u32 nruns, u32* cnt, const u32* flags /* the runs to take; null = all */) {
    const u32 i = blockIdx.x * 64 + threadIdx.x;
    if (i >= nruns) return;
    if (flags && !flags[i]) return;
    const u64 r0 = (u64)i * stride;
    if (r0 >= nrec) return;
    const u32 n = (u32)(nrec - r0 < run ? nrec - r0 : run);
    RecCountEnc cd; cd.cnt = cnt + (size_t)(blockIdx.x % REC_COUNT_COPIES) * PR_REC_ROWS * 256; cd.kc = 0;
    XfEnc x_rec; x_rec.init(nullptr, 0, XF_REC_X);
    PwTab none; none.slots = nullptr; none.hdr = nullptr; none.epoch = 0;
    u32 hb; int bad;
    rec_encode_lane(a, r0, r0, n, cd, x_rec, none, hb, bad);
}
// the counting pass counts into REC_COUNT_COPIES copies of the table (a workgroup takes copy blockIdx % copies): nearly every
// header puts its symbols on the same few dozen counters, and atomics on ONE address run one after the other in the L2
// (1.6 M of them on ~50 addresses were most of the pass's 2.5 ms).  cnt[0] += the other copies.
__global__ __launch_bounds__(256) void k_rec_count_sum(u32* __restrict__ cnt) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= PR_REC_ROWS * 256u) return;
    u32 s = 0;
#pragma unroll 8
    for (u32 k = 1; k < REC_COUNT_COPIES; k++) s += cnt[(size_t)k * PR_REC_ROWS * 256 + i];
    if (s) cnt[i] += s;
}
// frozen rows from the prior's scaled frequencies f[row][256]: x = f + 1, g = max(1, floor(x * 65536 / sum x)), the
// remainder to the largest g (the first of them); entry = cum | g << 16.  rdec (may be null; the decoder's form):
// [row][272] u16 -- the cum at every 16th symbol (16 of them), then the cum of all 256 (RDEC_ROW)
#define RDEC_ROW 272u
__global__ __launch_bounds__(256) void k_rec_frozen_rows(const u32* __restrict__ f, u32 nrows, u32* __restrict__ rrows, u16* __restrict__ rdec) {
    const u32 lane = threadIdx.x & 63;
    const u32 row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const uint4 fv = *reinterpret_cast<const uint4*>(f + (size_t)row * 256 + lane * 4);
    const u32 x[4] = { fv.x + 1, fv.y + 1, fv.z + 1, fv.w + 1 };
    u32 S = x[0] + x[1] + x[2] + x[3];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) S += (u32)__shfl_xor((int)S, d, 64);
    u32 g[4]; u32 sum = 0, best = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        g[j] = (u32)(((u64)x[j] << 16) / S); g[j] = g[j] ? g[j] : 1u;
        sum += g[j];
        const u32 key = (g[j] << 8) | (255u - (lane * 4 + j));
        best = key > best ? key : best;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        sum += (u32)__shfl_xor((int)sum, d, 64);
        const u32 o = (u32)__shfl_xor((int)best, d, 64); best = o > best ? o : best;
    }
    const u32 bsym = 255u - (best & 255u);
#pragma unroll
    for (int j = 0; j < 4; j++) if (lane * 4 + j == bsym) g[j] += 65536u - sum;
    const u32 mine = g[0] + g[1] + g[2] + g[3];
    u32 incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 o = (u32)__shfl_up((int)incl, d, 64); if (lane >= (u32)d) incl += o; }
    u32 cum = incl - mine;
    if (rdec) {
        u16* rd = rdec + (size_t)row * RDEC_ROW;
        if ((lane & 3) == 0) rd[lane >> 2] = (u16)cum;
        const u32 c1 = cum + g[0], c2 = c1 + g[1], c3 = c2 + g[2];
        *reinterpret_cast<uint2*>(rd + 16 + lane * 4) = make_uint2(cum | (c1 << 16), c2 | (c3 << 16));
    }
    uint4 e;
    e.x = cum | (g[0] << 16); cum += g[0];
    e.y = cum | (g[1] << 16); cum += g[1];
    e.z = cum | (g[2] << 16); cum += g[2];
    e.w = cum | (g[3] << 16);
    *reinterpret_cast<uint4*>(rrows + (size_t)row * 256 + lane * 4) = e;
}
// The transmitted header prior from the sample's counts, on the device (round 4: the host did this between two waits, 0.3 ms of
// every call with the header chains behind it): a row's counts x 14, shifted down until the largest is at most 32000 --
// api.cpp pack_rec_prior_f writes "rec.pri" from the result, unpack_rec_prior reads it back.  rtot[row] = the row's sum.
__global__ __launch_bounds__(256) void k_rec_prior_freqs(const u32* __restrict__ cnt, u32 nrows, u32* __restrict__ f, u32* __restrict__ rtot) {
    const u32 lane = threadIdx.x & 63;
    const u32 row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const uint4 cv = *reinterpret_cast<const uint4*>(cnt + (size_t)row * 256 + lane * 4);
    u32 mx = max(max(cv.x, cv.y), max(cv.z, cv.w));
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { const u32 o = (u32)__shfl_xor((int)mx, d, 64); mx = o > mx ? o : mx; }
    u32 sh = 0;
    while ((((u64)mx * 14u) >> sh) > 32000u) sh++;
    uint4 fv;
    fv.x = (u32)(((u64)cv.x * 14u) >> sh); fv.y = (u32)(((u64)cv.y * 14u) >> sh); fv.z = (u32)(((u64)cv.z * 14u) >> sh); fv.w = (u32)(((u64)cv.w * 14u) >> sh);
    *reinterpret_cast<uint4*>(f + (size_t)row * 256 + lane * 4) = fv;
    u32 t = fv.x + fv.y + fv.z + fv.w;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) t += (u32)__shfl_xor((int)t, d, 64);
    if (lane == 0) rtot[row] = t;
}
// the rows worth staging in LDS: the nh the prior gives the most weight, the lower row first among equals.  map[row] = its
// place among them or 0xFFFF, hot[place] = row.  (One workgroup; a row's place = how many rows come before it.)
#define RHR_PER ((PR_REC_ROWS + 255u) / 256u)         /* rows per thread of k_rec_hot_rows */
__global__ __launch_bounds__(256) void k_rec_hot_rows(const u32* __restrict__ rtot, u32 nrows, u32 nh, u16* __restrict__ map, u16* __restrict__ hot) {
    // nh rounds of "the heaviest row not taken yet" (key = sum << 11 | 2047 - row: the lower row first among equals): a thread holds
    // a few rows' keys, a round is a wave maximum and four words of LDS.  (A row's place counted against all 1056 others, a
    // thousand LDS reads per thread: 0.07 ms alone, 1.2 ms on a CU the chains keep busy; and ONE workgroup of 1024 threads waits
    // that long for a CU with sixteen free wave slots once the chains run -- 256 threads find room at once.)
    __shared__ u64 wmax[4];
    const u32 t = threadIdx.x, lane = t & 63, wave = t >> 6;
    u64 k[RHR_PER];
#pragma unroll
    for (u32 j = 0; j < RHR_PER; j++) {
        const u32 row = t + 256u * j;
        k[j] = row < nrows ? ((u64)rtot[row] << 11) | (2047u - row) : 0ull;
        if (row < nrows) map[row] = 0xFFFFu;
    }
    for (u32 i = 0; i < nh; i++) {
        u64 m = 0;
#pragma unroll
        for (u32 j = 0; j < RHR_PER; j++) m = k[j] > m ? k[j] : m;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const u64 o = ((u64)(u32)__shfl_xor((int)(u32)(m >> 32), d, 64) << 32) | (u32)__shfl_xor((int)(u32)m, d, 64);
            m = o > m ? o : m;
        }
        __syncthreads();
        if (lane == 0) wmax[wave] = m;
        __syncthreads();
        u64 best = wmax[0];
        for (u32 w = 1; w < 4; w++) best = wmax[w] > best ? wmax[w] : best;
        if (best == 0) break;                                   // (fewer rows than nh)
        const u32 row = 2047u - (u32)(best & 2047u);
#pragma unroll
        for (u32 j = 0; j < RHR_PER; j++) if (k[j] == best) { k[j] = 0; hot[i] = (u16)row; map[row] = (u16)i; }
    }
}
void launch_rec_prior_freqs(const u32* cnt, u32 nrows, u32* f, u32* rtot, u32 nh, u16* map, u16* hot, hipStream_t st) {
    hipLaunchKernelGGL(k_rec_prior_freqs, dim3((nrows + 3) / 4), dim3(256), 0, st, cnt, nrows, f, rtot);
    hipLaunchKernelGGL(k_rec_hot_rows, dim3(1), dim3(256), 0, st, (const u32*)rtot, nrows, nh, map, hot);
}
void launch_rec_frozen_rows(const u32* f, u32 nrows, u32* rrows, u16* rdec, hipStream_t st) {
    hipLaunchKernelGGL(k_rec_frozen_rows, dim3((nrows + 3) / 4), dim3(256), 0, st, f, nrows, rrows, rdec);
}

// header encode, general path: one chain per lane; only the chains the fast kernel below has handed over (flags[c] != 0)
__global__ __launch_bounds__(64) void k_rec_encode_c(ChainArgs a, const u32* flags) {
    __shared__ u32 lrows[REC_LDS_ROWS * 256];
    __shared__ u16 lmap[PR_REC_ROWS];
    __shared__ u32 ltext[64 * 2 * REC_HBUF / 4];
    const u32 c = blockIdx.x * 64 + threadIdx.x;
    const bool mine = c < a.rgeo.nchains && (!flags || flags[c] != 0);             // (flags null: every chain)
    if (!__any(mine)) return;
    for (u32 i = threadIdx.x; i < PR_REC_ROWS; i += 64) { const u32 sl = a.rmap[i]; lmap[i] = (u16)(sl < REC_LDS_ROWS ? sl : 0xFFFFu); }
    for (u32 i = threadIdx.x; i < a.r_hot * 256; i += 64) lrows[i] = a.rrows[(size_t)a.rhot[i >> 8] * 256 + (i & 255)];
    __syncthreads();
    if (!mine) return;
    const RecChainPos cp = rec_chain_pos(a, c);
    BlockDesc* d = &a.m.blocks[cp.b];
    u32 cap = 0;
    u8* outp = rec_chain_region(a, cp, cap);
    RecFrozenEnc cd; cd.rows = a.rrows; cd.rc.init(outp, cap);
    cd.lbuf = reinterpret_cast<u8*>(ltext) + threadIdx.x * 2 * REC_HBUF; cd.lmap = lmap; cd.lrows = lrows;
    XfEnc x_rec; x_rec.init(nullptr, 0, XF_REC_X);
    PwTab none; none.slots = nullptr; none.hdr = nullptr; none.epoch = 0;
    u32 hdr_bytes = 0; int bad = 0;
    rec_encode_lane(a.m, d->rec0, cp.r0, cp.nrec, cd, x_rec, none, hdr_bytes, bad);
    a.rhb[c] = hdr_bytes;
    a.csz[c] = cd.rc.finish();
    if (cd.rc.err & 2) atomicMax(&d->status, (u32)(-SFQ_E_OVERFLOW));
    if (cd.rc.err & 1) atomicMax(&d->status, (u32)(-SFQ_E_CORRUPT));
    if (bad) atomicMax(&d->status, (u32)(-bad));
}

// header encode, fast path: the same model (recs.cpp:277-372) for chains whose headers are at most RF_MAXLEN bytes and
// RF_NF fields, with everything a lane touches per record in LDS, laid out [..][lane]: the header text (current and
// previous), the field tables of both, the field types / values, and the hottest frozen rows.  The general path above
// keeps those in per-lane scratch and reads the text through generic pointers: ~300 memory instructions per record,
// which is what its time is made of.  A chain with a longer header or more fields is handed over (flags[c] = 1).
// LDS is what limits how many of these waves a CU holds (and, beside them, the quality and base chains): the image is
// sized by the call's longest header -- ML = 62: 31.9 KiB, five waves per CU; 94: 35.9 KiB; 127: 40.1 KiB, three.
#define RF_NF 16u
template <u32 ML>
struct RecFastLds {
    static constexpr u32 maxlen = ML;
    u64 fvalue[RF_NF][64];
    u32 rows[REC_LDS_ROWS * 256];
    u8  text[2][ML + 1][64];
    u8  off[2][RF_NF][64], wln[2][RF_NF][64], sep[2][RF_NF][64];
    u8  fkind[RF_NF][64];
    u8  map[PR_REC_ROWS];                // row -> LDS slot, 0xFF = not staged
};
// a field's type and value (dev_rec.h field_type) over text[buf][off ..][lane]
template <typename LT>
__device__ __forceinline__ u32 nw_lds(const LT& L, u32 buf, u32 lane, u32 off, int len, u64& num, u32 pctype) {
    return field_type([&](u32 j) -> u32 { return L.text[buf][off + j][lane]; }, (u32)len, num, pctype);
}
template <typename LT>
struct RecFastEnc {
    const u32* rows; const LT* L; LaneEnc rc;
    __device__ __forceinline__ void put(u32 row, u32 sym) {
        const u32 slot = L->map[row];
        const u32 e = slot != 0xFFu ? L->rows[slot * 256 + sym] : rows[(size_t)row * 256 + sym];
        rc.encode16(FZ_CUM(e), FZ_FREQ(e));
    }
    __device__ __forceinline__ void put_u(u32 row0, u64 num) { put_u_rows(*this, row0, num); }
    __device__ __forceinline__ void record(u32) {}
};
// a header line into LDS, tokenised on the way (map_space, recs.cpp:141-157): a separator closes a field; the byte behind
// the text is the line's '\n', the last separator; a NUL ends the scan.  Eight dwords are fetched at a time, so a header
// costs a memory round trip or two, not one per dword.  Returns the number of fields (more than RF_NF: not all recorded;
// 0xFFFF: a NUL inside).
template <typename LT>
__device__ __forceinline__ u32 rf_stage(LT& L, u32 buf, u32 lane, const u8* text, u32 n) {
    const u32_any* g = reinterpret_cast<const u32_any*>(text);     // (vector global loads need no alignment on gfx9; the text goes on behind the line)
    const u32 nw = (n + 4) / 4;
    u32 nf = 0, start = 0; bool stop = false;
    for (u32 i0 = 0; i0 < nw; i0 += 8) {
        u32 wv[8];
#pragma unroll
        for (u32 q = 0; q < 8; q++) wv[q] = i0 + q < nw ? g[i0 + q] : 0u;
#pragma unroll
        for (u32 q = 0; q < 8; q++) {
#pragma unroll
            for (u32 j = 0; j < 4; j++) {
                const u32 pos = 4 * (i0 + q) + j, c = (wv[q] >> (8 * j)) & 0xffu;
                if (pos <= n) {
                    L.text[buf][pos][lane] = (u8)c;
                    if (!stop && !isword(c)) {
                        if (nf < RF_NF) { L.off[buf][nf][lane] = (u8)start; L.wln[buf][nf][lane] = (u8)(pos - start); L.sep[buf][nf][lane] = (u8)c; }
                        nf++; start = pos + 1;
                        if (c == 0) stop = true;
                    }
                }
            }
        }
    }
    return stop ? 0xFFFFu : nf;             // a NUL inside the header: not for the fast path (the general one sends it whole)
}
// one chain (or one run of the counting pass) on the fast path; returns false where a header is too long / has too many
// fields for it (the coder has then seen part of the chain: an encoder starts over on the general path, the counting
// pass looks at its run beforehand)
template <typename LT, typename CD>
__device__ __forceinline__ bool rec_fast_lane(const ModelArgs& m, LT& L, u32 lane, u64 base_rec, u64 r0, u32 nrec, CD& cd, u32& hdr_bytes_out) {
    u32 cur = 0, nf_prev = 0, hdr_bytes = 0, coded = 0;
    {                                                                         // recs.cpp:279-287: the base
        const u64 h0 = m.line_off[4 * base_rec] + 1, h1 = m.line_off[4 * base_rec + 1] - 1;
        const u32 n = h1 > h0 ? (u32)(h1 - h0) : 0;
        if (n > LT::maxlen) return false;
        nf_prev = rf_stage(L, cur, lane, m.fq + h0, n);
        if (nf_prev > RF_NF) return false;
        for (u32 f = 0; f < RF_NF; f++) L.fkind[f][lane] = 0;
        cur ^= 1u;
    }
    u64 nh0 = 0, nh1 = 0;                                                     // the next record's header line, fetched a record ahead
    if (nrec) { nh0 = m.line_off[4 * r0] + 1; nh1 = m.line_off[4 * r0 + 1] - 1; }
    for (u32 k = 0; k < nrec; k++) {
        const u64 r = r0 + k;
        const u64 h0 = nh0, h1 = nh1;
        if (k + 1 < nrec) { nh0 = m.line_off[4 * (r + 1)] + 1; nh1 = m.line_off[4 * (r + 1) + 1] - 1; }
        const u32 n = h1 > h0 ? (u32)(h1 - h0) : 0;
        hdr_bytes += n;
        if (r == base_rec) continue;                                          // the base itself
        cd.record(coded++);
        if (n > LT::maxlen) return false;
        const u32 nf = rf_stage(L, cur, lane, m.fq + h0, n);
        if (nf > RF_NF) return false;
        const u32 prv = cur ^ 1u;
        bool shape = nf != nf_prev;
        for (u32 f = 0; !shape && f < nf; f++) shape = L.sep[cur][f][lane] != L.sep[prv][f][lane];
        cd.put(REC_FLAG_ROW, shape ? 1u : 0u);
        if (shape) {                                                          // recs.cpp:292-305, in the chain itself
            cd.put_u(REC_FLAG_ROW + 2, n);
            for (u32 j = 0; j < n; j++) cd.put(REC_FLAG_ROW + 1, L.text[cur][j][lane]);
            for (u32 f = 0; f < RF_NF; f++) L.fkind[f][lane] = 0;
            nf_prev = nf; cur = prv;
            continue;
        }
        u64 map = 0;
        for (u32 f = 0; f < nf; f++) {
            const u32 wl = L.wln[cur][f][lane];
            bool ch = wl != L.wln[prv][f][lane];
            if (!ch) {
                const u32 o = L.off[cur][f][lane], po = L.off[prv][f][lane];
                for (u32 j = 0; j < wl; j++) if (L.text[cur][o + j][lane] != L.text[prv][po + j][lane]) { ch = true; break; }
            }
            if (ch) map |= 1ull << f;
        }
        cd.put_u(0 * 16 + 2, map);                                            // put_num(0, map) recs.cpp:313
        for (u32 f = 0; f < nf; f++) {
            if (!((map >> f) & 1)) continue;
            const u32 o = L.off[cur][f][lane], wl = L.wln[cur][f][lane], pct = L.fkind[f][lane];
            u64 fnum;
            u32 type = nw_lds(L, cur, lane, o, (int)wl, fnum, pct);
            if (type != ST_STR && !rec_number_prints_back(type, wl, L.text[cur][o][lane])) type = ST_STR;      // (chains are block format only: lossless)
            const u32 rr = (f + 1) * 16;
            if (type == ST_STR) {                                             // recs.cpp:324-331
                cd.put(rr + 0, type);
                cd.put_u(rr + 2, wl);
                for (u32 j = 0; j < wl; j++) cd.put(rr + 1, L.text[cur][o + j][lane]);
                L.fkind[f][lane] = 0;
                continue;
            }
            const u64 was = pct ? L.fvalue[f][lane] : 0;                       // recs.cpp:333-348
            u64 gap;
            L.fkind[f][lane] = (type < ST_STR || type >= ST_DGT_Z) ? 1 : 2;
            L.fvalue[f][lane] = fnum;
            if (fnum < was) { gap = was - fnum; type++; }
            else gap = fnum - was;
            cd.put(rr + 0, type);
            cd.put_u(rr + 2, gap);
        }
        nf_prev = nf; cur = prv;
    }
    hdr_bytes_out = hdr_bytes;
    return true;
}
template <u32 ML>
__global__ __launch_bounds__(64) void k_rec_encode_f(ChainArgs a, u32* flags, const u32* only /* the chains to take; null = all */) {
    __builtin_amdgcn_s_setprio(3);
    __shared__ RecFastLds<ML> L;
    const u32 lane = threadIdx.x;
    {   // (a workgroup none of whose chains is wanted leaves before it stages anything)
        const u32 c0 = blockIdx.x * 64 + lane;
        if (only && !__any(c0 < a.rgeo.nchains && only[c0] == 1)) return;      // (2: marked by the host for the general kernel, its flag set there)
    }
    for (u32 i = lane; i < PR_REC_ROWS; i += 64) { const u32 sl = a.rmap[i]; L.map[i] = (u8)(sl < REC_LDS_ROWS ? sl : 0xFFu); }
    for (u32 i = lane; i < a.r_hot * 256; i += 64) L.rows[i] = a.rrows[(size_t)a.rhot[i >> 8] * 256 + (i & 255)];
    __syncthreads();
    const u32 c = blockIdx.x * 64 + lane;
    if (c >= a.rgeo.nchains) return;
    if (only && !only[c]) return;
    const RecChainPos cp = rec_chain_pos(a, c);
    BlockDesc* d = &a.m.blocks[cp.b];
    u32 cap = 0;
    u8* outp = rec_chain_region(a, cp, cap);
    RecFastEnc<RecFastLds<ML>> cd; cd.rows = a.rrows; cd.L = &L; cd.rc.init(outp, cap);
    u32 hdr_bytes = 0;
    if (!rec_fast_lane(a.m, L, lane, d->rec0, cp.r0, cp.nrec, cd, hdr_bytes)) { flags[c] = 1; return; }     // the general kernel starts this chain over
    a.rhb[c] = hdr_bytes;
    a.csz[c] = cd.rc.finish();
    if (cd.rc.err & 2) atomicMax(&d->status, (u32)(-SFQ_E_OVERFLOW));
    if (cd.rc.err & 1) atomicMax(&d->status, (u32)(-SFQ_E_CORRUPT));
}
// the counting pass on the fast path: a run whose headers all fit (looked at first: counts cannot be taken back) is
// counted here, the others are marked for k_rec_count
struct RecFastCount {
    u32* cnt; u32 kc;                                                     // kc: which coded record of the run this is
    __device__ __forceinline__ void record(u32 k) { kc = k; }
    // the run's first coded record only warms the field types up: it adds 0.  (Not `if (counted) atomicAdd(.., 1)`: the
    // compiler branched on that as a wave-uniform condition whose mask held zeros for the lanes that were inactive where it
    // was computed -- a lane coding a field alone counted its warm-up record.  tests: ..._outside_the_fast_kernels_envelope)
    __device__ __forceinline__ void put(u32 row, u32 sym) { atomicAdd(&cnt[(size_t)row * 256 + sym], kc >= 1u ? 1u : 0u); }
    __device__ __forceinline__ void put_u(u32 row0, u64 num) { put_u_rows(*this, row0, num); }
};
__global__ __launch_bounds__(64) void k_rec_count_f(ModelArgs a, u64 nrec, u64 stride, u32 run, u32 nruns, u32* cnt, u32* flags) {
    __shared__ RecFastLds<127> L;
    const u32 lane = threadIdx.x;
    const u32 i = blockIdx.x * 64 + lane;
    if (i >= nruns) return;
    const u64 r0 = (u64)i * stride;
    if (r0 >= nrec) return;
    const u32 n = (u32)(nrec - r0 < run ? nrec - r0 : run);
    bool fits = true;
    for (u32 k = 0; k < n && fits; k++) {
        const u64 h0 = a.line_off[4 * (r0 + k)] + 1, h1 = a.line_off[4 * (r0 + k) + 1] - 1;
        const u32 len = h1 > h0 ? (u32)(h1 - h0) : 0;
        fits = len <= 127u && rf_stage(L, 0, lane, a.fq + h0, len) <= RF_NF;
    }
    if (!fits) { flags[i] = 1; return; }
    RecFastCount cd; cd.cnt = cnt + (size_t)(blockIdx.x % REC_COUNT_COPIES) * PR_REC_ROWS * 256; cd.kc = 0;
    u32 hb = 0;
    rec_fast_lane(a, L, lane, r0, r0, n, cd, hb);
}
// flags: one dword per run, zeroed by the caller
void launch_rec_count(const ModelArgs& a, u64 nrec, u64 stride, u32 run, u32 nruns, u32* cnt, u32* flags, hipStream_t st) {
    hipLaunchKernelGGL(k_rec_count_f, dim3((nruns + 63) / 64), dim3(64), 0, st, a, nrec, stride, run, nruns, cnt, flags);
    hipLaunchKernelGGL(k_rec_count, dim3((nruns + 63) / 64), dim3(64), 0, st, a, nrec, stride, run, nruns, cnt, (const u32*)flags);
    hipLaunchKernelGGL(k_rec_count_sum, dim3((PR_REC_ROWS * 256u + 255) / 256), dim3(256), 0, st, cnt);
}
// ---- header chains in two steps: tokens, then the coder ---------------------------------------------------------------------
// k_rec_encode_f above runs the whole header model a chain per LANE: every lane keeps two headers, two field tables and the field
// values in LDS (32-41 KiB per wave, one wave per SIMD) and walks them byte by byte -- 5400 wave instructions per record step,
// nothing to hide an LDS round trip behind, and 147 KiB of every CU's LDS gone while it runs.  But what a record CODES depends on
// its own text, the text of the record before it and very little history:
//   * the shape flag, the change map and the changed fields' texts: the two headers alone;
//   * a changed field's previous value (recs.cpp:333): the number in the previous header's field -- if that field has changed at
//     all since the chain began or the shape last changed (else 0: field types start cold), and if its text is a number;
//   * hexadecimal fields are sticky (numberwang's pctype == 2, recs.cpp:208): a chain in which a changed field types as hex is
//     left to the kernels above.
// So step 1 takes a RECORD per lane, a chain per wave: 64 consecutive headers staged and tokenised at once, each compared with its
// left neighbour's column, "has changed since" carried as an OR-scan over the lanes, the symbols (row, byte) written to a token
// buffer; step 2 is a coder per lane over the tokens, as cheap as the quality chains' coder.  The bytes are those of k_rec_encode_f.
#define RT_TOK_PER_REC 40u               /* token room per record, pooled over a chain; a chain that needs more goes to k_rec_encode_f */
#define RT_STAGED 24u                    /* symbols of a record that wait in LDS for their place (a record with more is walked a second time) */
template <u32 ML>
struct RecTokLds {                       // column l + 1 = lane l's record, column 0 = the record before lane 0's
    static constexpr u32 maxlen = ML;
    static constexpr u32 NW = (ML + 4) / 4;               // dwords that hold bytes 0 .. ML of a header (the '\n' behind the text is one of them)
    static constexpr u32 NWP = NW | 1u;                   // a column's stride in dwords: odd, so the 64 columns' dword k lie in 64 banks
    u32 tw[66][NWP];                                      // the text, a header per column, staged a dword at a time
    u8 off[1][RF_NF][66], wln[1][RF_NF][66], sep[1][RF_NF][66];
    u8 nf[66];
    u32 tokst[RT_STAGED][64];                             // a record's symbols until the wave knows where they go
    __device__ __forceinline__ u32 byte(u32 col, u32 pos) const { return reinterpret_cast<const u8*>(tw[col])[pos]; }
};
// Where a header's fields end: bit p = byte p is neither a letter nor a digit (map_space, recs.cpp:141-157), p = 0 .. n (bit n: the line's '\n')
struct HdrMask {
    u64 lo, hi;
    __device__ __forceinline__ u32 count() const { return (u32)__popcll(lo) + (u32)__popcll(hi); }
    __device__ __forceinline__ u32 take() {               // the lowest set bit, cleared
        if (lo) { const u32 p = (u32)__ffsll((long long)lo) - 1u; lo &= lo - 1; return p; }
        const u32 p = 63u + (u32)__ffsll((long long)hi); hi &= hi - 1; return p;
    }
};
// A header into its column, and its field ends as a mask.  Four bytes a step: a byte is a field character if it lies in '0'-'9' or, with bit 5 set, in
// 'a'-'z' -- two range tests on all four bytes at once (x + (0x80 - lo) carries into bit 7 where x >= lo; x + (0x7f - hi) does not where x <= hi; the
// bytes are below 0x80 there, so nothing carries across) --, and a multiplication gathers the four flags into a nibble of the mask.  Round 4 walked
// the bytes one at a time, each with its own test, its own LDS store and -- the lanes' fields ending at different bytes -- its own divergent
// "a field ends here" branch.  Returns false where a NUL lies inside (map_space stops there: not for this kernel).
template <typename LT>
__device__ __forceinline__ bool rt_stage(LT& L, u32 col, const u8* text, u32 n, HdrMask& M) {
    const u32_any* g = reinterpret_cast<const u32_any*>(text);     // (the text goes on behind the line)
    const u32 kn = n >> 2, nw = kn + 1;
    const u32 fill = (n & 3u) == 3u ? 0u : ~0u << (8u * ((n & 3u) + 1u));      // the last dword's bytes behind the '\n': made 0xFF (no NUL, no field character)
    u32 zero = 0, mw[4] = {};
#pragma unroll
    for (u32 k0 = 0; k0 < LT::NW; k0 += 4) {
        if (!__any(k0 < nw)) break;
        u32 wv[4];
#pragma unroll
        for (u32 q = 0; q < 4; q++) wv[q] = k0 + q < nw && k0 + q < LT::NW ? g[k0 + q] : ~0u;
#pragma unroll
        for (u32 q = 0; q < 4; q++) {
            const u32 k = k0 + q;
            if (k >= LT::NW) break;
            L.tw[col][k] = wv[q];
            const u32 w = wv[q] | (k == kn ? fill : 0u);
            const u32 x = w & 0x7f7f7f7fu, y = x | 0x20202020u;
            const u32 dig = (x + 0x50505050u) & ~(x + 0x46464646u);
            const u32 let = (y + 0x1f1f1f1fu) & ~(y + 0x05050505u);
            const u32 t = ~((dig | let) & ~w) & 0x80808080u;                     // bit 7 of each byte: not a field character
            mw[k / 8] |= (((t >> 7) * 0x01020408u) >> 24) << (4u * (k % 8));
            zero |= ~(((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u;      // bit 7 of each byte: the byte is 0
        }
    }
    M.lo = mw[0] | ((u64)mw[1] << 32);
    M.hi = mw[2] | ((u64)mw[3] << 32);
    // only bytes 0 .. n count
    if (n < 63) { M.lo &= (2ull << n) - 1; M.hi = 0; }
    else if (n == 63) M.hi = 0;
    else M.hi &= (2ull << (n - 64)) - 1;
    return zero == 0;
}
// the field tables of the header in column col from its mask: field f = bytes off .. off + wln, closed by the byte sep
template <typename LT>
__device__ __forceinline__ void rt_fields(LT& L, u32 col, HdrMask m, u32 nf) {
    u32 start = 0;
    for (u32 f = 0; f < nf; f++) {
        const u32 p = m.take();
        L.off[0][f][col] = (u8)start; L.wln[0][f][col] = (u8)(p - start); L.sep[0][f][col] = (u8)L.byte(col, p);
        start = p + 1;
    }
}
struct TokCount { u32 n; __device__ __forceinline__ void put(u32, u32) { n++; } __device__ __forceinline__ void put_u(u32 row0, u64 num) { put_u_rows(*this, row0, num); } };
struct TokStage {                        // into the lane's column of RecTokLds::tokst, counting on beyond it
    u32* p; u32 n;
    __device__ __forceinline__ void put(u32 row, u32 sym) { if (n < RT_STAGED) p[n * 64u] = (row << 8) | sym; n++; }
    __device__ __forceinline__ void put_u(u32 row0, u64 num) { put_u_rows(*this, row0, num); }
};
struct TokStore { u32* p; __device__ __forceinline__ void put(u32 row, u32 sym) { *p++ = (row << 8) | sym; } __device__ __forceinline__ void put_u(u32 row0, u64 num) { put_u_rows(*this, row0, num); } };
// the symbols of the record in column col (n bytes, nf fields), against the record in column col - 1; since: the fields that have
// changed since the chain began / the shape last changed.  false: a field types as hexadecimal.
// a field's type and value (dev_rec.h field_type) over the header in column col.  The usual field is a few decimal digits: one multiply-add a
// digit, in 32 bits; field_type's general pass -- both readings carried in 64 bits, the classes, the wrap test: 35 instructions a byte, and with
// two fields parsed per changed field and two passes four fifths of this kernel's instructions -- is for the lanes whose field is anything else
template <typename LT>
__device__ __forceinline__ u32 nw_tok(const LT& L, u32 col, u32 off, u32 len, u64& num) {
    u32 v = 0, lead = 0;
    bool plain = len <= 9u;
    if (plain)
        for (u32 j = 0; j < len; j++) {
            const u32 d = L.byte(col, off + j) - '0';
            plain = plain && d <= 9u;
            lead |= (j < 2u && d == 0u) ? 1u << j : 0u;
            v = v * 10u + d;
        }
    if (plain) {
        num = (lead & 3u) == 3u ? 0u : v;                                       // "00...": field_type's first test
        return (lead & 3u) == 3u ? ST_STR : (lead & 1u) ? ST_DGT_Z : ST_DGT;
    }
    return field_type([&](u32 j) -> u32 { return L.byte(col, off + j); }, len, num, 0);
}
template <typename LT, typename EM>
__device__ __forceinline__ bool rec_tokens(const LT& L, u32 col, u32 n, u32 nf, bool shape, u64 map, u64 since, EM& em) {
    em.put(REC_FLAG_ROW, shape ? 1u : 0u);
    if (shape) {                                                              // recs.cpp:292-305, in the chain itself
        em.put_u(REC_FLAG_ROW + 2, n);
        for (u32 j = 0; j < n; j++) em.put(REC_FLAG_ROW + 1, L.byte(col, j));
        return true;
    }
    em.put_u(0 * 16 + 2, map);                                                // put_num(0, map) recs.cpp:313
    for (u64 todo = map; todo; todo &= todo - 1) {                            // the changed fields, in order
        const u32 f = (u32)__ffsll((long long)todo) - 1u;
        const u32 o = L.off[0][f][col], wl = L.wln[0][f][col];
        u64 fnum;
        u32 type = nw_tok(L, col, o, wl, fnum);
        if (type != ST_STR && !rec_number_prints_back(type, wl, L.byte(col, o))) type = ST_STR;
        if (type >= ST_HGT && type <= ST_HLTC_Z) return false;
        const u32 rr = (f + 1) * 16;
        if (type == ST_STR) {                                                 // recs.cpp:324-331
            em.put(rr + 0, type);
            em.put_u(rr + 2, wl);
            for (u32 j = 0; j < wl; j++) em.put(rr + 1, L.byte(col, o + j));
            continue;
        }
        u64 was = 0;                                                         // recs.cpp:333: the previous VALUE, if the field has one
        if ((since >> f) & 1) {
            const u32 po = L.off[0][f][col - 1], pwl = L.wln[0][f][col - 1];
            u64 pn;
            u32 pt = nw_tok(L, col - 1, po, pwl, pn);
            if (pt != ST_STR && !rec_number_prints_back(pt, pwl, L.byte(col - 1, po))) pt = ST_STR;
            if (pt == ST_DGT || pt == ST_DGT_Z) was = pn;
        }
        u64 gap;
        if (fnum < was) { gap = was - fnum; type++; }
        else gap = fnum - was;
        em.put(rr + 0, type);
        em.put_u(rr + 2, gap);
    }
    return true;
}
template <u32 ML>
__global__ __launch_bounds__(64) void k_rec_tokens(ChainArgs a, u32* __restrict__ tok, u32* __restrict__ ntok, u32* __restrict__ flags) {
#ifdef PRIO_REC
    __builtin_amdgcn_s_setprio(PRIO_REC);
#endif
    __shared__ RecTokLds<ML> L;
    const u32 lane = threadIdx.x, col = lane + 1;
    const u32 c = blockIdx.x;
    const RecChainPos cp = rec_chain_pos(a, c);
    if (cp.nrec == 0) { if (lane == 0) { ntok[c] = 0; a.rhb[c] = 0; } return; }
    const u64 base = a.m.blocks[cp.b].rec0;
    u32 bad = 0, base_n = 0;
    if (lane == 0) {                                                          // recs.cpp:279-287: the block's first header is what a chain starts from
        const u64 h0 = a.m.line_off[4 * base] + 1, h1 = a.m.line_off[4 * base + 1] - 1;
        base_n = h1 > h0 ? (u32)(h1 - h0) : 0;
        if (base_n > ML) bad = 1;
        else {
            HdrMask m0;
            const bool ok = rt_stage(L, 0, a.m.fq + h0, base_n, m0);
            const u32 nf = m0.count();
            if (!ok || nf > RF_NF) bad = 1; else { rt_fields(L, 0, m0, nf); L.nf[0] = (u8)nf; }
        }
    }
    if (__any(bad != 0)) { if (lane == 0) flags[c] = 1; return; }
    const u64 end = cp.r0 + cp.nrec;
    const u64 first = cp.r0 == base ? cp.r0 + 1 : cp.r0;                      // the base itself is not coded
    u32 hdr_bytes = cp.r0 == base ? rl(base_n, 0) : 0u;
    u64 since0 = 0;                                                           // fields changed since the chain began / the last shape change, before lane 0's record
    u32 total = 0;
    u32* const out = tok + cp.r0 * RT_TOK_PER_REC;
    const u32 cap = cp.nrec * RT_TOK_PER_REC;
    for (u64 rb = first; rb < end; rb += 64) {
        const u64 r = rb + lane;
        const bool valid = r < end;
        u32 n = 0, nf = 0;
        if (valid) {
            const u64 h0 = a.m.line_off[4 * r] + 1, h1 = a.m.line_off[4 * r + 1] - 1;
            n = h1 > h0 ? (u32)(h1 - h0) : 0;
            if (n > ML) bad = 1;
            else {
                HdrMask m;
                const bool ok = rt_stage(L, col, a.m.fq + h0, n, m);
                nf = m.count();
                if (!ok || nf > RF_NF) bad = 1; else { rt_fields(L, col, m, nf); L.nf[col] = (u8)nf; }
            }
        }
        if (__any(bad != 0)) { if (lane == 0) flags[c] = 1; return; }
        __syncthreads();
        // against the left neighbour
        bool shape = false; u64 map = 0;
        if (valid) {
            shape = nf != L.nf[col - 1];
            for (u32 f = 0; !shape && f < nf; f++) shape = L.sep[0][f][col] != L.sep[0][f][col - 1];
            if (!shape)
                for (u32 f = 0; f < nf; f++) {
                    const u32 wl = L.wln[0][f][col];
                    bool ch = wl != L.wln[0][f][col - 1];
                    if (!ch) {
                        const u32 o = L.off[0][f][col], po = L.off[0][f][col - 1];
                        for (u32 j = 0; j < wl; j++) if (L.byte(col, o + j) != L.byte(col - 1, po + j)) { ch = true; break; }
                    }
                    if (ch) map |= 1ull << f;
                }
        }
        // "changed since": a scan over the lanes of (reset, mask) pairs -- later . earlier = (reset: later's mask; else both), a lane beyond the chain is the identity
        u32 rs = shape ? 1u : 0u; u64 mm = shape ? 0ull : map;
#pragma unroll
        for (u32 dd = 1; dd < 64; dd <<= 1) {
            const u32 ors = (u32)__shfl_up((int)rs, dd, 64);
            const u64 omm = (u64)__shfl_up((unsigned long long)mm, dd, 64);
            if (lane >= dd) { mm = rs ? mm : (mm | omm); rs |= ors; }
        }
        u32 ers = (u32)__shfl_up((int)rs, 1, 64); u64 emm = (u64)__shfl_up((unsigned long long)mm, 1, 64);
        if (lane == 0) { ers = 0; emm = 0; }
        const u64 since = ers ? emm : (since0 | emm);
        {   // what the next round's lane 0 starts from
            const u32 lrs = (u32)__shfl((int)rs, 63, 64); const u64 lmm = (u64)__shfl((unsigned long long)mm, 63, 64);
            since0 = lrs ? lmm : (since0 | lmm);
        }
        // the symbols: made once, into LDS; placed by a wave scan of their numbers; copied out.  (Round 4 walked every record twice -- count, then write --,
        // each walk typing the changed fields and their left neighbours' again: 47 + 27 of this kernel's 100 M instructions per 2 M records.)
        TokStage tl; tl.p = &L.tokst[0][lane]; tl.n = 0;
        bool ok = true;
        if (valid) ok = rec_tokens(L, col, n, nf, shape, map, since, tl);
        if (__any(!ok)) { if (lane == 0) flags[c] = 1; return; }
        const u32 cnt = tl.n;
        const u32 incl = wave_incl_scan(cnt);
        const u32 round = rl(incl, 63);
        if (total + round > cap) { if (lane == 0) flags[c] = 1; return; }
        u32* const dst = out + total + (incl - cnt);
        if (__any(cnt > RT_STAGED)) {                                         // (a record that spells its header out: the walk again, to memory)
            if (valid) { TokStore ts; ts.p = dst; rec_tokens(L, col, n, nf, shape, map, since, ts); }
        } else
            for (u32 j = 0; __any(j < cnt); j++) if (j < cnt) dst[j] = L.tokst[j][lane];
        total += round;
        hdr_bytes += rl(wave_incl_scan(n), 63);
        __syncthreads();
        if (rb + 64 < end) {                                                  // lane 63's record is the next round's left neighbour
            for (u32 p = lane; p < RecTokLds<ML>::NWP; p += 64) L.tw[0][p] = L.tw[64][p];
            if (lane < RF_NF) { L.off[0][lane][0] = L.off[0][lane][64]; L.wln[0][lane][0] = L.wln[0][lane][64]; L.sep[0][lane][0] = L.sep[0][lane][64]; }
            if (lane == 0) L.nf[0] = L.nf[64];
            __syncthreads();
        }
    }
    if (lane == 0) { ntok[c] = total; a.rhb[c] = hdr_bytes; }
}
// step 2: a chain per lane codes its tokens through the frozen rows; sixteen of them staged in LDS for the workgroup
#define RC_LDS_ROWS 16u
struct RecCodeLds { u32 rows[RC_LDS_ROWS * 256]; u8 map[PR_REC_ROWS]; };
__device__ __forceinline__ u32 rec_code_entry(const RecCodeLds& L, const u32* __restrict__ grows, u32 row, u32 sym) {
    const u32 slot = L.map[row];
    return slot != 0xFFu ? L.rows[slot * 256 + sym] : grows[(size_t)row * 256 + sym];
}
__global__ __launch_bounds__(256) void k_rec_code(ChainArgs a, const u32* __restrict__ tok, const u32* __restrict__ ntok, const u32* __restrict__ flags, u32 n_hot) {
#ifdef PRIO_REC
    __builtin_amdgcn_s_setprio(PRIO_REC);
#endif
    __shared__ RecCodeLds L;
    for (u32 i = threadIdx.x; i < PR_REC_ROWS; i += 256) { const u32 sl = a.rmap[i]; L.map[i] = (u8)(sl < n_hot ? sl : 0xFFu); }
    for (u32 i = threadIdx.x; i < n_hot * 256; i += 256) L.rows[i] = a.rrows[(size_t)a.rhot[i >> 8] * 256 + (i & 255)];
    __syncthreads();
    const u32 c = blockIdx.x * 256 + threadIdx.x;
    if (c >= a.rgeo.nchains || flags[c]) return;
    const RecChainPos cp = rec_chain_pos(a, c);
    BlockDesc* d = &a.m.blocks[cp.b];
    u32 cap = 0;
    u8* outp = rec_chain_region(a, cp, cap);
    RecFastEnc<RecCodeLds> cd; cd.rows = a.rrows; cd.L = &L; cd.rc.init(outp, cap);
    const uint4* t = reinterpret_cast<const uint4*>(tok + cp.r0 * RT_TOK_PER_REC);       // (160 bytes per record: 16-byte aligned)
    const u32 n = ntok[c];
    // Tokens sixteen at a time, two groups ahead; their row entries one group ahead of the coder -- the rows are frozen, so an
    // entry depends on nothing the coder does.  (Round 4: written as fetch-then-code per token the coder's own stores kept the
    // compiler from moving a fetch up, and a lane paid a memory round trip per symbol: 0.93 ms for the 128-record chains of
    // a 600 k-read call, 600 ns a symbol.  Round 5: groups of four made it a round trip per four symbols -- 640 of them down a chain of 171
    // records, 2 ms with the chip to itself and 4.9 at the tail of a call whose CUs the quality chains hold; sixteen entries in flight now.)
    const u32* const grows = a.rrows;
#define RC_ENTRY(tk, on) rec_code_entry(L, grows, (on) ? (tk) >> 8 : 0u, (tk) & 0xffu)
#define RC_ENTRIES(v, i) make_uint4(RC_ENTRY((v).x, (i) < n), RC_ENTRY((v).y, (i) + 1 < n), RC_ENTRY((v).z, (i) + 2 < n), RC_ENTRY((v).w, (i) + 3 < n))
    // (a token group is loaded under an `if`, not chosen by `? :` against a zero constant: that becomes a load through a chosen
    //  POINTER, with the constant in scratch)
    constexpr u32 Q = 4;                                                      // uint4s a group
    uint4 t1[Q], e[Q];
#pragma unroll
    for (u32 k = 0; k < Q; k++) { uint4 t0 = make_uint4(0, 0, 0, 0); t1[k] = t0; if (4 * k < n) t0 = t[k]; if (4 * (Q + k) < n) t1[k] = t[Q + k]; e[k] = RC_ENTRIES(t0, 4u * k); }
    for (u32 i = 0; i < n; i += 4 * Q) {
        uint4 t2[Q], e2[Q];
#pragma unroll
        for (u32 k = 0; k < Q; k++) { t2[k] = make_uint4(0, 0, 0, 0); if (i + 4 * (2 * Q + k) < n) t2[k] = t[(i >> 2) + 2 * Q + k]; }
#pragma unroll
        for (u32 k = 0; k < Q; k++) e2[k] = RC_ENTRIES(t1[k], i + 4u * (Q + k));
#pragma unroll
        for (u32 k = 0; k < Q; k++) {
            const u32 j = i + 4 * k;
            if (j < n) cd.rc.encode16(FZ_CUM(e[k].x), FZ_FREQ(e[k].x));
            if (j + 1 < n) cd.rc.encode16(FZ_CUM(e[k].y), FZ_FREQ(e[k].y));
            if (j + 2 < n) cd.rc.encode16(FZ_CUM(e[k].z), FZ_FREQ(e[k].z));
            if (j + 3 < n) cd.rc.encode16(FZ_CUM(e[k].w), FZ_FREQ(e[k].w));
        }
#pragma unroll
        for (u32 k = 0; k < Q; k++) { t1[k] = t2[k]; e[k] = e2[k]; }
    }
#undef RC_ENTRIES
#undef RC_ENTRY
    a.csz[c] = cd.rc.finish();
    if (cd.rc.err & 2) atomicMax(&d->status, (u32)(-SFQ_E_OVERFLOW));
    if (cd.rc.err & 1) atomicMax(&d->status, (u32)(-SFQ_E_CORRUPT));
}
// flags, flags2: one dword per header chain each, zeroed by the caller; tok: RT_TOK_PER_REC dwords per record of the call; ntok: [chains]
void launch_rec_encode_c(const ChainArgs& a, u32* flags, u32* flags2, u32* tok, u32* ntok, u32 n_hot, u32 max_hdr, hipStream_t st, u32 min_hdr, hipEvent_t after_tokens) {
    const u32 nc = a.rgeo.nchains;
    if (!nc) return;
    if (min_hdr > 127) {           // every header is past the token step's and the fast lane kernel's 127 bytes (long reads' UUID headers): the general kernel alone
        hipLaunchKernelGGL(k_rec_encode_c, dim3((nc + 63) / 64), dim3(64), 0, st, a, (const u32*)nullptr);
        return;
    }
    if (max_hdr <= 62) hipLaunchKernelGGL(k_rec_tokens<62>, dim3(nc), dim3(64), 0, st, a, tok, ntok, flags);
    else if (max_hdr <= 94) hipLaunchKernelGGL(k_rec_tokens<94>, dim3(nc), dim3(64), 0, st, a, tok, ntok, flags);
    else hipLaunchKernelGGL(k_rec_tokens<127>, dim3(nc), dim3(64), 0, st, a, tok, ntok, flags);
    if (after_tokens) (void)hipEventRecord(after_tokens, st);
    hipLaunchKernelGGL(k_rec_code, dim3((nc + 255) / 256), dim3(256), 0, st, a, (const u32*)tok, (const u32*)ntok, (const u32*)flags, n_hot < RC_LDS_ROWS ? n_hot : RC_LDS_ROWS);
    // what the token step left: a chain per lane with everything in LDS, then the general kernel for what that leaves
    const dim3 grid((nc + 63) / 64);
    if (max_hdr <= 62) hipLaunchKernelGGL(k_rec_encode_f<62>, grid, dim3(64), 0, st, a, flags2, (const u32*)flags);
    else if (max_hdr <= 94) hipLaunchKernelGGL(k_rec_encode_f<94>, grid, dim3(64), 0, st, a, flags2, (const u32*)flags);
    else hipLaunchKernelGGL(k_rec_encode_f<127>, grid, dim3(64), 0, st, a, flags2, (const u32*)flags);
    hipLaunchKernelGGL(k_rec_encode_c, grid, dim3(64), 0, st, a, (const u32*)flags2);
}
u64 rec_token_bytes(u64 nrec) { return nrec * RT_TOK_PER_REC * 4; }

// ---- "chn.idx": the size lists as bytes, on the device ------------------------------------------------------------------
// The four lists of a call's chain sizes (quality, bases, header chains, the header bytes they restore: csz[0 .. n), list k
// = [b[k], b[k + 1])) each as zigzag differences to the entry before, every difference a LEB128 varint -- what api.cpp wrote on
// the host from a copy of csz: half a million values, 1.0 ms AFTER the device had finished (round 4).  Two kernels around a
// scan: the lengths, then the bytes; out[0 .. off[n]) = the lists back to back, info[0] = off[b[2]] (where the header chains'
// lists begin), info[1] = off[n].
__device__ __forceinline__ u32 chn_zz(const u32* __restrict__ csz, u32 i, u32 b1, u32 b2, u32 b3) {
    const u32 prev = (i == 0 || i == b1 || i == b2 || i == b3) ? 0u : csz[i - 1];
    const i32 d = (i32)(csz[i] - prev);
    return ((u32)d << 1) ^ (u32)(d >> 31);
}
__global__ __launch_bounds__(256) void k_chn_len(const u32* __restrict__ csz, u32 n, u32 b1, u32 b2, u32 b3, u32* __restrict__ len) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32 z = chn_zz(csz, i, b1, b2, b3);
    len[i] = z < (1u << 7) ? 1u : z < (1u << 14) ? 2u : z < (1u << 21) ? 3u : z < (1u << 28) ? 4u : 5u;
}
__global__ __launch_bounds__(256) void k_chn_bytes(const u32* __restrict__ csz, u32 n, u32 b1, u32 b2, u32 b3, const u64* __restrict__ off, u8* __restrict__ out, u64* __restrict__ info) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) { info[0] = off[b2]; info[1] = off[n]; }
    if (i >= n) return;
    u32 z = chn_zz(csz, i, b1, b2, b3);
    u8* w = out + off[i];
    while (z >= 0x80u) { *w++ = (u8)(z | 0x80u); z >>= 7; }
    *w = (u8)z;
}
void launch_chain_index_bytes(const u32* csz, u32 n, u32 b1, u32 b2, u32 b3, u32* len, u64* off, u64* scan_tmp, u8* out, u64* info, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_chn_len, dim3((n + 255) / 256), dim3(256), 0, st, csz, n, b1, b2, b3, len);
    launch_scan_u32((const u32*)len, off, n, scan_tmp, st);
    hipLaunchKernelGGL(k_chn_bytes, dim3((n + 255) / 256), dim3(256), 0, st, csz, n, b1, b2, b3, (const u64*)off, out, info);
}

// header decode: one chain per lane.  DecodeArgs::hdr_stage_off / hdr_stage_cap are per CHAIN here.
// largest s with cum[s] <= prob in a row of the decoder's form (RDEC_ROW): the sixteenth of the row, then the symbol in
// it.  Two round trips of 32 bytes (a search that fetched entry by entry was nine dependent fetches).
// (P names the row's address space -- LDS or global: a search through a generic pointer that may be either is four flat loads)
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) u16* lds_row;
typedef const __attribute__((address_space(1))) u16* glb_row;
template <typename V> __device__ __forceinline__ void unpack8v(V v, u32* o) {
    o[0] = v.x & 0xffffu; o[1] = v.x >> 16; o[2] = v.y & 0xffffu; o[3] = v.y >> 16; o[4] = v.z & 0xffffu; o[5] = v.z >> 16; o[6] = v.w & 0xffffu; o[7] = v.w >> 16;
}
__device__ __forceinline__ u32x4 row16(lds_row p) { return *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(p); }
__device__ __forceinline__ u32x4 row16(glb_row p) { return *reinterpret_cast<const __attribute__((address_space(1))) u32x4*>(p); }
template <typename P>
__device__ __forceinline__ u32 rdec_search(P rd, u32 prob, u32& cum, u32& next) {
    u32 cc[16], ff[16];
    unpack8v(row16(rd), cc);
    unpack8v(row16(rd + 8), cc + 8);
    u32 k = 0;
#pragma unroll
    for (u32 j = 1; j < 16; j++) k += cc[j] <= prob;
    P fr = rd + 16 + k * 16;
    unpack8v(row16(fr), ff);
    unpack8v(row16(fr + 8), ff + 8);
    u32 s = 0;
    cum = ff[0]; next = 65536u;
#pragma unroll
    for (u32 j = 15; j >= 1; j--) next = cc[j] > prob ? cc[j] : next;
#pragma unroll
    for (u32 j = 15; j >= 1; j--) next = ff[j] > prob ? ff[j] : next;
#pragma unroll
    for (u32 j = 1; j < 16; j++) { const bool le = ff[j] <= prob; s += le; cum = le ? ff[j] : cum; }
    return k * 16 + s;
}
struct RecFrozenDec {
    static constexpr bool inband = true;
    const u16* rdec; LaneDec rc;
    __device__ __forceinline__ u32 get(u32 row) {
        const u32 prob = rc.get_freq16();
        u32 cum, next;
        const u32 s = rdec_search((glb_row)(rdec + (size_t)row * RDEC_ROW), prob, cum, next);
        rc.decode(cum, next - cum);
        return s;
    }
    __device__ __forceinline__ u64 get_u(u32 row0) { return get_u_rows(*this, row0); }
    __device__ __forceinline__ u32 err() const { return rc.err; }
};
// general path: one chain per lane; with `flags` only the chains the fast kernel below has handed over
__global__ __launch_bounds__(64) void k_rec_decode_c(ChainArgs a, DecodeArgs da, const u32* flags) {
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.rgeo.nchains) return;
    if (flags && !flags[c]) return;
    const RecChainPos cp = rec_chain_pos(a, c);
    BlockDesc* d = &da.m.blocks[cp.b];
    RecFrozenDec cd; cd.rdec = a.rdec;
    cd.rc.init(da.streams + a.coff[c], a.csz[c]);
    XfDec x_rec; x_rec.init(nullptr, 0, XF_REC_X);
    PwTab none; none.slots = nullptr; none.hdr = nullptr; none.epoch = 0;
    rec_decode_lane(da, d, cp.r0, cp.nrec, c, cd, x_rec, none);
}

// header decode, fast path (the decoder's side of k_rec_encode_f): RecLoad::load (recs.cpp:374-461) for chains whose
// headers stay within RF_DML bytes and RF_NF fields, with the previous and the current header, the previous header's
// field table, the field types / values and the hottest rows (in the decoder's form) in LDS, laid out [..][lane], and the
// headers leaving through LaneOut (sixteen bytes a store).  The general path keeps these in per-lane scratch, re-reads
// the previous header from the staging area byte by byte and stores single bytes: ~460 memory instructions per record
// against ~10 here.  Anything unusual -- a longer header, more fields, a value that prints with a sign, a damaged
// stream -- hands the chain over (flags[c] = 1): the general kernel starts it again and reports what it finds.
#define RF_DML 127u
struct RecFastDecLds {
    u64 fvalue[RF_NF][64];
    u16 drows[RDEC_LDS_ROWS][RDEC_ROW];
    u8  text[2][RF_DML + 1][64];
    u8  off[RF_NF][64], wln[RF_NF][64], sep[RF_NF][64];        // of the previous header
    u8  fkind[RF_NF][64];
    u8  map[PR_REC_ROWS];                                      // row -> LDS slot, 0xFF = not staged
};
struct RecFastDecSrc {
    const u16* rdec; const RecFastDecLds* L; LaneDec rc;
    __device__ __forceinline__ u32 get(u32 row) {
        const u32 prob = rc.get_freq16();
        const u32 slot = L->map[row];
        u32 cum, next, s;
        if (slot != 0xFFu) s = rdec_search((lds_row)L->drows[slot], prob, cum, next);
        else s = rdec_search((glb_row)(rdec + (size_t)row * RDEC_ROW), prob, cum, next);
        rc.decode(cum, next - cum);
        return s;
    }
    __device__ __forceinline__ u64 get_u(u32 row0) { return get_u_rows(*this, row0); }
};
// the header in text[buf][0 .. n) goes out (with its '\n') and is tokenised for the record after it (d_map_space,
// recs.cpp:141-157); returns the number of fields
__device__ __forceinline__ u32 rfd_emit(RecFastDecLds& L, u32 buf, u32 lane, u32 n, LaneOut* out) {
    u32 nf = 0, start = 0; bool stop = false;
    for (u32 pos = 0; pos <= n; pos++) {
        const u32 c = pos < n ? L.text[buf][pos][lane] : '\n';
        if (out) out->put(c);
        if (!stop && !isword(c)) {
            if (nf < RF_NF) { L.off[nf][lane] = (u8)start; L.wln[nf][lane] = (u8)(pos - start); L.sep[nf][lane] = (u8)c; }
            nf++; start = pos + 1;
            if (c == 0) stop = true;
        }
    }
    return nf;
}
__device__ __forceinline__ bool rec_fast_decode_lane(const DecodeArgs& a, const BlockDesc* d, RecFastDecLds& L, u32 lane, u64 r0, u32 nrec, u32 chain,
                                                     RecFastDecSrc& cd) {
    const u64 cap = a.hdr_stage_cap[chain], stage_off = a.hdr_stage_off[chain];
    u32 n_prev = d->first_hdr_len;
    if (n_prev > RF_DML) return false;
    {                                                                          // the base: the block's first header (recs.cpp:113-119)
        const u8* fh = a.first_hdrs + d->first_hdr_off;
        for (u32 j = 0; j < n_prev; j++) L.text[0][j][lane] = fh[j];
    }
    u32 prv = 0, cur = 1;
    u32 nf_prev = rfd_emit(L, prv, lane, n_prev, nullptr);
    if (nf_prev > RF_NF) return false;
    for (u32 f = 0; f < RF_NF; f++) L.fkind[f][lane] = 0;
    LaneOut out; out.begin(a.hdr_stage + stage_off);
    u64 pos = 0;
    for (u32 k = 0; k < nrec; k++) {
        const u64 r = r0 + k;
        if (pos + SFQ_MAX_ID_LLEN + 2 > cap) return false;                     // (the general path reports it)
        u32 n;
        if (r == d->rec0) {                                                    // the base itself
            n = n_prev;
            for (u32 j = 0; j < n; j++) out.put(L.text[prv][j][lane]);
            out.put('\n');
        } else {
            if (cd.get(REC_FLAG_ROW) != 0) {                                   // a header of another shape, whole
                const u64 len = cd.get_u(REC_FLAG_ROW + 2);
                if (len > RF_DML) return false;
                for (u32 j = 0; j < (u32)len; j++) L.text[cur][j][lane] = (u8)cd.get(REC_FLAG_ROW + 1);
                n = (u32)len;
                for (u32 f = 0; f < RF_NF; f++) L.fkind[f][lane] = 0;
            } else {
                const u64 map = cd.get_u(0 * 16 + 2);
                u32 b = 0;
                for (u32 i = 0; i < nf_prev; i++) {
                    const u32 wl = L.wln[i][lane], o = L.off[i][lane];
                    if (b + wl + 24 > RF_DML) return false;                    // room for the field as it was or as a number (<= 21 bytes), and its separator
                    const u32 rr = (i + 1) * 16;
                    if (!((map >> i) & 1)) {
                        for (u32 j = 0; j < wl; j++) L.text[cur][b + j][lane] = L.text[prv][o + j][lane];
                        b += wl;
                    } else {
                        const u32 type = cd.get(rr + 0);
                        if (type == ST_STR) {                                  // recs.cpp:421-428
                            const u64 len = cd.get_u(rr + 2);
                            if (len > RF_DML || b + len + 2 > RF_DML) return false;
                            for (u32 j = 0; j < (u32)len; j++) L.text[cur][b + j][lane] = (u8)cd.get(rr + 1);
                            b += (u32)len;
                            L.fkind[i][lane] = 0;
                        } else {                                               // recs.cpp:430-456
                            if (type > ST_DLT_Z) return false;
                            const u64 pval = L.fkind[i][lane] ? L.fvalue[i][lane] : 0;
                            const u64 gap = cd.get_u(rr + 2);
                            const bool less = type == ST_DLT || type == ST_HLT || type == ST_HLT_Z || type == ST_HLTC ||
                                              type == ST_HLTC_Z || type == ST_DLT_Z;
                            const u64 val = less ? pval - gap : pval + gap;
                            const bool deci = type < ST_STR || type >= ST_DGT_Z;
                            const bool lead = type == ST_HGT_Z || type == ST_HLT_Z || type == ST_HGTC_Z || type == ST_HLTC_Z ||
                                              type == ST_DGT_Z || type == ST_DLT_Z;
                            const bool upper = type >= ST_HGTC && type <= ST_HLTC_Z;
                            L.fkind[i][lane] = deci ? 1 : 2;
                            L.fvalue[i][lane] = val;
                            if (val == 0) L.text[cur][b++][lane] = '0';        // recs.cpp:453-454
                            else {
                                if (lead) L.text[cur][b++][lane] = '0';
                                if (deci) {                                    // "%lld"
                                    if ((i64)val < 0) return false;            // (a sign changes the fields: the general path's business)
                                    u32 nd = 0;
                                    for (u64 t = val; t; t /= 10) nd++;
                                    u64 t = val;
                                    for (u32 j = nd; j-- > 0;) { L.text[cur][b + j][lane] = (u8)('0' + t % 10); t /= 10; }
                                    b += nd;
                                } else {                                       // "%llx" / "%llX"
                                    int sh = 60;
                                    while (sh > 0 && ((val >> sh) & 0xf) == 0) sh -= 4;
                                    for (; sh >= 0; sh -= 4) {
                                        const u32 dg = (u32)(val >> sh) & 0xf;
                                        L.text[cur][b++][lane] = (u8)(dg < 10 ? '0' + dg : (upper ? 'A' : 'a') + dg - 10);
                                    }
                                }
                            }
                        }
                    }
                    L.text[cur][b++][lane] = L.sep[i][lane];
                }
                n = b - 1;                                                     // recs.cpp:460
            }
            if (cd.rc.err) return false;
            nf_prev = rfd_emit(L, cur, lane, n, &out);
            if (nf_prev > RF_NF) return false;
            prv = cur; cur ^= 1u; n_prev = n;
        }
        a.hlen[r] = n; a.hoff[r] = stage_off + pos;
        pos += (u64)n + 1;
    }
    out.end();
    return true;
}
__global__ __launch_bounds__(64) void k_rec_decode_f(ChainArgs a, DecodeArgs da, u32* flags, const u32* only /* the chains to take; null = all */) {
    __builtin_amdgcn_s_setprio(3);         // (a wave per SIMD with a long serial walk beside the quality decoder's many: 19.2 -> 16.2 ms)
    __shared__ RecFastDecLds L;
    const u32 lane = threadIdx.x;
    {
        const u32 c0 = blockIdx.x * 64 + lane;
        if (only && !__any(c0 < a.rgeo.nchains && only[c0] == 1)) return;      // (2: marked by the host for the general kernel, its flag set there)
    }
    for (u32 i = lane; i < PR_REC_ROWS; i += 64) { const u32 sl = a.rmap[i]; L.map[i] = (u8)(sl < RDEC_LDS_ROWS ? sl : 0xFFu); }
    for (u32 i = lane; i < a.r_hot * RDEC_ROW; i += 64) L.drows[i / RDEC_ROW][i % RDEC_ROW] = a.rdec[(size_t)a.rhot[i / RDEC_ROW] * RDEC_ROW + i % RDEC_ROW];
    __syncthreads();
    const u32 c = blockIdx.x * 64 + lane;
    if (c >= a.rgeo.nchains) return;
    if (only && only[c] != 1) return;
    const RecChainPos cp = rec_chain_pos(a, c);
    const BlockDesc* d = &da.m.blocks[cp.b];
    RecFastDecSrc cd; cd.rdec = a.rdec; cd.L = &L;
    cd.rc.init(da.streams + a.coff[c], a.csz[c]);
    if (!rec_fast_decode_lane(da, d, L, lane, cp.r0, cp.nrec, c, cd)) flags[c] = 1;
}
// ---- header decode in two steps: the symbols, then the text --------------------------------------------------------------------
// k_rec_decode_f above walks a chain per lane with two headers, the field table and the field values of every lane in LDS
// (38 KiB per wave: one wave per SIMD) -- 5.5 ms alone and 16-19 ms beside the quality decoder, the longest kernel of a decode.
// But WHICH symbols a chain holds depends on the decoded values only, never on the text: flag, change map, then per changed field
// a type and a gap or a string.  So step 1 (k_rec_dsym) is a decoder per lane with nothing but the coder and a cursor: it turns a
// chain's stream into tokens -- per record the change map, then per changed field one word (type, a gap below 2^24) or more (a
// longer gap, a string's bytes).  Step 2 (k_rec_dtext) takes a RECORD per lane, a chain per wave, 64 records a round, and rebuilds
// the texts field by field: a changed number is the field's running value -- a segmented prefix sum of the signed gaps over the
// lanes, reset where the field was a string (recs.cpp:333: the previous value counts only while the field is numeric) --, an
// unchanged field is the text of its LAST WRITER -- a running maximum of lane numbers --, or of the round before (a carried copy of
// the last record).  Chains with a header of another shape, more than 16 fields, headers over 127 bytes or a sign in a number are
// left to the kernels above (dflags).
#define RD_TOK_PER_REC 16u
struct RecDsymLds { u16 drows[RDEC_LDS_ROWS][RDEC_ROW]; u8 map[PR_REC_ROWS]; };
struct RecSymSrc {
    const u16* rdec; const RecDsymLds* L; LaneDec rc;
    __device__ __forceinline__ u32 get(u32 row) {
        const u32 prob = rc.get_freq16();
        const u32 slot = L->map[row];
        u32 cum, next, s;
        if (slot != 0xFFu) s = rdec_search((lds_row)L->drows[slot], prob, cum, next);
        else s = rdec_search((glb_row)(rdec + (size_t)row * RDEC_ROW), prob, cum, next);
        rc.decode(cum, next - cum);
        return s;
    }
    __device__ __forceinline__ u64 get_u(u32 row0) { return get_u_rows(*this, row0); }
};
__global__ __launch_bounds__(256) void k_rec_dsym(ChainArgs a, DecodeArgs da, u32* __restrict__ dtok, u32* __restrict__ dtoff, u32* __restrict__ dflags) {
    __shared__ RecDsymLds L;
    for (u32 i = threadIdx.x; i < PR_REC_ROWS; i += 256) { const u32 sl = a.rmap[i]; L.map[i] = (u8)(sl < RDEC_LDS_ROWS ? sl : 0xFFu); }
    for (u32 i = threadIdx.x; i < a.r_hot * RDEC_ROW; i += 256) L.drows[i / RDEC_ROW][i % RDEC_ROW] = a.rdec[(size_t)a.rhot[i / RDEC_ROW] * RDEC_ROW + i % RDEC_ROW];
    __syncthreads();
    const u32 c = blockIdx.x * 256 + threadIdx.x;
    if (c >= a.rgeo.nchains) return;
    if (dflags[c]) return;                                  // (marked by the host: headers too long for the two steps, api.cpp)
    const RecChainPos cp = rec_chain_pos(a, c);
    const BlockDesc* d = &da.m.blocks[cp.b];
    RecSymSrc cd; cd.rdec = a.rdec; cd.L = &L;
    cd.rc.init(da.streams + a.coff[c], a.csz[c]);
    u32* const out = dtok + cp.r0 * RD_TOK_PER_REC;
    const u32 cap = cp.nrec * RD_TOK_PER_REC;
    u32 w = 0; bool bad = false;
    uint4 acc = make_uint4(0, 0, 0, 0);
    auto emit = [&](u32 word) {                                   // four words a store
        const u32 q = w & 3u;
        acc.x = q == 0 ? word : acc.x; acc.y = q == 1 ? word : acc.y; acc.z = q == 2 ? word : acc.z; acc.w = q == 3 ? word : acc.w;
        if (q == 3u && w < cap) *reinterpret_cast<uint4*>(out + (w - 3u)) = acc;
        w++;
    };
    for (u32 k = 0; k < cp.nrec && !bad; k++) {
        const u64 r = cp.r0 + k;
        if (r == d->rec0) { dtoff[r] = 0xFFFFFFFFu; continue; }           // the base itself: nothing is coded for it
        dtoff[r] = w;
        if (cd.get(REC_FLAG_ROW) != 0) { bad = true; break; }             // a header of another shape: the lane kernels' business
        const u64 map = cd.get_u(0 * 16 + 2);
        if (map >> RF_NF) { bad = true; break; }
        emit((u32)map);
        for (u32 f = 0; f < RF_NF && !bad; f++) {
            if (!((map >> f) & 1)) continue;
            const u32 rr = (f + 1) * 16;
            const u32 type = cd.get(rr + 0);
            if (type == ST_STR) {
                const u64 len = cd.get_u(rr + 2);
                if (len > RF_DML) { bad = true; break; }
                emit(type | (2u << 4) | ((u32)len << 8));
                u32 word = 0;
                for (u32 j = 0; j < (u32)len; j++) {
                    word |= cd.get(rr + 1) << (8 * (j & 3u));
                    if ((j & 3u) == 3u) { emit(word); word = 0; }
                }
                if (len & 3u) emit(word);
            } else if (type > ST_DLT_Z) bad = true;
            else {
                const u64 gap = cd.get_u(rr + 2);
                if (gap < (1u << 24)) emit(type | ((u32)gap << 8));
                else { emit(type | (1u << 4)); emit((u32)gap); emit((u32)(gap >> 32)); }
            }
        }
        if (cd.rc.err || w + 8 > cap) bad = true;                        // (room for the next record's first words is checked as it goes)
    }
    if (!bad) {                                                         // what the last store left
        const u32 q = w & 3u, b0 = w - q;
        if (q > 0 && b0 < cap) out[b0] = acc.x;
        if (q > 1 && b0 + 1 < cap) out[b0 + 1] = acc.y;
        if (q > 2 && b0 + 2 < cap) out[b0 + 2] = acc.z;
        if (w > cap) bad = true;
    }
    if (bad) dflags[c] = 1;
}
struct RecDtLds {
    u8 scratch[RF_DML + 1][64];                                         // the changed fields' texts of the round's records: [byte][lane]
    u8 soff[RF_NF][64], slen[RF_NF][64], wsrc[RF_NF][64];               // where a lane's field f lies in its scratch column; the lane that wrote field f last (0xFF: the round before)
    u8 ctext[2][RF_DML + 1];                                            // the record before lane 0's, whole (two copies take turns)
    u8 coff[2][RF_NF], cwln[2][RF_NF];
    u8 csep[RF_NF];                                                     // the separators: the same for every record of a chain without shape changes
    u64 cval[RF_NF];                                                    // the fields' running values behind that record (0: cold, or a string since)
};
__global__ __launch_bounds__(64) void k_rec_dtext(ChainArgs a, DecodeArgs da, const u32* __restrict__ dtok, const u32* __restrict__ dtoff, u32* __restrict__ dflags) {
    __shared__ RecDtLds L;
    const u32 lane = threadIdx.x;
    const u32 c = blockIdx.x;
    if (dflags[c]) return;
    const RecChainPos cp = rec_chain_pos(a, c);
    if (cp.nrec == 0) return;
    const BlockDesc* d = &da.m.blocks[cp.b];
    const u64 cap = da.hdr_stage_cap[c], stage_off = da.hdr_stage_off[c];
    // the base: the block's first header (recs.cpp:113-119), tokenised by every lane alike (d_map_space, recs.cpp:141-157)
    const u32 n0 = d->first_hdr_len;
    if (n0 > RF_DML) { if (lane == 0) dflags[c] = 1; return; }
    {
        const u8* fh = da.first_hdrs + d->first_hdr_off;
        for (u32 j = lane; j < n0; j += 64) L.ctext[0][j] = fh[j];
        if (lane < RF_NF) L.cval[lane] = 0;
    }
    __syncthreads();
    u32 nf = 0;
    {
        u32 start = 0; bool stop = false;
        for (u32 pos = 0; pos <= n0; pos++) {
            const u32 ch = pos < n0 ? L.ctext[0][pos] : '\n';
            if (!stop && !isword(ch)) {
                if (nf < RF_NF && lane == 0) { L.coff[0][nf] = (u8)start; L.cwln[0][nf] = (u8)(pos - start); L.csep[nf] = (u8)ch; }
                nf++; start = pos + 1;
                if (ch == 0) stop = true;
            }
        }
    }
    if (nf > RF_NF) { if (lane == 0) dflags[c] = 1; return; }
    __syncthreads();
    const u32* const T = dtok + cp.r0 * RD_TOK_PER_REC;
    const u64 end = cp.r0 + cp.nrec;
    u64 first = cp.r0;
    u64 pos = 0;
    if (cp.r0 == d->rec0) {                                             // the base itself is this chain's first record
        if (lane == 0) { da.hlen[cp.r0] = n0; da.hoff[cp.r0] = stage_off; }
        for (u32 j = lane; j < n0; j += 64) da.hdr_stage[stage_off + j] = L.ctext[0][j];
        if (lane == 0) da.hdr_stage[stage_off + n0] = '\n';
        pos = (u64)n0 + 1; first = cp.r0 + 1;
    }
    u32 cb = 0;
    for (u64 rb = first; rb < end; rb += 64) {
        const u64 r = rb + lane;
        const bool valid = r < end;
        u32 cur = 0, map = 0, spos = 0; bool bad = false;
        if (valid) { cur = dtoff[r]; map = T[cur]; cur++; }
        for (u32 f = 0; f < nf; f++) {
            const bool ch = valid && ((map >> f) & 1u);
            u32 type = 0, slen = 0, strw = 0; u64 gap = 0;
            if (ch) {
                const u32 t = T[cur++];
                type = t & 15u;
                const u32 kind = (t >> 4) & 3u;
                if (kind == 0) gap = t >> 8;
                else if (kind == 1) { gap = (u64)T[cur] | ((u64)T[cur + 1] << 32); cur += 2; }
                else { slen = (t >> 8) & 0xffu; strw = cur; cur += (slen + 3u) >> 2; }
            }
            const bool isstr = ch && type == ST_STR, isnum = ch && !isstr;
            const bool less = type == ST_DLT || type == ST_HLT || type == ST_HLT_Z || type == ST_HLTC || type == ST_HLTC_Z || type == ST_DLT_Z;
            // (a field that no record of the round changes -- most fields of most rounds -- has nothing to add up and no writer: the six-step scan of a
            //  64-bit sum, a flag and a lane number was a fifth of this kernel's instructions)
            if (!__any(ch)) { L.wsrc[f][lane] = 0xFFu; continue; }
            // the field's running value: a sum of the signed gaps over the lanes, starting over behind a string
            u32 rs = isstr ? 1u : 0u;
            u64 sv = isnum ? (less ? (u64)0 - gap : gap) : 0ull;
            int wl = ch ? (int)lane : -1;                                // ... and the lane that wrote the field last
#pragma unroll
            for (u32 dd = 1; dd < 64; dd <<= 1) {
                const u32 ors = (u32)__shfl_up((int)rs, dd, 64);
                const u64 osv = (u64)__shfl_up((unsigned long long)sv, dd, 64);
                const int owl = __shfl_up(wl, dd, 64);
                if (lane >= dd) { sv = rs ? sv : sv + osv; rs |= ors; wl = owl > wl ? owl : wl; }
            }
            const u64 before = L.cval[f];
            const u64 val = rs ? sv : before + sv;
            {
                const u32 lrs = (u32)__shfl((int)rs, 63, 64); const u64 lsv = (u64)__shfl((unsigned long long)sv, 63, 64);
                if (lane == 0) L.cval[f] = lrs ? lsv : before + lsv;
            }
            if (ch) {
                u32 len = 0;
                if (isstr) {
                    len = slen;
                    if (spos + len > RF_DML) bad = true;
                    else for (u32 j = 0; j < len; j++) L.scratch[spos + j][lane] = (u8)(T[strw + (j >> 2)] >> (8 * (j & 3u)));
                } else {                                                 // recs.cpp:430-456
                    const bool deci = type < ST_STR || type >= ST_DGT_Z;
                    const bool lead = type == ST_HGT_Z || type == ST_HLT_Z || type == ST_HGTC_Z || type == ST_HLTC_Z || type == ST_DGT_Z || type == ST_DLT_Z;
                    const bool upper = type >= ST_HGTC && type <= ST_HLTC_Z;
                    if (spos + 24 > RF_DML) bad = true;
                    else if (val == 0) { L.scratch[spos][lane] = '0'; len = 1; }      // recs.cpp:453-454
                    else {
                        if (lead) L.scratch[spos + len++][lane] = '0';
                        if (deci) {                                      // "%lld"
                            if ((i64)val < 0) bad = true;                // (a sign changes the fields: the general path's business)
                            u32 nd = 0;
                            if (!__any(!bad && (val >> 32) != 0)) {      // (the usual number fits 32 bits: a multiply-high a digit, not a 64-bit division)
                                for (u32 t = (u32)val; t; t /= 10u) nd++;
                                u32 t = (u32)val;
                                for (u32 j = nd; j-- > 0;) { const u32 q = t / 10u; L.scratch[spos + len + j][lane] = (u8)('0' + (t - q * 10u)); t = q; }
                            } else {
                                for (u64 t = val; t; t /= 10) nd++;
                                u64 t = val;
                                for (u32 j = nd; j-- > 0;) { L.scratch[spos + len + j][lane] = (u8)('0' + t % 10); t /= 10; }
                            }
                            len += nd;
                        } else {                                         // "%llx" / "%llX"
                            int sh = 60;
                            while (sh > 0 && ((val >> sh) & 0xf) == 0) sh -= 4;
                            for (; sh >= 0; sh -= 4) {
                                const u32 dg = (u32)(val >> sh) & 0xf;
                                L.scratch[spos + len++][lane] = (u8)(dg < 10 ? '0' + dg : (upper ? 'A' : 'a') + dg - 10);
                            }
                        }
                    }
                }
                L.soff[f][lane] = (u8)spos; L.slen[f][lane] = (u8)len;
                spos += len;
            }
            L.wsrc[f][lane] = (u8)(wl < 0 ? 0xFF : wl);
        }
        if (__any(bad)) { if (lane == 0) dflags[c] = 1; return; }
        __syncthreads();
        // the lengths, the places in the staging area, then the texts
        u32 n = 0;
        if (valid) {
            for (u32 f = 0; f < nf; f++) { const u32 src = L.wsrc[f][lane]; n += src == 0xFFu ? L.cwln[cb][f] : L.slen[f][src]; }
            n += nf - 1;                                                 // the separators between the fields (recs.cpp:460)
        }
        if (__any(valid && n > RF_DML)) { if (lane == 0) dflags[c] = 1; return; }
        const u32 mine = valid ? n + 1 : 0;
        const u32 incl = wave_incl_scan(mine);
        const u32 round = rl(incl, 63);
        if (pos + round + SFQ_MAX_ID_LLEN + 2 > cap) { if (lane == 0) dflags[c] = 1; return; }     // (the general path reports it)
        const u32 nvalid = (u32)(end - rb < 64 ? end - rb : 64);
        const bool keep = lane + 1 == nvalid && rb + 64 < end;           // this lane's record is what the next round starts from
        if (valid) {
            const u64 at = stage_off + pos + (incl - mine);
            da.hlen[r] = n; da.hoff[r] = at;
            LaneOut out; out.begin(da.hdr_stage + at);
            u32 b = 0;
            for (u32 f = 0; f < nf; f++) {
                const u32 src = L.wsrc[f][lane];
                u32 len;
                if (keep) L.coff[cb ^ 1][f] = (u8)b;
                if (src == 0xFFu) {
                    len = L.cwln[cb][f]; const u32 o = L.coff[cb][f];
                    for (u32 j = 0; j < len; j++) { const u32 ch = L.ctext[cb][o + j]; out.put(ch); if (keep) L.ctext[cb ^ 1][b + j] = (u8)ch; }
                } else {
                    len = L.slen[f][src]; const u32 o = L.soff[f][src];
                    for (u32 j = 0; j < len; j++) { const u32 ch = L.scratch[o + j][src]; out.put(ch); if (keep) L.ctext[cb ^ 1][b + j] = (u8)ch; }
                }
                if (keep) L.cwln[cb ^ 1][f] = (u8)len;
                b += len;
                const u32 sp = f + 1 < nf ? L.csep[f] : '\n';
                out.put(sp);
                if (keep && f + 1 < nf) L.ctext[cb ^ 1][b] = (u8)sp;
                b++;
            }
            out.end();
        }
        pos += round;
        cb ^= 1u;
        __syncthreads();
    }
}
// flags: one dword per header chain, zeroed by the caller (null: every chain on the general path -- archives before version 5)
void launch_rec_decode_c(const ChainArgs& a, const DecodeArgs& da, u32* flags, hipStream_t st, u32* dtok, u32* dtoff, u32* dflags) {
    const dim3 grid((a.rgeo.nchains + 63) / 64);
    if (flags && dtok) {                                                 // symbols, texts, then the lane kernels on the chains those left (dflags)
        hipLaunchKernelGGL(k_rec_dsym, dim3((a.rgeo.nchains + 255) / 256), dim3(256), 0, st, a, da, dtok, dtoff, dflags);
        hipLaunchKernelGGL(k_rec_dtext, dim3(a.rgeo.nchains), dim3(64), 0, st, a, da, (const u32*)dtok, (const u32*)dtoff, dflags);
        hipLaunchKernelGGL(k_rec_decode_f, grid, dim3(64), 0, st, a, da, flags, (const u32*)dflags);
        hipLaunchKernelGGL(k_rec_decode_c, grid, dim3(64), 0, st, a, da, (const u32*)flags);
        return;
    }
    if (flags) hipLaunchKernelGGL(k_rec_decode_f, grid, dim3(64), 0, st, a, da, flags, (const u32*)nullptr);
    hipLaunchKernelGGL(k_rec_decode_c, grid, dim3(64), 0, st, a, da, (const u32*)flags);
}
u64 rec_dtok_bytes(u64 nrec) { return nrec * RD_TOK_PER_REC * 4; }
