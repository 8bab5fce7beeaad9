// coder_l.hip -- stage 3 of the default quality / base kernels as its own launch: the range coder of 64
// record blocks per wavefront, one block per LANE.
//
// The range coder (coder.hpp:66-81) is a serial chain per block: low/range depend on every earlier symbol.
// Run inside the model kernels (models_w.hip WaveCoder) it costs a whole wavefront ~19 instructions per
// symbol with every lane computing the same value -- about half of those kernels' instruction issue, which is
// what bounds them.  The models do not need the coder's state, only the coder needs
// the models' (cum, freq, tot) triples, so the split kernels park the triples in HBM (8 bytes per quality
// symbol, 4 per base) and this kernel replays them with 64 independent chains per wave: ~1 instruction per
// symbol.  The bytes are those of RcEnc / WaveCoder / the reference; only the schedule differs.
#include "kernels.h"
#include "dev_coder.h"

// lane-private byte sink that stores whole dwords (regions are 16-byte aligned: frame.hip k_block_prepare)
struct LaneSink {
    u32* p;
    u32 pos, cap, acc;
    __device__ __forceinline__ void init(u8* ptr, u32 c) { p = reinterpret_cast<u32*>(ptr); pos = 0; cap = c; acc = 0; }
    __device__ __forceinline__ void put(u32 byte) {
        acc |= byte << ((pos & 3u) * 8u);
        pos++;
        if ((pos & 3u) == 0) { if (pos <= cap) p[(pos >> 2) - 1] = acc; acc = 0; }
    }
    __device__ __forceinline__ void flush() {
        const u32 pend = pos & 3u;
        u8* b = reinterpret_cast<u8*>(p);
        for (u32 i = 0; i < pend; i++) { const u32 at = pos - pend + i; if (at < cap) b[at] = (u8)(acc >> (8 * i)); }
    }
};

struct LaneCoder {
    u64 low;
    u32 range, err;
    __device__ __forceinline__ void init() { low = 0; range = 0xFFFFFFFFu; err = 0; }
    // one symbol on every lane with `on` (coder.hpp:66-81); the renormalisation loop runs while ANY lane needs it
    __device__ __forceinline__ void encode(LaneSink& s, bool on, u32 cum, u32 freq, u32 tot) {
        if (on) {
            const u32 r = range / tot;
            low += (u64)(u32)(cum * r);
            range = r * freq;
        }
        int guard = 0;
#pragma nounroll
        while (__any(on && range < RC_TOP)) {
            if (on && range < RC_TOP) {
                if ((low ^ (low + range)) >> 56) range = (((u32)low | (RC_TOP - 1)) - (u32)low);
                s.put((u32)(low >> 56));
                range <<= 8;
                low <<= 8;
            }
            // the reference spins forever if the clamp yields range 0; every chain here must drain
            if (++guard > 12) { if (on && range < RC_TOP) { err = 1; range = 0xFFFFFFFFu; } break; }
        }
    }
    __device__ __forceinline__ void done(LaneSink& s) { for (int i = 0; i < 8; i++) { s.put((u32)(low >> 56)); low <<= 8; } }   // coder.hpp:52-61
};

// QW = true: quality triples (u64, TRIP_Q_*), false: base triples (u32, TRIP_G_*).
// Lane j codes block 64 * blockIdx + j.  Its triples are fetched by the whole wave: per chunk, 256 bytes of each
// of the 64 blocks (16 lanes x 16 bytes per block, 4 blocks per load instruction) go through registers into an
// LDS row per block, the next chunk's loads in flight while the current one is coded.  The waves of this kernel
// are few and each is a long serial chain, so they raise their issue priority over the model kernels' waves
// that share their SIMDs.
template <bool QW>
__global__ __launch_bounds__(64) void k_rc_lanes(ModelArgs a, int stream) {
    constexpr u32 CH = 256, ROW = CH + 16;              // chunk bytes per block; LDS row stride (spreads the banks)
    constexpr u32 PER = QW ? 2u : 4u;                   // triples per 16-byte piece
    constexpr u32 TPC = CH / 16 * PER;                  // triples per chunk
    __shared__ __attribute__((aligned(16))) u8 stage[2][64 * ROW];
    __builtin_amdgcn_s_setprio(3);
    const u32 lane = threadIdx.x;
    const u32 b = blockIdx.x * 64 + lane;
    const bool act = b < a.nblocks;
    BlockDesc* d = &a.blocks[act ? b : 0];
    const u32 n = act ? (QW ? a.ntrip_q[b] : a.ntrip_g[b]) : 0u;
    const u64 tb = trip_base(a.line_off, d->rec0);
    LaneSink s; s.init(a.arena + d->out_off[stream], act ? d->out_cap[stream] : 0u);
    LaneCoder rc; rc.init();
    u32 nmax = n;
#pragma unroll
    for (int dd = 32; dd > 0; dd >>= 1) { const u32 o = (u32)__shfl_xor((int)nmax, dd, 64); nmax = o > nmax ? o : nmax; }
    const u64 src = (u64)(QW ? (const void*)(a.trip_q + tb) : (const void*)(a.trip_g + tb));
    const u32 src_lo = (u32)src, src_hi = (u32)(src >> 32);
    const u32 nchunks = (nmax + TPC - 1) / TPC;
    const u32 piece = lane & 15u, sub = lane >> 4;
    uint4 regs[16];
    auto fetch = [&](u32 c) {
#pragma unroll
        for (u32 k = 0; k < 16; k++) {
            const u32 blk = 4 * k + sub;
            const u64 base = ((u64)(u32)__shfl((int)src_hi, (int)blk, 64) << 32) | (u32)__shfl((int)src_lo, (int)blk, 64);
            const u32 nb = (u32)__shfl((int)n, (int)blk, 64);
            regs[k] = (c * TPC + piece * PER < nb) ? *reinterpret_cast<const uint4*>(base + (u64)c * CH + piece * 16) : make_uint4(0, 0, 0, 0);
        }
    };
    auto park = [&](u32 buf) {
#pragma unroll
        for (u32 k = 0; k < 16; k++) *reinterpret_cast<uint4*>(&stage[buf][(4 * k + sub) * ROW + piece * 16]) = regs[k];
    };
    if (nchunks) { fetch(0); park(0); }
#pragma nounroll
    for (u32 c = 0; c < nchunks; c++) {
        if (c + 1 < nchunks) fetch(c + 1);
        __syncthreads();
        const u8* row = &stage[c & 1][lane * ROW];
        const u32 k0 = c * TPC;
        const u32 left = nmax - k0 < TPC ? nmax - k0 : TPC;           // uniform
#pragma nounroll
        for (u32 i = 0; i * PER < left; i++) {
            const uint4 v = *reinterpret_cast<const uint4*>(row + i * 16);
            const u32 k = k0 + i * PER;
            if (QW) {
                rc.encode(s, k < n,     TRIP_Q_CUM(v.x, v.y), TRIP_Q_FREQ(v.x, v.y), TRIP_Q_TOT(v.x, v.y));
                rc.encode(s, k + 1 < n, TRIP_Q_CUM(v.z, v.w), TRIP_Q_FREQ(v.z, v.w), TRIP_Q_TOT(v.z, v.w));
            } else {
                rc.encode(s, k < n,     TRIP_G_CUM(v.x), TRIP_G_FREQ(v.x), TRIP_G_TOT(v.x));
                rc.encode(s, k + 1 < n, TRIP_G_CUM(v.y), TRIP_G_FREQ(v.y), TRIP_G_TOT(v.y));
                rc.encode(s, k + 2 < n, TRIP_G_CUM(v.z), TRIP_G_FREQ(v.z), TRIP_G_TOT(v.z));
                rc.encode(s, k + 3 < n, TRIP_G_CUM(v.w), TRIP_G_FREQ(v.w), TRIP_G_TOT(v.w));
            }
        }
        __syncthreads();
        if (c + 1 < nchunks) park((c + 1) & 1);
    }
    if (act) {
        rc.done(s);
        s.flush();
        d->size[stream] = s.pos;
        if (s.pos > s.cap) atomicMax(&d->status, (u32)(-SFQ_E_OVERFLOW));
        if (rc.err) atomicMax(&d->status, (u32)(-SFQ_E_CORRUPT));
    }
}
void launch_rc_lanes(const ModelArgs& a, bool quality, hipStream_t st) {
    const dim3 grid((a.nblocks + 63) / 64);
    if (quality) hipLaunchKernelGGL((k_rc_lanes<true>), grid, dim3(64), 0, st, a, (int)SFQ_S_QLT);
    else         hipLaunchKernelGGL((k_rc_lanes<false>), grid, dim3(64), 0, st, a, (int)SFQ_S_GEN);
}
