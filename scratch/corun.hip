// corun.hip -- does a lane-scattered text stream (16 bytes per lane from 64 different lines per wave instruction) slow an L2-resident
// gather kernel beside it more than the same bytes read coalesced?   (the question behind the chain kernels' mutual slowdown)
//   Q : per iteration one scattered 4-byte gather from a hot 1 MiB of a table + ALU work         (the quality chains' row lookups)
//   TS: per iteration one 16-byte load of the lane's OWN stream (lanes 16 KiB apart) + ALU work    (a chain kernel's text)
//   TC: the same bytes, but a wave's 64 lanes read 1 KiB contiguous                                (the text transposed)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32; typedef uint64_t u64;
__device__ __forceinline__ u32 alu(u32 x, int n) { for (int i = 0; i < n; i++) x = x * 1664525u + 1013904223u + (x >> 13); return x; }
__global__ __launch_bounds__(256) void kq(const u32* __restrict__ tab, u32 mask, u32 iters, int work, u32* out) {
    const u32 tid = blockIdx.x * 256 + threadIdx.x;
    u32 x = tid * 2654435761u, acc = 0;
    u32 nxt = tab[x & mask];
    for (u32 i = 0; i < iters; i++) {
        const u32 v = nxt;
        x = x * 1664525u + 1013904223u;
        nxt = tab[(x >> 7) & mask];                       // one gather in flight while the ALU work of the last runs
        acc += alu(v ^ x, work);
    }
    out[tid] = acc;
}
template <bool COAL> __global__ __launch_bounds__(256) void kt(const uint4* __restrict__ text, u64 lanes, u32 iters, int work, u32* out) {
    const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x;
    u32 acc = 0;
    const uint4* p = COAL ? text + tid : text + tid * 1024;          // own stream: 16 KiB = 1024 pieces per lane
    const u64 step = COAL ? lanes : 1;
    uint4 nxt = p[0];
    for (u32 i = 0; i < iters; i++) {
        const uint4 v = nxt;
        nxt = p[(u64)((i + 1) & 1023) * step];
        acc += alu(v.x ^ v.y ^ v.z ^ v.w, work);
    }
    out[tid] = acc;
}
int main(int argc, char** argv) {
    const int blocks = 800;                                   // 3200 waves = 204 800 lanes
    const u64 lanes = (u64)blocks * 256;
    const u32 qi = argc > 1 ? atoi(argv[1]) : 7000, ti = argc > 2 ? atoi(argv[2]) : 460;
    const int qw = argc > 3 ? atoi(argv[3]) : 20, tw = argc > 4 ? atoi(argv[4]) : 320;
    u32 *tab, *o1, *o2; uint4* text;
    (void)hipMalloc(&tab, 16 << 20); (void)hipMemset(tab, 1, 16 << 20);
    (void)hipMalloc(&text, lanes * 16384); (void)hipMemset(text, 2, lanes * 16384);
    (void)hipMalloc(&o1, lanes * 4); (void)hipMalloc(&o2, lanes * 4);
    hipStream_t s1, s2; (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
    hipEvent_t a1, b1, a2, b2; (void)hipEventCreate(&a1); (void)hipEventCreate(&b1); (void)hipEventCreate(&a2); (void)hipEventCreate(&b2);
    auto run = [&](const char* name, int q, int t) {        // t: 0 none, 1 scattered, 2 coalesced
        for (int rep = 0; rep < 2; rep++) {
            (void)hipDeviceSynchronize();
            if (q) { (void)hipEventRecord(a1, s1); hipLaunchKernelGGL(kq, dim3(blocks), dim3(256), 0, s1, tab, (1u << 18) - 1, qi, qw, o1); (void)hipEventRecord(b1, s1); }
            if (t) { (void)hipEventRecord(a2, s2);
                     if (t == 1) hipLaunchKernelGGL((kt<false>), dim3(blocks), dim3(256), 0, s2, text, lanes, ti, tw, o2);
                     else        hipLaunchKernelGGL((kt<true>), dim3(blocks), dim3(256), 0, s2, text, lanes, ti, tw, o2);
                     (void)hipEventRecord(b2, s2); }
            (void)hipDeviceSynchronize();
        }
        float m1 = 0, m2 = 0;
        if (q) (void)hipEventElapsedTime(&m1, a1, b1);
        if (t) (void)hipEventElapsedTime(&m2, a2, b2);
        printf("%-34s Q %7.3f ms   T %7.3f ms\n", name, m1, m2);
    };
    printf("Q: %u gathers per lane, %d alu each; T: %u pieces of 16 B per lane, %d alu each; %llu lanes\n", qi, qw, ti, tw, (unsigned long long)lanes);
    run("Q alone", 1, 0);
    run("T scattered alone", 0, 1);
    run("T coalesced alone", 0, 2);
    run("Q + T scattered", 1, 1);
    run("Q + T coalesced", 1, 2);
    return 0;
}
