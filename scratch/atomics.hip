// atomics.hip -- rate of scattered no-return u32 atomic adds (k-mer counting), and of scattered 4-byte gathers, by table size
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32; typedef uint64_t u64;
__device__ __forceinline__ u64 mix(u64 z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
template <int MODE> __global__ __launch_bounds__(256) void k(u32* tab, u64 mask, u32 iters, u32* out) {
    const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x;
    u64 x = mix(tid + 12345);
    u32 acc = 0;
    for (u32 i = 0; i < iters; i++) {
        x = mix(x + i);
        const u64 s = x & mask;
        if (MODE == 0) atomicAdd(&tab[s], 1u);            // no-return atomic
        if (MODE == 1) acc += tab[s];                     // gather
        if (MODE == 2) acc += atomicAdd(&tab[s], 1u);     // returning atomic
    }
    out[tid] = acc;
}
template <int MODE> static void run(const char* name, u32* tab, u64 entries, u32* out, int blocks, u32 iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, tab, entries - 1, 16u, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, tab, entries - 1, iters, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 256 * iters;
    printf("%-28s table %6.0f MiB  %8.2f ms  %7.2f G ops/s\n", name, entries * 4.0 / 1048576, ms, n / ms / 1e6);
}
int main() {
    u32* tab; const u64 maxb = 4ull << 30;
    if (hipMalloc(&tab, maxb) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(tab, 0, maxb);
    u32* out; (void)hipMalloc(&out, 8192ull * 256 * 4);
    for (u64 mib : {1ull, 16ull, 64ull, 256ull, 1024ull, 4096ull}) {
        const u64 entries = mib << 18;
        run<0>("scattered atomic add (no ret)", tab, entries, out, 8192, 256);
        run<1>("scattered 4-byte gather", tab, entries, out, 8192, 256);
    }
    run<2>("scattered atomic add (return)", tab, 64ull << 18, out, 8192, 256);
    return 0;
}
