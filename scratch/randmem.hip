// random 64-byte-sector read-modify-write throughput of HBM: what the adaptive tables can ask for at most
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32; typedef uint64_t u64;
__device__ __forceinline__ u64 mix(u64 z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
// MODE 0: independent addresses (throughput); MODE 1: next address depends on the loaded value (one chain per lane)
template <int MODE, int WRITE> __global__ __launch_bounds__(64, 8) void k(u32* tab, u64 nsect, u32 iters, u32* out) {
    const u64 tid = (u64)blockIdx.x * 64 + threadIdx.x;
    u64 x = mix(tid + 12345);
    u32 acc = 0;
    for (u32 i = 0; i < iters; i++) {
        const u64 s = x % nsect;
        u32* p = tab + s * 16;
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        acc += v.x;
        if (WRITE) p[1] = v.y + 1;
        x = MODE ? mix(x + v.x) : mix(x + i);
    }
    out[tid] = acc;
}
template <int MODE, int WRITE> static void run(const char* name, u32* tab, u64 nsect, u32* out, int waves, u32 iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, WRITE>), dim3(waves), dim3(64), 0, 0, tab, nsect, 64u, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, WRITE>), dim3(waves), dim3(64), 0, 0, tab, nsect, iters, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double acc = (double)waves * 64 * iters;
    printf("%-34s waves %6d  %8.2f ms  %7.2f G accesses/s  = %6.2f TB/s of 64-byte sectors%s\n", name, waves, ms, acc / ms / 1e6,
           acc * 64 * (WRITE ? 2 : 1) / ms / 1e9, WRITE ? " (read + write-back)" : " (read)");
}
int main() {
    const u64 bytes = 128ull << 30;                       // 128 GiB of table, far beyond L2 + Infinity Cache
    u32* tab; if (hipMalloc(&tab, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(tab, 1, bytes);
    u32* out; (void)hipMalloc(&out, 65536ull * 64 * 4);
    const u64 nsect = bytes / 64;
    for (int waves : {8192, 16384}) {
        run<0, 0>("independent reads", tab, nsect, out, waves, 2048);
        run<0, 1>("independent read-modify-write", tab, nsect, out, waves, 2048);
        run<1, 0>("dependent reads (chain per lane)", tab, nsect, out, waves, 512);
        run<1, 1>("dependent read-modify-write", tab, nsect, out, waves, 512);
    }
    return 0;
}
