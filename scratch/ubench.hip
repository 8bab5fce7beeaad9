// instruction-throughput microbenchmark: how many cycles does a wave64 instruction hold its SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32; typedef uint64_t u64;
#define REP 4096
template <int OP> __global__ __launch_bounds__(64) void k(u32* out, u32 seed) {
    u32 a[8]; u64 w[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x * 7 + i; w[i] = ((u64)a[i] << 20) | 5; }
    const u32 m = seed | 3;
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = a[i] + m;                                   // v_add_u32
            if (OP == 1) a[i] = a[i] * m;                                   // v_mul_lo_u32
            if (OP == 2) a[i] = __umulhi(a[i], m);                          // v_mul_hi_u32
            if (OP == 3) w[i] = w[i] + (u64)a[i] * m;                       // v_mad_u64_u32
            if (OP == 4) w[i] = w[i] << (m & 7);                            // v_lshlrev_b64
            if (OP == 5) a[i] = __umul24(a[i], m);          // v_mul_u32_u24
            if (OP == 6) a[i] = (u32)__builtin_amdgcn_readlane((int)a[i], 3) + a[i];   // v_readlane + add
            if (OP == 7) a[i] = (u32)__builtin_amdgcn_ds_bpermute((int)(threadIdx.x * 4 ^ 4), (int)a[i]);
            if (OP == 8) a[i] = (u32)__builtin_amdgcn_update_dpp(0, (int)a[i], 0x111, 0xf, 0xf, false) + 1;   // DPP row_shr1
            if (OP == 9) w[i] = w[i] + a[i];                                // 64-bit add
        }
    }
    u32 s = 0; for (int i = 0; i < 8; i++) s += a[i] + (u32)w[i] + (u32)(w[i] >> 32);
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int OP> __global__ __launch_bounds__(64) void ks(u32* out, u32 seed) {   // scalar flavours (uniform values)
    u32 a[8];
    for (int i = 0; i < 8; i++) a[i] = __builtin_amdgcn_readfirstlane(seed + i);
    const u32 m = __builtin_amdgcn_readfirstlane(seed | 3);
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = a[i] + m;
            if (OP == 1) a[i] = a[i] * m;
            if (OP == 2) a[i] = __umulhi(a[i], m);
            asm volatile("" : "+s"(a[i]));
        }
    }
    u32 s = 0; for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
// mixed: even workgroups run a scalar chain, odd ones a vector chain -- do SALU and VALU of different waves overlap?
__global__ __launch_bounds__(64) void kmix(u32* out, u32 seed, int mode) {
    const bool scalar = mode == 0 || (mode == 2 && (blockIdx.x & 1) == 0);
    u32 res = 0;
    if (scalar) {
        u32 a[8];
        for (int i = 0; i < 8; i++) a[i] = __builtin_amdgcn_readfirstlane(seed + i);
        const u32 m = __builtin_amdgcn_readfirstlane(seed | 3);
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) { a[i] = __umulhi(a[i], m) + 7; asm volatile("" : "+s"(a[i])); }
        }
        for (int i = 0; i < 8; i++) res += a[i];
    } else {
        u32 a[8];
        for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 7 + i;
        const u32 m = seed | 3;
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = __umulhi(a[i], m) + 7;
        }
        for (int i = 0; i < 8; i++) res += a[i];
    }
    out[blockIdx.x * 64 + threadIdx.x] = res;
}
static void run_mix(const char* name, int mode, u32* d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256 * 4 * 8;
    hipLaunchKernelGGL(kmix, dim3(grid), dim3(64), 0, 0, d, 12345u, mode); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL(kmix, dim3(grid), dim3(64), 0, 0, d, 12345u, mode); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.3f ms\n", name, ms);
}
template <typename F> static void run(const char* name, F f, u32* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 4 * 8;            // 8 waves per SIMD
    f(grid, d); hipDeviceSynchronize();
    hipEventRecord(e0); f(grid, d); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)grid * REP * 8;          // wave-instructions of the measured kind
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("%-22s %8.3f ms  %.2f cycles per wave-instruction per SIMD (clock %d kHz)\n", name, ms, ms * 1e-3 * clk * 1e3 / (inst / 1024.0), clk);
}
int main() {
    u32* d; hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
#define V(OP, NAME) run(NAME, [](int g, u32* p) { hipLaunchKernelGGL(k<OP>, dim3(g), dim3(64), 0, 0, p, 12345u); }, d)
#define S(OP, NAME) run(NAME, [](int g, u32* p) { hipLaunchKernelGGL(ks<OP>, dim3(g), dim3(64), 0, 0, p, 12345u); }, d)
    V(0, "v_add_u32"); V(1, "v_mul_lo_u32"); V(2, "v_mul_hi_u32"); V(3, "v_mad_u64_u32"); V(4, "v_lshlrev_b64"); V(5, "v_mul_u32_u24");
    V(6, "v_readlane+add"); V(7, "ds_bpermute"); V(8, "dpp mov+add"); V(9, "add u64+u32");
    S(0, "s_add_u32"); S(1, "s_mul_i32"); S(2, "s_mul_hi_u32");
    run_mix("all waves scalar (mul_hi + add)", 0, d);
    run_mix("all waves vector (mul_hi + add)", 1, d);
    run_mix("half scalar, half vector (same total work)", 2, d);
    return 0;
}
