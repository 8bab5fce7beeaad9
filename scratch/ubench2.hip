// ubench2.hip -- issue-rate microbenchmark for gfx950, exact instructions (inline asm, nothing the compiler can fold),
// timed by WALL time of long launches and by in-kernel clocks (s_memtime = shader cycles, s_memrealtime = 100 MHz).
// Settles what one SIMD can issue per cycle for the instruction kinds the entropy-coding kernels
// are made of, and which clock the chip holds while it does so.
//   cycles per wave-instruction per SIMD = wall_time x in-kernel clock / (instructions per wave x waves per SIMD)
// Build: hipcc -O3 --offload-arch=gfx950 -o scratch/ubench2 scratch/ubench2.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32; typedef uint64_t u64;
#define REP 8192
#define NCH 8

struct Stamp { u64 c0, c1, r0, r1; };

// one instruction on each of 8 independent registers per iteration
#define ASM8(STR) \
    asm volatile(STR : "+v"(a[0]) : "v"(m), "s"(sm)); asm volatile(STR : "+v"(a[1]) : "v"(m), "s"(sm)); \
    asm volatile(STR : "+v"(a[2]) : "v"(m), "s"(sm)); asm volatile(STR : "+v"(a[3]) : "v"(m), "s"(sm)); \
    asm volatile(STR : "+v"(a[4]) : "v"(m), "s"(sm)); asm volatile(STR : "+v"(a[5]) : "v"(m), "s"(sm)); \
    asm volatile(STR : "+v"(a[6]) : "v"(m), "s"(sm)); asm volatile(STR : "+v"(a[7]) : "v"(m), "s"(sm));
#define SASM8(STR) \
    asm volatile(STR : "+s"(sa[0]) : "s"(sm)); asm volatile(STR : "+s"(sa[1]) : "s"(sm)); \
    asm volatile(STR : "+s"(sa[2]) : "s"(sm)); asm volatile(STR : "+s"(sa[3]) : "s"(sm)); \
    asm volatile(STR : "+s"(sa[4]) : "s"(sm)); asm volatile(STR : "+s"(sa[5]) : "s"(sm)); \
    asm volatile(STR : "+s"(sa[6]) : "s"(sm)); asm volatile(STR : "+s"(sa[7]) : "s"(sm));

template <int OP> __global__ __launch_bounds__(64) void k(u32* out, Stamp* st, u32 seed) {
    __shared__ u32 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (i * 97u + seed) & 1023u;
    __syncthreads();
    u32 a[NCH]; u32 sa[NCH];
    for (int i = 0; i < NCH; i++) { a[i] = (seed + threadIdx.x * 7 + i) & 1023u; sa[i] = __builtin_amdgcn_readfirstlane(seed + i); }
    const u32 m = seed | 3;
    const u32 sm = __builtin_amdgcn_readfirstlane(seed | 3);
    u64 w = ((u64)seed << 20) | threadIdx.x;
    const u64 c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < REP; r++) {
        if (OP == 0)  { ASM8("v_add_u32 %0, %0, %1") }
        if (OP == 1)  { ASM8("v_mul_lo_u32 %0, %0, %1") }
        if (OP == 2)  { ASM8("v_mul_hi_u32 %0, %0, %1") }
        if (OP == 3)  { ASM8("v_mul_u32_u24 %0, %0, %1") }
        if (OP == 4)  { ASM8("v_and_b32 %0, %0, %1") }
        if (OP == 5)  { ASM8("v_cndmask_b32 %0, %0, %1, vcc") }
        if (OP == 6)  { ASM8("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf") }
        if (OP == 7)  { ASM8("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") }
        if (OP == 8)  { ASM8("v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf") }
        if (OP == 9)  { ASM8("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf") }
        if (OP == 10) { ASM8("v_mov_b32_dpp %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf") }
        if (OP == 11) { ASM8("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)") }
        if (OP == 12) { ASM8("ds_swizzle_b32 %0, %0 offset:swizzle(SWAP,16)\n\ts_waitcnt lgkmcnt(0)") }
        if (OP == 13) { ASM8("v_permlane32_swap_b32 %0, %0") }
        if (OP == 14) { ASM8("v_permlane16_swap_b32 %0, %0") }
        if (OP == 15) { SASM8("s_add_u32 %0, %0, %1") }
        if (OP == 16) { SASM8("s_mul_i32 %0, %0, %1") }
        if (OP == 17) { SASM8("s_lshl_b32 %0, %0, 1") }
        if (OP == 18) { ASM8("v_readlane_b32 s20, %0, 3\n\tv_add_u32 %0, s20, %0") }        // VALU -> SGPR -> VALU
        if (OP == 19) { ASM8("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc") }  // compare + select
        if (OP == 20) { ASM8("v_cmp_lt_u32 s[20:21], %0, %1\n\ts_ff1_i32_b64 s22, s[20:21]\n\tv_add_u32 %0, s22, %0") }   // ballot + ffs + use
        if (OP == 21) { ASM8("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %0, 0xffc, %0") }      // dependent LDS read (latency)
        if (OP == 22) { for (int i = 0; i < NCH; i++) a[i] = lds[(a[i] + r) & 1023u]; }                    // 8 independent LDS reads in flight
        if (OP == 23) { for (int i = 0; i < NCH; i++) { w = w + (u64)a[i] * m; asm volatile("" : "+v"(w)); } }   // v_mad_u64_u32 chain
        if (OP == 24) { for (int i = 0; i < NCH; i++) { w = w << (m & 7); asm volatile("" : "+v"(w)); } }        // v_lshlrev_b64
        if (OP == 25) { ASM8("v_lshlrev_b32 %0, 1, %0") }
        if (OP == 26) { ASM8("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc") }
        if (OP == 27) { ASM8("v_ffbh_u32 %0, %0") }
    }
    const u64 c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    u32 s = (u32)w + (u32)(w >> 32); for (int i = 0; i < NCH; i++) s += a[i] + sa[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) { Stamp x; x.c0 = c0; x.c1 = c1; x.r0 = r0; x.r1 = r1; st[blockIdx.x] = x; }
}

template <typename F> static void run(const char* name, F launch, int wps, u32* d, Stamp* dst, int ninst_per_iter) {
    const int grid = 256 * 4 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(grid, d, dst); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); launch(grid, d, dst); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h(grid);
    (void)hipMemcpy(h.data(), dst, sizeof(Stamp) * grid, hipMemcpyDeviceToHost);
    std::vector<double> cyc(grid), clk(grid);
    for (int i = 0; i < grid; i++) { cyc[i] = (double)(h[i].c1 - h[i].c0); clk[i] = cyc[i] / (double)(h[i].r1 - h[i].r0) * 100.0; }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double inst = (double)REP * NCH * ninst_per_iter;                // per wave
    const double mhz = clk[grid / 2];
    const double per_simd_wall = ms * 1e-3 * mhz * 1e6 / (inst * wps);
    printf("%-34s waves/SIMD %d  wall %8.3f ms  clock %5.0f MHz  cycles/inst: one wave's view %6.2f  per SIMD (wall) %5.2f\n",
           name, wps, ms, mhz, cyc[grid / 2] / inst, per_simd_wall);
}
int main() {
    u32* d; Stamp* dst;
    (void)hipMalloc(&d, 256 * 4 * 8 * 64 * 4); (void)hipMalloc(&dst, sizeof(Stamp) * 256 * 4 * 8);
    int clk = 0; (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("hipDeviceAttributeClockRate %d kHz; REP %d x %d instructions per wave\n", clk, REP, NCH);
#define V(OP, NAME, NI) for (int w : {1, 2, 4, 8}) run(NAME, [](int g, u32* p, Stamp* s) { hipLaunchKernelGGL(k<OP>, dim3(g), dim3(64), 0, 0, p, s, 12345u); }, w, d, dst, NI)
    V(0, "v_add_u32", 1); V(1, "v_mul_lo_u32", 1); V(2, "v_mul_hi_u32", 1); V(3, "v_mul_u32_u24", 1); V(4, "v_and_b32", 1);
    V(5, "v_cndmask_b32", 1); V(25, "v_lshlrev_b32", 1); V(27, "v_ffbh_u32", 1); V(26, "v_add_co + v_addc_co (64-bit add)", 2);
    V(23, "v_mad_u64_u32 (dependent)", 1); V(24, "v_lshlrev_b64 (dependent)", 1);
    V(6, "v_add_u32_dpp row_shr:1", 1); V(7, "v_mov_dpp quad_perm", 1); V(8, "v_mov_dpp row_ror:8", 1);
    V(9, "v_mov_dpp wave_shr:1", 1); V(10, "v_mov_dpp row_bcast:15", 1);
    V(13, "v_permlane32_swap_b32", 1); V(14, "v_permlane16_swap_b32", 1);
    V(11, "ds_bpermute_b32 + wait", 1); V(12, "ds_swizzle_b32 + wait", 1);
    V(15, "s_add_u32", 1); V(16, "s_mul_i32", 1); V(17, "s_lshl_b32", 1);
    V(18, "v_readlane -> v_add (sgpr)", 2); V(19, "v_cmp + v_cndmask (vcc)", 2); V(20, "v_cmp sgpr + s_ff1 + v_add", 3);
    V(21, "ds_read_b32 dependent + and", 2); V(22, "ds_read_b32 x8 in flight", 3);
    return 0;
}
