// host stand-in for <hip/hip_runtime.h>: lets g++ compile the lane coders of dev_chain.h for scratch/host_chain_test.cpp
#pragma once
#include <stdint.h>
#include <string.h>
#define __device__
#define __host__
#define __global__
#define __shared__ static
#define __forceinline__ inline
#define __launch_bounds__(x)
struct uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { uint4 v = {x, y, z, w}; return v; }
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
static inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31)); }
static inline int __any(int x) { return x; }
