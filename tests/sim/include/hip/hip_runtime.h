// TEST INFRASTRUCTURE: stand-in for <hip/hip_runtime.h> when the kernel headers are compiled for the CPU
// SIMT executor (tests/sim).  Only what vmap_amd/csrc/step_kernels.h uses.
#pragma once
#include <cmath>
#include <cstdint>

#include "sim_runtime.h"

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__

struct sim_tid_proxy { unsigned x, y, z; };
#define threadIdx (sim_tid_proxy{sim::tid(), 0u, 0u})
#define blockIdx (sim::g_blockIdx)
#define blockDim (sim::g_blockDim)
#define gridDim (sim::g_gridDim)

inline void __syncthreads() { sim::barrier_wait(sim::g_block->block_bar); }
inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
inline int atomicOr(int* p, int v) { int o = *p; *p = o | v; return o; }
inline int atomicMax(int* p, int v) {                       // workgroups run on several OS threads: a real atomic
    int o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v > o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}

inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
#include <cstring>
inline unsigned __float_as_uint(float x) { unsigned u; std::memcpy(&u, &x, 4); return u; }
inline float __uint_as_float(unsigned u) { float x; std::memcpy(&x, &u, 4); return x; }
