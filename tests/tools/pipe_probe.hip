// Measurement probe (not part of the product): what do two waves on ONE SIMD of an MI355X share?
//   - does a wave issuing exact-fp32 matrix instructions (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32) overlap with a
//     partner wave on the same SIMD issuing plain VALU work, LDS reads, or more matrix instructions?
//   - the same for the bf16 matrix instruction (v_mfma_f32_32x32x16_bf16)
// Every kernel runs 256 workgroups (one per CU: 96 KiB of LDS each) of 256 or 512 threads; waves 0-3 take role A,
// waves 4-7 (if present) role B; wave w sits on SIMD w % 4, so A/B pairs share a SIMD.
// Build:  hipcc --offload-arch=gfx950 -O3 -o tests/tools/pipe_probe tests/tools/pipe_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum Role { NONE = 0, MFMA32 = 1, MFMA16 = 2, MFMABF = 3, VALU = 4, DSREAD = 5, MFMA32_VALU = 6, MFMA16_VALU = 7 };

template <int ROLE>
__device__ __forceinline__ float run_role(int iters, float seed, float* lds, int lane) {
    float out = 0.0f;
    if constexpr (ROLE == MFMA32) {          // 16 dependent 32x32x2 per iteration on one accumulator (like a layer chain)
        f32x16 acc = {};
        float a = seed, b = seed * 0.5f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        out = acc[0] + acc[7];
    } else if constexpr (ROLE == MFMA16) {   // 16 dependent 16x16x4 per iteration (half the matrix cycles of MFMA32)
        f32x4 acc = {};
        float a = seed, b = seed * 0.5f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
        out = acc[0] + acc[3];
    } else if constexpr (ROLE == MFMABF) {   // 16 dependent 32x32x16 bf16 per iteration
        f32x16 acc = {};
        bf16x8 a, b;
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + j); b[j] = (__bf16)(seed * 0.5f + j); }
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        out = acc[0] + acc[7];
    } else if constexpr (ROLE == VALU) {     // 256 fp32 FMAs per iteration on 8 independent chains
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = seed + j;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 32; ++k) {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], 0.999f, 0.001f);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) out += x[j];
    } else if constexpr (ROLE == DSREAD) {   // 32 ds_read_b128 per iteration (conflict-free: lane * 16 B)
        f32x4 s = {};
        const f32x4* p = reinterpret_cast<const f32x4*>(lds) + lane;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                f32x4 v = *reinterpret_cast<const volatile f32x4*>(p + 64 * (k & 7));
                s += v;
            }
        }
        out = s[0] + s[1] + s[2] + s[3];
    } else if constexpr (ROLE == MFMA32_VALU) {   // one wave: 16 x 32x32x2 AND 256 FMAs per iteration, interleaved by the compiler
        f32x16 acc = {};
        float a = seed, b = seed * 0.5f;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = seed + j;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], 0.999f, 0.001f);
                }
            }
        }
        out = acc[0];
#pragma unroll
        for (int j = 0; j < 8; ++j) out += x[j];
    } else if constexpr (ROLE == MFMA16_VALU) {   // one wave: 16 x 16x16x4 AND 128 FMAs per iteration (half of MFMA32_VALU)
        f32x4 acc = {};
        float a = seed, b = seed * 0.5f;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = seed + j;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], 0.999f, 0.001f);
            }
        }
        out = acc[0];
#pragma unroll
        for (int j = 0; j < 8; ++j) out += x[j];
    }
    return out;
}

template <int RA, int RB, int THREADS>
__global__ __launch_bounds__(THREADS) void probe(int iters, float seed, float* sink) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8 * 64 * 4; i += THREADS) lds[i] = seed;
    __syncthreads();
    float r;
    if (wave < 4) r = run_role<RA>(iters, seed, lds, lane);
    else r = run_role<RB>(iters, seed, lds, lane);
    if (r == 12345.678f) sink[blockIdx.x * THREADS + threadIdx.x] = r;     // never true: keeps the work alive
}

template <int RA, int RB, int THREADS>
static float time_probe(const char* name, int iters, float* sink) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t lds_bytes = 96 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<RA, RB, THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    std::vector<float> ms;
    for (int rep = 0; rep < 7; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<RA, RB, THREADS>), dim3(256), dim3(THREADS), lds_bytes, 0, iters, 1.0f, sink);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float t; (void)hipEventElapsedTime(&t, e0, e1);
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const float med = ms[ms.size() / 2];
    // cycles per iteration per wave at 2.4 GHz
    printf("{\"probe\": \"%s\", \"ms\": %.4f, \"us_per_iter\": %.4f, \"cycles_per_iter_at_2p4GHz\": %.1f}\n", name, med, med * 1e3 / iters,
           med * 1e-3 / iters * 2.4e9);
    return med;
}


// 16 x 32x32x2 per iteration with F independent FMAs (or DS reads if DS) after every matrix instruction; NA accumulators used
// round-robin (NA = 1: one dependent chain; NA = 2: two interleaved chains)
template <int F, int NA, bool DS>
__global__ __launch_bounds__(256) void gap_probe(int iters, float seed, float* sink) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8 * 64 * 4; i += 256) lds[i] = seed;
    __syncthreads();
    f32x16 acc[NA];
#pragma unroll
    for (int n = 0; n < NA; ++n) acc[n] = f32x16{};
    float a = seed, b = seed * 0.5f;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = seed + j;
    const f32x4* p = reinterpret_cast<const f32x4*>(lds) + lane;
    f32x4 dsum = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            acc[k % NA] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k % NA], 0, 0, 0);
            if constexpr (DS) {
#pragma unroll
                for (int j = 0; j < F; ++j) {
                    f32x4 v;
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(lane * 16)), "n"(1024 * ((j + 0) & 7)));
                    asm volatile("" : "+v"(v));
                    if (j == F - 1 && k == 15) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); dsum += v; }
                }
            } else {
#pragma unroll
                for (int j = 0; j < F; ++j) {
                    x[j & 7] = __builtin_fmaf(x[j & 7], 0.999f, 0.001f);
                    asm volatile("" : "+v"(x[j & 7]));      // pin the program order: no sinking / hoisting across the matrix instruction
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float r = dsum[0];
#pragma unroll
    for (int n = 0; n < NA; ++n) r += acc[n][0];
#pragma unroll
    for (int j = 0; j < 8; ++j) r += x[j];
    if (r == 12345.678f) sink[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int F, int NA, bool DS>
static void time_gap(int iters, float* sink) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t lds_bytes = 96 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gap_probe<F, NA, DS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    std::vector<float> ms;
    for (int rep = 0; rep < 7; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((gap_probe<F, NA, DS>), dim3(256), dim3(256), lds_bytes, 0, iters, 1.0f, sink);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float t; (void)hipEventElapsedTime(&t, e0, e1);
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const float med = ms[ms.size() / 2];
    printf("{\"probe\": \"gap: 16 x mfma32, %d %s per gap, %d accumulator(s)\", \"cycles_per_mfma_at_2p4GHz\": %.1f}\n", F, DS ? "ds_read_b128" : "fma", NA,
           med * 1e-3 / iters * 2.4e9 / 16);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    float* sink;
    (void)hipMalloc(&sink, 256 * 512 * sizeof(float));
    // single role, one wave per SIMD
    time_probe<MFMA32, NONE, 256>("A=16x mfma_32x32x2_f32 (1 wave/SIMD)", iters, sink);
    time_probe<MFMA16, NONE, 256>("A=16x mfma_16x16x4_f32 (1 wave/SIMD)", iters, sink);
    time_probe<MFMABF, NONE, 256>("A=16x mfma_32x32x16_bf16 (1 wave/SIMD)", iters, sink);
    time_probe<VALU, NONE, 256>("A=256 fma (1 wave/SIMD)", iters, sink);
    time_probe<DSREAD, NONE, 256>("A=32 ds_read_b128 (1 wave/SIMD)", iters, sink);
    time_probe<MFMA32_VALU, NONE, 256>("A=16x mfma32 + 256 fma in ONE wave", iters, sink);
    time_probe<MFMA16_VALU, NONE, 256>("A=16x mfma16 + 128 fma in ONE wave", iters, sink);
    // two waves per SIMD
    time_probe<MFMA32, VALU, 512>("A=16x mfma32 | B=256 fma (2 waves/SIMD)", iters, sink);
    time_probe<MFMA32, DSREAD, 512>("A=16x mfma32 | B=32 ds_read_b128", iters, sink);
    time_probe<MFMA32, MFMA32, 512>("A=16x mfma32 | B=16x mfma32", iters, sink);
    time_probe<MFMA16, MFMA16, 512>("A=16x mfma16 | B=16x mfma16", iters, sink);
    time_probe<MFMA16_VALU, MFMA16_VALU, 512>("A=B=16x mfma16 + 128 fma (same total work as mfma32+256fma in one wave)", iters, sink);
    time_probe<MFMA32_VALU, MFMA32_VALU, 512>("A=B=16x mfma32 + 256 fma", iters, sink);
    time_probe<MFMABF, VALU, 512>("A=16x mfma_bf16 | B=256 fma", iters, sink);
    time_probe<MFMABF, MFMABF, 512>("A=16x mfma_bf16 | B=16x mfma_bf16", iters, sink);
    time_probe<VALU, VALU, 512>("A=256 fma | B=256 fma", iters, sink);
    time_probe<VALU, DSREAD, 512>("A=256 fma | B=32 ds_read_b128", iters, sink);
    time_gap<0, 1, false>(iters, sink); time_gap<1, 1, false>(iters, sink); time_gap<2, 1, false>(iters, sink);
    time_gap<4, 1, false>(iters, sink); time_gap<8, 1, false>(iters, sink); time_gap<16, 1, false>(iters, sink);
    time_gap<0, 2, false>(iters, sink); time_gap<1, 2, false>(iters, sink); time_gap<2, 2, false>(iters, sink);
    time_gap<4, 2, false>(iters, sink); time_gap<8, 2, false>(iters, sink); time_gap<16, 2, false>(iters, sink);
    time_gap<1, 1, true>(iters, sink); time_gap<2, 1, true>(iters, sink); time_gap<4, 1, true>(iters, sink);
    time_gap<1, 2, true>(iters, sink); time_gap<2, 2, true>(iters, sink); time_gap<4, 2, true>(iters, sink);
    (void)hipFree(sink);
    return 0;
}
