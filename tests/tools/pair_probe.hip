// Measurement probe (not part of the product): does a SECOND wave on the same SIMD recover what one in-order wave cannot
// overlap?  256 workgroups (one per CU) of 256 or 512 threads; waves 0-3 run role A, waves 4-7 (if present) role B; wave w
// sits on SIMD w % 4.  Roles: M16 = chain of v_mfma_f32_16x16x32_bf16 (two accumulators), M32 = chain of
// v_mfma_f32_32x32x16_bf16, SPL = the 3-plane split of register pairs (cvt_pk / shift / and / sub: the step kernel's VALU
// mix), FMA = plain v_fma.  Output: clocks per role-iteration for every combination.
// Build:  hipcc --offload-arch=gfx950 -O3 -o tests/tools/pair_probe.out tests/tools/pair_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

enum Role { NONE = 0, M16 = 1, M32 = 2, SPL = 3, FMA = 4, MIX16 = 5, MIX32 = 6, M32N = 7, M32S = 8 };
constexpr int ITERS = 64;

__device__ __forceinline__ unsigned pk(float a, float b) { f32x2 v = {a, b}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }

// one "iteration" = 16 matrix instructions (M16: 32, same matrix time) or 16 pair-splits (176 VALU) or 176 FMAs, or both (MIX)
template <int ROLE>
__device__ __forceinline__ float run(float seed, int lane) {
    float out = 0.0f;
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + j + lane); b[j] = (__bf16)(seed * 0.5f + j); }
    float x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = seed * (1.0f + j) + lane;
    unsigned acc_u = 0;
    if constexpr (ROLE == M16 || ROLE == MIX16) {
        f32x4 c0 = {}, c1 = {};
        for (int i = 0; i < ITERS; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c1, 0, 0, 0);
                if constexpr (ROLE == MIX16) {
                    const unsigned ph = pk(x[k], x[(k + 1) & 15]);
                    const float ra = x[k] - __uint_as_float(ph << 16), rb = x[(k + 1) & 15] - __uint_as_float(ph & 0xffff0000u);
                    const unsigned pm = pk(ra, rb);
                    acc_u ^= ph ^ pm ^ pk(ra - __uint_as_float(pm << 16), rb - __uint_as_float(pm & 0xffff0000u));
                    x[k] += 1.0f;
                }
            }
        }
        out = c0[0] + c1[3];
    } else if constexpr (ROLE == M32 || ROLE == MIX32) {
        f32x16 c0 = {};
        for (int i = 0; i < ITERS; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                if constexpr (ROLE == MIX32) {
                    const unsigned ph = pk(x[k], x[(k + 1) & 15]);
                    const float ra = x[k] - __uint_as_float(ph << 16), rb = x[(k + 1) & 15] - __uint_as_float(ph & 0xffff0000u);
                    const unsigned pm = pk(ra, rb);
                    acc_u ^= ph ^ pm ^ pk(ra - __uint_as_float(pm << 16), rb - __uint_as_float(pm & 0xffff0000u));
                    x[k] += 1.0f;
                }
            }
        }
        out = c0[0] + c0[7];
    } else if constexpr (ROLE == M32N || ROLE == M32S) {
        // the same dependent chain, but the wave does not present the next matrix instruction to the issue stage while the
        // previous one occupies the pipe: M32N pads with s_nop (scalar class), M32S with s_sleep
        f32x16 c0 = {};
        for (int i = 0; i < ITERS; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ROLE == M32N) asm volatile("s_nop 15\n\ts_nop 9" ::: "memory");
                else asm volatile("s_sleep 0" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        out = c0[0] + c0[7];
    } else if constexpr (ROLE == SPL) {
        for (int i = 0; i < ITERS; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const unsigned ph = pk(x[k], x[(k + 1) & 15]);
                const float ra = x[k] - __uint_as_float(ph << 16), rb = x[(k + 1) & 15] - __uint_as_float(ph & 0xffff0000u);
                const unsigned pm = pk(ra, rb);
                acc_u ^= ph ^ pm ^ pk(ra - __uint_as_float(pm << 16), rb - __uint_as_float(pm & 0xffff0000u));
                x[k] += 1.0f;
            }
        }
    } else if constexpr (ROLE == FMA) {
        for (int i = 0; i < ITERS; ++i) {
#pragma unroll
            for (int k = 0; k < 11; ++k) {
#pragma unroll
                for (int j = 0; j < 16; ++j) x[j] = __builtin_fmaf(x[j], 0.999f, 0.001f);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) out += x[j];
    return out + (float)acc_u;
}

template <int RA, int RB>
__global__ __launch_bounds__(RB == NONE ? 256 : 512, 1) void probe(unsigned* clocks, float* sink, float seed) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float r;
    if (wave < 4) r = run<RA>(seed, lane);
    else r = run<RB>(seed, lane);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (r == 1234.5f) sink[tid] = r;
    if (lane == 0) clocks[blockIdx.x * 8 + wave] = (unsigned)(t1 - t0);
}

template <int RA, int RB>
void go(const char* name, unsigned* d_clk, float* d_sink) {
    std::vector<unsigned> h(2048);
    double a = 0, b = 0;
    for (int it = 0; it < 3; ++it) {
        hipMemset(d_clk, 0, 2048 * sizeof(unsigned));
        hipLaunchKernelGGL((probe<RA, RB>), dim3(256), dim3(RB == NONE ? 256 : 512), 0, 0, d_clk, d_sink, 1.0f + it);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d_clk, 2048 * sizeof(unsigned), hipMemcpyDeviceToHost);
        std::vector<unsigned> va, vb;
        for (int blk = 0; blk < 256; ++blk) for (int w = 0; w < 8; ++w) { if (w < 4) va.push_back(h[blk * 8 + w]); else if (RB != NONE) vb.push_back(h[blk * 8 + w]); }
        std::sort(va.begin(), va.end()); a = va[va.size() / 2];
        if (!vb.empty()) { std::sort(vb.begin(), vb.end()); b = vb[vb.size() / 2]; }
    }
    printf("{\"probe\": \"%s\", \"role_a_clocks_per_iter\": %.1f, \"role_b_clocks_per_iter\": %.1f}\n", name, a / ITERS, b / ITERS);
    fflush(stdout);
}

int main() {
    unsigned* d_clk; float* d_sink;
    hipMalloc(&d_clk, 2048 * sizeof(unsigned));
    hipMalloc(&d_sink, 2048 * sizeof(float));
    go<M32, NONE>("one wave/SIMD: 16 x mfma 32x32x16 bf16", d_clk, d_sink);
    go<M16, NONE>("one wave/SIMD: 32 x mfma 16x16x32 bf16 (two accumulators)", d_clk, d_sink);
    go<SPL, NONE>("one wave/SIMD: 16 pair-splits (3 planes)", d_clk, d_sink);
    go<FMA, NONE>("one wave/SIMD: 176 v_fma", d_clk, d_sink);
    go<MIX32, NONE>("one wave/SIMD: 16 x (mfma 32x32x16 + pair-split)", d_clk, d_sink);
    go<MIX16, NONE>("one wave/SIMD: 16 x (2 mfma 16x16x32 + pair-split)", d_clk, d_sink);
    go<M32N, NONE>("one wave/SIMD: 16 x (mfma 32x32x16 + s_nop 26)", d_clk, d_sink);
    go<M32S, NONE>("one wave/SIMD: 16 x (mfma 32x32x16 + s_sleep 0)", d_clk, d_sink);
    go<M32N, SPL>("two waves/SIMD: A = mfma 32x32x16 chain padded with s_nop, B = pair-splits", d_clk, d_sink);
    go<M32S, SPL>("two waves/SIMD: A = mfma 32x32x16 chain padded with s_sleep, B = pair-splits", d_clk, d_sink);
    go<SPL, M32>("two waves/SIMD: A = pair-splits, B = mfma 32x32x16 chain", d_clk, d_sink);
    go<M32N, FMA>("two waves/SIMD: A = mfma chain padded with s_nop, B = v_fma", d_clk, d_sink);
    go<M32, FMA>("two waves/SIMD: A = mfma chain, B = v_fma", d_clk, d_sink);
    go<M32, SPL>("two waves/SIMD: A = mfma 32x32x16 chain, B = pair-splits", d_clk, d_sink);
    go<M16, SPL>("two waves/SIMD: A = mfma 16x16x32 chain, B = pair-splits", d_clk, d_sink);
    go<SPL, SPL>("two waves/SIMD: both pair-splits", d_clk, d_sink);
    go<FMA, FMA>("two waves/SIMD: both v_fma", d_clk, d_sink);
    go<M32, M32>("two waves/SIMD: both mfma 32x32x16", d_clk, d_sink);
    go<MIX32, MIX32>("two waves/SIMD: both mfma 32x32x16 + pair-split", d_clk, d_sink);
    go<MIX16, MIX16>("two waves/SIMD: both 2 x mfma 16x16x32 + pair-split", d_clk, d_sink);
    return 0;
}
