// Measurement probe (not part of the product): issue cost of individual VALU instructions for ONE wave per SIMD (independent
// operands, 16 registers in rotation) - which instructions is the split of a float32 into bf16 planes best built from?
// Build:  hipcc --offload-arch=gfx950 -O3 -o tests/tools/valu_probe.out tests/tools/valu_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
constexpr int ITERS = 64, N = 64;
#define OP1(name, asmstr)                                                                                          \
    __global__ __launch_bounds__(256, 1) void k_##name(unsigned* clocks, float* sink, float seed) {               \
        float x[16]; unsigned u[16];                                                                               \
        for (int j = 0; j < 16; ++j) { x[j] = seed * (j + 1) + threadIdx.x; u[j] = threadIdx.x * 77 + j; }         \
        const unsigned mask = 0xffff0000u + (unsigned)(seed < -5.0f); const unsigned sel = 0x07060302u;              \
        unsigned msk_s = __builtin_amdgcn_readfirstlane(mask);                                                     \
        __syncthreads();                                                                                           \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                \
        for (int i = 0; i < ITERS; ++i) {                                                                          \
            _Pragma("unroll") for (int k = 0; k < N; ++k) {                                                        \
                const int a = k & 15, b = (k + 5) & 15, c = (k + 9) & 15;                                          \
                asm volatile(asmstr : "+v"(x[a]), "+v"(u[a]) : "v"(x[b]), "v"(x[c]), "v"(u[b]), "s"(msk_s), "v"(mask), "v"(sel)); \
            }                                                                                                      \
        }                                                                                                          \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                \
        float s = 0; for (int j = 0; j < 16; ++j) s += x[j] + (float)u[j];                                         \
        if (s == 1234.5f) sink[threadIdx.x] = s;                                                                   \
        if ((threadIdx.x & 63) == 0) clocks[blockIdx.x * 4 + (threadIdx.x >> 6)] = (unsigned)(t1 - t0);            \
    }
// operands: %0 x[a] (rw), %1 u[a] (rw), %2 x[b], %3 x[c], %4 u[b], %5 sgpr mask, %6 vgpr mask, %7 vgpr perm selector
OP1(fma, "v_fma_f32 %0, %2, %3, %0")
OP1(cvt_pk_bf16, "v_cvt_pk_bf16_f32 %1, %2, %3")
OP1(perm, "v_perm_b32 %1, %2, %3, %7")
OP1(and_lit, "v_and_b32 %1, 0xffff0000, %2")
OP1(and_sgpr, "v_and_b32 %1, %5, %2")
OP1(and_vgpr, "v_and_b32 %1, %6, %2")
OP1(lshl, "v_lshlrev_b32 %1, 16, %4")
OP1(sub, "v_sub_f32 %0, %2, %3")

OP1(and_or, "v_and_or_b32 %1, %2, %6, %4")
OP1(bfi, "v_bfi_b32 %1, %6, %2, %3")
OP1(alignbit, "v_alignbit_b32 %1, %2, %3, 16")
OP1(med3, "v_med3_f32 %0, %2, 0, %3")
OP1(cvt_pkrtz_f16, "v_cvt_pkrtz_f16_f32 %1, %2, %3")
OP1(cvt_f32_f16, "v_cvt_f32_f16 %0, %4")
OP1(mul, "v_mul_f32 %0, %2, %3")
OP1(lshl_or, "v_lshl_or_b32 %1, %2, 16, %4")
OP1(mov, "v_mov_b32 %1, %4")
OP1(cndmask, "v_cndmask_b32 %0, %2, %3, vcc")
OP1(sub_sdwa_like, "v_sub_f32 %0, %2, %3\n\tv_and_b32 %1, %5, %2")

template <class K> void go(const char* name, K kern, unsigned* d_clk, float* d_sink, int per) {
    std::vector<unsigned> h(1024);
    double med = 0;
    for (int it = 0; it < 3; ++it) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, d_clk, d_sink, 1.0f + it);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d_clk, 1024 * sizeof(unsigned), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        med = h[512];
    }
    printf("{\"instruction\": \"%s\", \"clocks_per_instruction\": %.2f}\n", name, med / (ITERS * N * per));
    fflush(stdout);
}
int main() {
    unsigned* d_clk; float* d_sink;
    hipMalloc(&d_clk, 1024 * sizeof(unsigned));
    hipMalloc(&d_sink, 1024 * sizeof(float));
#define GO(n) go(#n, k_##n, d_clk, d_sink, 1)
    GO(fma); GO(cvt_pk_bf16); GO(perm); GO(and_lit); GO(and_sgpr); GO(and_vgpr); GO(lshl); GO(sub); GO(and_or); GO(bfi);
    GO(alignbit); GO(med3); GO(cvt_pkrtz_f16); GO(cvt_f32_f16); GO(mul); GO(lshl_or); GO(mov); GO(cndmask);
    go("sub+and_sgpr (2)", k_sub_sdwa_like, d_clk, d_sink, 2);
    return 0;
}
