// Measurement probe (not part of the product): what does straight-line code that is executed ONCE cost per instruction,
// against the same instructions in a loop that stays in the instruction cache?  4 waves per CU on 200 CUs (the step
// kernel's geometry) execute BODY x REPS VALU instructions: as a loop of REPS iterations over BODY instructions, or fully
// unrolled (BODY * REPS * 8 bytes of code).
// Build:  hipcc --offload-arch=gfx950 -O3 -o tests/tools/icache_probe.out tests/tools/icache_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int BODY, int REPS, bool UNROLL>
__global__ __launch_bounds__(256, 1) void k(unsigned* clocks, float* sink, float seed) {
    float x[16];
    for (int j = 0; j < 16; ++j) x[j] = seed * (j + 1) + threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (UNROLL) {
#pragma unroll
        for (int i = 0; i < REPS; ++i) {
#pragma unroll
            for (int q = 0; q < BODY; ++q) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[q & 15]) : "v"(x[(q + 5) & 15]), "v"(x[(q + 9) & 15]));
        }
    } else {
#pragma unroll 1
        for (int i = 0; i < REPS; ++i) {
#pragma unroll
            for (int q = 0; q < BODY; ++q) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[q & 15]) : "v"(x[(q + 5) & 15]), "v"(x[(q + 9) & 15]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int j = 0; j < 16; ++j) s += x[j];
    if (s == 1234.5f) sink[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clocks[blockIdx.x * 4 + (threadIdx.x >> 6)] = (unsigned)(t1 - t0);
}

template <int BODY, int REPS, bool UNROLL>
void go(unsigned* d_clk, float* d_sink) {
    std::vector<unsigned> h(800);
    double med = 0, mx = 0;
    for (int it = 0; it < 3; ++it) {
        hipLaunchKernelGGL((k<BODY, REPS, UNROLL>), dim3(200), dim3(256), 0, 0, d_clk, d_sink, 1.0f + it);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d_clk, 800 * sizeof(unsigned), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        med = h[400]; mx = h[799];
    }
    printf("{\"instructions\": %d, \"code_bytes\": %d, \"form\": \"%s\", \"clocks_per_instruction_median\": %.2f, \"max\": %.2f}\n",
           BODY * REPS, (UNROLL ? BODY * REPS : BODY) * 8, UNROLL ? "straight line, executed once" : "loop", med / (BODY * REPS), mx / (BODY * REPS));
    fflush(stdout);
}
int main() {
    unsigned* d_clk; float* d_sink;
    hipMalloc(&d_clk, 800 * sizeof(unsigned));
    hipMalloc(&d_sink, 1024 * sizeof(float));
    go<128, 64, false>(d_clk, d_sink);
    go<128, 8, true>(d_clk, d_sink);
    go<128, 16, true>(d_clk, d_sink);
    go<128, 32, true>(d_clk, d_sink);
    go<128, 64, true>(d_clk, d_sink);
    go<1024, 8, false>(d_clk, d_sink);
    go<4096, 2, false>(d_clk, d_sink);
    go<8192, 2, false>(d_clk, d_sink);
    return 0;
}
