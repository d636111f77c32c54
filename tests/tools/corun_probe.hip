// Measurement probe (not part of the product): do two kernels on two streams run side by side on an MI355X when the first
// leaves CUs idle, where do the second one's workgroups land, and what do a kernel boundary and a cross-stream event cost?
// Kernel A mimics step_main's footprint (240 workgroups of 256 threads, 132 KB of LDS: one per CU, 16 idle CUs... grid is an
// argument); kernel B is a small finalize-like kernel.  Times come from s_memrealtime (100 MHz, one clock for the whole chip).
// Build:  hipcc --offload-arch=gfx950 -O3 -o corun_probe tests/tools/corun_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <set>

struct Rec { unsigned long long t0, t1; unsigned xcc, hwid; };

__global__ __launch_bounds__(256) void spin(Rec* out, unsigned ticks) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = 1.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
    if (threadIdx.x == 0) { Rec r; r.t0 = t0; r.t1 = __builtin_amdgcn_s_memrealtime(); r.xcc = xcc & 0xF; r.hwid = hwid; out[blockIdx.x] = r; }
}

static unsigned cu_key(const Rec& r) { return (r.xcc << 16) | (((r.hwid >> 13) & 7) << 8) | (((r.hwid >> 12) & 1) << 4) | ((r.hwid >> 8) & 0xF); }

int main(int argc, char** argv) {
    const int gridA = argc > 1 ? atoi(argv[1]) : 200, gridB = argc > 2 ? atoi(argv[2]) : 112;
    const unsigned ticksA = 3000, ticksB = 300;          // 30 us, 3 us
    Rec *dA, *dB;
    (void)hipMalloc(&dA, gridA * sizeof(Rec)); (void)hipMalloc(&dB, gridB * sizeof(Rec));
    const size_t ldsA = 132 * 1024, ldsB = 2 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spin), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsA);
    hipStream_t s1, s2;
    (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
    std::vector<Rec> hA(gridA), hB(gridB);
    auto fetch = [&]() {
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(hA.data(), dA, gridA * sizeof(Rec), hipMemcpyDeviceToHost);
        (void)hipMemcpy(hB.data(), dB, gridB * sizeof(Rec), hipMemcpyDeviceToHost);
    };
    auto report = [&](const char* name) {
        unsigned long long a0 = ~0ull, a0max = 0, a1 = 0, b0 = ~0ull, b0max = 0, b1 = 0;
        std::set<unsigned> cuA, cuB, both;
        for (auto& r : hA) { a0 = std::min(a0, r.t0); a0max = std::max(a0max, r.t0); a1 = std::max(a1, r.t1); cuA.insert(cu_key(r)); }
        for (auto& r : hB) { b0 = std::min(b0, r.t0); b0max = std::max(b0max, r.t0); b1 = std::max(b1, r.t1); cuB.insert(cu_key(r)); }
        for (auto k : cuB) if (cuA.count(k)) both.insert(k);
        int b_during_a = 0;
        for (auto& r : hB) b_during_a += (r.t0 >= a0 && r.t0 < a1) ? 1 : 0;
        printf("{\"case\": \"%s\", \"A_first_start\": 0, \"A_last_start_us\": %.2f, \"A_end_us\": %.2f, \"B_first_start_us\": %.2f, \"B_last_start_us\": %.2f, "
               "\"B_end_us\": %.2f, \"B_workgroups_started_while_A_ran\": %d, \"cus_A\": %zu, \"cus_B\": %zu, \"cus_shared\": %zu}\n",
               name, (a0max - a0) / 100.0, (a1 - a0) / 100.0, ((double)b0 - (double)a0) / 100.0, ((double)b0max - (double)a0) / 100.0,
               ((double)b1 - (double)a0) / 100.0, b_during_a, cuA.size(), cuB.size(), both.size());
    };
    for (int rep = 0; rep < 2; ++rep) {
        // 1. A then B on two streams
        hipLaunchKernelGGL(spin, dim3(gridA), dim3(256), ldsA, s1, dA, ticksA);
        hipLaunchKernelGGL(spin, dim3(gridB), dim3(256), ldsB, s2, dB, ticksB);
        fetch(); report("A on stream 1, then B on stream 2");
        // 2. B then A on two streams (B's workgroups get the first pick of CUs)
        hipLaunchKernelGGL(spin, dim3(gridB), dim3(256), ldsB, s2, dB, ticksA);      // B long this time
        hipLaunchKernelGGL(spin, dim3(gridA), dim3(256), ldsA, s1, dA, ticksB);
        fetch(); report("long B on stream 2 first, then short A on stream 1");
        // 3. same stream: A then B (the kernel boundary)
        hipLaunchKernelGGL(spin, dim3(gridA), dim3(256), ldsA, s1, dA, ticksA);
        hipLaunchKernelGGL(spin, dim3(gridB), dim3(256), ldsB, s1, dB, ticksB);
        fetch(); report("A then B on ONE stream (B_first_start - A_end = the boundary)");
        // 4. two streams with an event: B waits for A's completion event
        hipEvent_t ev; (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        hipLaunchKernelGGL(spin, dim3(gridA), dim3(256), ldsA, s1, dA, ticksA);
        (void)hipEventRecord(ev, s1); (void)hipStreamWaitEvent(s2, ev, 0);
        hipLaunchKernelGGL(spin, dim3(gridB), dim3(256), ldsB, s2, dB, ticksB);
        fetch(); report("A on stream 1, event, B on stream 2 after the event");
        (void)hipEventDestroy(ev);
    }
    return 0;
}
