// Measurement probe (not part of the product): which XCD / SE / CU does workgroup b of a launch land on?
// step_main's XCD-affine block map (an object's workgroups share an L2) assumes block b runs on XCD b % 8.
// Build:  hipcc --offload-arch=gfx950 -O3 -o xcd_probe tests/tools/xcd_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
#include <map>

__global__ __launch_bounds__(256) void probe(unsigned* out) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = 1.0f;
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    // keep the workgroup alive long enough for the whole grid to be resident at once
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 200000ull) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 240;
    unsigned* d;
    (void)hipMalloc(&d, grid * 2 * sizeof(unsigned));
    const size_t lds_bytes = 132 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(grid), dim3(256), lds_bytes, 0, d);
        (void)hipDeviceSynchronize();
        std::vector<unsigned> h(grid * 2);
        (void)hipMemcpy(h.data(), d, grid * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
        int match = 0;
        std::map<unsigned, std::set<unsigned>> cus;
        std::map<unsigned, int> per_xcd;
        for (int b = 0; b < grid; ++b) {
            const unsigned xcc = h[2 * b] & 0xF, hw = h[2 * b + 1];
            const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;     // gfx9 HW_ID fields
            match += (int)(xcc == (unsigned)(b % 8));
            cus[xcc].insert((se << 8) | (sh << 4) | cu);
            per_xcd[xcc]++;
        }
        printf("{\"grid\": %d, \"rep\": %d, \"blocks_on_xcd_b_mod_8\": %d, \"workgroups_per_xcd\": [", grid, rep, match);
        for (auto& kv : per_xcd) printf("%d%s", kv.second, kv.first == per_xcd.rbegin()->first ? "" : ", ");
        printf("], \"distinct_cus_per_xcd\": [");
        for (auto& kv : cus) printf("%zu%s", kv.second.size(), kv.first == cus.rbegin()->first ? "" : ", ");
        printf("]}\n");
    }
    (void)hipFree(d);
    return 0;
}
