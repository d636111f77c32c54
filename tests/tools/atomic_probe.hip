// Measurement probe (not part of the product): what does it cost to accumulate the workgroups' weight-gradient blocks with
// float32 atomic adds in the XCD's L2 (one row per XCD, indexed by the hardware XCC id) instead of storing one row per
// workgroup?  Shapes of the background step: rows of 94 464 floats (377 KB), 150 / 256 / 300 workgroups.
// Build:  hipcc --offload-arch=gfx950 -O3 -o atomic_probe tests/tools/atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kRow = 94464;

__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xF;
}

// MODE 0: plain stores to the workgroup's own row; 1: L2 atomics (workgroup scope: no cache-bypass bits) to the XCD's row;
// 2: agent-scope atomics to ONE shared row; 3: read-modify-write of the workgroup's own row (a second round today)
template <int MODE>
__global__ __launch_bounds__(256) void accumulate(float* rows, int rounds, float v) {
    const unsigned xcc = xcc_id();
    float* row = rows + (size_t)(MODE == 1 ? xcc : MODE == 2 ? 0 : blockIdx.x) * kRow;
    for (int r = 0; r < rounds; ++r) {
        // the access shape of a 32x32 block store: lane = column, 16 rows per lane -> here simply 16 consecutive passes
        for (int i = threadIdx.x; i < kRow; i += 256) {
            if (MODE == 0) row[i] = v;
            else if (MODE == 1) __hip_atomic_fetch_add(row + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 2) __hip_atomic_fetch_add(row + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else row[i] = row[i] + v;
        }
    }
}

__global__ void count_xcc(unsigned* cnt) {
    if (threadIdx.x == 0) atomicAdd(cnt + xcc_id(), 1u);
}

template <int MODE>
float run(float* rows, int grid, int rounds, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(accumulate<MODE>, dim3(grid), dim3(256), 0, 0, rows, rounds, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(accumulate<MODE>, dim3(grid), dim3(256), 0, 0, rows, rounds, 1.0f);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}

int main() {
    float* rows;
    const size_t bytes = (size_t)300 * kRow * sizeof(float);
    (void)hipMalloc(&rows, bytes);
    for (int grid : {150, 256, 300}) {
        for (int rounds : {1, 2}) {
            const float t0 = run<0>(rows, grid, rounds, 20), t3 = run<3>(rows, grid, rounds, 20);
            const float t1 = run<1>(rows, grid, rounds, 20), t2 = run<2>(rows, grid, rounds, 20);
            printf("{\"workgroups\": %d, \"rounds\": %d, \"us_store_own_row\": %.1f, \"us_rmw_own_row\": %.1f, \"us_l2_atomic_xcd_row\": %.1f, \"us_agent_atomic_one_row\": %.1f}\n",
                   grid, rounds, t0, t3, t1, t2);
        }
    }
    // correctness of the XCD-row scheme: every workgroup adds 1.0 once -> row x holds the number of workgroups that ran on XCD x
    (void)hipMemset(rows, 0, bytes);
    unsigned* cnt;
    (void)hipMalloc(&cnt, 16 * sizeof(unsigned));
    (void)hipMemset(cnt, 0, 16 * sizeof(unsigned));
    hipLaunchKernelGGL(accumulate<1>, dim3(300), dim3(256), 0, 0, rows, 1, 1.0f);
    (void)hipDeviceSynchronize();
    std::vector<float> h((size_t)8 * kRow);
    (void)hipMemcpy(h.data(), rows, h.size() * sizeof(float), hipMemcpyDeviceToHost);
    printf("{\"xcd_row_values\": [");
    bool uniform = true;
    float total = 0;
    for (int x = 0; x < 8; ++x) {
        for (int i = 1; i < kRow; ++i) uniform = uniform && h[(size_t)x * kRow + i] == h[(size_t)x * kRow];
        total += h[(size_t)x * kRow];
        printf("%.0f%s", h[(size_t)x * kRow], x == 7 ? "" : ", ");
    }
    printf("], \"every_element_of_a_row_equal\": %s, \"sum_over_xcds\": %.0f, \"expected\": 300}\n", uniform ? "true" : "false", total);
    (void)hipFree(rows);
    return 0;
}
