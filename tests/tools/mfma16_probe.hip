// Measurement probe (not part of the product): operand / result lane layout of v_mfma_f32_16x16x32_bf16 on gfx950, checked
// against a host matrix product under the layout the 16-point-tile design assumes:
//   A (16 x 32): lane l holds A[l & 15][8 (l >> 4) + t], t = 0..7;  B (32 x 16): lane l holds B[8 (l >> 4) + t][l & 15];
//   D (16 x 16): lane l, register r holds D[4 (l >> 4) + r][l & 15].
// Build:  hipcc --offload-arch=gfx950 -O3 -o tests/tools/mfma16_probe.out tests/tools/mfma16_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void k(const float* A, const float* B, float* D) {
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int t = 0; t < 8; ++t) { a[t] = (__bf16)A[(l & 15) * 32 + 8 * (l >> 4) + t]; b[t] = (__bf16)B[(8 * (l >> 4) + t) * 16 + (l & 15)]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
int main() {
    float hA[16 * 32], hB[32 * 16], hD[256], ref[256];
    for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5); }
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float s = 0; for (int kk = 0; kk < 32; ++kk) s += hA[m * 32 + kk] * hB[kk * 16 + n]; ref[m * 16 + n] = s; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    double err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(hD[i] - ref[i]));
    printf("{\"probe\": \"mfma_f32_16x16x32_bf16 layout\", \"max_abs_err_vs_host_product\": %g}\n", err);
    return 0;
}
