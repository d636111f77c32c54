// Stand-alone reproducer (no other file of this repository needed) for the MI355X behaviour vmap_amd/csrc/gfx950_errata.py works around:
//   v_pk_mul_f32 d, a, b op_sel:[0,1]   (low result = a.lo * b.HI)
// returns a wrong LOW result in lanes 48..63 (b's high register read as zero) about once per 1e3 executions while ANOTHER wave of the same
// SIMD runs a matrix-instruction -> VALU -> LDS mix; the same product with the sources swapped (op_sel:[1,0]) never does, and neither
// form fails with the SIMD to itself.  Build + run:
//   hipcc --offload-arch=gfx950 -O3 -o pk_erratum_min tests/tools/pk_erratum_min.hip && ./pk_erratum_min
// Prints four lines {aggressor off / on} x {op_sel:[0,1] / sources swapped, op_sel:[1,0]}: executions, wrong low results, wrong results
// per 16-lane quarter.  Expected on MI355X (committed: profiles/round6_pk_erratum_min.txt): zeros except "aggressor on, op_sel:[0,1]".
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256, 2) void probe(int swapped, int aggressor, int iters, unsigned long long* bad, float* sink) {
    __shared__ float L[2048];
    const int tid = threadIdx.x, lane = tid & 63;
    if (blockIdx.x < 256) {                                  // first-resident workgroups: the aggressor (or nothing)
        float x = 0.001f * tid;
        if (aggressor) {
            f16v acc = {0};
            bf8v a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (lane + i)); b[i] = (__bf16)(0.02f * (lane - i)); }
            for (int it = 0; it < iters * 3; ++it) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc, 0, 0, 0);
                L[(tid + 256 * (it & 7)) & 2047] = acc[3] + x;
                x = __builtin_fmaf(x, 1.0001f, L[(tid * 5 + it) & 2047]);
            }
            x += acc[0] + acc[7];
        }
        sink[blockIdx.x * 256 + tid] = x;
        return;
    }
    unsigned h = (blockIdx.x * 256u + tid) * 2654435761u + 12345u;      // the victims: one packed product per iteration, checked against v_mul_f32
    unsigned long long wrong = 0, wq[4] = {0, 0, 0, 0};
    float keep = 0.0f;
    for (int it = 0; it < iters; ++it) {
        float v[4];
        for (int k = 0; k < 4; ++k) { h = h * 1664525u + 1013904223u; v[k] = (float)(int)(h >> 8) * (1.0f / 4194304.0f) - 2.0f; }
        float dlo, dhi, elo;
        if (!swapped)
            asm volatile("v_mov_b32 v4, %2\n v_mov_b32 v5, %3\n v_mov_b32 v10, %4\n v_mov_b32 v11, %5\n s_nop 4\n"
                         "v_pk_mul_f32 v[18:19], v[4:5], v[10:11] op_sel:[0,1] op_sel_hi:[1,0]\n s_nop 4\n v_mov_b32 %0, v18\n v_mov_b32 %1, v19\n"
                         : "=v"(dlo), "=v"(dhi) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "v4", "v5", "v10", "v11", "v18", "v19");
        else
            asm volatile("v_mov_b32 v4, %2\n v_mov_b32 v5, %3\n v_mov_b32 v10, %4\n v_mov_b32 v11, %5\n s_nop 4\n"
                         "v_pk_mul_f32 v[18:19], v[10:11], v[4:5] op_sel:[1,0] op_sel_hi:[0,1]\n s_nop 4\n v_mov_b32 %0, v18\n v_mov_b32 %1, v19\n"
                         : "=v"(dlo), "=v"(dhi) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "v4", "v5", "v10", "v11", "v18", "v19");
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(elo) : "v"(v[0]), "v"(v[3]));     // a.lo * b.hi, single width
        const bool w = __float_as_uint(dlo) != __float_as_uint(elo);
        wrong += w;
        wq[lane >> 4] += w;
        keep += dlo + dhi;
    }
    if (wrong) atomicAdd(&bad[0], wrong);
    for (int q = 0; q < 4; ++q) if (wq[q]) atomicAdd(&bad[1 + q], wq[q]);
    sink[blockIdx.x * 256 + tid] = keep;
}

int main() {
    const int iters = 4000;
    unsigned long long *bad, hb[5];
    float* sink;
    if (hipMalloc(&bad, sizeof(hb)) != hipSuccess || hipMalloc(&sink, 1024 * 256 * sizeof(float)) != hipSuccess) return 1;
    for (int aggressor = 0; aggressor < 2; ++aggressor)
        for (int swapped = 0; swapped < 2; ++swapped) {
            (void)hipMemset(bad, 0, sizeof(hb));
            probe<<<1024, 256>>>(swapped, aggressor, iters, bad, sink);
            if (hipDeviceSynchronize() != hipSuccess) return 1;
            (void)hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost);
            printf("aggressor %-3s  %-52s executions %.0f  wrong_low %llu  by lanes [0-15, 16-31, 32-47, 48-63] = [%llu, %llu, %llu, %llu]\n", aggressor ? "on" : "off",
                   swapped ? "v_pk_mul_f32 d, b, a op_sel:[1,0] op_sel_hi:[0,1]" : "v_pk_mul_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]", 768.0 * 256.0 * iters, hb[0], hb[1], hb[2], hb[3], hb[4]);
        }
    return 0;
}
