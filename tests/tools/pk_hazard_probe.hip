// Measurement probe (not part of the product; round 5): is the packed-float32 instruction stream the compiler emitted for the
// encoding's double-angle recurrence (HISTORY.md round 5, "a code object that was not repeatable") wrong ON ITS OWN when two waves
// share a SIMD?  The 24 instructions below are copied verbatim - registers, op_sel forms, the compiler's own s_nop placement - from the
// code object that was not repeatable (step_main_wp<2> with the ray prologue, hipcc 7.2 -O3, first slot of the encoding); every lane
// runs them on fresh (s0, c0) per iteration and compares the eleven values the kernel goes on to use with the same recurrence in
// single-lane-width instructions.
//   variant 0: verbatim            variant 1: + s_nop 1 behind every packed instruction
//   lds_kb: dynamic LDS per 256-thread workgroup (160 -> one workgroup per CU, 80 -> two, 40 -> four)
//   noise 1: the first-resident workgroups (block < 256) run matrix + LDS + VALU work instead of the test
// Build:  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o pk_hazard_probe.out tests/tools/pk_hazard_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));

#define PK_BLOCK(NOP)                                                                                        \
    "v_pk_add_f32 v[4:5], v[10:11], v[10:11]\n" NOP                                                          \
    "v_mov_b32_e32 v8, v10\n"                                                                                \
    "v_pk_mul_f32 v[18:19], v[4:5], v[10:11] op_sel:[0,1] op_sel_hi:[1,0]\n" NOP                             \
    "s_nop 0\n"                                                                                              \
    "v_pk_add_f32 v[6:7], v[18:19], v[18:19]\n" NOP                                                          \
    "v_mov_b32_e32 v9, v18\n"                                                                                \
    "v_mov_b32_e32 v5, v6\n"                                                                                 \
    "v_pk_fma_f32 v[16:17], v[4:5], v[8:9], 1.0 op_sel_hi:[1,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n" NOP       \
    "s_nop 0\n"                                                                                              \
    "v_pk_mul_f32 v[20:21], v[16:17], v[6:7]\n" NOP                                                          \
    "s_nop 0\n"                                                                                              \
    "v_pk_add_f32 v[4:5], v[20:21], v[20:21]\n" NOP                                                          \
    "s_nop 0\n"                                                                                              \
    "v_pk_mul_f32 v[24:25], v[16:17], v[4:5] op_sel:[1,0] op_sel_hi:[0,1]\n" NOP                             \
    "v_pk_add_f32 v[6:7], v[24:25], v[24:25]\n" NOP                                                          \
    "v_mov_b32_e32 v21, v24\n"                                                                               \
    "v_mov_b32_e32 v5, v6\n"                                                                                 \
    "v_pk_fma_f32 v[22:23], v[4:5], v[20:21], 1.0 op_sel_hi:[1,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n" NOP     \
    "s_nop 0\n"                                                                                              \
    "v_pk_mul_f32 v[12:13], v[22:23], v[6:7]\n" NOP                                                          \
    "s_nop 0\n"                                                                                              \
    "v_pk_add_f32 v[26:27], v[12:13], v[12:13]\n" NOP                                                        \
    "s_nop 0\n"                                                                                              \
    "v_pk_mul_f32 v[14:15], v[22:23], v[26:27] op_sel_hi:[1,0]\n" NOP

// the same recurrence in single-width instructions (same registers for the values the kernel keeps)
#define SC_BLOCK                                          \
    "v_add_f32_e32 v4, v10, v10\n"                        \
    "v_mul_f32_e32 v18, v4, v11\n"                        \
    "v_add_f32_e32 v6, v18, v18\n"                        \
    "v_fma_f32 v16, -v4, v10, 1.0\n"                      \
    "v_fma_f32 v17, -v6, v18, 1.0\n"                      \
    "v_mul_f32_e32 v20, v16, v6\n"                        \
    "v_add_f32_e32 v4, v20, v20\n"                        \
    "v_mul_f32_e32 v24, v17, v4\n"                        \
    "v_add_f32_e32 v6, v24, v24\n"                        \
    "v_fma_f32 v22, -v4, v20, 1.0\n"                      \
    "v_fma_f32 v23, -v6, v24, 1.0\n"                      \
    "v_mul_f32_e32 v12, v22, v6\n"                        \
    "v_add_f32_e32 v26, v12, v12\n"                       \
    "v_mul_f32_e32 v15, v23, v26\n"

template <int V>
__device__ __forceinline__ void pk_block(float s0, float c0, float (&o)[11]) {
    // outputs: s1 c1 s2 c2 s3 c3 s4 c4 s5 (2 s4) + c0 as the block's own registers held it
#define PK_IO                                                                                                             \
    : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3]), "=v"(o[4]), "=v"(o[5]), "=v"(o[6]), "=v"(o[7]), "=v"(o[8]), "=v"(o[9]), \
      "=v"(o[10])                                                                                                         \
    : "v"(s0), "v"(c0)                                                                                                    \
    : "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", \
      "v22", "v23", "v24", "v25", "v26", "v27"
#define PK_HEAD "v_mov_b32_e32 v10, %11\n v_mov_b32_e32 v11, %12\n s_nop 4\n"
#define PK_TAIL                                                                                               \
    "s_nop 4\n v_mov_b32_e32 %0, v18\n v_mov_b32_e32 %1, v16\n v_mov_b32_e32 %2, v20\n v_mov_b32_e32 %3, v17\n"  \
    "v_mov_b32_e32 %4, v24\n v_mov_b32_e32 %5, v22\n v_mov_b32_e32 %6, v12\n v_mov_b32_e32 %7, v23\n"          \
    "v_mov_b32_e32 %8, v15\n v_mov_b32_e32 %9, v26\n v_mov_b32_e32 %10, v11\n"
    if (V == 0) asm volatile(PK_HEAD PK_BLOCK("") PK_TAIL PK_IO);
    else if (V == 1) asm volatile(PK_HEAD PK_BLOCK("s_nop 1\n") PK_TAIL PK_IO);
    else asm volatile(PK_HEAD SC_BLOCK PK_TAIL PK_IO);
}

__device__ __forceinline__ void ref_block(float s0, float c0, float (&o)[11]) {
#pragma clang fp contract(off)
    float s = s0, c = c0;
    float sv[6], cv[6];
    sv[0] = s; cv[0] = c;
#pragma unroll
    for (int f = 1; f < 6; ++f) {
        const float t = sv[f - 1] + sv[f - 1];
        sv[f] = t * cv[f - 1];
        cv[f] = __builtin_fmaf(-t, sv[f - 1], 1.0f);
    }
    o[0] = sv[1]; o[1] = cv[1]; o[2] = sv[2]; o[3] = cv[2]; o[4] = sv[3]; o[5] = cv[3]; o[6] = sv[4]; o[7] = cv[4]; o[8] = sv[5];
    o[9] = sv[4] + sv[4];
    o[10] = c0;
}

struct Rec { unsigned block, lane, iter, which, got, want; };

template <int V, bool SQRT>
__global__ __launch_bounds__(256, 2) void probe(int iters, int noise, unsigned long long* bad, Rec* recs, unsigned* nrec, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    float* L = reinterpret_cast<float*>(lds);
    if (noise && blockIdx.x < 256) {
        // first residents: other work for about as long as the test runs.  noise 1 = matrix + LDS + VALU with a run-time index into the
        // accumulator (compiled to s_set_gpr_idx_on: VGPR index mode); 2 = the same with a fixed accumulator element (no index mode);
        // 3 = matrix instructions alone; 4 = LDS + VALU alone; 5 = VGPR index mode alone (run-time index into a register array)
        f16v acc = {0};
        bf8v a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (lane + i)); b[i] = (__bf16)(0.02f * (lane - i)); }
        float x = 0.001f * tid;
        if (noise == 1) {
            for (int it = 0; it < iters * 3; ++it) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc, 0, 0, 0);
                L[tid + 256 * (it & 7)] = acc[it & 15] + x;
                x = __builtin_fmaf(x, 1.0001f, L[(tid * 5 + it) & 2047]);
            }
        } else if (noise == 2) {
            for (int it = 0; it < iters * 3; ++it) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc, 0, 0, 0);
                L[tid + 256 * (it & 7)] = acc[3] + x;
                x = __builtin_fmaf(x, 1.0001f, L[(tid * 5 + it) & 2047]);
            }
        } else if (noise == 3) {
            for (int it = 0; it < iters * 6; ++it) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc, 0, 0, 0);
            }
        } else if (noise == 4) {
            for (int it = 0; it < iters * 3; ++it) {
                L[tid + 256 * (it & 7)] = x;
                x = __builtin_fmaf(x, 1.0001f, L[(tid * 5 + it) & 2047]);
            }
        } else {
            for (int i = 0; i < 16; ++i) acc[i] = 0.5f * i + x;
            for (int it = 0; it < iters * 6; ++it) {
                x = __builtin_fmaf(x, 0.999f, acc[it & 15]);
                acc[(it + 5) & 15] = x;
            }
        }
        sink[blockIdx.x * 256 + tid] = x + acc[0] + acc[9];
        return;
    }
    unsigned h = (blockIdx.x * 256u + tid) * 2654435761u + 12345u;
    unsigned long long nb = 0;
    float keep = 0.0f;
    for (int it = 0; it < iters; ++it) {
        h = h * 1664525u + 1013904223u;
        const float s0 = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;          // [-1, 1)
        float c0;
        if (SQRT) c0 = __builtin_sqrtf(__builtin_fmaxf(0.0f, 1.0f - s0 * s0)) * ((h & 1) ? 1.0f : -1.0f);     // a transcendental-unit producer
        else c0 = (float)(int)((h * 2246822519u) >> 8) * (1.0f / 8388608.0f) - 1.0f;                       // plain VALU producers only
        float o[11], r[11];
        pk_block<V>(s0, c0, o);
        ref_block(s0, c0, r);
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const unsigned g = __float_as_uint(o[k]), w = __float_as_uint(r[k]);
            if (g != w) {
                ++nb;
                atomicAdd(&bad[1 + k], 1ull);
                const unsigned slot = atomicAdd(nrec, 1u);
                if (slot < 64) recs[slot] = Rec{blockIdx.x, (unsigned)tid, (unsigned)it, (unsigned)k, g, w};
            }
            keep += o[k];
        }
        if ((it & 15) == 0) L[tid] = keep;          // a little LDS traffic between blocks, like the kernel's image stores
    }
    if (nb) atomicAdd(&bad[0], nb);
    sink[blockIdx.x * 256 + tid] = keep + L[(tid + 1) & 255];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef void (*Kern)(int, int, unsigned long long*, Rec*, unsigned*, float*);

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    unsigned long long* bad; Rec* recs; unsigned* nrec; float* sink;
    CK(hipMalloc(&bad, 16 * sizeof(unsigned long long)));
    CK(hipMalloc(&recs, 64 * sizeof(Rec)));
    CK(hipMalloc(&nrec, sizeof(unsigned)));
    CK(hipMalloc(&sink, 4096 * 256 * sizeof(float)));
    const Kern kerns[3][2] = {{probe<0, false>, probe<0, true>}, {probe<1, false>, probe<1, true>}, {probe<2, false>, probe<2, true>}};
    const char* nnames[6] = {"none", "mfma+lds+valu+vgpr_index_mode", "mfma+lds+valu", "mfma", "lds+valu", "vgpr_index_mode"};
    const char* vnames[3] = {"packed_verbatim", "packed_nop_after_each", "single_width"};
    for (int v = 0; v < 3; ++v)
        for (int q = 0; q < 2; ++q) CK(hipFuncSetAttribute((const void*)kerns[v][q], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int cfgs[8][2] = {{160, 0}, {80, 0}, {40, 0}, {80, 1}, {40, 1}, {40, 2}, {40, 3}, {40, 4}};      // {LDS KB per workgroup, noise}
    for (int v = 0; v < 3; ++v)
        for (int q = 0; q < 2; ++q)
            for (int ci = 0; ci < 9; ++ci) {
                const int lds_kb = ci < 8 ? cfgs[ci][0] : 40, noise = ci < 8 ? cfgs[ci][1] : 5;
                const int per_cu = 160 / lds_kb;
                const int blocks = 256 * per_cu;
                CK(hipMemset(bad, 0, 16 * sizeof(unsigned long long)));
                CK(hipMemset(nrec, 0, sizeof(unsigned)));
                kerns[v][q]<<<blocks, 256, lds_kb * 1024 - 512>>>(iters, noise, bad, recs, nrec, sink);
                CK(hipDeviceSynchronize());
                unsigned long long hb[16]; unsigned hn; std::vector<Rec> hr(64);
                CK(hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost));
                CK(hipMemcpy(&hn, nrec, sizeof(hn), hipMemcpyDeviceToHost));
                CK(hipMemcpy(hr.data(), recs, 64 * sizeof(Rec), hipMemcpyDeviceToHost));
                const double tests = (double)(blocks - (noise ? 256 : 0)) * 256.0 * iters;
                unsigned quarter[4] = {0, 0, 0, 0};
                for (unsigned i = 0; i < hn && i < 64; ++i) ++quarter[(hr[i].lane & 63) >> 4];
                printf("{\"recurrence\": \"%s\", \"c0_from\": \"%s\", \"workgroups_per_cu\": %d, \"other_waves\": \"%s\", \"recurrences\": %.0f, \"wrong_values\": %llu, "
                       "\"by_output\": {\"s1\": %llu, \"c1\": %llu, \"s2\": %llu, \"c2\": %llu, \"s3\": %llu, \"c3\": %llu, \"s4\": %llu, \"c4\": %llu, \"s5\": %llu, "
                       "\"2s4\": %llu, \"c0_as_seen\": %llu}, \"lane_quarter_of_first_64\": [%u, %u, %u, %u]",
                       vnames[v], q ? "v_sqrt_f32" : "integer_hash", per_cu, nnames[noise], tests, hb[0], hb[1], hb[2], hb[3], hb[4], hb[5], hb[6], hb[7], hb[8], hb[9],
                       hb[10], hb[11], quarter[0], quarter[1], quarter[2], quarter[3]);
                printf(", \"first\": [");
                for (unsigned i = 0; i < hn && i < 4; ++i)
                    printf("%s{\"block\": %u, \"tid\": %u, \"iter\": %u, \"output\": %u, \"got\": \"%08x\", \"want\": \"%08x\"}", i ? ", " : "", hr[i].block,
                           hr[i].lane, hr[i].iter, hr[i].which, hr[i].got, hr[i].want);
                printf("]}\n");
                fflush(stdout);
            }
    return 0;
}
