// Measurement probe (not part of the product; round 5), second half of pk_hazard_probe.hip: WHICH packed-float32 instruction forms return
// wrong lanes next to another wave's work on the same SIMD, and what the other wave has to be doing.  Every op_sel / op_sel_hi combination of
// v_pk_mul_f32, v_pk_add_f32 and v_pk_fma_f32 (third source at its default selection) on fixed registers, both result halves compared with
// single-width instructions; the first-resident workgroups (block < 256) run ONE kind of work: matrix instructions, LDS traffic, VALU
// arithmetic or nothing (they exit).  Four 256-thread workgroups per CU (two waves of each kind per SIMD).
// Build:  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o pk_form_probe.out tests/tools/pk_form_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));

// two register sets (different VGPR banks; the second writes its result over its first source, as compiled code often does)
#define HEAD0 "v_mov_b32_e32 v4, %2\n v_mov_b32_e32 v5, %3\n v_mov_b32_e32 v10, %4\n v_mov_b32_e32 v11, %5\n v_mov_b32_e32 v8, %6\n v_mov_b32_e32 v9, %7\n s_nop 4\n"
#define TAIL0 "s_nop 4\n v_mov_b32_e32 %0, v18\n v_mov_b32_e32 %1, v19\n"
#define REGS0(OP, TAILSEL) OP " v[18:19], v[4:5], v[10:11]" TAILSEL
#define REGS0F(OP, TAILSEL) OP " v[18:19], v[4:5], v[10:11], v[8:9]" TAILSEL
#define HEAD1 "v_mov_b32_e32 v6, %2\n v_mov_b32_e32 v7, %3\n v_mov_b32_e32 v12, %4\n v_mov_b32_e32 v13, %5\n v_mov_b32_e32 v16, %6\n v_mov_b32_e32 v17, %7\n s_nop 4\n"
#define TAIL1 "s_nop 4\n v_mov_b32_e32 %0, v6\n v_mov_b32_e32 %1, v7\n"
#define REGS1(OP, TAILSEL) OP " v[6:7], v[6:7], v[12:13]" TAILSEL
#define REGS1F(OP, TAILSEL) OP " v[6:7], v[6:7], v[12:13], v[16:17]" TAILSEL
#define IO : "=v"(dlo), "=v"(dhi) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1) \
    : "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v16", "v17", "v18", "v19"
#define SEL2(x, y, z, w) " op_sel:[" #x "," #y "] op_sel_hi:[" #z "," #w "]\n"
#define SEL3(x, y, z, w) " op_sel:[" #x "," #y ",0] op_sel_hi:[" #z "," #w ",1]\n"
#define SELC(k, m) " op_sel:[0,0," #k "] op_sel_hi:[1,1," #m "]\n"
#define ONE(base, i, H, R, T, OP, S, x, y, z, w) case base + i: asm volatile(H R(OP, S(x, y, z, w)) T IO); break;
#define ALL16(base, H, R, T, OP, S)                                                                                              \
    ONE(base, 0, H, R, T, OP, S, 0, 0, 0, 0) ONE(base, 1, H, R, T, OP, S, 0, 0, 0, 1) ONE(base, 2, H, R, T, OP, S, 0, 0, 1, 0)     \
    ONE(base, 3, H, R, T, OP, S, 0, 0, 1, 1) ONE(base, 4, H, R, T, OP, S, 0, 1, 0, 0) ONE(base, 5, H, R, T, OP, S, 0, 1, 0, 1)     \
    ONE(base, 6, H, R, T, OP, S, 0, 1, 1, 0) ONE(base, 7, H, R, T, OP, S, 0, 1, 1, 1) ONE(base, 8, H, R, T, OP, S, 1, 0, 0, 0)     \
    ONE(base, 9, H, R, T, OP, S, 1, 0, 0, 1) ONE(base, 10, H, R, T, OP, S, 1, 0, 1, 0) ONE(base, 11, H, R, T, OP, S, 1, 0, 1, 1)   \
    ONE(base, 12, H, R, T, OP, S, 1, 1, 0, 0) ONE(base, 13, H, R, T, OP, S, 1, 1, 0, 1) ONE(base, 14, H, R, T, OP, S, 1, 1, 1, 0)  \
    ONE(base, 15, H, R, T, OP, S, 1, 1, 1, 1)
#define SRC2(base, H, R, T)                                                                                                      \
    case base + 0: asm volatile(H R("v_pk_fma_f32", SELC(0, 0)) T IO); break;                                                    \
    case base + 1: asm volatile(H R("v_pk_fma_f32", SELC(0, 1)) T IO); break;                                                    \
    case base + 2: asm volatile(H R("v_pk_fma_f32", SELC(1, 0)) T IO); break;                                                    \
    case base + 3: asm volatile(H R("v_pk_fma_f32", SELC(1, 1)) T IO); break;
constexpr int kFormsPerSet = 52;

__global__ __launch_bounds__(256, 2) void probe(int form, int noise, int iters, unsigned long long* bad, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    float* L = reinterpret_cast<float*>(lds);
    if (blockIdx.x < 256) {
        float x = 0.001f * tid;
        if (noise == 1) {                                   // matrix instructions, their results through VALU into LDS, LDS reads (a kernel's usual mix)
            f16v acc = {0};
            bf8v a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (lane + i)); b[i] = (__bf16)(0.02f * (lane - i)); }
            for (int it = 0; it < iters * 3; ++it) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc, 0, 0, 0);
                L[tid + 256 * (it & 7)] = acc[3] + x;
                x = __builtin_fmaf(x, 1.0001f, L[(tid * 5 + it) & 2047]);
            }
            x += acc[0] + acc[7];
        } else if (noise == 2) {                            // LDS traffic only
            for (int it = 0; it < iters * 4; ++it) {
                L[tid + 256 * (it & 7)] = x;
                x += L[(tid * 5 + it) & 2047];
            }
        } else if (noise == 3) {                            // VALU arithmetic only
            float y = 1.0f + x;
            for (int it = 0; it < iters * 16; ++it) { x = __builtin_fmaf(x, 1.0001f, y); y = __builtin_fmaf(y, 0.9999f, x); }
            x += y;
        } else if (noise == 4) {                            // float32 matrix instructions (the exact kernels' pipe)
            f16v acc = {0};
            float a = 0.01f * lane, b = 0.02f * lane;
            for (int it = 0; it < iters * 2; ++it) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc, 0, 0, 0);
            }
            x += acc[0] + acc[7];
        }
        sink[blockIdx.x * 256 + tid] = x;
        return;
    }
    unsigned h = (blockIdx.x * 256u + tid) * 2654435761u + 12345u;
    unsigned long long wlo = 0, whi = 0, wq[4] = {0, 0, 0, 0};
    const int f = form < 2 * kFormsPerSet ? form % kFormsPerSet : form - 2 * kFormsPerSet;
    const int op = form >= 2 * kFormsPerSet ? 4 : f < 48 ? f >> 4 : 3, sel = f < 48 ? f & 15 : f - 48;
    const int x = (sel >> 3) & 1, y = (sel >> 2) & 1, z = (sel >> 1) & 1, w = sel & 1;
    float keep = 0.0f;
    for (int it = 0; it < iters; ++it) {
        float v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { h = h * 1664525u + 1013904223u; v[k] = (float)(int)(h >> 8) * (1.0f / 4194304.0f) - 2.0f; }
        const float a0 = v[0], a1 = v[1], b0 = v[2], b1 = v[3], c0 = v[4], c1 = v[5];
        float dlo, dhi;
        switch (form) {
            ALL16(0, HEAD0, REGS0, TAIL0, "v_pk_mul_f32", SEL2)
            ALL16(16, HEAD0, REGS0, TAIL0, "v_pk_add_f32", SEL2)
            ALL16(32, HEAD0, REGS0F, TAIL0, "v_pk_fma_f32", SEL3)
            SRC2(48, HEAD0, REGS0F, TAIL0)
            ALL16(52, HEAD1, REGS1, TAIL1, "v_pk_mul_f32", SEL2)
            ALL16(68, HEAD1, REGS1, TAIL1, "v_pk_add_f32", SEL2)
            ALL16(84, HEAD1, REGS1F, TAIL1, "v_pk_fma_f32", SEL3)
            SRC2(100, HEAD1, REGS1F, TAIL1)
            ALL16(104, HEAD0, REGS0, TAIL0, "v_pk_mov_b32", SEL2)      // D.lo = S0[op_sel[0]], D.hi = S1[op_sel[1]] (op_sel_hi unused)
            default: dlo = dhi = 0.0f;
        }
        float elo, ehi;
        {
#pragma clang fp contract(off)
            const float al = x ? a1 : a0, bl = y ? b1 : b0, ah = z ? a1 : a0, bh = w ? b1 : b0;
            if (op == 0) { elo = al * bl; ehi = ah * bh; }
            else if (op == 1) { elo = al + bl; ehi = ah + bh; }
            else if (op == 2) { elo = __builtin_fmaf(al, bl, c0); ehi = __builtin_fmaf(ah, bh, c1); }
            else if (op == 4) { elo = al; ehi = y ? b1 : b0; }
            else { elo = __builtin_fmaf(a0, b0, z ? c1 : c0); ehi = __builtin_fmaf(a1, b1, w ? c1 : c0); }    // third source selected: op_sel[2] = z, op_sel_hi[2] = w
        }
        const bool bl_ = __float_as_uint(dlo) != __float_as_uint(elo), bh_ = __float_as_uint(dhi) != __float_as_uint(ehi);
        wlo += bl_; whi += bh_;
        if (bl_ || bh_) ++wq[lane >> 4];
        keep += dlo + dhi;
    }
    if (wlo) atomicAdd(&bad[0], wlo);
    if (whi) atomicAdd(&bad[1], whi);
    for (int q = 0; q < 4; ++q) if (wq[q]) atomicAdd(&bad[2 + q], wq[q]);
    sink[blockIdx.x * 256 + tid] = keep;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    unsigned long long* bad; float* sink;
    CK(hipMalloc(&bad, 8 * sizeof(unsigned long long)));
    CK(hipMalloc(&sink, 1024 * 256 * sizeof(float)));
    CK(hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const char* ops[5] = {"v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_pk_fma_f32 third source: op_sel [0,0,k] op_sel_hi [1,1,m], printed as op_sel_hi [k, m]", "v_pk_mov_b32"};
    const char* noises[5] = {"none", "mfma+lds+valu", "lds", "valu", "mfma_f32"};
    for (int noise = 0; noise < 2; ++noise)
        for (int form = 0; form < 2 * kFormsPerSet + 16; ++form) {
            CK(hipMemset(bad, 0, 8 * sizeof(unsigned long long)));
            probe<<<1024, 256, 40 * 1024 - 512>>>(form, noise, iters, bad, sink);
            CK(hipDeviceSynchronize());
            unsigned long long hb[8];
            CK(hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost));
            const int f = form < 2 * kFormsPerSet ? form % kFormsPerSet : form - 2 * kFormsPerSet, sel = f < 48 ? f & 15 : ((f - 48) >> 1) * 4 + ((f - 48) & 1) * 1 + 16;
            // sel printed as (op_sel[0], op_sel[1]) (op_sel_hi[0], op_sel_hi[1]); third-source forms: op_sel [0,0,k], op_sel_hi [1,1,m]
            printf("{\"other_waves\": \"%s\", \"registers\": \"%s\", \"op\": \"%s\", \"op_sel\": [%d, %d], \"op_sel_hi\": [%d, %d], \"tested\": %.0f, \"wrong_lo\": %llu, \"wrong_hi\": %llu, "
                   "\"wrong_by_lane_quarter\": [%llu, %llu, %llu, %llu]}\n",
                   noises[noise], form < kFormsPerSet ? "d=v[18:19] a=v[4:5] b=v[10:11] c=v[8:9]" : "d=a=v[6:7] b=v[12:13] c=v[16:17]", ops[form >= 2 * kFormsPerSet ? 4 : f < 48 ? f >> 4 : 3], (sel >> 3) & 1, (sel >> 2) & 1, (sel >> 1) & 1, sel & 1, 768.0 * 256.0 * iters, hb[0], hb[1], hb[2], hb[3], hb[4],
                   hb[5]);
            fflush(stdout);
        }
    return 0;
}
