// Measurement probe (not part of the product): what does ONE wave per SIMD pay for work placed next to a chain of
// v_mfma_f32_32x32x16_bf16, and what exactly does ds_read_b64_tr_b16 deliver?  Inputs for the split-bf16 step kernel
// (DESIGN.md section 3.1e).
//   chain<KIND, NF, NACC>: REPS x 16 matrix instructions on NACC accumulators in turn, NF fillers of KIND after each
//   tr16 dump:             every lane reads 8 bytes at its own address; LDS holds the 16-bit element index
// 256 workgroups (one per CU, 96 KiB of LDS each) x 256 threads: one wave per SIMD.  Output: JSON lines on stdout.
// Build:  hipcc --offload-arch=gfx950 -O3 -o tests/tools/bf16_probe.out tests/tools/bf16_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define SB() __builtin_amdgcn_sched_barrier(0)

enum Kind { K_NONE = 0, K_FMA = 1, K_CVT = 2, K_PERM = 3, K_DSR128 = 4, K_DSW16 = 5, K_DSW32 = 6, K_ANDSUB = 7,
            K_DEPB = 8, K_TR16 = 9, K_F32MFMA = 10, K_DSW64 = 11, K_SPLIT3 = 12 };

constexpr int REPS = 8;

template <int KIND, int NF, int NACC>
__global__ __launch_bounds__(256, 1) void chain(unsigned* clocks, float* sink, float seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float* lds = reinterpret_cast<float*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* mine = lds + wave * 4096 + lane * 4;          // 16 KiB per wave
    for (int i = tid; i < 4 * 4096; i += 256) lds[i] = seed + i;
    __syncthreads();
    f32x16 acc[NACC];
#pragma unroll
    for (int n = 0; n < NACC; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + j + lane); b[j] = (__bf16)(seed * 0.5f + j); }
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = seed + j + lane;
    f32x4 rd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rd[j] = f32x4{0, 0, 0, 0};
    unsigned u[4] = {1, 2, 3, 4};
    s16x4 tr = {0, 0, 0, 0};
    float fa = seed, fb = seed * 0.25f;
    SB();
    const unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime();
    SB();
    for (int rep = 0; rep < REPS; ++rep) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if constexpr (KIND == K_F32MFMA) acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[k % NACC], 0, 0, 0);
            else acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k % NACC], 0, 0, 0);
            SB();
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int j = (k * NF + f) & 7;
                if constexpr (KIND == K_FMA || KIND == K_F32MFMA) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(0.999f), "v"(0.001f));
                } else if constexpr (KIND == K_CVT) {
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[j & 3]) : "v"(x[j]), "v"(x[(j + 1) & 7]));
                } else if constexpr (KIND == K_PERM) {
                    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[j & 3]) : "v"(x[j]), "v"(x[(j + 1) & 7]), "v"(0x07060302u));
                } else if constexpr (KIND == K_ANDSUB) {     // one filler = and + sub (the residual of a split)
                    asm volatile("v_and_b32 %0, 0xffff0000, %1\n\tv_sub_f32 %1, %1, %0" : "=&v"(u[j & 3]), "+v"(x[j]));
                } else if constexpr (KIND == K_SPLIT3) {     // one filler = full 3-plane split of a PAIR (11 VALU)
                    unsigned h, m, l; float r0, r1, s0, s1;
                    asm volatile(
                        "v_cvt_pk_bf16_f32 %0, %7, %8\n\t"
                        "v_lshlrev_b32 %3, 16, %0\n\t"
                        "v_and_b32 %4, 0xffff0000, %0\n\t"
                        "v_sub_f32 %3, %7, %3\n\t"
                        "v_sub_f32 %4, %8, %4\n\t"
                        "v_cvt_pk_bf16_f32 %1, %3, %4\n\t"
                        "v_lshlrev_b32 %5, 16, %1\n\t"
                        "v_and_b32 %6, 0xffff0000, %1\n\t"
                        "v_sub_f32 %5, %3, %5\n\t"
                        "v_sub_f32 %6, %4, %6\n\t"
                        "v_cvt_pk_bf16_f32 %2, %5, %6"
                        : "=&v"(h), "=&v"(m), "=&v"(l), "=&v"(r0), "=&v"(r1), "=&v"(s0), "=&v"(s1)
                        : "v"(x[j]), "v"(x[(j + 1) & 7]));
                    u[j & 3] ^= h ^ m ^ l;
                } else if constexpr (KIND == K_DSR128) {
                    rd[j & 3] = *reinterpret_cast<volatile f32x4*>(mine + 256 * ((k * NF + f) & 15));
                } else if constexpr (KIND == K_DSW16) {
                    *reinterpret_cast<volatile unsigned short*>(reinterpret_cast<unsigned short*>(lds + wave * 4096) + lane + 64 * ((k * NF + f) & 15)) = (unsigned short)u[j & 3];
                } else if constexpr (KIND == K_DSW32) {
                    *reinterpret_cast<volatile float*>(lds + wave * 4096 + lane + 64 * ((k * NF + f) & 15)) = x[j];
                } else if constexpr (KIND == K_DSW64) {
                    *reinterpret_cast<volatile f32x2*>(lds + wave * 4096 + 2 * lane + 128 * ((k * NF + f) & 15)) = f32x2{x[j], x[(j + 1) & 7]};
                } else if constexpr (KIND == K_TR16) {
                    tr += __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(reinterpret_cast<short*>(lds + wave * 4096) + lane * 4 + 256 * ((k * NF + f) & 15)));
                } else if constexpr (KIND == K_DEPB) {       // the NEXT matrix instruction's B operand comes out of this VALU op
                    f32x2 v = {x[j], x[(j + 1) & 7]};
                    bf16x2 p = __builtin_convertvector(v, bf16x2);
                    b[2 * (f & 3)] = p[0]; b[2 * (f & 3) + 1] = p[1];
                }
            }
            SB();
        }
    }
    SB();
    const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime();
    SB();
    float s = 0.0f;
#pragma unroll
    for (int n = 0; n < NACC; ++n) s += acc[n][0] + acc[n][5] + acc[n][15];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j];
#pragma unroll
    for (int j = 0; j < 4; ++j) s += rd[j][0] + rd[j][3] + (float)u[j];
    s += (float)(tr[0] + tr[1] + tr[2] + tr[3]) + (float)b[3];
    if (s == 12345.678f) sink[tid] = s;
    if (lane == 0) clocks[blockIdx.x * 4 + wave] = t1 - t0;
}

// ds_read_b64_tr_b16: LDS element e (16 bit) holds e; lane l reads 8 bytes at element address addr[l]
__global__ void tr16_dump(const int* addr_elems, short* out) {
    __shared__ __attribute__((aligned(16))) short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (short)i;
    __syncthreads();
    if (threadIdx.x < 64) {
        const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr_elems[threadIdx.x]));
#pragma unroll
        for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
    }
}

// A/B operand k-map of v_mfma_f32_32x32x16_bf16: A = one-hot rows, B = k index -> D[i][j] tells which k a (lane, t) slot is
__global__ void kmap_dump(float* out) {
    const int lane = threadIdx.x & 63;
    for (int slot = 0; slot < 16; ++slot) {          // slot = (hi, t): A has a 1 only at lanes with l>>5 == slot>>3, element slot&7
        bf16x8 a, b;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            a[t] = (__bf16)(((lane >> 5) == (slot >> 3) && t == (slot & 7)) ? 1.0f : 0.0f);
            b[t] = (__bf16)(float)(8 * (lane >> 5) + t + 1);      // B[(hi,t)][j] = 8*hi + t + 1 for every column
        }
        f32x16 c = {};
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
        if (lane == 0) out[slot] = c[0];              // = B value of the k the A slot pairs with (expect slot + 1)
    }
}

template <int KIND, int NF, int NACC>
void run(const char* name, unsigned* d_clk, float* d_sink) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(chain<KIND, NF, NACC>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    std::vector<unsigned> h(1024);
    double best = 1e30, med = 0;
    for (int it = 0; it < 3; ++it) {
        hipLaunchKernelGGL((chain<KIND, NF, NACC>), dim3(256), dim3(256), 96 * 1024, 0, d_clk, d_sink, 1.0f + it);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d_clk, 1024 * sizeof(unsigned), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        best = std::min(best, (double)h[0]);
        med = h[512];
    }
    printf("{\"probe\": \"%s\", \"fillers_per_mfma\": %d, \"accumulators\": %d, \"clocks_per_mfma_min\": %.2f, \"clocks_per_mfma_median\": %.2f}\n",
           name, NF, NACC, best / (REPS * 16), med / (REPS * 16));
    fflush(stdout);
}

int main() {
    unsigned* d_clk; float* d_sink;
    hipMalloc(&d_clk, 1024 * sizeof(unsigned));
    hipMalloc(&d_sink, 1024 * sizeof(float));
    run<K_NONE, 0, 1>("bf16 chain, one accumulator", d_clk, d_sink);
    run<K_NONE, 0, 2>("bf16 chain, two accumulators", d_clk, d_sink);
    run<K_NONE, 0, 4>("bf16 chain, four accumulators", d_clk, d_sink);
    run<K_FMA, 1, 1>("v_fma fillers", d_clk, d_sink);
    run<K_FMA, 2, 1>("v_fma fillers", d_clk, d_sink);
    run<K_FMA, 4, 1>("v_fma fillers", d_clk, d_sink);
    run<K_FMA, 6, 1>("v_fma fillers", d_clk, d_sink);
    run<K_FMA, 8, 1>("v_fma fillers", d_clk, d_sink);
    run<K_FMA, 12, 1>("v_fma fillers", d_clk, d_sink);
    run<K_FMA, 16, 1>("v_fma fillers", d_clk, d_sink);
    run<K_FMA, 4, 2>("v_fma fillers", d_clk, d_sink);
    run<K_FMA, 8, 2>("v_fma fillers", d_clk, d_sink);
    run<K_FMA, 16, 2>("v_fma fillers", d_clk, d_sink);
    run<K_CVT, 4, 1>("v_cvt_pk_bf16_f32 fillers", d_clk, d_sink);
    run<K_CVT, 8, 1>("v_cvt_pk_bf16_f32 fillers", d_clk, d_sink);
    run<K_PERM, 4, 1>("v_perm_b32 fillers", d_clk, d_sink);
    run<K_PERM, 8, 1>("v_perm_b32 fillers", d_clk, d_sink);
    run<K_ANDSUB, 2, 1>("and+sub pair fillers (2 VALU each)", d_clk, d_sink);
    run<K_ANDSUB, 4, 1>("and+sub pair fillers (2 VALU each)", d_clk, d_sink);
    run<K_SPLIT3, 1, 1>("3-plane split of a pair (11 VALU each)", d_clk, d_sink);
    run<K_SPLIT3, 2, 1>("3-plane split of a pair (11 VALU each)", d_clk, d_sink);
    run<K_DSR128, 1, 1>("ds_read_b128 fillers", d_clk, d_sink);
    run<K_DSR128, 2, 1>("ds_read_b128 fillers", d_clk, d_sink);
    run<K_DSR128, 4, 1>("ds_read_b128 fillers", d_clk, d_sink);
    run<K_DSW16, 2, 1>("ds_write_b16 fillers", d_clk, d_sink);
    run<K_DSW16, 4, 1>("ds_write_b16 fillers", d_clk, d_sink);
    run<K_DSW16, 8, 1>("ds_write_b16 fillers", d_clk, d_sink);
    run<K_DSW32, 2, 1>("ds_write_b32 fillers", d_clk, d_sink);
    run<K_DSW32, 4, 1>("ds_write_b32 fillers", d_clk, d_sink);
    run<K_DSW64, 2, 1>("ds_write_b64 fillers", d_clk, d_sink);
    run<K_TR16, 1, 1>("ds_read_b64_tr_b16 fillers", d_clk, d_sink);
    run<K_TR16, 2, 1>("ds_read_b64_tr_b16 fillers", d_clk, d_sink);
    run<K_TR16, 4, 1>("ds_read_b64_tr_b16 fillers", d_clk, d_sink);
    run<K_DEPB, 1, 1>("cvt_pk feeding the next B operand", d_clk, d_sink);
    run<K_DEPB, 4, 1>("cvt_pk feeding the next B operand", d_clk, d_sink);
    run<K_DEPB, 4, 2>("cvt_pk feeding the next B operand", d_clk, d_sink);
    run<K_F32MFMA, 0, 1>("fp32 32x32x2 chain", d_clk, d_sink);
    run<K_F32MFMA, 4, 1>("fp32 32x32x2 chain + v_fma fillers", d_clk, d_sink);

    // ---- ds_read_b64_tr_b16 semantics ----
    int* d_addr; short* d_out;
    hipMalloc(&d_addr, 64 * sizeof(int));
    hipMalloc(&d_out, 256 * sizeof(short));
    for (int variant = 0; variant < 3; ++variant) {
        int addr[64];
        for (int l = 0; l < 64; ++l)
            addr[l] = variant == 0 ? 4 * l                       // contiguous: lane l -> elements 4l..4l+3
                    : variant == 1 ? 64 * l                      // one 128-byte row per lane
                                   : 4 * (l & 3) + 100 * (l >> 2);   // 4 lanes per 200-byte row
        hipMemcpy(d_addr, addr, sizeof(addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(tr16_dump, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        short out[256];
        hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost);
        printf("{\"probe\": \"tr16_dump\", \"variant\": %d, \"addr_elems\": [", variant);
        for (int l = 0; l < 64; ++l) printf("%d%s", addr[l], l < 63 ? ", " : "");
        printf("], \"out\": [");
        for (int i = 0; i < 256; ++i) printf("%d%s", (int)out[i], i < 255 ? ", " : "");
        printf("]}\n");
    }
    float* d_k;
    hipMalloc(&d_k, 16 * sizeof(float));
    hipLaunchKernelGGL(kmap_dump, dim3(1), dim3(64), 0, 0, d_k);
    float hk[16];
    hipMemcpy(hk, d_k, sizeof(hk), hipMemcpyDeviceToHost);
    printf("{\"probe\": \"mfma_32x32x16_bf16 operand k pairing: B value seen by A slot (hi,t)\", \"values\": [");
    for (int i = 0; i < 16; ++i) printf("%.0f%s", hk[i], i < 15 ? ", " : "");
    printf("]}\n");
    return 0;
}
