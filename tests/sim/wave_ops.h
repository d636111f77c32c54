// TEST INFRASTRUCTURE: CPU model of vmap_amd/csrc/wave_ops.h (same interface, fibers instead of lanes).
// Selected by include-path order when tests/sim builds the kernel headers for the host.
#pragma once
#include <hip/hip_runtime.h>

#define WV_WAVES_PER_SIMD(n)

namespace wv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// v_mfma_f32_32x32x2_f32 lane maps (cdna_hip_programming.md section 3):
//   A[i][k]: lane = 32*k + i;  B[k][j]: lane = 32*k + j;  D[i][j]: lane = j + 32*((i>>2)&1), reg = (i&3) + 4*(i>>3)
inline f32x16 mfma32(float a, float b, f32x16 c) {
    const int w = sim::wave_id(), l = sim::lane_id();
    sim::Block* B = sim::g_block;
    B->xa[w][l] = a;
    B->xb[w][l] = b;
    sim::wave_barrier();
    const int j = l & 31, hi = l >> 5;
    f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        acc = fmaf(B->xa[w][i], B->xb[w][j], acc);             // k = 0
        acc = fmaf(B->xa[w][32 + i], B->xb[w][32 + j], acc);   // k = 1
        d[r] = acc;
    }
    sim::wave_barrier();
    return d;
}

// ---- bf16 matrix pipe model ----
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

inline float bf16_to_f32(unsigned h) { unsigned u = (h & 0xFFFFu) << 16; float f; std::memcpy(&f, &u, 4); return f; }
inline unsigned f32_to_bf16_rne(float x) {
    unsigned u; std::memcpy(&u, &x, 4);
    if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) return (u >> 16) | 0x40u;     // NaN stays NaN
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
inline unsigned pack_bf16(float lo, float hi) { return f32_to_bf16_rne(lo) | (f32_to_bf16_rne(hi) << 16); }

// v_mfma_f32_32x32x16_bf16: exact products, sum of the 16 products and C formed in double and rounded once (the hardware's
// internal summation order is not architected; this is the most accurate model, the gpu tier checks the real thing)
inline f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    const int w = sim::wave_id(), l = sim::lane_id();
    sim::Block* B = sim::g_block;
    for (int i = 0; i < 4; ++i) { B->xa4[w][l][i] = a[i]; B->xb4[w][l][i] = b[i]; }
    sim::wave_barrier();
    const int j = l & 31, hi = l >> 5;
    f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        double acc = c[r];
        for (int h = 0; h < 2; ++h)
            for (int t = 0; t < 8; ++t) {
                const unsigned ua = B->xa4[w][32 * h + i][t >> 1], ub = B->xb4[w][32 * h + j][t >> 1];
                const float fa = bf16_to_f32((t & 1) ? ua >> 16 : ua), fb = bf16_to_f32((t & 1) ? ub >> 16 : ub);
                acc += (double)fa * (double)fb;
            }
        d[r] = (float)acc;
    }
    sim::wave_barrier();
    return d;
}

// v_mfma_f32_16x16x32_bf16 (same model as mfma_bf16)
typedef float f32x4m __attribute__((ext_vector_type(4)));
inline f32x4m mfma16_bf16(u32x4 a, u32x4 b, f32x4m c) {
    const int w = sim::wave_id(), l = sim::lane_id();
    sim::Block* B = sim::g_block;
    for (int i = 0; i < 4; ++i) { B->xa4[w][l][i] = a[i]; B->xb4[w][l][i] = b[i]; }
    sim::wave_barrier();
    const int n = l & 15, g = l >> 4;
    f32x4m d = c;
    for (int r = 0; r < 4; ++r) {
        const int m = 4 * g + r;
        double acc = c[r];
        for (int kg = 0; kg < 4; ++kg)
            for (int t = 0; t < 8; ++t) {
                const unsigned ua = B->xa4[w][16 * kg + m][t >> 1], ub = B->xb4[w][16 * kg + n][t >> 1];
                const float fa = bf16_to_f32((t & 1) ? ua >> 16 : ua), fb = bf16_to_f32((t & 1) ? ub >> 16 : ub);
                acc += (double)fa * (double)fb;
            }
        d[r] = (float)acc;
    }
    sim::wave_barrier();
    return d;
}

// ds_read_b64_tr_b16 (see the device header): lane c of a 16-lane group gets element (c & 3) of the 4 words at lane 4j + (c >> 2)'s address
inline u32x2 lds_tr16(const void* p) {
    const int w = sim::wave_id(), l = sim::lane_id();
    sim::Block* B = sim::g_block;
    B->xp[w][l] = p;
    sim::wave_barrier();
    const int g = l & ~15, c = l & 15;
    unsigned short v[4];
    for (int j = 0; j < 4; ++j) {
        const unsigned short* src = static_cast<const unsigned short*>(B->xp[w][g + 4 * j + (c >> 2)]);
        v[j] = src[c & 3];
    }
    sim::wave_barrier();
    u32x2 out;
    out[0] = (unsigned)v[0] | ((unsigned)v[1] << 16);
    out[1] = (unsigned)v[2] | ((unsigned)v[3] << 16);
    return out;
}

inline float swap_half(float x) {
    const int w = sim::wave_id(), l = sim::lane_id();
    sim::g_block->xa[w][l] = x;
    sim::wave_barrier();
    float y = sim::g_block->xa[w][l ^ 32];
    sim::wave_barrier();
    return y;
}

inline float shfl(float x, int src) {
    const int w = sim::wave_id(), l = sim::lane_id();
    sim::g_block->xa[w][l] = x;
    sim::wave_barrier();
    float y = sim::g_block->xa[w][src & 63];
    sim::wave_barrier();
    return y;
}

inline void glds16(const float* g, float* lds_wave_base) {
    float* d = lds_wave_base + 4 * sim::lane_id();
    for (int i = 0; i < 4; ++i) d[i] = g[i];
}

template <int J>
inline float row_bcast(float x) { return shfl(x, (sim::lane_id() & ~15) + J); }

inline float row_sum16(float x) {       // same butterfly order as the device DPP sequence
    const int l = sim::lane_id();
    x += shfl(x, l ^ 1);
    x += shfl(x, l ^ 2);
    x += shfl(x, (l & ~7) + (7 - (l & 7)));
    x += shfl(x, (l & ~15) + (15 - (l & 15)));
    return x;
}

inline float half_sum32_hi_row(float x) {   // row sums, then rows 1 / 3 add the sum of rows 0 / 2 (own row's sum first, as the device does)
    const int l = sim::lane_id();
    x = row_sum16(x);
    const float t = shfl(x, (l & ~31) + 15);
    return (l & 16) ? x + t : x;
}

inline bool wave_any(bool pred) {
    const int w = sim::wave_id(), l = sim::lane_id();
    sim::g_block->xa[w][l] = pred ? 1.0f : 0.0f;
    sim::wave_barrier();
    bool any = false;
    for (int i = 0; i < sim::kWave; ++i) any |= sim::g_block->xa[w][i] != 0.0f;
    sim::wave_barrier();
    return any;
}

inline void wave_lds_fence() { sim::wave_barrier(); }

inline float* lds_base() { return reinterpret_cast<float*>(sim::g_block->lds.data()); }

template <class T>
inline const T& kernarg_late(const T& a) { return a; }

inline float relu(float x) { return x > 0.0f ? (x < 3.4028234663852886e38f ? x : 3.4028234663852886e38f) : 0.0f; }
inline float opaque(float x) { return x; }
inline int opaque_iter(int x) { return x; }
inline int uniform(int x) { return x; }
inline unsigned opaque_uzero() { return 0u; }
inline unsigned opaque_u(unsigned x) { return x; }
inline float after(float x, float) { return x; }
inline unsigned after_u(unsigned x, unsigned) { return x; }

inline void sched_fence() {}
template <int NM, int NV>
inline void interleave_mfma_valu() {}

inline unsigned clock32() { return (unsigned)sim::g_yields; }

inline void lds_add(float* p, float v) { *p += v; }

// in-launch hand-off primitives: the executor runs one workgroup at a time, so these are plain accesses and a wait that
// is not already satisfied fails (the carried-finalize kernels are exercised on the GPU tier only)
inline void touch_scalar4(unsigned&, unsigned&, unsigned&, unsigned&) {}
inline void store_wt(float* p, float v) { *p = v; }
inline float load_wt(const float* p) { return *p; }
inline void glds16_wt(const float* g, float* lds_wave_base) { glds16(g, lds_wave_base); }
inline void drain_vm() {}
inline void signal_add(unsigned* c) { ++*c; }
inline bool wait_ge(const unsigned* c, unsigned target, int) { return *c >= target; }

}  // namespace wv
