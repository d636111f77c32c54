// gen_kernels.h - the fused training step for ANY hidden width H = 32*NB (background model H=128, iMAP H=256,
// stress config H=64): same algorithm, same matrix-instruction building blocks and the same numerics as
// step_main_h32, but sized for widths whose parameters (up to 1.3 MB) and activations do not fit LDS/registers:
//
//   * weights are read straight from the object's packed parameter image in global memory (L2-resident; the
//     image's 16-byte-aligned rows make the forward operand a global_load_dwordx4 per four MFMAs);
//   * activations, their F-forms (lane = feature) and the cos factors of the encoding live as "register images"
//     (16 registers x 64 lanes = 4 KiB, lane-contiguous => coalesced) in a per-wave scratch area of the workspace;
//     every lane only ever re-reads what it wrote itself, so no cross-lane visibility protocol is involved;
//   * loops over the NB feature blocks are runtime loops (one kernel for every width).
//
// Workgroup shape, compositing phase, atomic-free gradient reduction (reduce_block + partials + step_finalize) and
// the C ABI are those of the H=32 kernel.  This is the general path (about one global round trip per 16 matrix
// instructions); hidden = 32 keeps its LDS/register-resident specialisation.
#pragma once
#include "step_kernels.h"

namespace vk {

constexpr int kBlk = 16 * 64;            // floats in one stored register block

struct GenArgs {
    StepArgs s;
    float* scratch;                      // [workgroups][kWaves][wave_blocks][kBlk]
    int wave_blocks;                     // 20 + 14 * NB
};

__host__ __device__ constexpr int gen_wave_blocks(int NB) { return 20 + 14 * NB; }

// LDS map of the generic kernel (floats)
struct LdsGen {
    static constexpr int SCR = 0;                                   // per-wave transpose scratch: kWaves x 2 x [32][33]
    static constexpr int SCR_WAVE = Lds32::SCR_WAVE;                // same tile strides as the hidden-32 kernel (shared helpers)
    static constexpr int STG_TILE = Lds32::STG_TILE;
    static constexpr int STG = SCR + kWaves * SCR_WAVE;             // 2 buffers x kWaves tiles
    static constexpr int CB = STG + 2 * kWaves * STG_TILE;          // composite buffer [kMaxPts][8]
    static constexpr int LOSS = CB + kMaxPts * 8;                   // [kWaves][4]
    static constexpr int VEC = LOSS + kWaves * 4;                   // kWaves x small_n (runtime size)
    __host__ __device__ static constexpr int bytes(int small_n) { return (VEC + kWaves * small_n) * 4; }
};

// Register image of 16 values per lane (4 KiB): four 1 KiB chunks, chunk j = registers 4j..4j+3 of all 64 lanes, 16 bytes
// per lane - every access is one fully coalesced 16-byte-per-lane instruction (the first layout, [register][lane] dwords,
// cost 16 memory instructions per image: 8900 loads per wave per step at the background shape, 57 % of the time waiting)
__device__ __forceinline__ void ldb(float (&v)[16], const float* blk, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const wv::f32x4 t = *reinterpret_cast<const wv::f32x4*>(blk + j * 256 + lane * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * j + e] = t[e];
    }
}
__device__ __forceinline__ void stb(float* blk, const float (&v)[16], int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        wv::f32x4 t;
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = v[4 * j + e];
        *reinterpret_cast<wv::f32x4*>(blk + j * 256 + lane * 4) = t;
    }
}
__device__ __forceinline__ void stacc(float* blk, const f32x16& a, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        wv::f32x4 t;
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = a[4 * j + e];
        *reinterpret_cast<wv::f32x4*>(blk + j * 256 + lane * 4) = t;
    }
}
// ---- software-pipelined matrix chains over register images in global memory ----
// A wave of these kernels spends most of its time waiting for memory (background shape: 57 % of its cycles in
// s_waitcnt): every 16-deep matrix chain needs one register image (4 x 16 B per lane) and 16 weights per lane, all
// through global loads of L2 latency, and the feature-block loops are runtime loops the compiler does not pipeline.
// Here the operands of segment i+1 are requested before the chain of segment i is issued (two named operand sets).
struct FSeg { const float* w; const float* x; };      // w: this lane's weight row (32 consecutive k), x: register image
template <class F>
__device__ __forceinline__ void chain_fwd(f32x16& acc, int n, F seg, int lane) {
    float xa[16], xb[16];
    wv::f32x4 wa[4], wb[4];
    auto ld = [&](int i, float (&x)[16], wv::f32x4 (&w)[4]) {
        const FSeg s = seg(i);
        ldb(x, s.x, lane);
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = *reinterpret_cast<const wv::f32x4*>(s.w + 8 * q);
    };
    auto mm = [&](const float (&x)[16], const wv::f32x4 (&w)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = wv::mfma32(w[q][i], x[4 * q + i], acc);
        }
    };
    ld(0, xa, wa);
    int i = 0;
    for (; i + 1 < n; i += 2) {
        ld(i + 1, xb, wb);
        mm(xa, wa);
        if (i + 2 < n) ld(i + 2, xa, wa);
        mm(xb, wb);
    }
    if (i < n) mm(xa, wa);
}
struct BSeg { const float* wcol; const float* x; };   // wcol: &W[32*ob + 4*hi][this lane's column], x: delta image (P-form)
template <class F>
__device__ __forceinline__ void chain_bwd(f32x16& acc, int n, int ld_w, F seg, int lane) {
    float xa[16], xb[16], wa[16], wb[16];
    auto ld = [&](int i, float (&x)[16], float (&w)[16]) {
        const BSeg s = seg(i);
        ldb(x, s.x, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) w[r] = s.wcol[((r & 3) + 8 * (r >> 2)) * ld_w];
    };
    auto mm = [&](const float (&x)[16], const float (&w)[16]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc = wv::mfma32(w[r], x[r], acc);
    };
    ld(0, xa, wa);
    int i = 0;
    for (; i + 1 < n; i += 2) {
        ld(i + 1, xb, wb);
        mm(xa, wa);
        if (i + 2 < n) ld(i + 2, xa, wa);
        mm(xb, wb);
    }
    if (i < n) mm(xa, wa);
}

// Paired forms: TWO output blocks per loaded operand image (the kernels are bound by the traffic of the register images:
// ~1.3 MB read per 32-point tile at hidden 128 when every output block re-reads its inputs).  The two accumulators are
// independent, so their matrix instructions alternate.
template <class F>
__device__ __forceinline__ void chain_fwd2(f32x16& a0, f32x16& a1, int n, long long wskip, F seg, int lane) {
    float xa[16], xb[16];
    wv::f32x4 wa0[4], wa1[4], wb0[4], wb1[4];
    auto ld = [&](int i, float (&x)[16], wv::f32x4 (&w0)[4], wv::f32x4 (&w1)[4]) {
        const FSeg s = seg(i);
        ldb(x, s.x, lane);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            w0[q] = *reinterpret_cast<const wv::f32x4*>(s.w + 8 * q);
            w1[q] = *reinterpret_cast<const wv::f32x4*>(s.w + wskip + 8 * q);
        }
    };
    auto mm = [&](const float (&x)[16], const wv::f32x4 (&w0)[4], const wv::f32x4 (&w1)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a0 = wv::mfma32(w0[q][i], x[4 * q + i], a0);
                a1 = wv::mfma32(w1[q][i], x[4 * q + i], a1);
            }
        }
    };
    ld(0, xa, wa0, wa1);
    int i = 0;
    for (; i + 1 < n; i += 2) {
        ld(i + 1, xb, wb0, wb1);
        mm(xa, wa0, wa1);
        if (i + 2 < n) ld(i + 2, xa, wa0, wa1);
        mm(xb, wb0, wb1);
    }
    if (i < n) mm(xa, wa0, wa1);
}
struct BSeg2 { const float* wcol0; const float* wcol1; const float* x; };
template <class F>
__device__ __forceinline__ void chain_bwd2(f32x16& a0, f32x16& a1, int n, int ld_w, F seg, int lane) {
    float xa[16], xb[16], wa0[16], wa1[16], wb0[16], wb1[16];
    auto ld = [&](int i, float (&x)[16], float (&w0)[16], float (&w1)[16]) {
        const BSeg2 s = seg(i);
        ldb(x, s.x, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            w0[r] = s.wcol0[((r & 3) + 8 * (r >> 2)) * ld_w];
            w1[r] = s.wcol1[((r & 3) + 8 * (r >> 2)) * ld_w];
        }
    };
    auto mm = [&](const float (&x)[16], const float (&w0)[16], const float (&w1)[16]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            a0 = wv::mfma32(w0[r], x[r], a0);
            a1 = wv::mfma32(w1[r], x[r], a1);
        }
    };
    ld(0, xa, wa0, wa1);
    int i = 0;
    for (; i + 1 < n; i += 2) {
        ld(i + 1, xb, wb0, wb1);
        mm(xa, wa0, wa1);
        if (i + 2 < n) ld(i + 2, xa, wa0, wa1);
        mm(xb, wb0, wb1);
    }
    if (i < n) mm(xa, wa0, wa1);
}

// d-prop with a runtime row pitch
__device__ __forceinline__ void bwd_mm_rt(f32x16& acc, const float* wcol, int ld, const float (&dy)[16]) {
    float w[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) w[r] = wcol[((r & 3) + 8 * (r >> 2)) * ld];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc = wv::mfma32(w[r], dy[r], acc);
}
// quarter store with a runtime row pitch; add = accumulate onto an earlier pass of the same workgroup
__device__ __forceinline__ void store_quarter_rt(float* out, int K, const float (&q)[4], int col0, int ncols, bool add,
                                                 int wave, int p31, int hi) {
    if (p31 < ncols) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float* o = out + (8 * wave + 4 * hi + i) * K + col0 + p31;
            *o = add ? *o + q[i] : q[i];
        }
    }
}

template <bool BWD>
__global__ __launch_bounds__(kWG, 1) void step_main_gen(const GenArgs ga) {
    const StepArgs& a = ga.s;
    const GenLayout L = gen_layout(a.hidden);
    const int H = L.H, NB = L.NB;
    float* lds = wv::lds_base();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p31 = lane & 31, hi = lane >> 5;
    const int obj = blockIdx.x / a.NW, wgo = blockIdx.x - obj * a.NW;
    const float* Wg = a.wimg + (long long)obj * L.imgp;                 // this object's parameter image (global)
    float* Gv = lds + LdsGen::VEC + wave * L.small_n - L.b_in;           // private small-vector gradients, image-indexed
    float* sb = ga.scratch + ((long long)blockIdx.x * kWaves + wave) * ga.wave_blocks * kBlk;
    // scratch block indices
    const int E_P = 0, E_F = 5, CFB = 10, H_P = 15, H_F = 15 + 5 * NB, D_P = 15 + 10 * NB, D_F = 15 + 12 * NB, DE = 15 + 14 * NB;
#define BLK(i) (sb + (long long)(i) * kBlk)

    if (BWD) {
        for (int i = tid; i < kWaves * L.small_n; i += kWG) lds[LdsGen::VEC + i] = 0.0f;
    }
    if (tid < kWaves * 4) lds[LdsGen::LOSS + tid] = 0.0f;
    float* out = a.part_grad + ((long long)(obj * a.NW + wgo)) * a.PP;
    float* stg0 = lds + LdsGen::STG;
    float* stg1 = stg0 + kWaves * LdsGen::STG_TILE;
    float* scrX = lds + LdsGen::SCR + wave * LdsGen::SCR_WAVE;
    float* scrD = scrX + Lds32::SCR_TILE;
    float* cb = lds + LdsGen::CB;
    const float* cbw = cb + wave * 32 * 8;
    const float scale = a.pe_scale.p[obj * a.pe_scale.stride];
    const float* Bg = Wg + L.pe_b;
    int stage_toggle = 0;                                               // alternates the two staging buffers

    const int tid_k = tid;
    for (int grp = wgo; grp < a.NG; grp += a.NW) {
    // lane coordinates of this pass, opaque per iteration (see step_main_h32: keeps the body's invariants in the body)
    const int tid = wv::opaque_iter(tid_k), lane = tid & 63, wave = tid >> 6, p31 = lane & 31, hi = lane >> 5;
    const bool first_pass = grp == wgo;
    __syncthreads();
    for (int i = tid; i < kMaxPts * 8; i += kWG) cb[i] = 0.0f;

    const int ray0 = grp * a.G;
    const int nrays = min(a.G, a.R - ray0);
    const int npts = nrays * a.S;
    const int pt = wave * 32 + p31;
    const bool valid = pt < npts;
    const int lray = valid ? pt / a.S : 0;
    const int smp = valid ? pt - lray * a.S : 0;
    const int ray = ray0 + lray;
    float t[3] = {0.0f, 0.0f, 0.0f};
    if (valid) {
        float x0, x1, x2;
        load_point(a, obj, ray, smp, x0, x1, x2);
        t[0] = x0 / scale;
        t[1] = x1 / scale;
        t[2] = x2 / scale;
    }
    float xv[16], yv[16];
    f32x16 acc;
    // ---- encoding: P-form, F-form and cos factors of the five encoding blocks go to scratch ----
    {
        float proj[kDirs];
#pragma unroll
        for (int d = 0; d < kDirs; ++d)
            proj[d] = fmaf(t[2], Bg[3 * d + 2], fmaf(t[1], Bg[3 * d + 1], t[0] * Bg[3 * d]));
        float amax = 0.0f;
#pragma unroll
        for (int d = 0; d < kDirs; ++d) amax = fmaxf(amax, fabsf(proj[d]));
        const bool big = wv::wave_any(!(amax * (32.0f * kPi) < kSinCosFastLimit));
#define ENC(i, NS, base, limit, kb)                                                                      \
        if (__builtin_expect(!big, 1)) pe_block<NS, false>(xv, yv, base, limit, kb, t, proj, hi);                             \
        else pe_block<NS, true>(xv, yv, base, limit, kb, t, proj, hi);                                   \
        stb(BLK(E_P + i), xv, lane); stb(BLK(CFB + i), yv, lane);                                        \
        toF_put(scrX, xv, p31, hi); toF_get(yv, scrX, p31, hi); stb(BLK(E_F + i), yv, lane);
        ENC(0, 16, 0, kEmb1, 0)
        ENC(1, 16, 0, kEmb1, 1)
        ENC(2, 12, 0, kEmb1, 2)
        ENC(3, 16, kEmb1, kEmb2, 0)
        ENC(4, 6, kEmb1, kEmb2, 1)
#undef ENC
    }
    __syncthreads();        // composite buffer zeroed

    // ---- field MLP forward (model.py:59-83): layer l output block ob -> H_P(l, ob) ----
    f32x16 acc1;
    auto finish = [&](int l, int ob, const f32x16& av) {          // ReLU, store P-form and F-form
        relu_to(xv, av);
        stb(BLK(H_P + l * NB + ob), xv, lane);
        toF_put(scrX, xv, p31, hi); toF_get(yv, scrX, p31, hi);
        stb(BLK(H_F + l * NB + ob), yv, lane);
    };
    for (int ob = 0; ob < NB; ob += 2) {           // :59 in_layer (zero weights / encodings pad block 2)
        const float* w = Wg + L.w_in + (32 * ob + p31) * L.ld_in + 4 * hi;
        load_bias(acc, Wg + L.b_in + 32 * ob, hi);
        if (ob + 1 < NB) {
            load_bias(acc1, Wg + L.b_in + 32 * ob + 32, hi);
            chain_fwd2(acc, acc1, 3, 32LL * L.ld_in, [&](int i) { return FSeg{w + 32 * i, BLK(E_P + i)}; }, lane);
            finish(0, ob, acc);
            finish(0, ob + 1, acc1);
        } else {
            chain_fwd(acc, 3, [&](int i) { return FSeg{w + 32 * i, BLK(E_P + i)}; }, lane);
            finish(0, ob, acc);
        }
    }
    for (int ob = 0; ob < NB; ob += 2) {           // :60 mid1
        const float* w = Wg + L.w_m1 + (32 * ob + p31) * L.ld_m + 4 * hi;
        load_bias(acc, Wg + L.b_m1 + 32 * ob, hi);
        if (ob + 1 < NB) {
            load_bias(acc1, Wg + L.b_m1 + 32 * ob + 32, hi);
            chain_fwd2(acc, acc1, NB, 32LL * L.ld_m, [&](int i) { return FSeg{w + 32 * i, BLK(H_P + 0 * NB + i)}; }, lane);
            finish(1, ob, acc);
            finish(1, ob + 1, acc1);
        } else {
            chain_fwd(acc, NB, [&](int i) { return FSeg{w + 32 * i, BLK(H_P + 0 * NB + i)}; }, lane);
            finish(1, ob, acc);
        }
    }
    for (int ob = 0; ob < NB; ob += 2) {           // :63-64 cat_layer
        const float* w = Wg + L.w_cat + (32 * ob + p31) * L.ld_cat + 4 * hi;
        load_bias(acc, Wg + L.b_cat + 32 * ob, hi);
        if (ob + 1 < NB) {
            load_bias(acc1, Wg + L.b_cat + 32 * ob + 32, hi);
            chain_fwd2(acc, acc1, NB + 3, 32LL * L.ld_cat, [&](int i) { return i < NB ? FSeg{w + 32 * i, BLK(H_P + 1 * NB + i)} : FSeg{w + H + 32 * (i - NB), BLK(E_P + (i - NB))}; }, lane);
            finish(2, ob, acc);
            finish(2, ob + 1, acc1);
        } else {
            chain_fwd(acc, NB + 3, [&](int i) { return i < NB ? FSeg{w + 32 * i, BLK(H_P + 1 * NB + i)} : FSeg{w + H + 32 * (i - NB), BLK(E_P + (i - NB))}; }, lane);
            finish(2, ob, acc);
        }
    }
    for (int ob = 0; ob < NB; ob += 2) {           // :67 mid2
        const float* w = Wg + L.w_m2 + (32 * ob + p31) * L.ld_m + 4 * hi;
        load_bias(acc, Wg + L.b_m2 + 32 * ob, hi);
        if (ob + 1 < NB) {
            load_bias(acc1, Wg + L.b_m2 + 32 * ob + 32, hi);
            chain_fwd2(acc, acc1, NB, 32LL * L.ld_m, [&](int i) { return FSeg{w + 32 * i, BLK(H_P + 2 * NB + i)}; }, lane);
            finish(3, ob, acc);
            finish(3, ob + 1, acc1);
        } else {
            chain_fwd(acc, NB, [&](int i) { return FSeg{w + 32 * i, BLK(H_P + 2 * NB + i)}; }, lane);
            finish(3, ob, acc);
        }
    }
    for (int ob = 0; ob < NB; ob += 2) {           // :81 color_linear
        const float* w = Wg + L.w_c + (32 * ob + p31) * L.ld_c + 4 * hi;
        load_bias(acc, Wg + L.b_c + 32 * ob, hi);
        if (ob + 1 < NB) {
            load_bias(acc1, Wg + L.b_c + 32 * ob + 32, hi);
            chain_fwd2(acc, acc1, NB + 2, 32LL * L.ld_c, [&](int i) { return i < NB ? FSeg{w + 32 * i, BLK(H_P + 3 * NB + i)} : FSeg{w + H + 32 * (i - NB), BLK(E_P + 3 + (i - NB))}; }, lane);
            finish(4, ob, acc);
            finish(4, ob + 1, acc1);
        } else {
            chain_fwd(acc, NB + 2, [&](int i) { return i < NB ? FSeg{w + 32 * i, BLK(H_P + 3 * NB + i)} : FSeg{w + H + 32 * (i - NB), BLK(E_P + 3 + (i - NB))}; }, lane);
            finish(4, ob, acc);
        }
    }
    {   // heads (model.py:71,77,82-83)
        float ra = 0.0f, r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
        for (int kb = 0; kb < NB; ++kb) {
            ldb(xv, BLK(H_P + 3 * NB + kb), lane);
            ldb(yv, BLK(H_P + 4 * NB + kb), lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = 32 * kb + phi(r, hi);
                ra = fmaf(Wg[L.w_a + j], xv[r], ra);
                r0 = fmaf(Wg[L.w_oc + j], yv[r], r0);
                r1 = fmaf(Wg[L.w_oc + H + j], yv[r], r1);
                r2 = fmaf(Wg[L.w_oc + 2 * H + j], yv[r], r2);
            }
        }
        ra += wv::swap_half(ra); r0 += wv::swap_half(r0); r1 += wv::swap_half(r1); r2 += wv::swap_half(r2);
        ra += Wg[L.b_a]; r0 += Wg[L.b_oc]; r1 += Wg[L.b_oc + 1]; r2 += Wg[L.b_oc + 2];
        if (valid && hi == 0) {
            float* row = cb + pt * 8;
            row[6] = a.z[obj * a.z_so + ray * a.z_sr + smp * a.z_ss];
            row[0] = sigmoidf_acc(ra * 10.0f);
            row[1] = sigmoidf_acc(r0);
            row[2] = sigmoidf_acc(r1);
            row[3] = sigmoidf_acc(r2);
        }
    }
    __syncthreads();
    {
        const StepArgs& al = wv::kernarg_late(ga).s;
        composite_phase<BWD>(al, cb, lds + LdsGen::LOSS, obj, ray0, nrays, wave, lane, tid,
                             load_ray_meta(al, obj, ray0 + min(4 * wave + (lane >> 4), nrays - 1)));
    }
    __syncthreads();

    if (BWD) {
    float d_raw, d_c0, d_c1, d_c2;
    {
        const float* row = cb + pt * 8;
        d_raw = row[0]; d_c0 = row[1]; d_c1 = row[2]; d_c2 = row[3];
    }
    float dproj[kDirs];
#pragma unroll
    for (int d = 0; d < kDirs; ++d) dproj[d] = 0.0f;

    // one reduced weight-gradient block: acc -> staged cross-wave sum -> this wave's quarter -> partial buffer
    auto emit = [&](float* tens, int K, int row0, int col0, int ncols) {
        float q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        { float* stg_ = (stage_toggle & 1) ? stg1 : stg0; stage_put(stg_, acc, wave, p31, hi); __syncthreads(); stage_get(q, stg_, wave, p31, hi); }
        ++stage_toggle;
        store_quarter_rt(tens + (long long)row0 * K, K, q, col0, ncols, !first_pass, wave, p31, hi);
    };
    // weight gradient of one layer: delta blocks D_F(ds, ob) x input blocks; bias gradient from the same F-forms
    // weight-gradient blocks of one output-row block: delta image (F-form) loaded once, the input image of block i+1
    // requested before the chain of block i
    struct XSeg { int blk, col0, ncols; };
    auto dw_row = [&](int dfblk, int n, auto xs, float* tens, int K, int row0) {
        float df[16], xa[16], xb[16];
        ldb(df, BLK(dfblk), lane);
        XSeg s = xs(0), sn = s;
        ldb(xa, BLK(s.blk), lane);
        int i = 0;
        for (; i + 1 < n; i += 2) {
            sn = xs(i + 1);
            ldb(xb, BLK(sn.blk), lane);
            zero_acc(acc); dw_mm(acc, df, xa);
            emit(tens, K, row0, s.col0, s.ncols);
            s = sn;
            if (i + 2 < n) { sn = xs(i + 2); ldb(xa, BLK(sn.blk), lane); }
            zero_acc(acc); dw_mm(acc, df, xb);
            emit(tens, K, row0, s.col0, s.ncols);
            s = sn;
        }
        if (i < n) {
            zero_acc(acc); dw_mm(acc, df, xa);
            emit(tens, K, row0, s.col0, s.ncols);
        }
    };
    // store a delta block (already masked) as P-form + F-form and add its bias gradient
    auto put_delta = [&](int ds, int kb, int bias_off) {
        stb(BLK(D_P + ds * NB + kb), xv, lane);
        toF_put(scrD, xv, p31, hi); toF_get(yv, scrD, p31, hi);
        stb(BLK(D_F + ds * NB + kb), yv, lane);
        add_db(Gv + bias_off + 32 * kb, yv, p31, hi);
    };
    // two delta rows share every input image (paired form of dw_row)
    auto dw_row2 = [&](int dfblk0, int dfblk1, int n, auto xs, float* tens, int K, int row0) {
        float df0[16], df1[16], xa[16], xb[16];
        ldb(df0, BLK(dfblk0), lane);
        ldb(df1, BLK(dfblk1), lane);
        XSeg s = xs(0), sn = s;
        ldb(xa, BLK(s.blk), lane);
        for (int i = 0; i < n; i += 2) {
            if (i + 1 < n) { sn = xs(i + 1); ldb(xb, BLK(sn.blk), lane); }
            zero_acc(acc); dw_mm(acc, df0, xa);
            emit(tens, K, row0, s.col0, s.ncols);
            zero_acc(acc); dw_mm(acc, df1, xa);
            emit(tens, K, row0 + 32, s.col0, s.ncols);
            if (i + 1 >= n) break;
            s = sn;
            if (i + 2 < n) { sn = xs(i + 2); ldb(xa, BLK(sn.blk), lane); }
            zero_acc(acc); dw_mm(acc, df0, xb);
            emit(tens, K, row0, s.col0, s.ncols);
            zero_acc(acc); dw_mm(acc, df1, xb);
            emit(tens, K, row0 + 32, s.col0, s.ncols);
            s = sn;
        }
    };
    auto dw_rows = [&](int dfbase, int n, auto xs, float* tens, int K) {          // all NB delta rows of a layer, in pairs
        for (int ob = 0; ob < NB; ob += 2) {
            if (ob + 1 < NB) dw_row2(dfbase + ob, dfbase + ob + 1, n, xs, tens, K, 32 * ob);
            else dw_row(dfbase + ob, n, xs, tens, K, 32 * ob);
        }
    };
    // delta of a hidden layer: d h(l) block kb = init + sum_ob W[ob rows][kb cols]^T D(dsrc, ob), masked by h(l, kb); two
    // output blocks per pass over the delta images
    auto d_hidden = [&](auto init, const float* wb, int ldw, int dsrc, int hl, int ddst, int bias_off) {
        for (int kb = 0; kb < NB; kb += 2) {
            init(acc, kb);
            if (kb + 1 < NB) {
                init(acc1, kb + 1);
                chain_bwd2(acc, acc1, NB, ldw, [&](int ob) {
                    const float* c = wb + (32 * ob + 4 * hi) * ldw + 32 * kb + p31;
                    return BSeg2{c, c + 32, BLK(D_P + dsrc * NB + ob)}; }, lane);
            } else {
                chain_bwd(acc, NB, ldw, [&](int ob) { return BSeg{wb + (32 * ob + 4 * hi) * ldw + 32 * kb + p31, BLK(D_P + dsrc * NB + ob)}; }, lane);
            }
            for (int u = 0; u < 2 && kb + u < NB; ++u) {
                ldb(yv, BLK(H_P + hl * NB + kb + u), lane);
#pragma unroll
                for (int r = 0; r < 16; ++r) xv[r] = yv[r] > 0.0f ? (u ? acc1[r] : acc[r]) : 0.0f;
                put_delta(ddst, kb + u, bias_off);
            }
        }
    };
    auto zero_init = [&](f32x16& v, int) { zero_acc(v); };

    // ---- heads: out_alpha / out_color gradients; delta of color_linear's output -> D(0) ----
    for (int kb = 0; kb < NB; ++kb) {
        ldb(xv, BLK(H_F + 3 * NB + kb), lane);      // h4 F-form
        ldb(yv, BLK(H_F + 4 * NB + kb), lane);      // hc F-form
        float gA = 0.0f, g0 = 0.0f, g1 = 0.0f, g2 = 0.0f, sa = 0.0f, s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* row = cbw + (r + 16 * hi) * 8;
            const float da = row[0], q0 = row[1], q1 = row[2], q2 = row[3];
            gA = fmaf(da, xv[r], gA);
            g0 = fmaf(q0, yv[r], g0); g1 = fmaf(q1, yv[r], g1); g2 = fmaf(q2, yv[r], g2);
            sa += da; s0 += q0; s1 += q1; s2 += q2;
        }
        gA += wv::swap_half(gA); g0 += wv::swap_half(g0); g1 += wv::swap_half(g1); g2 += wv::swap_half(g2);
        sa += wv::swap_half(sa); s0 += wv::swap_half(s0); s1 += wv::swap_half(s1); s2 += wv::swap_half(s2);
        if (hi == 0) {
            const int j = 32 * kb + p31;
            Gv[L.w_a + j] += gA;
            Gv[L.w_oc + j] += g0;
            Gv[L.w_oc + H + j] += g1;
            Gv[L.w_oc + 2 * H + j] += g2;
            if (p31 == 0 && kb == 0) {
                Gv[L.b_a] += sa; Gv[L.b_oc + 0] += s0; Gv[L.b_oc + 1] += s1; Gv[L.b_oc + 2] += s2;
            }
        }
        ldb(yv, BLK(H_P + 4 * NB + kb), lane);      // hc P-form for the ReLU mask
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = 32 * kb + phi(r, hi);
            const float v = Wg[L.w_oc + j] * d_c0 + Wg[L.w_oc + H + j] * d_c1 + Wg[L.w_oc + 2 * H + j] * d_c2;
            xv[r] = yv[r] > 0.0f ? v : 0.0f;
        }
        put_delta(0, kb, L.b_c);
    }
    // ---- color_linear: dW = D(0)^T [h4 | e2] ; d h4 -> D(1) ; d e2 -> dproj ----
    {
        float* tens = out + L.f[10];
        const int K = H + kEmb2;
        dw_rows(D_F + 0 * NB, NB + 2, [&](int i) { return i < NB ? XSeg{H_F + 3 * NB + i, 32 * i, 32}
                                                                 : XSeg{E_F + 3 + (i - NB), H + 32 * (i - NB), i == NB ? 32 : kEmb2 - 32}; }, tens, K);
        // d h4 = W_a d raw + W_c[:, :H]^T D(0), masked by h4
        d_hidden([&](f32x16& v, int kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = Wg[L.w_a + 32 * kb + phi(r, hi)] * d_raw; }, Wg + L.w_c, L.ld_c, 0, 3, 1, L.b_m2);
        {   // d e2: both blocks in one pass over D(0)
            zero_acc(acc); zero_acc(acc1);
            chain_bwd2(acc, acc1, NB, L.ld_c, [&](int ob) {
                const float* c = Wg + L.w_c + (32 * ob + 4 * hi) * L.ld_c + H;
                return BSeg2{c + p31, c + min(32 + p31, 46), BLK(D_P + 0 * NB + ob)}; }, lane);
            ldb(yv, BLK(CFB + 3), lane);
            pe_block_bwd<16>(dproj, acc, yv, kEmb1, kEmb2, 0, hi);
            ldb(yv, BLK(CFB + 4), lane);
            pe_block_bwd<6>(dproj, acc1, yv, kEmb1, kEmb2, 1, hi);
        }
    }
    // ---- mid2: delta D(1), input h3 ; d h3 -> D(0) ----
    {
        float* tens = out + L.f[6];
        dw_rows(D_F + 1 * NB, NB, [&](int i) { return XSeg{H_F + 2 * NB + i, 32 * i, 32}; }, tens, H);
        d_hidden(zero_init, Wg + L.w_m2, L.ld_m, 1, 2, 0, L.b_cat);
    }
    // ---- cat_layer: delta D(0), input [h2 | e1] ; d h2 -> D(1) ; d e1 -> DE (accumulators) ----
    {
        float* tens = out + L.f[4];
        const int K = H + kEmb1;
        dw_rows(D_F + 0 * NB, NB + 3, [&](int i) { return i < NB ? XSeg{H_F + 1 * NB + i, 32 * i, 32}
                                                                 : XSeg{E_F + (i - NB), H + 32 * (i - NB), i - NB < 2 ? 32 : kEmb1 - 64}; }, tens, K);
        d_hidden(zero_init, Wg + L.w_cat, L.ld_cat, 0, 1, 1, L.b_m1);
        {   // d e1 (cat_layer part): blocks 0, 1 in one pass over D(0), block 2 in another
            zero_acc(acc); zero_acc(acc1);
            chain_bwd2(acc, acc1, NB, L.ld_cat, [&](int ob) {
                const float* c = Wg + L.w_cat + (32 * ob + 4 * hi) * L.ld_cat + H + p31;
                return BSeg2{c, c + 32, BLK(D_P + 0 * NB + ob)}; }, lane);
            stacc(BLK(DE + 0), acc, lane);
            stacc(BLK(DE + 1), acc1, lane);
            zero_acc(acc);
            chain_bwd(acc, NB, L.ld_cat, [&](int ob) { return BSeg{Wg + L.w_cat + (32 * ob + 4 * hi) * L.ld_cat + H + min(64 + p31, 88), BLK(D_P + 0 * NB + ob)}; }, lane);
            stacc(BLK(DE + 2), acc, lane);
        }
    }
    // ---- mid1: delta D(1), input h1 ; d h1 -> D(0) ----
    {
        float* tens = out + L.f[2];
        dw_rows(D_F + 1 * NB, NB, [&](int i) { return XSeg{H_F + 0 * NB + i, 32 * i, 32}; }, tens, H);
        d_hidden(zero_init, Wg + L.w_m1, L.ld_m, 1, 0, 0, L.b_in);
    }
    // ---- in_layer: delta D(0), input e1 ; d e1 += ... ; encoding backward ----
    {
        float* tens = out + L.f[0];
        dw_rows(D_F + 0 * NB, 3, [&](int i) { return XSeg{E_F + i, 32 * i, i < 2 ? 32 : kEmb1 - 64}; }, tens, kEmb1);
        ldb(xv, BLK(DE + 0), lane);
        ldb(yv, BLK(DE + 1), lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = xv[r]; acc1[r] = yv[r]; }
        chain_bwd2(acc, acc1, NB, L.ld_in, [&](int ob) {
            const float* c = Wg + L.w_in + (32 * ob + 4 * hi) * L.ld_in + p31;
            return BSeg2{c, c + 32, BLK(D_P + 0 * NB + ob)}; }, lane);
        ldb(yv, BLK(CFB + 0), lane);
        pe_block_bwd<16>(dproj, acc, yv, 0, kEmb1, 0, hi);
        ldb(yv, BLK(CFB + 1), lane);
        pe_block_bwd<16>(dproj, acc1, yv, 0, kEmb1, 1, hi);
        ldb(xv, BLK(DE + 2), lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = xv[r];
        chain_bwd(acc, NB, L.ld_in, [&](int ob) { return BSeg{Wg + L.w_in + (32 * ob + 4 * hi) * L.ld_in + min(64 + p31, 88), BLK(D_P + 0 * NB + ob)}; }, lane);
        ldb(yv, BLK(CFB + 2), lane);
        pe_block_bwd<12>(dproj, acc, yv, 0, kEmb1, 2, hi);
    }
    // ---- B_layer.weight gradient ----
    {
#pragma unroll
        for (int d = 0; d < kDirs; ++d) dproj[d] += wv::swap_half(dproj[d]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f0 = phi(r, 0), f1 = phi(r, 1);
            const float v0 = f0 < kDirs ? dproj[f0 < kDirs ? f0 : 0] : 0.0f;
            const float v1 = f1 < kDirs ? dproj[f1 < kDirs ? f1 : 0] : 0.0f;
            xv[r] = hi ? v1 : v0;
        }
        toF_put(scrD, xv, p31, hi); toF_get(yv, scrD, p31, hi);
        ldb(xv, BLK(E_F + 0), lane);
        zero_acc(acc);
        dw_mm(acc, yv, xv);
        float q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        { float* stg_ = (stage_toggle & 1) ? stg1 : stg0; stage_put(stg_, acc, wave, p31, hi); __syncthreads(); stage_get(q, stg_, wave, p31, hi); }
        ++stage_toggle;
        if (p31 < 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d = 8 * wave + 4 * hi + i;
                if (d < kDirs) {
                    float* o = out + L.f[14] + 3 * d + p31;
                    *o = first_pass ? q[i] : *o + q[i];
                }
            }
        }
    }
    }   // BWD
    }   // pass loop
    __syncthreads();
    if (tid == 0) {
        float* pl = a.part_loss + (obj * a.NW + wgo) * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            pl[k] = (lds[LdsGen::LOSS + k] + lds[LdsGen::LOSS + 4 + k]) + (lds[LdsGen::LOSS + 8 + k] + lds[LdsGen::LOSS + 12 + k]);
        pl[3] = 0.0f;
    }
    if (!BWD) return;
    // small vectors: sum of the four waves' private accumulators, image order -> flat order
    for (int sv = tid; sv < L.small_n; sv += kWG) {
        const float* v = lds + LdsGen::VEC + sv;
        const float g = (v[0] + v[L.small_n]) + (v[2 * L.small_n] + v[3 * L.small_n]);
        const int i = L.b_in + sv;
        int o = -1;
        if (i < L.b_m1) o = L.f[1] + (i - L.b_in);
        else if (i < L.b_cat) o = L.f[3] + (i - L.b_m1);
        else if (i < L.b_m2) o = L.f[5] + (i - L.b_cat);
        else if (i < L.b_c) o = L.f[7] + (i - L.b_m2);
        else if (i < L.w_a) o = L.f[11] + (i - L.b_c);
        else if (i < L.w_oc) o = L.f[8] + (i - L.w_a);
        else if (i < L.b_a) o = L.f[12] + (i - L.w_oc);
        else if (i == L.b_a) o = L.f[9];
        else if (i >= L.b_oc && i < L.b_oc + 3) o = L.f[13] + (i - L.b_oc);
        if (o >= 0) out[o] = g;
    }
#undef BLK
}

}  // namespace vk
