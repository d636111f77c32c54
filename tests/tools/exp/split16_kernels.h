// split16_kernels.h - PROTOTYPE (forward / render only): hidden 32 on 16-point tiles, v_mfma_f32_16x16x32_bf16, eight waves per
// workgroup = two independent tiles per SIMD.  Same numerics as split_kernels.h (float32 = hi + mid + lo bfloat16 planes, six
// products).  Why: a SIMD's time is the sum of its waves' matrix and VALU time, and step_main_s32's wave spends
// 36 % of its pass parked on LDS / memory / barrier waits that only a second, independent tile on the same SIMD can fill.
// This file measures how much of that is real before the backward pass is re-derived for 16-wide tiles.
//
// Tile: lane (p = l & 15, g = l >> 4) = point p, feature group g.  An activation (32 features) is 8 registers per lane:
// feature 16 ob + 4 g + r (the result layout of the two 16-row matrix products ob = 0, 1) - and read as a K = 32 operand the
// same 8 registers ARE the lane's slice k = 8 g + t  <->  feature 16 (t >> 2) + 4 g + (t & 3): layers chain without data
// movement, one 32-deep step per hidden layer.  The encoding is owned by lane group: g = 0 directions 0..5, g = 1 6..10 + (x, y,
// z, 1), g = 2 11..15, g = 3 16..20.
#pragma once
#include "split_kernels.h"

namespace vk {

constexpr int kWaves16 = 8, kWG16 = 64 * kWaves16;

struct Img16 {
    // forward image: 1 KiB chunks [64 lanes][8 bf16] per (layer, output half ob, 32-deep step s, plane); hidden step first
    static constexpr int ST_IN = 3, ST_M = 1, ST_CAT = 4, ST_C = 3;
    static constexpr int C_IN = 0, C_M1 = C_IN + 2 * ST_IN, C_CAT = C_M1 + 2 * ST_M, C_M2 = C_CAT + 2 * ST_CAT, C_C = C_M2 + 2 * ST_M,
                         C_N = C_C + 2 * ST_C;                              // 24 chunks
    static constexpr int W_BYTES = C_N * 3 * 1024;                          // 73 728
    static constexpr int SMALL = W_BYTES;                                   // float32: B_M1 32 | B_M2 32 | W_A 32 | W_OC 96 | B_A 4 | B_OC 4 | PE_B 64
    static constexpr int B_M1 = 0, B_M2 = 32, W_A = 64, W_OC = 96, B_A = 192, B_OC = 196, PE_B = 200, SMALL_N = 264;
    static constexpr int BYTES = 81920;                                     // 10 rounds of 8 x 1 KiB LDS-DMA
    static constexpr int ROUNDS = BYTES / (kWaves16 * 1024);
    static constexpr int ELEMS = C_N * 512;                                 // bf16 elements per plane
    // LDS map
    static constexpr int CB = BYTES;                                        // composite buffer [128 points][8]
    static constexpr int LOSS = CB + kMaxPts * 8 * 4;
    static constexpr int LDS_BYTES = LOSS + kWaves * 4 * 4;
};
static_assert(Img16::SMALL + Img16::SMALL_N * 4 <= Img16::BYTES, "image size");

// owner-lane encoding slots: lane group g, register R of the first group (24 per lane: R = 4 i + f, local direction i, octave f) /
// of the second group (16 per lane: R = 2 i + (f - 4), R >= 12 padding).  Returns the column inside the group's part of the
// embedding (embedding.py:85-89 order), kSlotOne / kSlotPad as in split_kernels.h.
__host__ __device__ constexpr int dir16(int g, int i) { return g == 0 ? i : (i < 5 ? 6 + 5 * (g - 1) + i : -1); }
__host__ __device__ constexpr int e1_slot16(int R, int g) {
    const int i = R >> 2, f = R & 3, d = dir16(g, i);
    if (d >= 0) return 3 + 21 * f + d;
    if (g == 1) return f < 3 ? f : kSlotOne;                                // (x, y, z, 1) behind g = 1's five directions
    return kSlotPad;
}
__host__ __device__ constexpr int e2_slot16(int R, int g) {
    if (R >= 12) return kSlotPad;
    const int i = R >> 1, f = R & 1, d = dir16(g, i);
    if (d >= 0) return 21 * f + d;
    if (g == 1) return f == 0 ? kSlotOne : kSlotPad;
    return kSlotPad;
}
__host__ __device__ constexpr int hidden_f16(int g, int t) { return 16 * (t >> 2) + 4 * g + (t & 3); }

// element x of a plane -> (tensor, offset); false = zero padding
__host__ __device__ inline bool img16_source(int x, int& t, int& o) {
    using I = Img16;
    const int chunk = x >> 9, lane = (x >> 3) & 63, tt = x & 7, j = lane & 15, g = lane >> 4;
    int base, steps, kind;
    if (chunk < I::C_M1) { base = I::C_IN; steps = I::ST_IN; kind = 0; }
    else if (chunk < I::C_CAT) { base = I::C_M1; steps = I::ST_M; kind = 1; }
    else if (chunk < I::C_M2) { base = I::C_CAT; steps = I::ST_CAT; kind = 2; }
    else if (chunk < I::C_C) { base = I::C_M2; steps = I::ST_M; kind = 3; }
    else { base = I::C_C; steps = I::ST_C; kind = 4; }
    const int ob = (chunk - base) / steps, s = (chunk - base) - ob * steps, row = 16 * ob + j;
    const int hf = hidden_f16(g, tt);
    switch (kind) {
        case 0: {
            const int c = e1_slot16(8 * s + tt, g);
            if (c == kSlotPad) return false;
            if (c == kSlotOne) { t = 1; o = row; } else { t = 0; o = row * kEmb1 + c; }
            return true;
        }
        case 1: t = 2; o = row * 32 + hf; return true;
        case 2: {
            if (s == 0) { t = 4; o = row * (32 + kEmb1) + hf; return true; }
            const int c = e1_slot16(8 * (s - 1) + tt, g);
            if (c == kSlotPad) return false;
            if (c == kSlotOne) { t = 5; o = row; } else { t = 4; o = row * (32 + kEmb1) + 32 + c; }
            return true;
        }
        case 3: t = 6; o = row * 32 + hf; return true;
        default: {
            if (s == 0) { t = 10; o = row * (32 + kEmb2) + hf; return true; }
            const int c = e2_slot16(8 * (s - 1) + tt, g);
            if (c == kSlotPad) return false;
            if (c == kSlotOne) { t = 11; o = row; } else { t = 10; o = row * (32 + kEmb2) + 32 + c; }
            return true;
        }
    }
}
__host__ __device__ inline bool img16_small_source(int i, int& t, int& o) {
    using I = Img16;
    if (i < I::B_M2) { t = 3; o = i - I::B_M1; return true; }
    if (i < I::W_A) { t = 7; o = i - I::B_M2; return true; }
    if (i < I::W_OC) { t = 8; o = i - I::W_A; return true; }
    if (i < I::B_A) { t = 12; o = i - I::W_OC; return true; }
    if (i < I::B_OC) { t = 9; o = i - I::B_A; return o < 1; }
    if (i < I::PE_B) { t = 13; o = i - I::B_OC; return o < 3; }
    t = 14; o = i - I::PE_B;
    return o < 63;
}

// step_prep_s16: mask statistics (blocks [0, prep_steps)) + image build, one thread per 4 plane elements
constexpr int kPack16Blocks = (Img16::ELEMS / 4 + 128 + kWG - 1) / kWG;
__global__ __launch_bounds__(kWG) void step_prep_s16(const StepArgs a) {
    using I = Img16;
    if ((int)blockIdx.x < a.prep_steps) {
        prep_stats(a, blockIdx.x, Flat32::P, a.PP);
        return;
    }
    const int b = blockIdx.x - a.prep_steps;
    const int k = b / kPack16Blocks;
    const int q = (b - k * kPack16Blocks) * kWG + threadIdx.x;
    char* img = reinterpret_cast<char*>(a.wimg) + (long long)k * I::BYTES;
    auto fetch = [&](int t, int o) { return t < kNFc ? a.fc[t].p[k * a.fc[t].stride + o] : a.pe_B.p[k * a.pe_B.stride + o]; };
    if (q < I::ELEMS / 4) {
        unsigned h[4], m[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int t, o;
            float f = 0.0f;
            if (img16_source(4 * q + e, t, o)) f = fetch(t, o);
            split3_scalar(f, h[e], m[e], l[e]);
            if (a.weights_bf16) { m[e] = 0u; l[e] = 0u; }
        }
        const int x = 4 * q, chunk = x >> 9, within = x & 511;
        char* base = img + chunk * 3 * 1024 + within * 2;
        *reinterpret_cast<u32x2*>(base) = u32x2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
        *reinterpret_cast<u32x2*>(base + 1024) = u32x2{m[0] | (m[1] << 16), m[2] | (m[3] << 16)};
        *reinterpret_cast<u32x2*>(base + 2048) = u32x2{l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
    } else {
        const int s0 = (q - I::ELEMS / 4) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (s0 + e < I::SMALL_N) {
                int t, o;
                float f = 0.0f;
                if (img16_small_source(s0 + e, t, o)) f = fetch(t, o);
                reinterpret_cast<float*>(img + I::SMALL)[s0 + e] = a.weights_bf16 ? round_bf16(f) : f;
            }
        }
    }
}

typedef wv::f32x4m f32x4;
// one 32-deep step for both output halves: acc[ob] += W[ob] . x; wchunk = chunk (ob = 0, step) of the layer in LDS (+ lane * 16),
// ob 1 is `ob_stride` bytes further
template <bool W3>
__device__ __forceinline__ void step16(f32x4 (&acc)[2], const char* wchunk, int ob_stride, u32x4 xh, u32x4 xm, u32x4 xl) {
    u32x4 wh[2], wm[2], wl[2];
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
        const char* c = wchunk + ob * ob_stride;
        wh[ob] = *reinterpret_cast<const u32x4*>(c);
        if (W3) { wm[ob] = *reinterpret_cast<const u32x4*>(c + 1024); wl[ob] = *reinterpret_cast<const u32x4*>(c + 2048); }
    }
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) acc[ob] = wv::mfma16_bf16(wh[ob], xl, acc[ob]);
    if (W3) {
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) acc[ob] = wv::mfma16_bf16(wl[ob], xh, acc[ob]);
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) acc[ob] = wv::mfma16_bf16(wm[ob], xm, acc[ob]);
    }
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) acc[ob] = wv::mfma16_bf16(wh[ob], xm, acc[ob]);
    if (W3) {
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) acc[ob] = wv::mfma16_bf16(wm[ob], xh, acc[ob]);
    }
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) acc[ob] = wv::mfma16_bf16(wh[ob], xh, acc[ob]);
}
__device__ __forceinline__ u32x4 quad(const unsigned* u) { return u32x4{u[0], u[1], u[2], u[3]}; }

// forward + compositing + loss of one pass (128 points = eight 16-point tiles); no backward (prototype)
template <bool W3>
__global__ __launch_bounds__(kWG16, 1) void step_main_s16_fwd(const StepArgs a) {
    using I = Img16;
    constexpr int H = 32;
    char* lds = reinterpret_cast<char*>(wv::lds_base());
    const float* SM = reinterpret_cast<const float*>(lds + I::SMALL);
    const int tid_k = threadIdx.x;
    const int obj = blockIdx.x / a.NW, wgo = blockIdx.x - obj * a.NW;
    float* loss_cells = reinterpret_cast<float*>(lds + I::LOSS);
    if (tid_k < kWaves * 4) loss_cells[tid_k] = 0.0f;
    float* cb = reinterpret_cast<float*>(lds + I::CB);
    const float scale = a.pe_scale.p[obj * a.pe_scale.stride];
    const char* gimg = reinterpret_cast<const char*>(a.wimg) + (long long)obj * I::BYTES;
    const float* Bg = reinterpret_cast<const float*>(gimg + I::SMALL) + I::PE_B;

    for (int grp = wgo; grp < a.NG; grp += a.NW) {
    const int tid = wv::opaque_iter(tid_k), lane = tid & 63, wave = tid >> 6, p = lane & 15, g = lane >> 4;
    __syncthreads();
    for (int i = tid; i < kMaxPts * 8; i += kWG16) cb[i] = 0.0f;
    const int ray0 = grp * a.G;
    const int nrays = min(a.G, a.R - ray0);
    const int npts = nrays * a.S;
    const int pt = wave * 16 + p;
    const bool valid = pt < npts;
    const int lray = valid ? pt / a.S : 0, smp = valid ? pt - lray * a.S : 0, ray = ray0 + lray;
    float px3[3] = {0.0f, 0.0f, 0.0f};
    if (valid) {
        const float* px = a.pcs + obj * a.pcs_so + ray * a.pcs_sr + smp * a.pcs_ss;
        px3[0] = px[0]; px3[1] = px[a.pcs_sc]; px3[2] = px[2 * a.pcs_sc];
    }
    if (grp == wgo) {                                                    // the parameter image -> LDS (asynchronous, lands during the encoding)
        const char* src = gimg + wave * 1024 + lane * 16;
#pragma unroll
        for (int c = 0; c < I::ROUNDS; ++c)
            wv::glds16(reinterpret_cast<const float*>(src + c * kWaves16 * 1024), reinterpret_cast<float*>(lds + c * kWaves16 * 1024 + wave * 1024));
    }
    const float t[3] = {px3[0] / scale, px3[1] / scale, px3[2] / scale};          // embedding.py:83
    // ---- encoding (embedding.py:82-91): this lane group's directions, octaves by double-angle recurrence ----
    unsigned e1h[12], e1m[12], e1l[12], e2h[8], e2m[8], e2l[8];
    {
        float proj[6];
        float amax = 0.0f;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int d = g == 0 ? i : min(6 + 5 * (g - 1) + i, 20);
            proj[i] = fmaf(t[2], Bg[3 * d + 2], fmaf(t[1], Bg[3 * d + 1], t[0] * Bg[3 * d]));      // embedding.py:84
            amax = fmaxf(amax, fabsf(proj[i]));
        }
        const bool fast = !wv::wave_any(!(amax * (32.0f * kPi) < kSinCosFastLimit));
        float e1[24], e2[16];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float s[6], c[6];
            const float a0 = proj[i] * kPi;
            if (__builtin_expect(fast, 1)) octave_sincos<false>(a0, s, c);
            else octave_sincos<true>(a0, s, c);
            const bool own = g == 0 || i < 5;
#pragma unroll
            for (int f = 0; f < 4; ++f) e1[4 * i + f] = own ? s[f] : 0.0f;
            e2[2 * i] = own ? s[4] : 0.0f; e2[2 * i + 1] = own ? s[5] : 0.0f;
        }
        if (g == 1) { e1[20] = t[0]; e1[21] = t[1]; e1[22] = t[2]; e1[23] = 1.0f; e2[10] = 1.0f; }
#pragma unroll
        for (int i = 12; i < 16; ++i) e2[i] = 0.0f;
        split_planes<24, 3>(e1, e1h, e1m, e1l);
        split_planes<16, 3>(e2, e2h, e2m, e2l);
    }
    __syncthreads();                                                     // parameter image landed, composite buffer zeroed
    // ---- field MLP forward (model.py:59-83): per layer two output halves, 32-deep steps ----
    const char* W = lds + lane * 16;
    auto chunk = [&](int base, int steps, int s) { return W + (base + s) * 3072; };
    f32x4 acc[2];
    unsigned hh[4], hm[4], hl[4];                                        // planes of the current hidden activation (8 values)
    float hf[8];
    auto zero2 = [&]() {
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) acc[ob] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    };
    auto bias2 = [&](int off) {
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ob][r] = SM[off + 16 * ob + 4 * g + r];
    };
    auto relu_split = [&]() {
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int r = 0; r < 4; ++r) hf[4 * ob + r] = wv::relu(acc[ob][r]);
        split_planes<8, 3>(hf, hh, hm, hl);
    };
    zero2();                                                             // :59 in_layer (bias rides in the constant-1 column)
#pragma unroll
    for (int s = 0; s < 3; ++s) step16<W3>(acc, chunk(I::C_IN, I::ST_IN, s), I::ST_IN * 3072, quad(e1h + 4 * s), quad(e1m + 4 * s), quad(e1l + 4 * s));
    relu_split();
    bias2(I::B_M1);                                                      // :60 mid1
    step16<W3>(acc, chunk(I::C_M1, I::ST_M, 0), I::ST_M * 3072, quad(hh), quad(hm), quad(hl));
    relu_split();
    zero2();                                                             // :63-64 cat_layer
    step16<W3>(acc, chunk(I::C_CAT, I::ST_CAT, 0), I::ST_CAT * 3072, quad(hh), quad(hm), quad(hl));
#pragma unroll
    for (int s = 0; s < 3; ++s) step16<W3>(acc, chunk(I::C_CAT, I::ST_CAT, 1 + s), I::ST_CAT * 3072, quad(e1h + 4 * s), quad(e1m + 4 * s), quad(e1l + 4 * s));
    relu_split();
    bias2(I::B_M2);                                                      // :67 mid2
    step16<W3>(acc, chunk(I::C_M2, I::ST_M, 0), I::ST_M * 3072, quad(hh), quad(hm), quad(hl));
    relu_split();
    float ra = 0.0f;                                                     // :71 out_alpha
#pragma unroll
    for (int i = 0; i < 8; ++i) ra = fmaf(SM[I::W_A + hidden_f16(g, i)], hf[i], ra);
    zero2();                                                             // :81 color_linear
    step16<W3>(acc, chunk(I::C_C, I::ST_C, 0), I::ST_C * 3072, quad(hh), quad(hm), quad(hl));
#pragma unroll
    for (int s = 0; s < 2; ++s) step16<W3>(acc, chunk(I::C_C, I::ST_C, 1 + s), I::ST_C * 3072, quad(e2h + 4 * s), quad(e2m + 4 * s), quad(e2l + 4 * s));
    float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;                               // :82 out_color
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float hc = wv::relu(acc[ob][r]);
            const int j = 16 * ob + 4 * g + r;
            r0 = fmaf(SM[I::W_OC + j], hc, r0);
            r1 = fmaf(SM[I::W_OC + H + j], hc, r1);
            r2 = fmaf(SM[I::W_OC + 2 * H + j], hc, r2);
        }
    // sums over the four lane groups of a point
    ra += wv::shfl(ra, lane ^ 16); r0 += wv::shfl(r0, lane ^ 16); r1 += wv::shfl(r1, lane ^ 16); r2 += wv::shfl(r2, lane ^ 16);
    ra += wv::shfl(ra, lane ^ 32); r0 += wv::shfl(r0, lane ^ 32); r1 += wv::shfl(r1, lane ^ 32); r2 += wv::shfl(r2, lane ^ 32);
    if (valid && g == 0) {
        float* row = cb + pt * 8;
        row[6] = a.z[obj * a.z_so + ray * a.z_sr + smp * a.z_ss];
        row[0] = sigmoidf_acc((ra + SM[I::B_A]) * 10.0f);                 // :77 raw*10 ; render_rays.py:6 sigmoid
        row[1] = sigmoidf_acc(r0 + SM[I::B_OC]);                          // :83
        row[2] = sigmoidf_acc(r1 + SM[I::B_OC + 1]);
        row[3] = sigmoidf_acc(r2 + SM[I::B_OC + 2]);
    }
    __syncthreads();
    if (wave < kWaves) {
        const StepArgs& al = wv::kernarg_late(a);
        composite_phase<false>(al, cb, loss_cells, obj, ray0, nrays, wave, lane, tid,
                               load_ray_meta(al, obj, ray0 + min(4 * wave + (lane >> 4), nrays - 1)));
    }
    }   // passes
    __syncthreads();
    if (tid_k == 0) {
        float* pl = a.part_loss + (obj * a.NW + wgo) * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            pl[k] = (loss_cells[k] + loss_cells[4 + k]) + (loss_cells[8 + k] + loss_cells[12 + k]);
        pl[3] = 0.0f;
    }
}

}  // namespace vk
