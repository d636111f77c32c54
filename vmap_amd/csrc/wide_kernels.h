// wide_kernels.h - fused training step for WIDE fields (hidden = 128, 256: the background model, iMAP), gfx950.
//
// step_main_gen gives every wave a 32-point tile of its own: at hidden 128 that is 4368 exact-fp32 matrix instructions
// on ONE wave (>= 116 us) while the background model's 525 tiles leave half of the chip's 1024 SIMDs idle.  Here a
// tile belongs to the WORKGROUP and the four waves split every layer by 32-wide OUTPUT block (wave w owns blocks
// w, w+4, ...): forward block ob, the delta block ob, the weight-gradient rows of block ob.  Inputs a wave does not own
// come from the other waves through the same lane-contiguous "register images" step_main_gen keeps in its scratch
// area (all four waves hold the same 32 points on the same lanes, so an image written by one wave is directly another
// wave's matrix operand), separated by workgroup barriers; the per-wave LDS transpose tiles stay private.  A weight-
// gradient block now has ONE producer per tile, so there is no cross-wave reduction: blocks are stored (first pass)
// or accumulated (later passes of the same workgroup) straight into the workgroup's partial buffer.
// Same arithmetic and summation orders per block as step_main_gen except for the order in which a workgroup's tiles
// are added (per-pass accumulation instead of a 4-tile staged sum).  model.py:54-85, loss.py:5-62 as cited there.
#pragma once
#include "gen_kernels.h"

namespace vk {

constexpr int kWideTile = 32;            // sample points per workgroup pass with one tile per workgroup (SPLIT = 4)

// full 32x32 block (lane = column, register r <-> row phi(r,hi)) -> row-major tensor; add = accumulate
__device__ __forceinline__ void store_block_rt(float* tens, int K, const f32x16& acc, int col0, int ncols, bool add,
                                               int p31, int hi) {
    if (p31 < ncols) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float* o = tens + (long long)phi(r, hi) * K + col0 + p31;
            *o = add ? *o + acc[r] : acc[r];
        }
    }
}

// LDS map (floats).  SPLIT = waves per tile, TPW = tiles per workgroup.
template <int SPLIT>
struct LdsWide {
    static constexpr int TPW = SPLIT == 4 ? 1 : 4;
    static constexpr int NWAVES = SPLIT * TPW;
    static constexpr int SCR = 0;                                        // one transpose tile per wave
    static constexpr int DPX = NWAVES * kDirs * 64;                      // d(proj) partials of all waves
    static constexpr int STG_N = TPW > 1 ? SPLIT * 2 * TPW * Lds32::STG_TILE : 0;   // per group: 2 buffers x TPW staged tiles
    static constexpr int STG = SCR + NWAVES * Lds32::SCR_TILE;           // staging; also holds DPX (used after the last block)
    static constexpr int CB = STG + (STG_N > DPX ? STG_N : DPX);
    static constexpr int LOSS = CB + kMaxPts * 8;
    static constexpr int VEC = LOSS + kWaves * 4;                        // TPW x small_n (every entry has one owner per tile)
    __host__ __device__ static constexpr int bytes(int small_n) { return (VEC + TPW * small_n) * 4; }
};

// SPLIT = 4: one 32-point tile per 256-thread workgroup, wave w owns output blocks w, w+4, ...; a weight-gradient block
//            has one producer per workgroup pass and is stored / accumulated directly.
// SPLIT = 2: four tiles per 512-thread workgroup (two waves per SIMD), the two waves of a tile own the even / odd output
//            blocks; weight-gradient blocks are summed over the four tiles through staged LDS tiles like step_main_gen.
template <bool BWD, int SPLIT>
__global__ __launch_bounds__(64 * LdsWide<SPLIT>::NWAVES, 1) void step_main_wide(const GenArgs ga) {
    using LW = LdsWide<SPLIT>;
    constexpr int TPW = LW::TPW, NWAVES = LW::NWAVES, NT = 64 * NWAVES;
    const StepArgs& a = ga.s;
    const GenLayout L = gen_layout(a.hidden);
    const int H = L.H, NB = L.NB;
    float* lds = wv::lds_base();
    const int tid_k = threadIdx.x;
    const int obj = blockIdx.x / a.NW, wgo = blockIdx.x - obj * a.NW;
    const float* Wg = a.wimg + (long long)obj * L.imgp;
    const int E_P = 0, E_F = 5, CFB = 10, H_P = 15, H_F = 15 + 5 * NB, D_P = 15 + 10 * NB, D_F = 15 + 12 * NB, DE = 15 + 14 * NB;
#define BLK(i) (sb + (long long)(i) * kBlk)
    if (BWD) {
        for (int i = tid_k; i < TPW * L.small_n; i += NT) lds[LW::VEC + i] = 0.0f;
    }
    if (tid_k < kWaves * 4) lds[LW::LOSS + tid_k] = 0.0f;
    float* out = a.part_grad + ((long long)(obj * a.NW + wgo)) * a.PP;
    float* cb = lds + LW::CB;
    float* dpx = lds + LW::STG;                                          // [NWAVES][kDirs][64]
    const float scale = a.pe_scale.p[obj * a.pe_scale.stride];
    const float* Bg = Wg + L.pe_b;
    int stage_toggle = 0;

    for (int grp = wgo; grp < a.NG; grp += a.NW) {
    const int tid = wv::opaque_iter(tid_k), lane = tid & 63, wave = tid >> 6, p31 = lane & 31, hi = lane >> 5;
    const int tile = wave / SPLIT, sw = wave - tile * SPLIT;             // this wave: sub-wave sw of tile `tile`
    float* sb = ga.scratch + ((long long)blockIdx.x * TPW + tile) * ga.wave_blocks * kBlk;   // one image set per tile
    float* Gv = lds + LW::VEC + tile * L.small_n - L.b_in;               // small-vector gradients of this tile
    float* scrX = lds + LW::SCR + wave * Lds32::SCR_TILE;
    float* scrD = scrX;                                                  // put / get pairs are immediate: one tile suffices
    float* stg = lds + LW::STG + sw * 2 * TPW * Lds32::STG_TILE;         // this group's two staging buffers (TPW > 1)
    const bool first_pass = grp == wgo;
    __syncthreads();                                                     // previous pass done with cb and the images
    for (int i = tid; i < kMaxPts * 8; i += NT) cb[i] = 0.0f;

    const int ray0 = grp * a.G;
    const int nrays = min(a.G, a.R - ray0);
    const int npts = nrays * a.S;                                        // <= 32 * TPW
    const int pt = tile * 32 + p31;                                      // the SPLIT waves of a tile hold the same 32 points
    const bool valid = pt < npts;
    const int lray = valid ? pt / a.S : 0;
    const int smp = valid ? pt - lray * a.S : 0;
    const int ray = ray0 + lray;
    float xv[16], yv[16];
    f32x16 acc;
    // ---- encoding: wave 0..2 the three blocks of e1, wave 3 the two of e2; P-form, F-form, cos factors -> images ----
    {
        float t[3] = {0.0f, 0.0f, 0.0f};
        if (valid) {
            float x0, x1, x2;
            load_point(a, obj, ray, smp, x0, x1, x2);
            t[0] = x0 / scale;
            t[1] = x1 / scale;
            t[2] = x2 / scale;
        }
        float proj[kDirs];
#pragma unroll
        for (int d = 0; d < kDirs; ++d)
            proj[d] = fmaf(t[2], Bg[3 * d + 2], fmaf(t[1], Bg[3 * d + 1], t[0] * Bg[3 * d]));
        float amax = 0.0f;
#pragma unroll
        for (int d = 0; d < kDirs; ++d) amax = fmaxf(amax, fabsf(proj[d]));
        const bool big = wv::wave_any(!(amax * (32.0f * kPi) < kSinCosFastLimit));
#define ENC(i, NS, base, limit, kb)                                                                      \
        {                                                                                                \
            if (__builtin_expect(!big, 1)) pe_block<NS, false>(xv, yv, base, limit, kb, t, proj, hi);                         \
            else pe_block<NS, true>(xv, yv, base, limit, kb, t, proj, hi);                               \
            stb(BLK(E_P + i), xv, lane); stb(BLK(CFB + i), yv, lane);                                    \
            toF_put(scrX, xv, p31, hi); toF_get(yv, scrX, p31, hi); stb(BLK(E_F + i), yv, lane);         \
        }
        // the five encoding blocks are dealt round-robin to the SPLIT waves of the tile
        if (0 % SPLIT == sw) ENC(0, 16, 0, kEmb1, 0)
        if (1 % SPLIT == sw) ENC(1, 16, 0, kEmb1, 1)
        if (2 % SPLIT == sw) ENC(2, 12, 0, kEmb1, 2)
        if (3 % SPLIT == sw) ENC(3, 16, kEmb1, kEmb2, 0)
        if (4 % SPLIT == sw) ENC(4, 6, kEmb1, kEmb2, 1)
#undef ENC
    }
    __syncthreads();

    // ---- field MLP forward: wave w produces output blocks ob = w, w+4, ... of every layer ----
    auto finish = [&](int l, int ob) {
        relu_to(xv, acc);
        stb(BLK(H_P + l * NB + ob), xv, lane);
        toF_put(scrX, xv, p31, hi); toF_get(yv, scrX, p31, hi);
        stb(BLK(H_F + l * NB + ob), yv, lane);
    };
    for (int ob = sw; ob < NB; ob += SPLIT) {           // :59 in_layer
        const float* w = Wg + L.w_in + (32 * ob + p31) * L.ld_in + 4 * hi;
        load_bias(acc, Wg + L.b_in + 32 * ob, hi);
        chain_fwd(acc, 3, [&](int i) { return FSeg{w + 32 * i, BLK(E_P + i)}; }, lane);   // zero weights/encodings pad block 2
        finish(0, ob);
    }
    __syncthreads();
    for (int ob = sw; ob < NB; ob += SPLIT) {           // :60 mid1
        const float* w = Wg + L.w_m1 + (32 * ob + p31) * L.ld_m + 4 * hi;
        load_bias(acc, Wg + L.b_m1 + 32 * ob, hi);
        chain_fwd(acc, NB, [&](int i) { return FSeg{w + 32 * i, BLK(H_P + 0 * NB + i)}; }, lane);
        finish(1, ob);
    }
    __syncthreads();
    for (int ob = sw; ob < NB; ob += SPLIT) {           // :63-64 cat_layer
        const float* w = Wg + L.w_cat + (32 * ob + p31) * L.ld_cat + 4 * hi;
        load_bias(acc, Wg + L.b_cat + 32 * ob, hi);
        chain_fwd(acc, NB + 3, [&](int i) { return i < NB ? FSeg{w + 32 * i, BLK(H_P + 1 * NB + i)} : FSeg{w + H + 32 * (i - NB), BLK(E_P + (i - NB))}; }, lane);
        finish(2, ob);
    }
    __syncthreads();
    for (int ob = sw; ob < NB; ob += SPLIT) {           // :67 mid2
        const float* w = Wg + L.w_m2 + (32 * ob + p31) * L.ld_m + 4 * hi;
        load_bias(acc, Wg + L.b_m2 + 32 * ob, hi);
        chain_fwd(acc, NB, [&](int i) { return FSeg{w + 32 * i, BLK(H_P + 2 * NB + i)}; }, lane);
        finish(3, ob);
    }
    __syncthreads();
    for (int ob = sw; ob < NB; ob += SPLIT) {           // :81 color_linear
        const float* w = Wg + L.w_c + (32 * ob + p31) * L.ld_c + 4 * hi;
        load_bias(acc, Wg + L.b_c + 32 * ob, hi);
        chain_fwd(acc, NB + 2, [&](int i) { return i < NB ? FSeg{w + 32 * i, BLK(H_P + 3 * NB + i)} : FSeg{w + H + 32 * (i - NB), BLK(E_P + 3 + (i - NB))}; }, lane);
        finish(4, ob);
    }
    __syncthreads();
    if (sw == 0) {   // heads (model.py:71,77,82-83): 4 dot products over all H features, one wave per tile
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
        if (wave < kWaves)                 // the compositing helper is written for four waves (16 rays per round)
            composite_phase<BWD>(al, cb, lds + LW::LOSS, obj, ray0, nrays, wave, lane, tid,
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

    // weight-gradient block (ob, x-block): one producer per tile -> straight into the workgroup's partial buffer
    // one weight-gradient block of this wave: with one tile per workgroup it has a single producer and goes straight to
    // the partial buffer; with four tiles it is summed over the tiles through this group's staged LDS tiles (the other
    // group emits its own block in the same barrier)
    auto emit = [&](float* tens, int K, int row0, int col0, int ncols) {
        if constexpr (TPW == 1) {
            store_block_rt(tens + (long long)row0 * K, K, acc, col0, ncols, !first_pass, p31, hi);
        } else {
            float q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            float* st = stg + (stage_toggle & 1) * TPW * Lds32::STG_TILE;
            stage_put(st, acc, tile, p31, hi);
            __syncthreads();
            stage_get(q, st, tile, p31, hi);
            ++stage_toggle;
            store_quarter_rt(tens + (long long)row0 * K, K, q, col0, ncols, !first_pass, tile, p31, hi);
        }
    };
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
    auto put_delta = [&](int ds, int kb, int bias_off) {
        stb(BLK(D_P + ds * NB + kb), xv, lane);
        toF_put(scrD, xv, p31, hi); toF_get(yv, scrD, p31, hi);
        stb(BLK(D_F + ds * NB + kb), yv, lane);
        add_db(Gv + bias_off + 32 * kb, yv, p31, hi);
    };
    const float* cbw = cb + tile * 32 * 8;                  // this tile's 32 rows of the composite buffer

    // ---- heads: gradients of out_alpha / out_color; delta of color_linear's output -> D(0) ----
    for (int kb = sw; kb < NB; kb += SPLIT) {
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
    __syncthreads();
    // ---- color_linear: dW = D(0)^T [h4 | e2] ; d h4 -> D(1) ; d e2 -> dproj (waves 0, 1) ----
    {
        float* tens = out + L.f[10];
        const int K = H + kEmb2;
        for (int ob = sw; ob < NB; ob += SPLIT) {
            dw_row(D_F + 0 * NB + ob, NB + 2, [&](int i) { return i < NB ? XSeg{H_F + 3 * NB + i, 32 * i, 32}
                                                                        : XSeg{E_F + 3 + (i - NB), H + 32 * (i - NB), i == NB ? 32 : kEmb2 - 32}; }, tens, K, 32 * ob);
        }
        for (int kb = sw; kb < NB; kb += SPLIT) {           // d h4 = W_a d raw + W_c[:, :H]^T D(0), masked by h4
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = Wg[L.w_a + 32 * kb + phi(r, hi)] * d_raw;
            chain_bwd(acc, NB, L.ld_c, [&](int ob) { return BSeg{Wg + L.w_c + (32 * ob + 4 * hi) * L.ld_c + 32 * kb + p31, BLK(D_P + 0 * NB + ob)}; }, lane);
            ldb(yv, BLK(H_P + 3 * NB + kb), lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) xv[r] = yv[r] > 0.0f ? acc[r] : 0.0f;
            put_delta(1, kb, L.b_m2);
        }
        for (int eb = 0; eb < 2; ++eb) {                        // d e2: the two blocks dealt to the tile's waves
            if ((eb + 1) % SPLIT != sw) continue;
            zero_acc(acc);
            const int col = eb == 0 ? p31 : min(32 + p31, 46);
            chain_bwd(acc, NB, L.ld_c, [&](int ob) { return BSeg{Wg + L.w_c + (32 * ob + 4 * hi) * L.ld_c + H + col, BLK(D_P + 0 * NB + ob)}; }, lane);
            ldb(yv, BLK(CFB + 3 + eb), lane);
            if (eb == 0) pe_block_bwd<16>(dproj, acc, yv, kEmb1, kEmb2, 0, hi);
            else pe_block_bwd<6>(dproj, acc, yv, kEmb1, kEmb2, 1, hi);
        }
    }
    __syncthreads();
    // ---- mid2: delta D(1), input h3 ; d h3 -> D(0) ----
    {
        float* tens = out + L.f[6];
        for (int ob = sw; ob < NB; ob += SPLIT)
            dw_row(D_F + 1 * NB + ob, NB, [&](int i) { return XSeg{H_F + 2 * NB + i, 32 * i, 32}; }, tens, H, 32 * ob);
        for (int kb = sw; kb < NB; kb += SPLIT) {
            zero_acc(acc);
            chain_bwd(acc, NB, L.ld_m, [&](int ob) { return BSeg{Wg + L.w_m2 + (32 * ob + 4 * hi) * L.ld_m + 32 * kb + p31, BLK(D_P + 1 * NB + ob)}; }, lane);
            ldb(yv, BLK(H_P + 2 * NB + kb), lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) xv[r] = yv[r] > 0.0f ? acc[r] : 0.0f;
            put_delta(0, kb, L.b_cat);
        }
    }
    __syncthreads();
    // ---- cat_layer: delta D(0), input [h2 | e1] ; d h2 -> D(1) ; d e1 -> DE images (waves 0..2) ----
    {
        float* tens = out + L.f[4];
        const int K = H + kEmb1;
        for (int ob = sw; ob < NB; ob += SPLIT) {
            dw_row(D_F + 0 * NB + ob, NB + 3, [&](int i) { return i < NB ? XSeg{H_F + 1 * NB + i, 32 * i, 32}
                                                                        : XSeg{E_F + (i - NB), H + 32 * (i - NB), i - NB < 2 ? 32 : kEmb1 - 64}; }, tens, K, 32 * ob);
        }
        for (int kb = sw; kb < NB; kb += SPLIT) {
            zero_acc(acc);
            chain_bwd(acc, NB, L.ld_cat, [&](int ob) { return BSeg{Wg + L.w_cat + (32 * ob + 4 * hi) * L.ld_cat + 32 * kb + p31, BLK(D_P + 0 * NB + ob)}; }, lane);
            ldb(yv, BLK(H_P + 1 * NB + kb), lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) xv[r] = yv[r] > 0.0f ? acc[r] : 0.0f;
            put_delta(1, kb, L.b_m1);
        }
        for (int eb = 0; eb < 3; ++eb) {
            if (eb % SPLIT != sw) continue;
            zero_acc(acc);
            const int col = eb < 2 ? 32 * eb + p31 : min(64 + p31, 88);
            chain_bwd(acc, NB, L.ld_cat, [&](int ob) { return BSeg{Wg + L.w_cat + (32 * ob + 4 * hi) * L.ld_cat + H + col, BLK(D_P + 0 * NB + ob)}; }, lane);
            stacc(BLK(DE + eb), acc, lane);                     // re-read by the same wave in the in_layer phase
        }
    }
    __syncthreads();
    // ---- mid1: delta D(1), input h1 ; d h1 -> D(0) ----
    {
        float* tens = out + L.f[2];
        for (int ob = sw; ob < NB; ob += SPLIT)
            dw_row(D_F + 1 * NB + ob, NB, [&](int i) { return XSeg{H_F + 0 * NB + i, 32 * i, 32}; }, tens, H, 32 * ob);
        for (int kb = sw; kb < NB; kb += SPLIT) {
            zero_acc(acc);
            chain_bwd(acc, NB, L.ld_m, [&](int ob) { return BSeg{Wg + L.w_m1 + (32 * ob + 4 * hi) * L.ld_m + 32 * kb + p31, BLK(D_P + 1 * NB + ob)}; }, lane);
            ldb(yv, BLK(H_P + 0 * NB + kb), lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) xv[r] = yv[r] > 0.0f ? acc[r] : 0.0f;
            put_delta(0, kb, L.b_in);
        }
    }
    __syncthreads();
    // ---- in_layer: delta D(0), input e1 ; d e1 += ... (waves 0..2) ; encoding backward ----
    {
        float* tens = out + L.f[0];
        for (int ob = sw; ob < NB; ob += SPLIT) {
            dw_row(D_F + 0 * NB + ob, 3, [&](int i) { return XSeg{E_F + i, 32 * i, i < 2 ? 32 : kEmb1 - 64}; }, tens, kEmb1, 32 * ob);
        }
        for (int eb = 0; eb < 3; ++eb) {
            if (eb % SPLIT != sw) continue;
            ldb(xv, BLK(DE + eb), lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = xv[r];
            const int col = eb < 2 ? 32 * eb + p31 : min(64 + p31, 88);
            chain_bwd(acc, NB, L.ld_in, [&](int ob) { return BSeg{Wg + L.w_in + (32 * ob + 4 * hi) * L.ld_in + col, BLK(D_P + 0 * NB + ob)}; }, lane);
            ldb(yv, BLK(CFB + eb), lane);
            if (eb == 0) pe_block_bwd<16>(dproj, acc, yv, 0, kEmb1, 0, hi);
            else if (eb == 1) pe_block_bwd<16>(dproj, acc, yv, 0, kEmb1, 1, hi);
            else pe_block_bwd<12>(dproj, acc, yv, 0, kEmb1, 2, hi);
        }
    }
    // ---- B_layer.weight gradient: sum the d(proj) partials of the tile's waves, dB = d(proj)^T t on its first wave ----
    __syncthreads();                                    // (TPW > 1) the last staged block is consumed: the area is reused
#pragma unroll
    for (int d = 0; d < kDirs; ++d) dpx[(wave * kDirs + d) * 64 + lane] = dproj[d];
    __syncthreads();
    if (sw == 0) {
        const float* px = dpx + (long long)tile * SPLIT * kDirs * 64 + lane;
#pragma unroll
        for (int d = 0; d < kDirs; ++d) {
            if constexpr (SPLIT == 4)
                dproj[d] = (px[d * 64] + px[(kDirs + d) * 64]) + (px[(2 * kDirs + d) * 64] + px[(3 * kDirs + d) * 64]);
            else
                dproj[d] = px[d * 64] + px[(kDirs + d) * 64];
        }
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
        dw_mm(acc, yv, xv);                             // rows = direction d (21 valid), cols 0..2 = xyz
    }
    if constexpr (TPW == 1) {
        if (sw == 0 && p31 < 3) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = phi(r, hi);
                if (d < kDirs) {
                    float* o = out + L.f[14] + 3 * d + p31;
                    *o = first_pass ? acc[r] : *o + acc[r];
                }
            }
        }
    } else {
        __syncthreads();                                // every tile has read its d(proj) partials
        float* st = lds + LW::STG;                      // group 0, buffer 0
        if (sw == 0) stage_put(st, acc, tile, p31, hi);
        __syncthreads();
        if (sw == 0 && p31 < 3) {
            float q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            stage_get(q, st, tile, p31, hi);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d = 8 * tile + 4 * hi + i;
                if (d < kDirs) {
                    float* o = out + L.f[14] + 3 * d + p31;
                    *o = first_pass ? q[i] : *o + q[i];
                }
            }
        }
    }
    }   // BWD
    }   // pass loop
#undef BLK
    __syncthreads();
    const int tid = tid_k;
    if (tid == 0) {
        float* pl = a.part_loss + (obj * a.NW + wgo) * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            pl[k] = (lds[LW::LOSS + k] + lds[LW::LOSS + 4 + k]) + (lds[LW::LOSS + 8 + k] + lds[LW::LOSS + 12 + k]);
        pl[3] = 0.0f;
    }
    if (!BWD) return;
    // small vectors: sum over the tiles of the workgroup, image order -> flat order
    for (int sv = tid; sv < L.small_n; sv += NT) {
        const float* v = lds + LW::VEC + sv;
        float g = v[0];
        if constexpr (TPW == 4) g = (v[0] + v[L.small_n]) + (v[2 * L.small_n] + v[3 * L.small_n]);
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
}

}  // namespace vk
