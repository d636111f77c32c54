// query_kernels.h - inference query of one object field at arbitrary points.
//
// Replaces the chunked torch.no_grad() loop of the reference's Trainer.eval_points (trainer.py:77-95: pe -> fc_occ_map
// -> sigmoid, 100 000 points per chunk) that mesh extraction calls on a dense grid (render_rays.py:98-122, up to 256^3
// points per object).  Hidden 32: field_query_s32 (query_split_kernels.h: the forward of step_main_s32 on the bf16 matrix
// pipe, float32-equivalent; it replaced the exact-fp32 field_query_h32 of rounds 1-2).  Any other width 32 k <= 256:
// field_query_gen below (exact-fp32 matrix instruction, weights streamed from the object's L2-resident image).
// Persistent workgroups, each wave streams 32-point tiles; 16 B written per point against 22.3 kFLOP at hidden 32.
#pragma once
#include "step_kernels.h"

namespace vk {

struct QueryArgs {
    const float* wimg;                 // packed parameter image of the object (Lds32 layout)
    const float* scale;                // pe.scale (1 float)
    const float* pts; long long pts_sn, pts_sc;    // [N,3] with element strides
    long long n_pts;
    float* occ;                        // [N]    sigmoid(alpha)   (render_rays.py:4-8 occupancy_activation)
    float* rgb;                        // [N,3]  sigmoid(raw colour)
};

// Any hidden width H = 32 * NB (NB = 2..8: the background model's 128, iMAP's 256, ...).  Same arithmetic; the weights
// are read straight from the object's parameter image in global memory (L2-resident: 0.4-1.3 MB), 16 bytes per lane and
// four matrix instructions, and a tile's activations stay on chip: two ping-pong sets of NB x 16 values per lane, both in
// registers up to NB = 4; from NB = 5 (where two sets no longer fit the 512-register file) the second set lives in a
// wave-private LDS area in register-image form (NB x 4 KiB per wave; a lane only re-reads what it wrote itself).
template <int NB>
__global__ __launch_bounds__(kWG, 1) void field_query_gen(const QueryArgs a) {
    constexpr GenLayout L = gen_layout(32 * NB);
    constexpr int H = 32 * NB;
    constexpr bool LDSB = NB > 4;
    const int tid_k = threadIdx.x;
    const float* Wg = a.wimg;
    const float scale = a.scale[0];
    const float* Bg = Wg + L.pe_b;
    for (long long chunk = blockIdx.x; chunk * kMaxPts < a.n_pts; chunk += gridDim.x) {
        // lane coordinates opaque per tile: otherwise the ~5 NB^2 weight row addresses of the body are loop-invariant,
        // hoisted in front of the loop and spilled (same effect as in step_main_h32's multi-pass loop)
        const int tid = wv::opaque_iter(tid_k), lane = tid & 63, wave = tid >> 6, p31 = lane & 31, hi = lane >> 5;
        float* hBl = wv::lds_base() + wave * (NB * 1024);      // LDSB: set B, block kb at hBl + kb * 1024, [r][lane]
        const long long pt = chunk * kMaxPts + wave * 32 + p31;
        const bool valid = pt < a.n_pts;
        float t[3] = {0.0f, 0.0f, 0.0f};
        if (valid) {
            const float* px = a.pts + pt * a.pts_sn;
            t[0] = px[0] / scale;
            t[1] = px[a.pts_sc] / scale;
            t[2] = px[2 * a.pts_sc] / scale;
        }
        float proj[kDirs];
#pragma unroll
        for (int d = 0; d < kDirs; ++d)
            proj[d] = fmaf(t[2], Bg[3 * d + 2], fmaf(t[1], Bg[3 * d + 1], t[0] * Bg[3 * d]));
        float amax = 0.0f;
#pragma unroll
        for (int d = 0; d < kDirs; ++d) amax = fmaxf(amax, fabsf(proj[d]));
        const bool big = wv::wave_any(!(amax * (32.0f * kPi) < kSinCosFastLimit));
        float e1a[16], e1b[16], e1c[16], cf[16];
        if (__builtin_expect(!big, 1)) {
            pe_block<16, false>(e1a, cf, 0, kEmb1, 0, t, proj, hi);
            pe_block<16, false>(e1b, cf, 0, kEmb1, 1, t, proj, hi);
            pe_block<12, false>(e1c, cf, 0, kEmb1, 2, t, proj, hi);
        } else {
            pe_block<16, true>(e1a, cf, 0, kEmb1, 0, t, proj, hi);
            pe_block<16, true>(e1b, cf, 0, kEmb1, 1, t, proj, hi);
            pe_block<12, true>(e1c, cf, 0, kEmb1, 2, t, proj, hi);
        }
        float hA[NB][16], hB[LDSB ? 1 : NB][16];
        float tmp[16];
        f32x16 acc;
        // set B accessors: registers or the wave's LDS area
        auto putB = [&](int ob, const f32x16& v) {
            if constexpr (LDSB) {
                relu_to(tmp, v);
#pragma unroll
                for (int r = 0; r < 16; ++r) hBl[ob * 1024 + (r >> 2) * 256 + lane * 4 + (r & 3)] = tmp[r];   // 16-byte chunks per lane
            } else {
                relu_to(hB[ob], v);
            }
        };
        auto mmB = [&](f32x16& c, const float* w, int kb) {
            if constexpr (LDSB) {
#pragma unroll
                for (int r = 0; r < 16; ++r) tmp[r] = hBl[kb * 1024 + (r >> 2) * 256 + lane * 4 + (r & 3)];
                fwd_mm<4>(c, w, tmp);
            } else {
                fwd_mm<4>(c, w, hB[kb]);
            }
        };
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {           // in_layer -> hA
            const float* w = Wg + L.w_in + (32 * ob + p31) * L.ld_in + 4 * hi;
            load_bias(acc, Wg + L.b_in + 32 * ob, hi);
            fwd_mm<4>(acc, w, e1a); fwd_mm<4>(acc, w + 32, e1b); fwd_mm<3>(acc, w + 64, e1c);
            relu_to(hA[ob], acc);
        }
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {           // mid1 -> hB
            const float* w = Wg + L.w_m1 + (32 * ob + p31) * L.ld_m + 4 * hi;
            load_bias(acc, Wg + L.b_m1 + 32 * ob, hi);
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) fwd_mm<4>(acc, w + 32 * kb, hA[kb]);
            putB(ob, acc);
        }
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {           // cat_layer -> hA
            const float* w = Wg + L.w_cat + (32 * ob + p31) * L.ld_cat + 4 * hi;
            load_bias(acc, Wg + L.b_cat + 32 * ob, hi);
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) mmB(acc, w + 32 * kb, kb);
            fwd_mm<4>(acc, w + H, e1a); fwd_mm<4>(acc, w + H + 32, e1b); fwd_mm<3>(acc, w + H + 64, e1c);
            relu_to(hA[ob], acc);
        }
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {           // mid2 -> hB (= fc4)
            const float* w = Wg + L.w_m2 + (32 * ob + p31) * L.ld_m + 4 * hi;
            load_bias(acc, Wg + L.b_m2 + 32 * ob, hi);
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) fwd_mm<4>(acc, w + 32 * kb, hA[kb]);
            putB(ob, acc);
        }
        {
            float e2a[16], e2b[16];
            if (__builtin_expect(!big, 1)) {
                pe_block<16, false>(e2a, cf, kEmb1, kEmb2, 0, t, proj, hi);
                pe_block<6, false>(e2b, cf, kEmb1, kEmb2, 1, t, proj, hi);
            } else {
                pe_block<16, true>(e2a, cf, kEmb1, kEmb2, 0, t, proj, hi);
                pe_block<6, true>(e2b, cf, kEmb1, kEmb2, 1, t, proj, hi);
            }
#pragma unroll
            for (int ob = 0; ob < NB; ++ob) {       // color_linear -> hA
                const float* w = Wg + L.w_c + (32 * ob + p31) * L.ld_c + 4 * hi;
                load_bias(acc, Wg + L.b_c + 32 * ob, hi);
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) mmB(acc, w + 32 * kb, kb);
                fwd_mm<4>(acc, w + H, e2a); fwd_mm<2>(acc, w + H + 32, e2b);
                relu_to(hA[ob], acc);
            }
        }
        float ra = 0.0f, r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            if constexpr (LDSB) {
#pragma unroll
                for (int r = 0; r < 16; ++r) tmp[r] = hBl[kb * 1024 + (r >> 2) * 256 + lane * 4 + (r & 3)];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = 32 * kb + phi(r, hi);
                ra = fmaf(Wg[L.w_a + j], LDSB ? tmp[r] : hB[LDSB ? 0 : kb][r], ra);
                r0 = fmaf(Wg[L.w_oc + j], hA[kb][r], r0);
                r1 = fmaf(Wg[L.w_oc + H + j], hA[kb][r], r1);
                r2 = fmaf(Wg[L.w_oc + 2 * H + j], hA[kb][r], r2);
            }
        }
        ra += wv::swap_half(ra); r0 += wv::swap_half(r0); r1 += wv::swap_half(r1); r2 += wv::swap_half(r2);
        ra += Wg[L.b_a]; r0 += Wg[L.b_oc]; r1 += Wg[L.b_oc + 1]; r2 += Wg[L.b_oc + 2];
        if (valid && hi == 0) {
            a.occ[pt] = sigmoidf_acc(ra * 10.0f);
            a.rgb[3 * pt + 0] = sigmoidf_acc(r0);
            a.rgb[3 * pt + 1] = sigmoidf_acc(r1);
            a.rgb[3 * pt + 2] = sigmoidf_acc(r2);
        }
    }
}

}  // namespace vk
