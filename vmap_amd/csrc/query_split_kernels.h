// query_split_kernels.h - the inference query at hidden 32 on the bf16 matrix pipe (round 3): the forward half of
// step_main_s32 (split_kernels.h: float32 = hi + mid + lo bfloat16 planes, six products per step = float32-equivalent;
// owner-lane encoding with the octave recurrence; the encoding halves of cat_layer / color_linear interleaved with the
// ReLU + split of the hidden layers) over arbitrary points of ONE object - what Trainer.eval_points (trainer.py:77-95)
// evaluates chunk by chunk for mesh extraction.  Replaces field_query_h32 (exact-fp32 matrix instruction: 174 x 64 clocks
// of matrix time per 32 points, on the vector pipe's lanes) by 138 x 32 clocks on the matrix pipe proper.
//
// Persistent workgroups: the object's split image (Img32s, 80 KiB, packed by step_prep_s32) is DMA'd into LDS once per
// workgroup, then every wave streams 32-point tiles.  28 B of traffic per point against 22.3 kFLOP: matrix / vector bound.
#pragma once
#include "query_kernels.h"
#include "split_kernels.h"

namespace vk {

constexpr int kQuerySplitLds = Img32s::BYTES;           // the image and nothing else: two workgroups fit a compute unit's 160 KiB

template <int = 0>
__global__ __launch_bounds__(kWG) WV_WAVES_PER_SIMD(2) void field_query_s32(const QueryArgs a) {
    using I = Img32s;
    constexpr int H = 32;
    char* lds = reinterpret_cast<char*>(wv::lds_base());
    const char* W = lds;
    const float* SM = reinterpret_cast<const float*>(lds + I::SMALL);
    const int tid_k = threadIdx.x;
    const float scale = a.scale[0];
    const char* gimg = reinterpret_cast<const char*>(a.wimg);
    const float* Bg = reinterpret_cast<const float*>(gimg + I::SMALL) + I::PE_B;     // B_layer.weight, from the global image
    {
        const int lane = tid_k & 63, wave = tid_k >> 6;
        const char* src = gimg + wave * 1024 + lane * 16;
#pragma unroll
        for (int c = 0; c < I::ROUNDS; ++c)
            wv::glds16(reinterpret_cast<const float*>(src + c * 4096), reinterpret_cast<float*>(lds + c * 4096 + wave * 1024));
    }
    bool first = true;
    for (long long chunk = blockIdx.x; chunk * kMaxPts < a.n_pts; chunk += gridDim.x) {
        // lane coordinates opaque per tile (keeps the body's LDS addresses and lane masks out of the loop pre-header)
        const int tid = wv::opaque_iter(tid_k), lane = tid & 63, wave = tid >> 6, p31 = lane & 31, hi = lane >> 5;
        const long long pt = chunk * kMaxPts + wave * 32 + p31;
        const bool valid = pt < a.n_pts;
        float t[3] = {0.0f, 0.0f, 0.0f};
        if (valid) {
            const float* px = a.pts + pt * a.pts_sn;
            t[0] = px[0] / scale;                              // embedding.py:83  x / self.scale
            t[1] = px[a.pts_sc] / scale;
            t[2] = px[2 * a.pts_sc] / scale;
        }
        // this lane's directions: hi = 0 -> 0..10, hi = 1 -> 11..20 (+ one dummy), as in step_main_s32
        float proj[11];
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            const int d0 = i, d1 = i < 10 ? 11 + i : 20;
            const float b0 = hi ? Bg[3 * d1] : Bg[3 * d0], b1 = hi ? Bg[3 * d1 + 1] : Bg[3 * d0 + 1], b2 = hi ? Bg[3 * d1 + 2] : Bg[3 * d0 + 2];
            proj[i] = fmaf(t[2], b2, fmaf(t[1], b1, t[0] * b0));          // embedding.py:84 B_layer(tensor)
        }
        float e1[48], e2[24];
        {
            float amax = 0.0f;
#pragma unroll
            for (int i = 0; i < 11; ++i) amax = fmaxf(amax, fabsf(proj[i]));
            const bool fast = !wv::wave_any(!(amax * (32.0f * kPi) < kSinCosFastLimit));
#pragma unroll
            for (int i = 0; i < 11; ++i) {
                float s[6], c[6];
                const float a0 = proj[i] * kPi;            // fl32(proj * fl32(pi)); the octaves 2^f * a0 are exact
                if (__builtin_expect(fast, 1)) octave_sincos<false>(a0, s, c);
                else octave_sincos<true>(a0, s, c);
                const bool own = i < 10 || hi == 0;        // the eleventh direction of the hi = 1 lanes is a dummy
#pragma unroll
                for (int f = 0; f < 6; ++f) {
                    const float sv = own ? s[f] : 0.0f;
                    if (f < 4) e1[4 * i + f] = sv;
                    else e2[2 * i + (f - 4)] = sv;
                }
            }
            e1[44] = hi ? 0.0f : t[0]; e1[45] = hi ? 0.0f : t[1]; e1[46] = hi ? 0.0f : t[2]; e1[47] = hi ? 0.0f : 1.0f;
            e2[22] = hi ? 0.0f : 1.0f; e2[23] = 0.0f;
        }
        unsigned e1h[24], e1m[24], e1l[24], e2h[12], e2m[12], e2l[12];
        split_planes<48, 3>(e1, e1h, e1m, e1l);
        split_planes<24, 3>(e2, e2h, e2m, e2l);
        if (first) {
            __syncthreads();            // parameter image landed (uniform: every workgroup has at least one chunk)
            first = false;
        }
        // ---- field MLP forward (model.py:59-83), the schedule of step_main_s32 ----
        unsigned h1h[8], h1m[8], h2h[8], h2m[8], h3h[8], h3m[8], h4h[8], h4m[8], xl[8];
        float h4[16], hc[16], hf[16];
        f32x16 acc, accE, accC;
        const char* w = W + I::O_IN + p31 * I::PIT_IN + 16 * hi;
        const char* wcat = W + I::O_CAT + p31 * I::PIT_CAT + 16 * hi;
        const char* wc = W + I::O_C + p31 * I::PIT_C + 16 * hi;
        zero_acc(acc);                                            // the bias rides in the column of the constant-1 slot
        fwd_chain<true, 6>(acc, w, e1h, e1m, e1l);
        zero_acc(accE);
        gap_fill<true, 3>(accE, wcat + 32 * 2, e1h, e1m, e1l, acc, hf, h1h, h1m, xl);                 // :59 in_layer -> h1
        w = W + I::O_M1 + p31 * I::PIT_M + 16 * hi;
        load_bias(acc, SM + I::B_M1, hi);
        fwd_chain<true, 2>(acc, w, h1h, h1m, xl);
        gap_fill<true, 3>(accE, wcat + 32 * 5, e1h + 12, e1m + 12, e1l + 12, acc, hf, h2h, h2m, xl);  // :60 mid1 -> h2
        fwd_chain<true, 2>(accE, wcat, h2h, h2m, xl);                                                  // :63 cat((fc2, x[:emb1]))
        zero_acc(accC);
        gap_fill<true, 2>(accC, wc + 32 * 2, e2h, e2m, e2l, accE, hf, h3h, h3m, xl);                  // :64 cat_layer -> h3
        w = W + I::O_M2 + p31 * I::PIT_M + 16 * hi;
        load_bias(acc, SM + I::B_M2, hi);
        fwd_chain<true, 2>(acc, w, h3h, h3m, xl);
        gap_fill<true, 1>(accC, wc + 32 * 4, e2h + 8, e2m + 8, e2l + 8, acc, h4, h4h, h4m, xl);       // :67 mid2 -> h4
        fwd_chain<true, 2>(accC, wc, h4h, h4m, xl);                                                    // :81 cat((fc4, x[emb1:]))
        relu_to(hc, accC);                                                                             // :81 color_linear
        float ra = 0.0f, r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = phi(r, hi);
            ra = fmaf(SM[I::W_A + j], h4[r], ra);                 // :71 out_alpha
            r0 = fmaf(SM[I::W_OC + j], hc[r], r0);                // :82 out_color
            r1 = fmaf(SM[I::W_OC + H + j], hc[r], r1);
            r2 = fmaf(SM[I::W_OC + 2 * H + j], hc[r], r2);
        }
        ra += wv::swap_half(ra); r0 += wv::swap_half(r0); r1 += wv::swap_half(r1); r2 += wv::swap_half(r2);
        ra += SM[I::B_A]; r0 += SM[I::B_OC]; r1 += SM[I::B_OC + 1]; r2 += SM[I::B_OC + 2];
        if (valid && hi == 0) {
            a.occ[pt] = sigmoidf_acc(ra * 10.0f);                 // :77 raw * 10 ; render_rays.py:6
            a.rgb[3 * pt + 0] = sigmoidf_acc(r0);                 // :83
            a.rgb[3 * pt + 1] = sigmoidf_acc(r1);
            a.rgb[3 * pt + 2] = sigmoidf_acc(r2);
        }
    }
}

}  // namespace vk
